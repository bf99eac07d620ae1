// small2.hip -- the sweep for small batches (16 < M <= 8192 candidates: DIRECT's ~100 batches of a few dozen
// sample points per maximisation, posteriors of a few hundred points), second design.
//
// The panel-split form of sweep_mfma_kernel gives each 64-row panel of W to one workgroup, which regenerates k* for
// all rows up to its diagonal and multiplies it alone: the workgroup of the last panel generates N rows and issues
// N/4 MFMAs per wave on one CU while 200 CUs are idle -- 55-60 us at N = 2048 for a batch of 30 points.  Here the
// batch goes through three short kernels, each spread over the chip:
//   kstar_small_kernel   every (32-candidate tile, 128-row stage) pair is one workgroup: k* by the exponent GEMM and
//                        the table exp of sweep2.hip, written to HBM in B-fragment order, plus that stage's part of the
//                        two mean dot products;
//   wk_small_kernel      every (tile, 16-row block of W) pair is one workgroup of 16 waves; wave w takes the 8-column
//                        steps j = w, w+16, .. of the row-block (at most 16 steps = 64 MFMAs), the 16 partial V tiles
//                        are summed through LDS in wave order, squared and reduced over the block's rows;
//   small_finish_kernel  sums the row-blocks' q and the stages' mean parts in index order, then the acquisition.
// All sums run in a fixed order (no atomics): the result does not depend on scheduling.  Candidates are read from,
// and results written to, wherever the caller's pointers lead -- abi.hip passes pinned host memory, so a batch costs
// no copy launches.
#include <atomic>
#include "sweep2_dev.h"
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>


#ifdef IBO_STAMPS   // diagnostic build (tools/stamp_small.py): the GPU-side timeline of a small batch, 100 MHz s_memrealtime ticks
__device__ unsigned long long g_sst[3][1024][4];
#define SST(k, i) do { const unsigned sb_ = blockIdx.x + gridDim.x * blockIdx.y; \
                       if (threadIdx.x == 0 && sb_ < 1024u) g_sst[k][sb_][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SST(k, i)
#endif

#define SM_TC 32                 // candidates per tile
#define SM_NW 16

// grid (ctiles, NA128 / 128); Kf[((ctile nk4 + s) 2 + cb) 64 + lane], nk4 = NA128 / 4; mupart[(stage 2 + which) Mp + c]
// A batch of at most SM_INLINE coordinates (DIRECT's: a few dozen points) travels in the kernel arguments: the candidates
// the caller hands over sit in pinned HOST memory, and reading them from there is a PCIe round trip at the head of the
// chain.  `ic` is the FIRST parameter and is never named in the body (indexing a by-value aggregate makes hipcc copy all
// of it to scratch in every thread): the kernel reads it through the kernarg segment pointer.
#define SM_INLINE 320
struct InlineCand { double v[SM_INLINE]; };
static_assert(sizeof(InlineCand) + sizeof(SweepArgs) + 32 <= 4096, "kernel arguments of kstar_small_kernel: 4 KiB at most");
// (the kernels' bodies are device functions of the item they work on -- candidate tile, stage, row-block -- so that the resident evaluation
// server of ibo_direct_max (server.hip) runs the SAME code on the same operands, item after item, without launches.  COH: the data another
// workgroup produces or consumes in the same launch -- k*, the partial sums -- goes through agent-scope accesses, past the XCDs' private L2s.)
template <bool COH> __device__ __forceinline__ double sm_ld(const double *p)
{
    if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool COH> __device__ __forceinline__ void sm_st(double *p, double v)
{
    if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
struct KstarLds {
    double *c;                 // SM_TC * (KA + 1)
    double *tab;               // 2048
    double (*al)[128];         // [2][128]
    double (*m)[SM_NW][16];    // [2][SM_NW][16]
};
// cands: where the tile's candidates are read from (nullptr: a.cand); cands_sys: they lie in host memory that changes under a resident kernel
template <int FAM, int KA4, bool COH>
__device__ __forceinline__ void kstar_small_body(const SweepArgs &a, const double *cands, double *__restrict__ Kf, double *__restrict__ mupart, int Mp,
                                                 int ctile, int t, const KstarLds &L, bool load_tab, int64_t Mo = -1)
{
    constexpr int KA = 4 * KA4;
    double *lds_c = L.c, *lds_tab = L.tab;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NA128 = (a.Npad + 127) & ~127;
    SST(0, 0);
    // the X fragments do not depend on the candidates: their L2 round trip runs beside the staging below
    const int rt = wave >> 1, gcb = wave & 1;
    const int tile = t * 8 + rt;
    double xav[KA4];
    {
        const double *xa = a.XA + (size_t)tile * KA4 * 64 + lane;
#pragma unroll
        for (int s = 0; s < KA4; s++) xav[s] = xa[s * 64];
    }
    if (load_tab) {
        lds_tab[tid] = a.exp_tab[tid];
        lds_tab[tid + 1024] = a.exp_tab[tid + 1024];
    }
    if (tid < 128) { L.al[0][tid] = a.alphaY[t * 128 + tid]; L.al[1][tid] = a.alpha1[t * 128 + tid]; }
    s2_stage_candidates<FAM, SM_TC, KA, SM_NW * 64, COH>(a, (int64_t)ctile * SM_TC, lds_c, cands, Mo);
    const double *cfrag = &lds_c[(16 * gcb + (lane & 15)) * (KA + 1) + (lane >> 4)];
    d4_t y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KA4; s++) y = mfma_f64(xav[s], cfrag[4 * s], y);
    SST(0, 1);
    double muY = 0.0, mu1 = 0.0;
    double *dst = Kf + (((size_t)ctile * (NA128 / 4) + tile * 4) * 2 + gcb) * 64 + lane;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const double kv = s2_kstar<FAM>(y[r], a.kp.sf2, lds_tab);
        const int kl = rt * 16 + 4 * r + (lane >> 4);
        muY = fma(L.al[0][kl], kv, muY);
        mu1 = fma(L.al[1][kl], kv, mu1);
        sm_st<COH>(dst + r * 128, kv);
    }
    muY += __shfl_xor(muY, 16); muY += __shfl_xor(muY, 32);
    mu1 += __shfl_xor(mu1, 16); mu1 += __shfl_xor(mu1, 32);
    if (lane < 16) { L.m[0][wave][lane] = muY; L.m[1][wave][lane] = mu1; }
    SST(0, 2);
    __syncthreads();
    if (tid < 2 * SM_TC) {
        const int which = tid >> 5, c = tid & 31;
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < SM_NW / 2; w++) s += L.m[which][2 * w + (c >> 4)][c & 15];
        sm_st<COH>(&mupart[(size_t)(t * 2 + which) * Mp + ctile * SM_TC + c], s);
    }
    SST(0, 3);
}
template <int FAM, int KA4>
__global__ __launch_bounds__(SM_NW * 64) void kstar_small_kernel(InlineCand ic, SweepArgs a, double *__restrict__ Kf, double *__restrict__ mupart, int Mp,
                                                                 int inlined)
{
    constexpr int KA = 4 * KA4;
    __shared__ double lds_c[SM_TC * (KA + 1)];
    __shared__ double lds_tab[2048];
    __shared__ double lds_al[2][128];
    __shared__ double lds_m[2][SM_NW][16];
    const KstarLds L{lds_c, lds_tab, lds_al, lds_m};
    kstar_small_body<FAM, KA4, false>(a, inlined ? (const double *)__builtin_amdgcn_kernarg_segment_ptr() : nullptr, Kf, mupart, Mp,
                                      (int)blockIdx.x, (int)blockIdx.y, L, true);
}

// grid (ctiles * 2 / CBS, Npad / 16); qpart[g Mp + c] = sum over the 16 rows of row-block g of (W K*)^2
// CBS = 2: a workgroup takes both 16-candidate blocks of its tile (batches of many tiles: W is read once per tile).
// CBS = 1: one block per workgroup -- for the few-tile batches of DIRECT, whose time is the LAST row-block's workgroup pulling its
// 2g + 2 steps x (1 KiB of W + 2 KiB of k*) through one CU (tools/stamp_small.py): half the k* bytes per CU, twice the workgroups
// on a chip that was half empty; a block that is all padding leaves at once.  Every candidate's sums are the same terms in the
// same order either way: identical bits.
template <int CBS, bool COH>
__device__ __forceinline__ void wk_small_body(const SweepArgs &a, const double *__restrict__ Kf, double *__restrict__ qpart, int Mp, int ctile, int cb0, int g,
                                              double (*lds_v)[CBS][256], double (*lds_s)[17])
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Npad = a.Npad, nk8 = Npad >> 3, NA128 = (Npad + 127) & ~127;
    SST(1, 0);
    const int nsteps = 2 * g + 2;                    // 8-column steps in which row-block g has non-zeros
    const double2 *Wp2 = (const double2 *)a.Wp + (size_t)g * nk8 * 64 + lane;
    const double *Kb = Kf + (size_t)ctile * (NA128 / 4) * 128 + cb0 * 64 + lane;
    d4_t acc[CBS];
#pragma unroll
    for (int c = 0; c < CBS; c++) acc[c] = (d4_t){0.0, 0.0, 0.0, 0.0};
    // this wave's steps j = wave, wave + 16, ..: the operands of TWO steps are in flight while a step's MFMAs issue (one step
    // ahead left an L2 round trip per step on the chain of the longest row-blocks: 16 steps at N = 2048)
    struct Ops { double2 av; double b0[CBS], b1[CBS]; };
    auto fetch = [&](int j) {
        Ops o;
        o.av = double2{0.0, 0.0};
#pragma unroll
        for (int c = 0; c < CBS; c++) o.b0[c] = o.b1[c] = 0.0;
        if (j < nsteps) {
            o.av = Wp2[(size_t)j * 64];
#pragma unroll
            for (int c = 0; c < CBS; c++) { o.b0[c] = sm_ld<COH>(&Kb[(size_t)(2 * j) * 128 + 64 * c]); o.b1[c] = sm_ld<COH>(&Kb[(size_t)(2 * j + 1) * 128 + 64 * c]); }
        }
        return o;
    };
    Ops c0 = fetch(wave), c1 = fetch(wave + SM_NW);
    for (int j = wave; j < nsteps; j += SM_NW) {
        const Ops c2 = fetch(j + 2 * SM_NW);
#pragma unroll
        for (int c = 0; c < CBS; c++) acc[c] = mfma_f64(c0.av.x, c0.b0[c], acc[c]);
#pragma unroll
        for (int c = 0; c < CBS; c++) acc[c] = mfma_f64(c0.av.y, c0.b1[c], acc[c]);
        c0 = c1; c1 = c2;
    }
#pragma unroll
    for (int c = 0; c < CBS; c++)
#pragma unroll
        for (int r = 0; r < 4; r++) lds_v[wave][c][lane * 4 + r] = acc[c][r];
    SST(1, 1);
    __syncthreads();
    SST(1, 2);
    if (tid < 256 * CBS) {
        // element e of candidate block cb: lane l = e >> 2, r = e & 3 -> row (l >> 4) + 4 r, candidate 16 cb + (l & 15)
        const int cb = tid >> 8, e = tid & 255, l = e >> 2, r = e & 3;
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < SM_NW; w++) v += lds_v[w][cb][e];
        lds_s[16 * cb + (l & 15)][(l >> 4) + 4 * r] = v * v;
    }
    __syncthreads();
    if (tid < 16 * CBS) {
        double q = 0.0;
#pragma unroll
        for (int rr = 0; rr < 16; rr++) q += lds_s[tid][rr];
        sm_st<COH>(&qpart[(size_t)g * Mp + ctile * SM_TC + cb0 * 16 + tid], q);
    }
    SST(1, 3);
}
template <int CBS>
__global__ __launch_bounds__(SM_NW * 64) void wk_small_kernel(SweepArgs a, const double *__restrict__ Kf, double *__restrict__ qpart, int Mp)
{
    __shared__ double lds_v[SM_NW][CBS][256];        // partial V tiles: [wave][cand-block][lane 64 x 4]
    __shared__ double lds_s[16 * CBS][17];           // squared sums [cand][row]
    const int ctile = CBS == 2 ? blockIdx.x : blockIdx.x >> 1, cb0 = CBS == 2 ? 0 : blockIdx.x & 1;
    if (CBS == 1 && (int64_t)ctile * SM_TC + cb0 * 16 >= a.M) return;
    const int g = gridDim.y - 1 - blockIdx.y;        // the longest rows of W first: the short ones fill the tail
    wk_small_body<CBS, false>(a, Kf, qpart, Mp, ctile, cb0, g, lds_v, lds_s);
}

// WAVE-LOCAL k*: the exponent GEMM's output layout (lane l: row (l>>4) + 4 r, candidate l & 15) IS the B-fragment layout of the
// product that follows, so a wave can make the k* of a 16-row tile and multiply it with its W fragments without anything
// leaving its registers -- no LDS stage, no barrier, no trip of k* through memory.  Workgroup (tile, row-block g): wave w takes
// the 16-row tiles rt = w, w + 16, .. <= g of both candidate blocks; per tile 2 (KA4 + 4) MFMAs and 8 exp() per lane, the next
// tile's fragments of X and W requested while this one is computed.  k* is regenerated by every row-block (N / 32 times on
// average) but a DIRECT batch is bound by its launches: one launch instead of two, ~4 us of kernel instead of 6 + 9 at N = 1024.
// The last row-block's workgroups see every row and also form the two mean dot products (one part per tile: nst = 1 for the
// finish kernel).  Sums run in a fixed order (tiles ascending per wave, waves ascending): deterministic.
struct WklLds {
    double *c;                        // SM_TC * (KA + 1)
    double *tab;                      // 2048
    double (*v)[2][256];              // [SM_NW][2][256]
    double (*s)[17];                  // [SM_TC][17]
    double (*m)[SM_NW][2][16];        // [2][SM_NW][2][16]
};
template <int FAM, int KA4, bool COH>
__device__ __forceinline__ void wkl_small_body(const SweepArgs &a, const double *cands, double *__restrict__ qpart, double *__restrict__ mupart, int Mp,
                                               int ctile, int g, bool means, const WklLds &L, bool load_tab, int64_t Mo = -1)
{
    constexpr int KA = 4 * KA4;
    double *lds_c = L.c, *lds_tab = L.tab;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk8 = a.Npad >> 3;
    if (load_tab) {
        lds_tab[tid] = a.exp_tab[tid];
        lds_tab[tid + 1024] = a.exp_tab[tid + 1024];
    }
    s2_stage_candidates<FAM, SM_TC, KA, SM_NW * 64, COH>(a, (int64_t)ctile * SM_TC, lds_c, cands, Mo);
    const double *cfrag0 = &lds_c[(lane & 15) * (KA + 1) + (lane >> 4)], *cfrag1 = cfrag0 + 16 * (KA + 1);
    const double2 *Wp2 = (const double2 *)a.Wp + (size_t)g * nk8 * 64 + lane;
    d4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    double mY0 = 0.0, mY1 = 0.0, m10 = 0.0, m11 = 0.0;
    double xan[KA4], aYn[4], a1n[4];
    double2 w0n = {0.0, 0.0}, w1n = {0.0, 0.0};
    auto request = [&](int rt) {
        const double *xa = a.XA + (size_t)rt * KA4 * 64 + lane;
#pragma unroll
        for (int s = 0; s < KA4; s++) xan[s] = xa[s * 64];
        w0n = Wp2[(size_t)(2 * rt) * 64];
        w1n = Wp2[(size_t)(2 * rt + 1) * 64];
        if (means) {
#pragma unroll
            for (int r = 0; r < 4; r++) { aYn[r] = a.alphaY[16 * rt + 4 * r + (lane >> 4)]; a1n[r] = a.alpha1[16 * rt + 4 * r + (lane >> 4)]; }
        }
    };
    if (wave <= g) request(wave);
    for (int rt = wave; rt <= g; rt += SM_NW) {
        double xa[KA4], aY[4], a1v[4];
#pragma unroll
        for (int s = 0; s < KA4; s++) xa[s] = xan[s];
        const double2 w0 = w0n, w1 = w1n;
#pragma unroll
        for (int r = 0; r < 4; r++) { aY[r] = aYn[r]; a1v[r] = a1n[r]; }
        if (rt + SM_NW <= g) request(rt + SM_NW);
        d4_t y0 = {0.0, 0.0, 0.0, 0.0}, y1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KA4; s++) { y0 = mfma_f64(xa[s], cfrag0[4 * s], y0); y1 = mfma_f64(xa[s], cfrag1[4 * s], y1); }
        const double wk[4] = {w0.x, w0.y, w1.x, w1.y};
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const double k0 = s2_kstar<FAM>(y0[r], a.kp.sf2, lds_tab), k1 = s2_kstar<FAM>(y1[r], a.kp.sf2, lds_tab);
            acc0 = mfma_f64(wk[r], k0, acc0);
            acc1 = mfma_f64(wk[r], k1, acc1);
            if (means) { mY0 = fma(aY[r], k0, mY0); mY1 = fma(aY[r], k1, mY1); m10 = fma(a1v[r], k0, m10); m11 = fma(a1v[r], k1, m11); }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) { L.v[wave][0][lane * 4 + r] = acc0[r]; L.v[wave][1][lane * 4 + r] = acc1[r]; }
    if (means) {
        mY0 += __shfl_xor(mY0, 16); mY0 += __shfl_xor(mY0, 32); mY1 += __shfl_xor(mY1, 16); mY1 += __shfl_xor(mY1, 32);
        m10 += __shfl_xor(m10, 16); m10 += __shfl_xor(m10, 32); m11 += __shfl_xor(m11, 16); m11 += __shfl_xor(m11, 32);
        if (lane < 16) { L.m[0][wave][0][lane] = mY0; L.m[0][wave][1][lane] = mY1; L.m[1][wave][0][lane] = m10; L.m[1][wave][1][lane] = m11; }
    }
    __syncthreads();
    if (tid < 512) {
        // element e of candidate block cb: lane l = e >> 2, r = e & 3 -> row (l >> 4) + 4 r, candidate 16 cb + (l & 15)
        const int cb = tid >> 8, e = tid & 255, l = e >> 2, r = e & 3;
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < SM_NW; w++) v += L.v[w][cb][e];
        L.s[16 * cb + (l & 15)][(l >> 4) + 4 * r] = v * v;
    } else if (means && tid < 512 + 2 * SM_TC) {
        const int which = (tid - 512) >> 5, c = (tid - 512) & 31;
        double sm = 0.0;
#pragma unroll
        for (int w = 0; w < SM_NW; w++) sm += L.m[which][w][c >> 4][c & 15];
        sm_st<COH>(&mupart[(size_t)which * Mp + ctile * SM_TC + c], sm);
    }
    __syncthreads();
    if (tid < SM_TC) {
        double q = 0.0;
#pragma unroll
        for (int rr = 0; rr < 16; rr++) q += L.s[tid][rr];
        sm_st<COH>(&qpart[(size_t)g * Mp + ctile * SM_TC + tid], q);
    }
}
template <int FAM, int KA4>
__global__ __launch_bounds__(SM_NW * 64) void wkl_small_kernel(InlineCand ic, SweepArgs a, double *__restrict__ qpart, double *__restrict__ mupart,
                                                               int Mp, int inlined)
{
    constexpr int KA = 4 * KA4;
    __shared__ double lds_c[SM_TC * (KA + 1)];
    __shared__ double lds_tab[2048];
    __shared__ double lds_v[SM_NW][2][256];          // partial V tiles: [wave][cand-block][lane 64 x 4]
    __shared__ double lds_s[SM_TC][17];
    __shared__ double lds_m[2][SM_NW][2][16];        // partial means [which][wave][cand-block][cand]
    const WklLds L{lds_c, lds_tab, lds_v, lds_s, lds_m};
    const int g = gridDim.y - 1 - blockIdx.y;        // the longest rows of W first
    wkl_small_body<FAM, KA4, false>(a, inlined ? (const double *)__builtin_amdgcn_kernarg_segment_ptr() : nullptr, qpart, mupart, Mp, (int)blockIdx.x, g,
                                    g == (int)gridDim.y - 1, L, true);
}

// 64 candidates per workgroup of 512 threads: eight threads per candidate each sum a contiguous eighth of the row-blocks'
// q (all their loads in flight at once: one L2 round trip instead of nrb / 8 dependent ones -- the kernel was 7 of a DIRECT
// batch's 22 us), the stages' mean parts likewise; the eight partial sums are added in index order by the candidate's first
// thread, which then evaluates the acquisition.  Every sum has a fixed order: the result does not depend on scheduling.
// (512 threads, not 1024: the acquisition's erf / exp chains want more than the 128 registers a 1024-thread workgroup leaves a lane.)
#define SM_FIN_P 8
// the finish of candidates [64 item, 64 item + 64): called by SM_FIN_P * 64 threads (waves 0 .. SM_FIN_P - 1 of the workgroup); returns in wave 0
// only, lane 0 holding the tile's (value, index).  Ends with the partials stored; the caller signals.
template <bool COH, bool SYSOUT = COH>
__device__ __forceinline__ bool small_finish_body(const SweepArgs &a, const double *__restrict__ qpart, const double *__restrict__ mupart, int Mp,
                                                  int nrb, int nst, int item, double (*lds_q)[64], double (*lds_y)[64], double (*lds_1)[64],
                                                  int64_t Mo = -1, const double *cands = nullptr)
{
    const int lane = threadIdx.x & 63, p = threadIdx.x >> 6;
    const int64_t Mtot = Mo >= 0 ? Mo : a.M;
    const int64_t li = (int64_t)item * 64 + lane;
    const bool valid = li < Mtot;
    const int64_t ci = valid ? li : Mtot - 1;
    if (!cands) cands = a.cand;
    SST(2, 0);
    if (p < SM_FIN_P) {                                  // (a caller with more waves than SM_FIN_P: the others only keep the barrier)
        const int per = (nrb + SM_FIN_P - 1) / SM_FIN_P, g0 = p * per;
        double v[16];
        double qs = 0.0;
        for (int gb = 0; gb < per; gb += 16) {           // (per <= 16 up to N = 2048: one round)
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = (gb + u < per && g0 + gb + u < nrb) ? sm_ld<COH>(&qpart[(size_t)(g0 + gb + u) * Mp + ci]) : 0.0;
#pragma unroll
            for (int u = 0; u < 16; u++) if (gb + u < per && g0 + gb + u < nrb) qs += v[u];
        }
        lds_q[p][lane] = qs;
        const int pers = (nst + SM_FIN_P - 1) / SM_FIN_P, t0 = p * pers;
        double ys = 0.0, os = 0.0;
        for (int t = t0; t < t0 + pers && t < nst; t++) { ys += sm_ld<COH>(&mupart[(size_t)(2 * t) * Mp + ci]); os += sm_ld<COH>(&mupart[(size_t)(2 * t + 1) * Mp + ci]); }
        lds_y[p][lane] = ys; lds_1[p][lane] = os;
    }
    __syncthreads();
    if (p != 0) return false;
    SST(2, 1);
    double q = 0.0, my = 0.0, m1 = 0.0;
#pragma unroll
    for (int u = 0; u < SM_FIN_P; u++) { q += lds_q[u][lane]; my += lds_y[u][lane]; m1 += lds_1[u][lane]; }
    bool excl;
    double val = s2_finish<SYSOUT>(a, cands + ci * a.kp.D, q, my, m1, li, valid, excl);
    int64_t idx = a.index_base + li;
    if (!valid || excl || !(val == val)) { val = -INFINITY; idx = INT64_MAX; }
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(val, o);
        const int64_t oi = __shfl_xor(idx, o);
        if (ov > val || (ov == val && oi < idx)) { val = ov; idx = oi; }
    }
    if (lane == 0 && a.part_val) { a.part_val[item] = val; a.part_idx[item] = idx; }
    SST(2, 2);
    return true;
}
template <bool SYSOUT>
__global__ __launch_bounds__(SM_FIN_P * 64) void small_finish_kernel(SweepArgs a, const double *__restrict__ qpart, const double *__restrict__ mupart, int Mp,
                                                                     int nrb, int nst)
{
    __shared__ double lds_q[SM_FIN_P][64], lds_y[SM_FIN_P][64], lds_1[SM_FIN_P][64];
    if (!small_finish_body<false, SYSOUT>(a, qpart, mupart, Mp, nrb, nst, (int)blockIdx.x, lds_q, lds_y, lds_1)) return;
    const int lane = threadIdx.x & 63;
    if (a.done_flag) {
        // every lane's results (possibly in host memory) are out before this workgroup takes its ticket; the last ticket
        // publishes the sequence number the host is waiting for.  SYSOUT (a host that spins on the flag): the results went out as system-scope
        // stores, which have no L2 line to be written back -- waiting for their acknowledgement is all the ordering the flag needs, where
        // __threadfence_system() writes back the whole L2
        if (SYSOUT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else __threadfence_system();
        if (gridDim.x == 1) {                            // at most 64 candidates: this wave IS the batch -- no ticket to take
            if (lane == 0) *(volatile unsigned long long *)a.done_flag = a.done_seq;
        } else if (lane == 0) {
            if (atomicAdd(a.done_count, 1u) == gridDim.x - 1) {
                *a.done_count = 0;
                if (!SYSOUT) __threadfence_system();
                *(volatile unsigned long long *)a.done_flag = a.done_seq;
            }
        }
    }
    SST(2, 3);
}

template <int FAM>
static int launch_kstar_small(const SweepArgs &a, double *Kf, double *mupart, int Mp, dim3 grid, hipStream_t s)
{
    InlineCand ic;
    int inl = 0;
    if (a.cand_host && a.M * a.kp.D <= SM_INLINE) {
        memcpy(ic.v, a.cand_host, sizeof(double) * (size_t)(a.M * a.kp.D));
        inl = 1;
    }
    switch ((a.kp.D + 2 + 3) / 4) {
    case 1: hipLaunchKernelGGL((kstar_small_kernel<FAM, 1>), grid, dim3(SM_NW * 64), 0, s, ic, a, Kf, mupart, Mp, inl); break;
    case 2: hipLaunchKernelGGL((kstar_small_kernel<FAM, 2>), grid, dim3(SM_NW * 64), 0, s, ic, a, Kf, mupart, Mp, inl); break;
    case 3: hipLaunchKernelGGL((kstar_small_kernel<FAM, 3>), grid, dim3(SM_NW * 64), 0, s, ic, a, Kf, mupart, Mp, inl); break;
    case 4: hipLaunchKernelGGL((kstar_small_kernel<FAM, 4>), grid, dim3(SM_NW * 64), 0, s, ic, a, Kf, mupart, Mp, inl); break;
    case 5: hipLaunchKernelGGL((kstar_small_kernel<FAM, 5>), grid, dim3(SM_NW * 64), 0, s, ic, a, Kf, mupart, Mp, inl); break;
    case 6: hipLaunchKernelGGL((kstar_small_kernel<FAM, 6>), grid, dim3(SM_NW * 64), 0, s, ic, a, Kf, mupart, Mp, inl); break;
    case 7: hipLaunchKernelGGL((kstar_small_kernel<FAM, 7>), grid, dim3(SM_NW * 64), 0, s, ic, a, Kf, mupart, Mp, inl); break;
    case 8: hipLaunchKernelGGL((kstar_small_kernel<FAM, 8>), grid, dim3(SM_NW * 64), 0, s, ic, a, Kf, mupart, Mp, inl); break;
    default: hipLaunchKernelGGL((kstar_small_kernel<FAM, 9>), grid, dim3(SM_NW * 64), 0, s, ic, a, Kf, mupart, Mp, inl); break;
    }
    return (int)hipGetLastError();
}

template <int FAM>
static int launch_wkl_small(const SweepArgs &a, double *qpart, double *mupart, int Mp, dim3 grid, hipStream_t s)
{
    InlineCand ic;
    int inl = 0;
    if (a.cand_host && a.M * a.kp.D <= SM_INLINE) {
        memcpy(ic.v, a.cand_host, sizeof(double) * (size_t)(a.M * a.kp.D));
        inl = 1;
    }
    switch ((a.kp.D + 2 + 3) / 4) {           // (up to 10 dimensions: beyond, the tile's X fragments push the kernel into scratch, and the two-kernel path runs)
    case 1: hipLaunchKernelGGL((wkl_small_kernel<FAM, 1>), grid, dim3(SM_NW * 64), 0, s, ic, a, qpart, mupart, Mp, inl); break;
    case 2: hipLaunchKernelGGL((wkl_small_kernel<FAM, 2>), grid, dim3(SM_NW * 64), 0, s, ic, a, qpart, mupart, Mp, inl); break;
    default: hipLaunchKernelGGL((wkl_small_kernel<FAM, 3>), grid, dim3(SM_NW * 64), 0, s, ic, a, qpart, mupart, Mp, inl); break;
    }
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------------------------
// The RESIDENT evaluation server of ibo_direct_max (round 6).  DIRECT hands the objective ~53 dependent batches of a few dozen to a few
// hundred points; as launches a batch costs ~25 us of which ~4 are kernel execution (10 us in three launch calls, 6 us from an empty launch
// to its flag, 2.5 us per further launch: profiles/r05_direct_batch_floor.txt), whatever the model's size (maximizeEI: 1.37 ms at N = 64).
// For the lifetime of ONE ibo_direct_max call this kernel stays on the chip -- one workgroup per CU -- and takes its batches from a mailbox
// in pinned host memory (1.75 us round trip): workgroup 0 polls the mailbox's sequence word and relays it through a device word the others
// poll; the batch then runs as the same items as the launches (kstar_small_body -> wk_small_body -> small_finish_body, or wkl_small_body ->
// small_finish_body on small models: the same code on the same operands, bit-identical values) with two device-wide hand-overs -- producers
// store through to the coherence point (agent-scope accesses), wait for their stores, count themselves in; consumers wait for the count --
// and the last finisher stores the batch's sequence number behind the results in host memory.
// EVERY WAIT IS BOUNDED (wall_clock64, 100 MHz): workgroups that do not all become resident within ~2 ms (another process's server holds
// part of the chip) give up before the first batch; a mailbox silent for `idle_ticks` (the host died, or stopped feeding) ends the
// kernel; a hand-over that does not complete in that time raises the abort word, which every poll loop watches.  The host side (abi_sweep.hip)
// bounds its own waits too and finishes the call on the launch path whenever the server is not there -- there is no hung-GPU mode.
struct ServerCtl { unsigned go, M, abort, arrive, cnt[3], pad; };
#define SRV_EXIT 0xffffffffu
#define SRV_BOX_CAND 8                               // the mailbox in doubles: [0] seq, [1] M, [2] done, [3] state, [8 ..) M x D candidates, then M values
enum { SRV_READY = 1, SRV_NOT_RESIDENT = 2, SRV_LEFT_ON_DEADLINE = 3, SRV_LEFT = 4 };

__device__ __forceinline__ unsigned srv_ld(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void srv_st(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long box_ld(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void box_st(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// every workgroup's stores have reached the coherence point, then it is counted in and waits for the count (cumulative over the batches).
// false: the abort word was raised, or the wait ran out (and raised it)
__device__ __forceinline__ bool srv_handover(ServerCtl *ctl, unsigned *cnt, unsigned target, long long limit, unsigned *sh)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned ok = 1;
        const long long t0 = wall_clock64();
        while ((int)(srv_ld(cnt) - target) < 0) {
            if (srv_ld(&ctl->abort)) { ok = 0; break; }
            if (wall_clock64() - t0 > limit) { srv_st(&ctl->abort, 1u); ok = 0; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        *sh = ok;
    }
    __syncthreads();
    return *sh != 0;
}

template <int FAM, int KA4>
__global__ __launch_bounds__(SM_NW * 64) void direct_server_kernel(SweepArgs a, ServerCtl *ctl, double *box, double *__restrict__ ws, int Mmax,
                                                                   long long idle_ticks, unsigned long long *stamps)
{
    // (diagnostics, tools/direct_server_ab.py with IBO_SRV_STAMPS=1: per batch < 64, eight 100 MHz stamps of workgroup 0 and of the last one)
#define SRV_STAMP(k) do { if (stamps && threadIdx.x == 0 && nbatch <= 64 && (wg == 0 || wg == G - 1)) \
                              stamps[((size_t)(nbatch - 1) * 2 + (wg ? 1 : 0)) * 8 + (k)] = wall_clock64(); } while (0)
    constexpr int KA = 4 * KA4;
    __shared__ double lds_tab[2048];
    __shared__ double pool[SM_TC * (4 * 9 + 1) + SM_NW * 2 * 256 + SM_TC * 17 + 2 * SM_NW * 2 * 16];     // the largest phase: wkl's (c, v, s, m)
    __shared__ unsigned sh_go, sh_M, sh_ok;
    const int tid = threadIdx.x, lane = tid & 63, wg = blockIdx.x, G = gridDim.x;
    unsigned long long *bx = (unsigned long long *)box;
    const double *cands = box + SRV_BOX_CAND;
    lds_tab[tid] = a.exp_tab[tid];
    lds_tab[tid + 1024] = a.exp_tab[tid + 1024];
    // the phases' views of the pool
    KstarLds Lk; Lk.c = pool; Lk.tab = lds_tab; Lk.al = (double (*)[128])(pool + SM_TC * (KA + 1)); Lk.m = (double (*)[SM_NW][16])(pool + SM_TC * (KA + 1) + 256);
    WklLds Ll; Ll.c = pool; Ll.tab = lds_tab; Ll.v = (double (*)[2][256])(pool + SM_TC * (KA + 1)); Ll.s = (double (*)[17])(pool + SM_TC * (KA + 1) + SM_NW * 512);
    Ll.m = (double (*)[SM_NW][2][16])(pool + SM_TC * (KA + 1) + SM_NW * 512 + SM_TC * 17);
    double (*wv1)[1][256] = (double (*)[1][256])pool; double (*wv2)[2][256] = (double (*)[2][256])pool;
    double (*ws1)[17] = (double (*)[17])(pool + SM_NW * 512);
    double (*fq)[64] = (double (*)[64])pool, (*fy)[64] = fq + SM_FIN_P, (*f1)[64] = fy + SM_FIN_P;
    const int Npad = a.Npad, NA128 = (Npad + 127) & ~127, nst = NA128 / 128, nrb = Npad / 16;
    // ---- is every workgroup on the chip?
    if (tid == 0) {
        __hip_atomic_fetch_add(&ctl->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned ok = 1;
        const long long t0 = wall_clock64();
        while (srv_ld(&ctl->arrive) < (unsigned)G) {
            if (srv_ld(&ctl->abort)) { ok = 0; break; }
            if (wall_clock64() - t0 > 200000LL) { srv_st(&ctl->abort, 1u); ok = 0; break; }       // 2 ms
            __builtin_amdgcn_s_sleep(2);
        }
        if (wg == 0) box_st(bx + 3, ok ? SRV_READY : SRV_NOT_RESIDENT);
        sh_ok = ok;
    }
    __syncthreads();
    if (!sh_ok) return;
    unsigned seq = 0, fin_total = 0, nbatch = 0;
    for (;;) {
        // ---- the next batch: workgroup 0 from the mailbox, the others from the word it relays
        if (tid == 0) {
            unsigned go = SRV_EXIT, M = 0;
            const long long t0 = wall_clock64();
            if (wg == 0) {
                unsigned long long s;
                bool timed_out = false;
                while ((s = box_ld(bx)) == (unsigned long long)seq) {
                    if (srv_ld(&ctl->abort) || wall_clock64() - t0 > idle_ticks) { timed_out = true; break; }
                }
                if (!timed_out && s != (unsigned long long)SRV_EXIT) { go = (unsigned)s; M = (unsigned)box_ld(bx + 1); if (M < 1 || M > (unsigned)Mmax) go = SRV_EXIT; }
                if (stamps && nbatch < 64) stamps[((size_t)nbatch * 2) * 8 + 6] = wall_clock64();
                // (no release / acquire operations anywhere in this kernel: each one is a write-back or an invalidation of the XCD's whole L2 --
                // with 255 workgroups polling `go` with acquire loads, an item of 2 us took 10.  Everything that crosses workgroups is accessed
                // past the L2 already; what remains is ORDER, and a wave's accesses to the coherence point complete in order once waited for)
                srv_st(&ctl->M, M);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                srv_st(&ctl->go, go);
                if (go == SRV_EXIT) box_st(bx + 3, timed_out ? SRV_LEFT_ON_DEADLINE : SRV_LEFT);
            } else {
                for (;;) {
                    go = srv_ld(&ctl->go);
                    if (go != seq) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); M = srv_ld(&ctl->M); break; }
                    if (srv_ld(&ctl->abort) || wall_clock64() - t0 > idle_ticks + 100000LL) { go = SRV_EXIT; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            sh_go = go; sh_M = M;
        }
        __syncthreads();
        if (sh_go == SRV_EXIT) return;
        seq = sh_go;
        nbatch++;
        SRV_STAMP(0);
        const int64_t M = sh_M;
        const int Mp = (int)((M + SM_TC - 1) / SM_TC) * SM_TC, ctiles = Mp / SM_TC;
        double *Kf = ws, *qpart = Kf + (size_t)Mp * NA128, *mupart = qpart + (size_t)nrb * Mp;
        int nst_fin = nst;
        const bool local = KA4 <= 3 && ctiles <= 4 && nrb <= 32;
        if (local) {
            if constexpr (KA4 <= 3) {
                const int n = ctiles * nrb;
                for (int i = wg; i < n; i += G) {
                    const int g = nrb - 1 - i / ctiles;
                    wkl_small_body<FAM, KA4, true>(a, cands, qpart, mupart, Mp, i % ctiles, g, g == nrb - 1, Ll, false, M);
                    __syncthreads();
                }
            }
            nst_fin = 1;
            SRV_STAMP(1);
        } else {
            const int n1 = ctiles * nst;
            for (int i = wg; i < n1; i += G) {
                kstar_small_body<FAM, KA4, true>(a, cands, Kf, mupart, Mp, i % ctiles, i / ctiles, Lk, false, M);
                __syncthreads();
            }
            SRV_STAMP(1);
            if (!srv_handover(ctl, &ctl->cnt[0], nbatch * (unsigned)G, idle_ticks, &sh_ok)) return;
            SRV_STAMP(2);
            if (ctiles <= 8) {
                const int nx = 2 * ctiles, n2 = nx * nrb;
                for (int i = wg; i < n2; i += G) {
                    const int x = i % nx, g = nrb - 1 - i / nx;
                    if ((int64_t)(x >> 1) * SM_TC + (x & 1) * 16 >= M) continue;
                    wk_small_body<1, true>(a, Kf, qpart, Mp, x >> 1, x & 1, g, wv1, ws1);
                    __syncthreads();
                }
            } else {
                const int n2 = ctiles * nrb;
                for (int i = wg; i < n2; i += G) {
                    wk_small_body<2, true>(a, Kf, qpart, Mp, i % ctiles, 0, nrb - 1 - i / ctiles, wv2, (double (*)[17])(pool + SM_NW * 512));
                    __syncthreads();
                }
            }
        }
        SRV_STAMP(3);
        if (!srv_handover(ctl, &ctl->cnt[1], nbatch * (unsigned)G, idle_ticks, &sh_ok)) return;
        SRV_STAMP(4);
        // ---- the finish; the last finisher of the batch tells the host
        const int nfin = (int)((M + 63) / 64);
        fin_total += (unsigned)nfin;
        for (int i = wg; i < nfin; i += G) {
            const bool w0 = small_finish_body<true>(a, qpart, mupart, Mp, nrb, nst_fin, i, fq, fy, f1, M, cands);
            if (w0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every lane's value has been accepted by the path to host memory before the ticket
                if (lane == 0 && __hip_atomic_fetch_add(&ctl->cnt[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fin_total - 1)
                    box_st(bx + 2, (unsigned long long)seq);
            }
            __syncthreads();
        }
        SRV_STAMP(5);
    }
#undef SRV_STAMP
}

template <int FAM>
static int launch_direct_server_fam(const SweepArgs &a, void *ctl, double *box, double *ws, int Mmax, int G, long long idle_ticks, unsigned long long *stamps, hipStream_t s)
{
#define SRV_LAUNCH(K) hipLaunchKernelGGL((direct_server_kernel<FAM, K>), dim3(G), dim3(SM_NW * 64), 0, s, a, (ServerCtl *)ctl, box, ws, Mmax, idle_ticks, stamps)
    switch ((a.kp.D + 2 + 3) / 4) {
    case 1: SRV_LAUNCH(1); break;
    case 2: SRV_LAUNCH(2); break;
    case 3: SRV_LAUNCH(3); break;
    case 4: SRV_LAUNCH(4); break;
    case 5: SRV_LAUNCH(5); break;
    default: return (int)hipErrorInvalidValue;
    }
#undef SRV_LAUNCH
    return (int)hipGetLastError();
}
// a.cand / a.out_acq: the mailbox's candidate and value areas (pinned host memory, device-visible); ctl: sizeof(ServerCtl) zeroed bytes
// of device memory; ws: small_sweep_workspace(Npad, Mmax) doubles
bool direct_server_takes(const SweepArgs &a) { return a.dot_form && a.kp.D <= 18 && a.n_excl == 0 && !a.out_mu && !a.out_s2; }
int launch_direct_server(const SweepArgs &a, void *ctl, double *box, double *ws, int Mmax, int G, double idle_ms, hipStream_t s, unsigned long long *stamps)
{
    const long long ticks = (long long)(idle_ms * 1e5);
    if (a.kp.family == FAM_SE) return launch_direct_server_fam<FAM_SE>(a, ctl, box, ws, Mmax, G, ticks, stamps, s);
    if (a.kp.family == FAM_M3) return launch_direct_server_fam<FAM_M3>(a, ctl, box, ws, Mmax, G, ticks, stamps, s);
    return launch_direct_server_fam<FAM_M5>(a, ctl, box, ws, Mmax, G, ticks, stamps, s);
}

// doubles of workspace: Kf (Mp NA128) + qpart (Npad/16 Mp) + mupart (2 NA128/128 Mp), Mp = M rounded up to 32
size_t small_sweep_workspace(int Npad, int64_t M)
{
    const size_t Mp = (size_t)((M + SM_TC - 1) / SM_TC) * SM_TC, NA128 = (size_t)((Npad + 127) & ~127);
    return Mp * NA128 + (size_t)(Npad / 16) * Mp + 2 * (NA128 / 128) * Mp;
}

int launch_sweep_small(const SweepArgs &a, double *ws, hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    const int Mp = (int)((a.M + SM_TC - 1) / SM_TC) * SM_TC, ctiles = Mp / SM_TC;
    const int NA128 = (a.Npad + 127) & ~127, nst = NA128 / 128, nrb = a.Npad / 16;
    double *Kf = ws, *qpart = Kf + (size_t)Mp * NA128, *mupart = qpart + (size_t)nrb * Mp;
    if (e0) (void)hipEventRecord(e0, s);
    int rc;
    int nst_fin = nst;                                   // mean parts per candidate the finish kernel sums
    // few tiles on a small model (DIRECT's batches, single posteriors): ONE launch, every wave making the k* it multiplies
    // (wkl_small_kernel).  Every row-block's workgroup regenerates k* (N / 32 times on average): measured under the tracer
    // 4.4 us at N = 64, but 15 us at N = 1024 and 28-34 us at N = 2048 against 6 + 8.6 / 6 + 16 for the two separate kernels --
    // its 14-instruction exp() chains are then the throughput of the 128 CUs it occupies.  (A stage-by-stage fusion through an
    // LDS stage and a barrier per 128 rows was measured too: 0.6 us per stage, slower from N = 512 on, and removed.)
    const bool local = ctiles <= 4 && nrb <= 32 && a.kp.D <= 10;
    if (local) {
        const dim3 gl(ctiles, nrb);
        if (a.kp.family == FAM_SE) rc = launch_wkl_small<FAM_SE>(a, qpart, mupart, Mp, gl, s);
        else if (a.kp.family == FAM_M3) rc = launch_wkl_small<FAM_M3>(a, qpart, mupart, Mp, gl, s);
        else rc = launch_wkl_small<FAM_M5>(a, qpart, mupart, Mp, gl, s);
        if (rc) return rc;
        nst_fin = 1;
    } else {
        const dim3 g1(ctiles, nst);
        if (a.kp.family == FAM_SE) rc = launch_kstar_small<FAM_SE>(a, Kf, mupart, Mp, g1, s);
        else if (a.kp.family == FAM_M3) rc = launch_kstar_small<FAM_M3>(a, Kf, mupart, Mp, g1, s);
        else rc = launch_kstar_small<FAM_M5>(a, Kf, mupart, Mp, g1, s);
        if (rc) return rc;
        // (round 6: whether a workgroup takes one 16-candidate block or the tile's two -- half or all of W's traffic -- makes no difference to a DIRECT run at
        // N = 2048 either: 2.19 .. 2.23 ms for every split point; the batch is latency, not W's bytes)
        if (ctiles <= 8) hipLaunchKernelGGL(wk_small_kernel<1>      // (one 16-candidate block per workgroup: twice the workgroups on a half-empty chip)
           , dim3(2 * ctiles, nrb), dim3(SM_NW * 64), 0, s, a, Kf, qpart, Mp);
        else hipLaunchKernelGGL(wk_small_kernel<2>, dim3(ctiles, nrb), dim3(SM_NW * 64), 0, s, a, Kf, qpart, Mp);
    }
    if (e1) (void)hipEventRecord(e1, s);
    const int64_t nfin = (a.M + 63) / 64;
    // (a caller that spins on the flag: results as system-scope stores and a wait for their acknowledgement instead of a fence that writes the L2 back -- ~1 us per batch)
    if (a.done_flag) hipLaunchKernelGGL(small_finish_kernel<true>, dim3((unsigned)nfin), dim3(SM_FIN_P * 64), 0, s, a, qpart, mupart, Mp, nrb, nst_fin);
    else hipLaunchKernelGGL(small_finish_kernel<false>, dim3((unsigned)nfin), dim3(SM_FIN_P * 64), 0, s, a, qpart, mupart, Mp, nrb, nst_fin);
    rc = (int)hipGetLastError();
    if (rc) return rc;
#ifdef IBO_STAMPS
    if (getenv("IBO_STAMP_FILE") && !local) {        // one record per batch: header (8 words), then the stamps of the three kernels
        static std::vector<unsigned long long> h(8 + 3 * 1024 * 4);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(h.data() + 8, HIP_SYMBOL(g_sst), sizeof(unsigned long long) * 3 * 1024 * 4);
        h[0] = (unsigned long long)a.M; h[1] = (unsigned long long)a.Npad; h[2] = (unsigned long long)ctiles; h[3] = (unsigned long long)nst;
        h[4] = (unsigned long long)nrb; h[5] = (unsigned long long)nfin; h[6] = h[7] = 0;
        FILE *f = fopen(getenv("IBO_STAMP_FILE"), "ab");
        if (f) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    }
#endif
    return launch_argmax_final(a, nfin, s);
}

// HIP loads a translation unit's code object when one of its kernels is first needed -- 0.5 .. 1.5 ms in the middle of whatever call that
// is (the first gallery call of a process paid 2.8 ms for two of them).  The library asks for one kernel of every unit when it makes its
// first allocation on a device (abi_core.hip: load_code_objects), beside the arena's first slab: start-up cost, paid once.
void ibo_touch_small2() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, (const void *)small_finish_kernel<false>); }
