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
template <int FAM, int KA4>
__global__ __launch_bounds__(SM_NW * 64) void kstar_small_kernel(InlineCand ic, SweepArgs a, double *__restrict__ Kf, double *__restrict__ mupart, int Mp,
                                                                 int inlined)
{
    constexpr int KA = 4 * KA4;
    __shared__ double lds_c[SM_TC * (KA + 1)];
    __shared__ double lds_tab[2048];
    __shared__ double lds_al[2][128];
    __shared__ double lds_m[2][SM_NW][16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ctile = blockIdx.x, t = blockIdx.y;
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
    lds_tab[tid] = a.exp_tab[tid];
    lds_tab[tid + 1024] = a.exp_tab[tid + 1024];
    if (tid < 128) { lds_al[0][tid] = a.alphaY[t * 128 + tid]; lds_al[1][tid] = a.alpha1[t * 128 + tid]; }
    s2_stage_candidates<FAM, SM_TC, KA, SM_NW * 64>(a, (int64_t)ctile * SM_TC, lds_c,
                                                    inlined ? (const double *)__builtin_amdgcn_kernarg_segment_ptr() : nullptr);
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
        muY = fma(lds_al[0][kl], kv, muY);
        mu1 = fma(lds_al[1][kl], kv, mu1);
        dst[r * 128] = kv;
    }
    muY += __shfl_xor(muY, 16); muY += __shfl_xor(muY, 32);
    mu1 += __shfl_xor(mu1, 16); mu1 += __shfl_xor(mu1, 32);
    if (lane < 16) { lds_m[0][wave][lane] = muY; lds_m[1][wave][lane] = mu1; }
    SST(0, 2);
    __syncthreads();
    if (tid < 2 * SM_TC) {
        const int which = tid >> 5, c = tid & 31;
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < SM_NW / 2; w++) s += lds_m[which][2 * w + (c >> 4)][c & 15];
        mupart[(size_t)(t * 2 + which) * Mp + ctile * SM_TC + c] = s;
    }
    SST(0, 3);
}

// grid (ctiles * 2 / CBS, Npad / 16); qpart[g Mp + c] = sum over the 16 rows of row-block g of (W K*)^2
// CBS = 2: a workgroup takes both 16-candidate blocks of its tile (batches of many tiles: W is read once per tile).
// CBS = 1: one block per workgroup -- for the few-tile batches of DIRECT, whose time is the LAST row-block's workgroup pulling its
// 2g + 2 steps x (1 KiB of W + 2 KiB of k*) through one CU (tools/stamp_small.py): half the k* bytes per CU, twice the workgroups
// on a chip that was half empty; a block that is all padding leaves at once.  Every candidate's sums are the same terms in the
// same order either way: identical bits.
template <int CBS>
__global__ __launch_bounds__(SM_NW * 64) void wk_small_kernel(SweepArgs a, const double *__restrict__ Kf, double *__restrict__ qpart, int Mp)
{
    __shared__ double lds_v[SM_NW][CBS][256];        // partial V tiles: [wave][cand-block][lane 64 x 4]
    __shared__ double lds_s[16 * CBS][17];           // squared sums [cand][row]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ctile = CBS == 2 ? blockIdx.x : blockIdx.x >> 1, cb0 = CBS == 2 ? 0 : blockIdx.x & 1;
    if (CBS == 1 && (int64_t)ctile * SM_TC + cb0 * 16 >= a.M) return;
    const int g = gridDim.y - 1 - blockIdx.y;        // the longest rows of W first: the short ones fill the tail
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
            for (int c = 0; c < CBS; c++) { o.b0[c] = Kb[(size_t)(2 * j) * 128 + 64 * c]; o.b1[c] = Kb[(size_t)(2 * j + 1) * 128 + 64 * c]; }
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
        qpart[(size_t)g * Mp + ctile * SM_TC + cb0 * 16 + tid] = q;
    }
    SST(1, 3);
}

// WAVE-LOCAL k*: the exponent GEMM's output layout (lane l: row (l>>4) + 4 r, candidate l & 15) IS the B-fragment layout of the
// product that follows, so a wave can make the k* of a 16-row tile and multiply it with its W fragments without anything
// leaving its registers -- no LDS stage, no barrier, no trip of k* through memory.  Workgroup (tile, row-block g): wave w takes
// the 16-row tiles rt = w, w + 16, .. <= g of both candidate blocks; per tile 2 (KA4 + 4) MFMAs and 8 exp() per lane, the next
// tile's fragments of X and W requested while this one is computed.  k* is regenerated by every row-block (N / 32 times on
// average) but a DIRECT batch is bound by its launches: one launch instead of two, ~4 us of kernel instead of 6 + 9 at N = 1024.
// The last row-block's workgroups see every row and also form the two mean dot products (one part per tile: nst = 1 for the
// finish kernel).  Sums run in a fixed order (tiles ascending per wave, waves ascending): deterministic.
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
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ctile = blockIdx.x, g = gridDim.y - 1 - blockIdx.y;       // the longest rows of W first
    const int nk8 = a.Npad >> 3;
    const bool means = g == (int)gridDim.y - 1;
    lds_tab[tid] = a.exp_tab[tid];
    lds_tab[tid + 1024] = a.exp_tab[tid + 1024];
    s2_stage_candidates<FAM, SM_TC, KA, SM_NW * 64>(a, (int64_t)ctile * SM_TC, lds_c,
                                                    inlined ? (const double *)__builtin_amdgcn_kernarg_segment_ptr() : nullptr);
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
    for (int r = 0; r < 4; r++) { lds_v[wave][0][lane * 4 + r] = acc0[r]; lds_v[wave][1][lane * 4 + r] = acc1[r]; }
    if (means) {
        mY0 += __shfl_xor(mY0, 16); mY0 += __shfl_xor(mY0, 32); mY1 += __shfl_xor(mY1, 16); mY1 += __shfl_xor(mY1, 32);
        m10 += __shfl_xor(m10, 16); m10 += __shfl_xor(m10, 32); m11 += __shfl_xor(m11, 16); m11 += __shfl_xor(m11, 32);
        if (lane < 16) { lds_m[0][wave][0][lane] = mY0; lds_m[0][wave][1][lane] = mY1; lds_m[1][wave][0][lane] = m10; lds_m[1][wave][1][lane] = m11; }
    }
    __syncthreads();
    if (tid < 512) {
        // element e of candidate block cb: lane l = e >> 2, r = e & 3 -> row (l >> 4) + 4 r, candidate 16 cb + (l & 15)
        const int cb = tid >> 8, e = tid & 255, l = e >> 2, r = e & 3;
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < SM_NW; w++) v += lds_v[w][cb][e];
        lds_s[16 * cb + (l & 15)][(l >> 4) + 4 * r] = v * v;
    } else if (means && tid < 512 + 2 * SM_TC) {
        const int which = (tid - 512) >> 5, c = (tid - 512) & 31;
        double sm = 0.0;
#pragma unroll
        for (int w = 0; w < SM_NW; w++) sm += lds_m[which][w][c >> 4][c & 15];
        mupart[(size_t)which * Mp + ctile * SM_TC + c] = sm;
    }
    __syncthreads();
    if (tid < SM_TC) {
        double q = 0.0;
#pragma unroll
        for (int rr = 0; rr < 16; rr++) q += lds_s[tid][rr];
        qpart[(size_t)g * Mp + ctile * SM_TC + tid] = q;
    }
}

// 64 candidates per workgroup of 512 threads: eight threads per candidate each sum a contiguous eighth of the row-blocks'
// q (all their loads in flight at once: one L2 round trip instead of nrb / 8 dependent ones -- the kernel was 7 of a DIRECT
// batch's 22 us), the stages' mean parts likewise; the eight partial sums are added in index order by the candidate's first
// thread, which then evaluates the acquisition.  Every sum has a fixed order: the result does not depend on scheduling.
// (512 threads, not 1024: the acquisition's erf / exp chains want more than the 128 registers a 1024-thread workgroup leaves a lane.)
#define SM_FIN_P 8
__global__ __launch_bounds__(SM_FIN_P * 64) void small_finish_kernel(SweepArgs a, const double *__restrict__ qpart, const double *__restrict__ mupart, int Mp,
                                                                     int nrb, int nst)
{
    __shared__ double lds_q[SM_FIN_P][64], lds_y[SM_FIN_P][64], lds_1[SM_FIN_P][64];
    const int lane = threadIdx.x & 63, p = threadIdx.x >> 6;
    const int64_t li = (int64_t)blockIdx.x * 64 + lane;
    const bool valid = li < a.M;
    const int64_t ci = valid ? li : a.M - 1;
    SST(2, 0);
    {
        const int per = (nrb + SM_FIN_P - 1) / SM_FIN_P, g0 = p * per;
        double v[16];
        double qs = 0.0;
        for (int gb = 0; gb < per; gb += 16) {           // (per <= 16 up to N = 2048: one round)
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = (gb + u < per && g0 + gb + u < nrb) ? qpart[(size_t)(g0 + gb + u) * Mp + ci] : 0.0;
#pragma unroll
            for (int u = 0; u < 16; u++) if (gb + u < per && g0 + gb + u < nrb) qs += v[u];
        }
        lds_q[p][lane] = qs;
        const int pers = (nst + SM_FIN_P - 1) / SM_FIN_P, t0 = p * pers;
        double ys = 0.0, os = 0.0;
        for (int t = t0; t < t0 + pers && t < nst; t++) { ys += mupart[(size_t)(2 * t) * Mp + ci]; os += mupart[(size_t)(2 * t + 1) * Mp + ci]; }
        lds_y[p][lane] = ys; lds_1[p][lane] = os;
    }
    __syncthreads();
    if (p != 0) return;
    SST(2, 1);
    double q = 0.0, my = 0.0, m1 = 0.0;
#pragma unroll
    for (int u = 0; u < SM_FIN_P; u++) { q += lds_q[u][lane]; my += lds_y[u][lane]; m1 += lds_1[u][lane]; }
    bool excl;
    double val = s2_finish(a, a.cand + ci * a.kp.D, q, my, m1, li, valid, excl);
    int64_t idx = a.index_base + li;
    if (!valid || excl || !(val == val)) { val = -INFINITY; idx = INT64_MAX; }
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(val, o);
        const int64_t oi = __shfl_xor(idx, o);
        if (ov > val || (ov == val && oi < idx)) { val = ov; idx = oi; }
    }
    if (lane == 0) { a.part_val[blockIdx.x] = val; a.part_idx[blockIdx.x] = idx; }
    SST(2, 2);
    if (a.done_flag) {
        // every lane's results (possibly in host memory) are out before this workgroup takes its ticket; the last ticket
        // publishes the sequence number the host is waiting for
        __threadfence_system();
        if (gridDim.x == 1) {                            // at most 64 candidates: this wave IS the batch -- no ticket to take
            if (lane == 0) *(volatile unsigned long long *)a.done_flag = a.done_seq;
        } else if (lane == 0) {
            if (atomicAdd(a.done_count, 1u) == gridDim.x - 1) {
                *a.done_count = 0;
                __threadfence_system();
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
        if (ctiles <= 8) hipLaunchKernelGGL(wk_small_kernel<1>      // (one 16-candidate block per workgroup: twice the workgroups on a half-empty chip)
           , dim3(2 * ctiles, nrb), dim3(SM_NW * 64), 0, s, a, Kf, qpart, Mp);
        else hipLaunchKernelGGL(wk_small_kernel<2>, dim3(ctiles, nrb), dim3(SM_NW * 64), 0, s, a, Kf, qpart, Mp);
    }
    if (e1) (void)hipEventRecord(e1, s);
    const int64_t nfin = (a.M + 63) / 64;
    hipLaunchKernelGGL(small_finish_kernel, dim3((unsigned)nfin), dim3(SM_FIN_P * 64), 0, s, a, qpart, mupart, Mp, nrb, nst_fin);
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
