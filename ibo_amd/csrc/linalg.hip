// linalg.hip -- the factorisation side of a fit on gfx950: blocked Cholesky (NB = 64, fp64 MFMA tiles), W = L^-1 riding along or
// by recursive doubling, K^-1 = W^T W.
//
// Replaces (reference, /root/reference):
//   linalg.cholesky(R)                     ego/gaussianprocess/__init__.py:299
//   linalg.inv(R) per maximize* call       ego/acquisition/__init__.py:385-388
//   linalg.solve / inv of the likelihood   ego/gaussianprocess/trainhyper.py:60-79
//
// Layout: every N x N matrix lives row-major with leading dimension Npad (a multiple of 64) so that all tile kernels run
// without bounds checks; the pad is the identity for matrices that get factored and zero for W.
// Orders (the order fixes the rounding, so it depends on the matrix size and the entry point, never on a batch size):
//   * single matrix, < 104 block columns: plain right-looking, one launch per block column, out of place, W riding along
//     (launch_cholesky_fused: chol_step8_kernel below four block columns, chol_pipe8_kernel -- software-pipelined -- from there);
//   * single matrix beyond: two-level (panels of four block columns, K = 256 updates), then launch_trinv (launch_cholesky_fused2);
//   * in place, any batch (the legacy inverse, ibo_spd_*, preference GPs beyond the first range): launch_cholesky_batched;
//   * the likelihood grid: left-looking from a packed copy of the factor (launch_cholesky_batched_left, update3.hip).
// Covariance assembly, packing, the alpha vectors, the gradient contraction and the small per-model kernels: assemble.hip.
#include "ibo_common.h"
#include <atomic>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

// ------------------------------------------------------------------------
// 64x64 output tiles on fp64 MFMA, 256 threads (2x2 waves, 32x32 each)
// ------------------------------------------------------------------------
// K = 64 in one stage (the factorisation's trsm / syrk tiles): both 64x64 operand tiles are fetched with
// every load in flight at once -- one global-memory latency instead of four.  acc += A * B^T.
// As, Bs: 64 * T64_LD doubles each.  The row stride is ODD: an MFMA fragment read takes, per 16-lane pass, the rows
// r = 0..15 at one k -- 8-byte words r * LD + k, which fall on 16 distinct bank pairs only if LD is odd (LD = 68, the
// first choice, put them on 4: a four-way conflict on every fragment read; fit 0.357 -> 0.345 ms at N = 1024, the
// C5 grid 45.3 -> 43.1 ms).  The price is scalar instead of 16-byte stores when a tile is stashed.
#define T64_LD 65
typedef double d2_t __attribute__((ext_vector_type(2)));   // (HIP's double2 struct arrays end up in scratch here)
__device__ __forceinline__ void tile64_fetch(const double *__restrict__ A, int lda, d2_t (&v)[8])
{
    const int t = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = *(const d2_t *)(A + (size_t)(8 * u + (t >> 5)) * lda + (t & 31) * 2);
}
template <bool NEG = false, int LD = T64_LD>
__device__ __forceinline__ void tile64_stash(double *As, const d2_t (&v)[8])
{
    const int t = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 8; u++) {
        double *dst = As + (8 * u + (t >> 5)) * LD + (t & 31) * 2;
        if (LD & 1) {                               // odd row stride: rows are 8-byte aligned only
            dst[0] = NEG ? -v[u].x : v[u].x;
            dst[1] = NEG ? -v[u].y : v[u].y;
        } else *(d2_t *)dst = NEG ? -v[u] : v[u];
    }
}
template <int LD = T64_LD>
__device__ __forceinline__ void tile64_mma_nt(const double *As, const double *Bs, d4_t (&acc)[2][2])
{
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wr = wv >> 1, wc = wv & 1;
#pragma unroll
    for (int k4 = 0; k4 < 16; k4++) {
        double a[2], b[2];
#pragma unroll
        for (int m = 0; m < 2; m++) a[m] = As[(wr * 32 + m * 16 + (lane & 15)) * LD + k4 * 4 + (lane >> 4)];
#pragma unroll
        for (int n = 0; n < 2; n++) b[n] = Bs[(wc * 32 + n * 16 + (lane & 15)) * LD + k4 * 4 + (lane >> 4)];
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 2; n++) acc[m][n] = mfma_f64(a[m], b[n], acc[m][n]);
    }
}

// The same product when B (64x64, row-major in Bs) is LOWER TRIANGULAR -- the inverse of a diagonal factor: column
// block cb of the result needs only k < 16 (cb + 1); the skipped terms are exact zeros, so the result equals the full
// product's.  The waves' column blocks are (0, 3) and (1, 2) instead of (0, 1) and (2, 3): 40 MFMAs per wave either
// way instead of 64.  Accumulator (m, n) is rows wr*32 + m*16.., columns 16 TRI_CB(n)...
#define TRI_CB(n) (wc ? ((n) ? 2 : 1) : ((n) ? 3 : 0))
template <int LD, int CB0, int CB1>
__device__ __forceinline__ void tile64_mma_nt_tri_body(const double *As, const double *Bs, d4_t (&acc)[2][2], int wr, int lane)
{
    // all fragments first (one LDS latency), then the MFMAs
    double a[16][2], b0[4 * (CB0 + 1)], b1[4 * (CB1 + 1)];
#pragma unroll
    for (int k4 = 0; k4 < 4 * (CB1 + 1); k4++) {
#pragma unroll
        for (int m = 0; m < 2; m++) a[k4][m] = As[(wr * 32 + m * 16 + (lane & 15)) * LD + k4 * 4 + (lane >> 4)];
        if (k4 < 4 * (CB0 + 1)) b0[k4] = Bs[(CB0 * 16 + (lane & 15)) * LD + k4 * 4 + (lane >> 4)];
        b1[k4] = Bs[(CB1 * 16 + (lane & 15)) * LD + k4 * 4 + (lane >> 4)];
    }
#pragma unroll
    for (int k4 = 0; k4 < 4 * (CB1 + 1); k4++) {
        if (k4 < 4 * (CB0 + 1)) {
#pragma unroll
            for (int m = 0; m < 2; m++) acc[m][0] = mfma_f64(a[k4][m], b0[k4], acc[m][0]);
        }
#pragma unroll
        for (int m = 0; m < 2; m++) acc[m][1] = mfma_f64(a[k4][m], b1[k4], acc[m][1]);
    }
}
template <int LD = T64_LD>
__device__ __forceinline__ void tile64_mma_nt_tri(const double *As, const double *Bs, d4_t (&acc)[2][2])
{
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wr = wv >> 1, wc = wv & 1;
    if (wc) tile64_mma_nt_tri_body<LD, 1, 2>(As, Bs, acc, wr, lane);
    else tile64_mma_nt_tri_body<LD, 0, 3>(As, Bs, acc, wr, lane);
}
#define TILE_COL_TRI(n) (TRI_CB(n) * 16 + (lane & 15))

// C[row][col] for accumulator element (m, n, r) of this lane
#define TILE_ROW(m, r) (wr * 32 + (m) * 16 + (lane >> 4) + 4 * (r))
#define TILE_COL(n) (wc * 32 + (n) * 16 + (lane & 15))
#define TILE_IDS const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wr = wv >> 1, wc = wv & 1

// ------------------------------------------------------------------------
// blocked right-looking Cholesky, NB = 64
// ------------------------------------------------------------------------
#define SD 65
__device__ __forceinline__ double lane_bcast(double x, int l)          // value of lane l, wave-uniform
{
    int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

// one wave: 16x16 (+)= A(16xK) * B(Kx16) with both operands in LDS (row stride SD).
// A element (r,k) = Am[r*SD + k];  B element (k,c) = TB ? Bm[c*SD + k] : Bm[k*SD + c].
// Result element r of the returned vector is row (lane>>4)+4r, column lane&15.
template <bool TB, int K, int LDA = SD, int LDB = SD>
__device__ __forceinline__ d4_t lds_mm16(const double *Am, const double *Bm)
{
    const int lane = threadIdx.x & 63, ar = lane & 15, q = lane >> 4;
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    double a[K / 4], b[K / 4];
#pragma unroll
    for (int s = 0; s < K / 4; s++) {               // all LDS reads in flight before the first MFMA
        a[s] = Am[ar * LDA + 4 * s + q];
        b[s] = TB ? Bm[ar * LDB + 4 * s + q] : Bm[(4 * s + q) * LDB + ar];
    }
#pragma unroll
    for (int s = 0; s < K / 4; s++) acc = mfma_f64(a[s], b[s], acc);
    return acc;
}
#define MM16_ROW(r) ((lane >> 4) + 4 * (r))
#define MM16_COL (lane & 15)

// Factor the 64x64 diagonal block and invert the factor inside one workgroup, blocked by 16.
// Everything here is instruction-issue bound on a single wave (~8 cycles per VALU instruction), so the
// sequential chain is kept as short as it can be:
//   per 16-column panel: (i) wave 0 factors the whole (64-o) x 16 panel with lane = row, the 16 column
//   values of a row in registers, the pivot row's values broadcast with v_readlane, fully unrolled;
//   1/sqrt(pivot) from v_rsq_f64 + one third-order correction (no fp64 divide / sqrt sequences);
//   (iii) the trailing sub-matrix gets its rank-16 update as 16x16 fp64-MFMA tiles out of LDS.
//   The four 16x16 diagonal factors are inverted afterwards, one per wave, in right-looking order
//   (independent updates instead of a dependent dot product), and the 64x64 inverse is assembled
//   from them by recursive doubling (MFMA).
// This kernel is the sequential chain of the whole factorisation (N/64 launches).
__device__ __forceinline__ double rcp_newton(double d)
{
    double y = __builtin_amdgcn_rcp(d);
    y = fma(y, fma(-d, y, 1.0), y);
    return fma(y, fma(-d, y, 1.0), y);
}

// the block's loads are issued by diag64_fetch and land in LDS by diag64_stash: a caller with more to request puts
// its other loads between the two (loads return in order, so the block is waited for alone)
__device__ __forceinline__ void diag64_fetch(const double *Lb, int Npad, double (&v)[16])
{
    const int t = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = Lb[(size_t)(4 * u + (t >> 6)) * Npad + (t & 63)];
}
template <int TD = SD>
__device__ __forceinline__ void diag64_stash(const double (&v)[16], double *S, double *V, double *T)
{
    const int t = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 16; u++) {
        S[(4 * u + (t >> 6)) * SD + (t & 63)] = v[u];
        V[(4 * u + (t >> 6)) * SD + (t & 63)] = 0.0;
    }
    T[(t >> 4) * TD + (t & 15)] = ((t >> 4) == (t & 15)) ? 1.0 : 0.0;       // diag64_dpp.h: the identity rows
}
template <int TD = SD>
__device__ __forceinline__ void diag64_load(const double *Lb, int Npad, double *S, double *V, double *T)
{
    double v[16];
    diag64_fetch(Lb, Npad, v);
    diag64_stash<TD>(v, S, V, T);
}

__device__ __forceinline__ void diag64_store(double *Lb, int Npad, double *Db, const double *S, const double *V)
{
    const int t = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 16; u++) {
        const int r = 4 * u + (t >> 6), c = t & 63;
        Lb[(size_t)r * Npad + c] = (c <= r) ? S[r * SD + c] : 0.0;
        Db[r * 64 + c] = V[r * SD + c];
    }
}

#include "diag64_dpp.h"

// Row stride of the chain's scratch array where nothing else uses it (chol_diag_kernel, chol_panel_diag_kernel): with 64 x 65 doubles each
// for S, V and T a workgroup needs 99 840 bytes of LDS, 1 536 more than fit beside a 65 536-byte workgroup of chol_update3_kernel on one CU
// (163 840): in the likelihood grid these one-workgroup-per-matrix kernels -- the sequential part of a panel -- then wait for a CU to drain
// COMPLETELY while the other sub-batch's update keeps backfilling it (measured: 0.8 .. 1.9 ms for a 90 us launch, profiles/r05_c5_timeline_before.txt).
#define CHAIN_TD 49
__global__ __launch_bounds__(256) void chol_diag_kernel(double *L, int Npad, int jb,
                                                        double *__restrict__ diag64, int *info,
                                                        size_t lstride, size_t dstride, double *Lout)
{
    __shared__ double S[64 * SD];          // the block; ends up holding L (lower)
    __shared__ double V[64 * SD];          // its inverse
    __shared__ double T[64 * CHAIN_TD];    // scratch (L21 * V11 products): 64 x 48 are used -- 91.6 KB in all, see CHAIN_TD
    L += blockIdx.z * lstride; diag64 += blockIdx.z * dstride; info += blockIdx.z;      // batch member
    if (!Lout) Lout = L; else Lout += blockIdx.z * lstride;
    const size_t off = (size_t)jb * 64 * Npad + jb * 64;
    diag64_load<CHAIN_TD>(L + off, Npad, S, V, T);
    __syncthreads();
    diag64_factor_invert<CHAIN_TD>(S, V, T, jb * 64, info);
    diag64_store(Lout + off, Npad, diag64 + (size_t)jb * 4096, S, V);
}

// ONE launch per block column for the plain right-looking order on few block columns (a fit of < 256 rows; the in-panel columns of the
// two-level order), OUT OF PLACE: the matrix being reduced (L) is only read in its panel column and updated in its trailing tiles (each
// by exactly one workgroup), the factor goes to a second matrix (Lout) -- overwriting the panel in place would race with the workgroups
// that still have to read it.  Every workgroup of the trailing update first repeats the 64 x 64 factorisation of the diagonal block (the
// chain costs the same on 120 workgroups as on one), then forms the two row blocks of L it needs (A_i V^T, A_k V^T) and updates its tile;
// workgroup 0 stores the diagonal block and its inverse, the workgroups of the first trailing column their row block of L.
// W = L^-1 RIDES ALONG: with E = I appended below the matrix, the factorisation's own "row block times inv(L_jj)^T, then update the
// trailing tiles" turns E into (L^-1)^T block column by block column (L21 L11^T = I).  Tile (i, k) of E is touched by step jb only for
// i <= jb < k, so a step carries (jb + 1)(nb - jb - 1) extra tiles: tiles >= nchol, number e -> (i = e / m, k = jb + 1 + e % m), A_i from
// Ework, X_i to Eout.  A workgroup takes the tiles blockIdx.x, + gridDim.x, ..: a second tile reuses the inverse in LDS, its operands
// requested before the first tile's products start.
// Eight waves: the chain is one wave's work whatever the workgroup's size, but a lone wave per SIMD issues an fp64 MFMA every 100-139
// cycles where the pipe takes one per 64 (tools/mfma_f64_peak); wave w owns the 16 x 32 strip (row block w >> 1, column half w & 1) of
// every product, each output element the same chain of MFMAs over ascending k whatever the wave count.
#define S8_ROW(r) (wr8 * 16 + (lane >> 4) + 4 * (r))
#define S8_COL(n) (wc8 * 32 + (n) * 16 + (lane & 15))
#define S8_CB(n) (wc8 ? ((n) ? 2 : 1) : ((n) ? 3 : 0))
#define S8_COL_TRI(n) (S8_CB(n) * 16 + (lane & 15))
__device__ __forceinline__ void s8_fetch(const double *__restrict__ A, int lda, d2_t (&v)[4])
{
    const int t = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = *(const d2_t *)(A + (size_t)(16 * u + (t >> 5)) * lda + (t & 31) * 2);
}
__device__ __forceinline__ void s8_stash(double *As, const d2_t (&v)[4])
{
    const int t = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        double *dst = As + (16 * u + (t >> 5)) * SD + (t & 31) * 2;
        dst[0] = v[u].x; dst[1] = v[u].y;
    }
}
// acc[n] += A(strip wr8) B^T(column block 32 wc8 + 16 n), K = 64, k ascending
__device__ __forceinline__ void s8_mma_nt(const double *As, const double *Bs, d4_t (&acc)[2], int wr8, int wc8, int lane)
{
#pragma unroll
    for (int k4 = 0; k4 < 16; k4++) {
        const double a = As[(wr8 * 16 + (lane & 15)) * SD + k4 * 4 + (lane >> 4)];
        double b[2];
#pragma unroll
        for (int n = 0; n < 2; n++) b[n] = Bs[(wc8 * 32 + n * 16 + (lane & 15)) * SD + k4 * 4 + (lane >> 4)];
#pragma unroll
        for (int n = 0; n < 2; n++) acc[n] = mfma_f64(a, b[n], acc[n]);
    }
}
// two products with the same lower-triangular B (tile64_mma_nt_tri2's arithmetic): column block cb needs k < 16 (cb + 1)
template <int CB0, int CB1>
__device__ __forceinline__ void s8_mma_nt_tri2_body(const double *As1, const double *As2, const double *Bs, d4_t (&acc1)[2], d4_t (&acc2)[2],
                                                    int wr8, int lane)
{
    double a1[4 * (CB1 + 1)], a2[4 * (CB1 + 1)], b0[4 * (CB0 + 1)], b1[4 * (CB1 + 1)];
#pragma unroll
    for (int k4 = 0; k4 < 4 * (CB1 + 1); k4++) {
        a1[k4] = As1[(wr8 * 16 + (lane & 15)) * SD + k4 * 4 + (lane >> 4)];
        a2[k4] = As2[(wr8 * 16 + (lane & 15)) * SD + k4 * 4 + (lane >> 4)];
        if (k4 < 4 * (CB0 + 1)) b0[k4] = Bs[(CB0 * 16 + (lane & 15)) * SD + k4 * 4 + (lane >> 4)];
        b1[k4] = Bs[(CB1 * 16 + (lane & 15)) * SD + k4 * 4 + (lane >> 4)];
    }
#pragma unroll
    for (int k4 = 0; k4 < 4 * (CB1 + 1); k4++) {
        if (k4 < 4 * (CB0 + 1)) { acc1[0] = mfma_f64(a1[k4], b0[k4], acc1[0]); acc2[0] = mfma_f64(a2[k4], b0[k4], acc2[0]); }
        acc1[1] = mfma_f64(a1[k4], b1[k4], acc1[1]); acc2[1] = mfma_f64(a2[k4], b1[k4], acc2[1]);
    }
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void chol_step8_kernel(double *__restrict__ L, double *__restrict__ Lout, int Npad, int jb,
                       double *__restrict__ diag64, int *info, int nchol, double *__restrict__ Ework,
                       double *__restrict__ Eout, int ntiles)
{
    __shared__ double S[64 * SD];
    __shared__ double V[64 * SD];
    __shared__ double T[64 * SD];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wr8 = wv >> 1, wc8 = wv & 1;
    const int nb = Npad / 64, m = nb - jb - 1;
    struct Tile { int k; const double *Ai, *Ak; double *Xi, *C; };
    auto decode = [&](int t) {
        Tile q;
        int i;
        if (t < nchol) {
            q.k = jb + 1;
            int rem = t;
            while (rem >= nb - q.k) { rem -= nb - q.k; q.k++; }
            i = q.k + rem;
            q.Ai = L + (size_t)i * 64 * Npad + jb * 64;
            q.Xi = Lout + (size_t)i * 64 * Npad + jb * 64;
            q.C = L + (size_t)i * 64 * Npad + q.k * 64;
        } else {
            const int e = t - nchol;
            i = e / m; q.k = jb + 1 + e % m;
            q.Ai = Ework + (size_t)i * 64 * Npad + jb * 64;
            q.Xi = Eout + (size_t)i * 64 * Npad + jb * 64;
            q.C = Ework + (size_t)i * 64 * Npad + q.k * 64;
        }
        q.Ak = L + (size_t)q.k * 64 * Npad + jb * 64;
        return q;
    };
    auto fetch = [&](const Tile &q, d2_t (&va)[4], d2_t (&vb)[4], d4_t (&c)[2]) {
        s8_fetch(q.Ai, Npad, va);
        s8_fetch(q.Ak, Npad, vb);
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) c[n][r] = q.C[(size_t)S8_ROW(r) * Npad + S8_COL(n)];
    };
    const size_t doff = (size_t)jb * 64 * Npad + jb * 64;
    int t = blockIdx.x;
    Tile cur = decode(t);
    // the diagonal block first (the chain starts when it is back), then this tile's operands
    double vd[8];
    {
        const int tt = threadIdx.x;
#pragma unroll
        for (int u = 0; u < 8; u++) vd[u] = L[doff + (size_t)(8 * u + (tt >> 6)) * Npad + (tt & 63)];
    }
    d2_t va[4], vb[4];
    d4_t c[2];
    fetch(cur, va, vb, c);
    {
        const int tt = threadIdx.x;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            S[(8 * u + (tt >> 6)) * SD + (tt & 63)] = vd[u];
            V[(8 * u + (tt >> 6)) * SD + (tt & 63)] = 0.0;
        }
        if (tt < 256) T[(tt >> 4) * SD + (tt & 15)] = ((tt >> 4) == (tt & 15)) ? 1.0 : 0.0;
    }
    __syncthreads();
    diag64_factor_invert(S, V, T, jb * 64, blockIdx.x == 0 ? info : nullptr);
    if (blockIdx.x == 0) {
        const int tt = threadIdx.x;
        double *Lb = Lout + doff, *Db = diag64 + (size_t)jb * 4096;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int r = 8 * u + (tt >> 6), cc = tt & 63;
            Lb[(size_t)r * Npad + cc] = (cc <= r) ? S[r * SD + cc] : 0.0;
            Db[r * 64 + cc] = V[r * SD + cc];
        }
    }
    for (;;) {
        __syncthreads();                                   // S (and T) are about to be reused
        s8_stash(S, va);
        s8_stash(T, vb);
        const int tn = t + gridDim.x;
        const bool more = tn < ntiles;
        Tile nxt = cur;
        d4_t cn[2];
        if (more) {
            nxt = decode(tn);
            fetch(nxt, va, vb, cn);
        }
        __syncthreads();
        d4_t xi[2] = {}, xk[2] = {};
        if (wc8) s8_mma_nt_tri2_body<1, 2>(S, T, V, xi, xk, wr8, lane);
        else s8_mma_nt_tri2_body<0, 3>(S, T, V, xi, xk, wr8, lane);
        __syncthreads();
        if (cur.k == jb + 1) {                             // first trailing column: this row block of L is final
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) cur.Xi[(size_t)S8_ROW(r) * Npad + S8_COL_TRI(n)] = xi[n][r];
        }
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                S[S8_ROW(r) * SD + S8_COL_TRI(n)] = -xi[n][r];
                T[S8_ROW(r) * SD + S8_COL_TRI(n)] = xk[n][r];
            }
        __syncthreads();
        s8_mma_nt(S, T, c, wr8, wc8, lane);
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) cur.C[(size_t)S8_ROW(r) * Npad + S8_COL(n)] = c[n][r];
        if (!more) break;
        t = tn; cur = nxt;
#pragma unroll
        for (int n = 0; n < 2; n++) c[n] = cn[n];
    }
}

// Diagonal block + row blocks of one block column in ONE launch: every row block's workgroup repeats the diagonal
// factorisation -- as chol_step_kernel does -- and multiplies its block by inv(L_jj)^T; workgroup 0 also stores the diagonal
// block and its inverse.  chol_diag_kernel + chol_trsm_kernel, same arithmetic.  Workgroups [0, m) take the matrix's row
// blocks jb + 1 .., workgroups m .. the row blocks 0 .. of the ride-along's E (chol_step_kernel's comment), if any.
// Used where a block column has no tile to update (a panel's last column in the two-level order) and where it has too many
// for one tile per workgroup (chol_update_step_kernel then does the updates).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void chol_diag_trsm_kernel(const double *__restrict__ A, double *__restrict__ Lout, int Npad, int jb, double *__restrict__ diag64, int *info,
                           int m, const double *__restrict__ Ework, double *__restrict__ Eout)
{
    __shared__ double S[64 * SD];
    __shared__ double V[64 * SD];
    __shared__ double T[64 * SD];
    TILE_IDS;
    const size_t doff = (size_t)jb * 64 * Npad + jb * 64;
    const bool erow = (int)blockIdx.x >= m;
    const int ib = erow ? (int)blockIdx.x - m : jb + 1 + (int)blockIdx.x;
    const size_t roff = (size_t)ib * 64 * Npad + jb * 64;
    double vd[16];
    diag64_fetch(A + doff, Npad, vd);
    d2_t va[8];
    tile64_fetch((erow ? Ework : A) + roff, Npad, va);
    diag64_stash(vd, S, V, T);
    __syncthreads();
    diag64_factor_invert(S, V, T, jb * 64, blockIdx.x == 0 ? info : nullptr);
    if (blockIdx.x == 0) diag64_store(Lout + doff, Npad, diag64 + (size_t)jb * 4096, S, V);
    __syncthreads();
    tile64_stash<false, SD>(S, va);
    __syncthreads();
    d4_t acc[2][2] = {};
    tile64_mma_nt_tri<SD>(S, V, acc);
    double *Ob = (erow ? Eout : Lout) + roff;
#pragma unroll
    for (int mm = 0; mm < 2; mm++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) Ob[(size_t)TILE_ROW(mm, r) * Npad + TILE_COL_TRI(n)] = acc[mm][n][r];
}

// The updates of a fused step on their own: tile numbering as in chol_step_kernel (t < nchol: tile (i, k) of the matrix,
// jb < k <= i; then tile (i, k) of E, i <= jb < k), C -= X_i X_k^T with the row blocks X = (row block) inv(L_jj)^T that
// chol_diag_trsm_kernel has stored in Lout / Eout.  X_i is negated on its way into LDS and the accumulators start as C:
// the arithmetic of chol_step_kernel's last product.  Two workgroups per CU.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void chol_update_step_kernel(double *__restrict__ A, const double *__restrict__ Lout, int Npad, int jb, int nchol,
                             double *__restrict__ Ework, const double *__restrict__ Eout)
{
    __shared__ double As[64 * T64_LD];
    __shared__ double Bs[64 * T64_LD];
    TILE_IDS;
    const int nb = Npad / 64, m = nb - jb - 1, t = blockIdx.x;
    int i, k;
    const double *Xi;
    double *C;
    if (t < nchol) {
        k = jb + 1;
        int rem = t;
        while (rem >= nb - k) { rem -= nb - k; k++; }
        i = k + rem;
        Xi = Lout + (size_t)i * 64 * Npad + jb * 64;
        C = A + (size_t)i * 64 * Npad + k * 64;
    } else {
        const int e = t - nchol;
        i = e / m; k = jb + 1 + e % m;
        Xi = Eout + (size_t)i * 64 * Npad + jb * 64;
        C = Ework + (size_t)i * 64 * Npad + k * 64;
    }
    const double *Xk = Lout + (size_t)k * 64 * Npad + jb * 64;
    d2_t va[8], vb[8];
    tile64_fetch(Xi, Npad, va);
    tile64_fetch(Xk, Npad, vb);
    d4_t acc[2][2];
#pragma unroll
    for (int mm = 0; mm < 2; mm++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[mm][n][r] = C[(size_t)TILE_ROW(mm, r) * Npad + TILE_COL(n)];
    tile64_stash<true>(As, va);
    tile64_stash(Bs, vb);
    __syncthreads();
    tile64_mma_nt(As, Bs, acc);
#pragma unroll
    for (int mm = 0; mm < 2; mm++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) C[(size_t)TILE_ROW(mm, r) * Npad + TILE_COL(n)] = acc[mm][n][r];
}

// rows below the diagonal block: A[ib][jb] <- A[ib][jb] * inv(L_jj)^T
__global__ __launch_bounds__(256) void chol_trsm_kernel(double *L, int Npad, int jb,
                                                        const double *__restrict__ diag64,
                                                        size_t lstride, size_t dstride, double *Lout, int row0 = -1)
{
    __shared__ double As[64 * T64_LD];
    __shared__ double Bs[64 * T64_LD];
    TILE_IDS;
    L += blockIdx.z * lstride; diag64 += blockIdx.z * dstride;
    if (!Lout) Lout = L; else Lout += blockIdx.z * lstride;
    int ib = (row0 < 0 ? jb + 1 : row0) + blockIdx.x;      // row0: the E rows of the W = L^-1 ride-along start at 0
    const double *Ab = L + (size_t)ib * 64 * Npad + jb * 64;
    double *Ob = Lout + (size_t)ib * 64 * Npad + jb * 64;
    d2_t va[8], vb[8];
    tile64_fetch(Ab, Npad, va);
    tile64_fetch(diag64 + (size_t)jb * 4096, 64, vb);
    tile64_stash(As, va);
    tile64_stash(Bs, vb);
    __syncthreads();
    d4_t acc[2][2] = {};
    tile64_mma_nt_tri(As, Bs, acc);
    // the tile is overwritten in place: it was read completely before the barrier
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) Ob[(size_t)TILE_ROW(m, r) * Npad + TILE_COL_TRI(n)] = acc[m][n][r];
}

// (amdgpu_waves_per_eu(2, 2) on the kernels with a K loop: with (1, 2) hipcc puts the accumulators in AGPRs and
// copies all 32 of them in and out of VGPRs on EVERY loop iteration -- one wasted VALU instruction per MFMA, on the
// pipe the MFMAs need; with the 256-register budget it keeps them in VGPRs and the copies disappear.)
// update with finished block columns [j0, j1):  A[i][k] -= sum_j L[i][j] L[k][j]^T  for the block columns
// k in [k0, k1) and the block rows i >= k.  One 64x64 tile per workgroup; the K loop runs in 64-wide stages
// with the next stage's operands (and, first, the tile itself) in flight.
// Tile order is XCD-aware: workgroups go round-robin to the 8 XCDs, each with its own 4 MiB L2, so
// workgroup b belongs to XCD b % 8 and that XCD's 64 consecutive workgroups are given one 8x8 super-block
// of tiles -- 16 operand strips of 64 x 64(j1-j0) serve 64 tiles out of L2 instead of being re-fetched
// from the Infinity Cache (with the operands also kept out of scratch, K = 256 updates went from 18 to 35 TFLOP/s at N = 4096).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void chol_update_kernel(double *L, int Npad, int j0, int j1,
                                                          int k0, int k1, int nsb, size_t lstride, const double *P, int iend)
{
    __shared__ double As[64 * T64_LD];
    __shared__ double Bs[64 * T64_LD];
    TILE_IDS;
    L += blockIdx.z * lstride;
    if (!P) P = L; else P += blockIdx.z * lstride;          // where the finished block columns live
    const int nb = iend > 0 ? iend : Npad / 64;             // block rows [k, nb) of every block column k
    int i, k;
    if (nsb > 0) {
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        const int sb = (q >> 6) * 8 + xcd, lt = q & 63;
        if (sb >= nsb) return;
        const int nsr = (nb - k0 + 7) / 8;          // super-block rows; (SK, SI >= SK) numbered column by column
        int SK = 0, rem = sb;
        while (rem >= nsr - SK) { rem -= nsr - SK; SK++; }
        k = k0 + 8 * SK + (lt & 7), i = k0 + 8 * (SK + rem) + (lt >> 3);
        if (k >= k1 || i >= nb || i < k) return;
    } else {                                        // few tiles (all resident at once): plain column-by-column numbering
        int rem = blockIdx.x;
        k = k0;
        while (rem >= nb - k) { rem -= nb - k; k++; }
        i = k + rem;
    }
    double *C = L + (size_t)i * 64 * Npad + k * 64;
    const double *Ai = P + (size_t)i * 64 * Npad, *Ak = P + (size_t)k * 64 * Npad;
    d2_t va[8], vb[8];
    tile64_fetch(Ai + j0 * 64, Npad, va);
    tile64_fetch(Ak + j0 * 64, Npad, vb);
    // the accumulators start as the tile itself (fetched alongside the first operands) and the A strip is
    // negated on its way into LDS: acc = C - A B^T with no second copy of the tile in registers
    d4_t acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[m][n][r] = C[(size_t)TILE_ROW(m, r) * Npad + TILE_COL(n)];
    for (int j = j0; j < j1; j++) {
        tile64_stash<true>(As, va);
        tile64_stash(Bs, vb);
        __syncthreads();
        if (j + 1 < j1) {
            tile64_fetch(Ai + (j + 1) * 64, Npad, va);
            tile64_fetch(Ak + (j + 1) * 64, Npad, vb);
        }
        tile64_mma_nt(As, Bs, acc);
        if (j + 1 < j1) __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) C[(size_t)TILE_ROW(m, r) * Npad + TILE_COL(n)] = acc[m][n][r];
}

// The rows below a panel of P <= 4 block columns [p0, pend) whose diagonal (64 P)^2 block is factored already: row block i (one
// workgroup) turns its P blocks A_i,j into the factor's blocks
//     X_i,j = (A_i,j - sum_{j' < j} X_i,j' L_j,j'^T) inv(L_jj)^T,   j = p0 .. pend-1,
// the updates of a block applied in ascending j', each as 16 k4-steps on accumulators that start as the block -- the arithmetic, in
// order, of the sequence "trsm of column j, K = 64 update of the panel's later columns" that it replaces, without that sequence's
// 2 P - 1 launches over all rows and its traffic (every update tile read and wrote 128 KB for half a megaflop): a row block is read
// once and written once, and its operands (6 blocks of the diagonal block, 4 inverses) are shared by all row blocks through L2.
// Pk (optional): the left-looking order's packed copy of the factor (update3.hip) -- the row block's finished columns go there straight
// from LDS, in fragment order, and only block rows >= rm_from are also stored row-major (the likelihood reads nothing else of the rows
// below a panel than its y row).  Eight waves (4 x 2 waves of 16 x 32): two waves per SIMD take turns at the MFMA pipe.
#define PR8_ROW(r) (wr8 * 16 + (lane >> 4) + 4 * (r))
#define PR8_COL(n) (wc8 * 32 + (n) * 16 + (lane & 15))
#define PR8_COL_TRI(n) ((wc8 ? ((n) ? 2 : 1) : ((n) ? 3 : 0)) * 16 + (lane & 15))
__device__ __forceinline__ void pr8_fetch(const double *__restrict__ A, int lda, d2_t (&v)[4])
{
    const int t = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = *(const d2_t *)(A + (size_t)(16 * u + (t >> 5)) * lda + (t & 31) * 2);
}
__device__ __forceinline__ void pr8_stash(double *As, const d2_t (&v)[4])
{
    const int t = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        double *dst = As + (16 * u + (t >> 5)) * SD + (t & 31) * 2;
        dst[0] = v[u].x; dst[1] = v[u].y;
    }
}
__device__ __forceinline__ void pr8_mma_nt(const double *As, const double *Bs, d4_t (&acc)[2], int wr8, int wc8, int lane)
{
#pragma unroll
    for (int k4 = 0; k4 < 16; k4++) {
        const double a = As[(wr8 * 16 + (lane & 15)) * SD + k4 * 4 + (lane >> 4)];
        double b[2];
#pragma unroll
        for (int n = 0; n < 2; n++) b[n] = Bs[(wc8 * 32 + n * 16 + (lane & 15)) * SD + k4 * 4 + (lane >> 4)];
#pragma unroll
        for (int n = 0; n < 2; n++) acc[n] = mfma_f64(a, b[n], acc[n]);
    }
}
template <int CB0, int CB1>
__device__ __forceinline__ void pr8_mma_nt_tri_body(const double *As, const double *Bs, d4_t (&acc)[2], int wr8, int lane)
{
    double a[4 * (CB1 + 1)], b0[4 * (CB0 + 1)], b1[4 * (CB1 + 1)];
#pragma unroll
    for (int k4 = 0; k4 < 4 * (CB1 + 1); k4++) {
        a[k4] = As[(wr8 * 16 + (lane & 15)) * SD + k4 * 4 + (lane >> 4)];
        if (k4 < 4 * (CB0 + 1)) b0[k4] = Bs[(CB0 * 16 + (lane & 15)) * SD + k4 * 4 + (lane >> 4)];
        b1[k4] = Bs[(CB1 * 16 + (lane & 15)) * SD + k4 * 4 + (lane >> 4)];
    }
#pragma unroll
    for (int k4 = 0; k4 < 4 * (CB1 + 1); k4++) {
        if (k4 < 4 * (CB0 + 1)) acc[0] = mfma_f64(a[k4], b0[k4], acc[0]);
        acc[1] = mfma_f64(a[k4], b1[k4], acc[1]);
    }
}
// the same product with the fragments read as they are used (the preloading version holds up to 96 VGPRs of fragments: too
// many beside four column blocks of accumulators at four waves per SIMD)
template <int CB0, int CB1>
__device__ __forceinline__ void pr8_mma_nt_tri_stream(const double *As, const double *Bs, d4_t (&acc)[2], int wr8, int lane)
{
    const double *ap = As + (wr8 * 16 + (lane & 15)) * SD + (lane >> 4);
    const double *b0p = Bs + (CB0 * 16 + (lane & 15)) * SD + (lane >> 4), *b1p = Bs + (CB1 * 16 + (lane & 15)) * SD + (lane >> 4);
#pragma unroll
    for (int k4 = 0; k4 < 4 * (CB1 + 1); k4 += 4) {
        double a[4], b0[4], b1[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            a[u] = ap[(k4 + u) * 4];
            if (k4 + u < 4 * (CB0 + 1)) b0[u] = b0p[(k4 + u) * 4];
            b1[u] = b1p[(k4 + u) * 4];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (k4 + u < 4 * (CB0 + 1)) acc[0] = mfma_f64(a[u], b0[u], acc[0]);
            acc[1] = mfma_f64(a[u], b1[u], acc[1]);
        }
    }
}

// The row block's panel RIGHT-LOOKING inside the workgroup: all P column blocks' accumulators stay in registers (16 VGPRs per
// block and wave), and as soon as column jj's X_jj = (block) inv(L_jj)^T is formed it is applied to the later columns,
// acc_j'' -= X_jj L_j''jj^T.  Only the current X sits in LDS (one 64 x 64 stage for it, one for the operand block), 66 KB
// instead of 133: TWO workgroups per CU, so one's barriers and operand fetches hide behind the other's MFMAs.  Every element
// still receives the updates of columns 0, 1, .. in that order, each as 16 ascending k4-steps: identical bits.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
void chol_panel_rows8r_kernel(double *L, int Npad, int p0, int pend, const double *__restrict__ diag64, size_t lstride,
                              size_t dstride, double *__restrict__ Pk, size_t pstride, int rm_from)
{
    __shared__ double Xc[64 * SD];
    __shared__ double Bs[64 * SD];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wr8 = wv >> 1, wc8 = wv & 1;
    L += blockIdx.z * lstride; diag64 += blockIdx.z * dstride;
    if (Pk) Pk += blockIdx.z * pstride;
    const int i = pend + blockIdx.x, P = pend - p0;
    const bool rowmajor = !Pk || i >= rm_from;
    double *Ai = L + (size_t)i * 64 * Npad + (size_t)p0 * 64;
    // operand blocks in the order they are used: for jj = 0 .. P-1: inv(L_{p0+jj}), then L_{p0+j2, p0+jj} for j2 = jj+1 .. P-1
    auto fetch_b = [&](int jj, int j2, d2_t (&vb)[4]) {
        if (j2 > jj) pr8_fetch(L + (size_t)(p0 + j2) * 64 * Npad + (size_t)(p0 + jj) * 64, Npad, vb);
        else pr8_fetch(diag64 + (size_t)(p0 + jj) * 4096, 64, vb);
    };
    auto pack_col = [&](int jj, const double *Xm) {       // column jj (held as -X in Xm) into the packed store (chol_panel_rows8_kernel)
        const int g = wv >> 1, nk8 = Npad >> 3;
        double *dst = Pk + (((size_t)(i * 4 + g) * nk8 + (size_t)(p0 + jj) * 8 + 4 * (wv & 1)) * 64 + lane) * 2;
        const double *src = Xm + (16 * g + (lane & 15)) * SD + 32 * (wv & 1) + (lane >> 4);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            d2_t v;
            v.x = -src[8 * j]; v.y = -src[8 * j + 4];
            *(d2_t *)(dst + (size_t)j * 128) = v;
        }
    };
    d2_t vb[4];
    d4_t acc[4][2];                                       // P <= 4 column blocks of this wave's 16 x 32 piece
    fetch_b(0, 0, vb);
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (c < P) {
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[c][n][r] = Ai[(size_t)PR8_ROW(r) * Npad + c * 64 + PR8_COL(n)];
        }
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
        if (jj >= P) break;
        // X_jj = acc_jj inv(L_jj)^T
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) Xc[PR8_ROW(r) * SD + PR8_COL(n)] = acc[jj][n][r];
        pr8_stash(Bs, vb);
        __syncthreads();
        if (jj + 1 < P) fetch_b(jj, jj + 1, vb); 
        d4_t x[2] = {};
        if (wc8) pr8_mma_nt_tri_stream<1, 2>(Xc, Bs, x, wr8, lane);
        else pr8_mma_nt_tri_stream<0, 3>(Xc, Bs, x, wr8, lane);
        if (rowmajor) {
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) Ai[(size_t)PR8_ROW(r) * Npad + jj * 64 + PR8_COL_TRI(n)] = x[n][r];
        }
        __syncthreads();                                  // everybody has read Xc and Bs
        if (jj + 1 < P || Pk) {
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) Xc[PR8_ROW(r) * SD + PR8_COL_TRI(n)] = -x[n][r];
        }
        // the later columns take column jj's update: acc_j2 -= X_jj L_{j2,jj}^T
#pragma unroll
        for (int j2 = 1; j2 < 4; j2++) {
            if (j2 <= jj || j2 >= P) continue;
            pr8_stash(Bs, vb);
            __syncthreads();                              // also: -X_jj is in place
            if (j2 + 1 < P) fetch_b(jj, j2 + 1, vb); else fetch_b(jj + 1, jj + 1, vb);
            if (Pk && j2 == jj + 1) pack_col(jj, Xc);
            pr8_mma_nt(Xc, Bs, acc[j2], wr8, wc8, lane);
            __syncthreads();
        }
        if (Pk && jj + 1 == P) {                          // the last column: nobody packs it later
            __syncthreads();
            pack_col(jj, Xc);
        }
    }
}

static void launch_update(double *L, int Npad, int j0, int j1, int k0, int k1, int batch, size_t lstride,
                          hipStream_t s, const double *P = nullptr, int iend = 0)
{
    const int nb = iend > 0 ? iend : Npad / 64, nsr = (nb - k0 + 7) / 8, nsc = (k1 - k0 + 7) / 8;
    int tiles = 0;
    for (int k = k0; k < k1; k++) tiles += nb - k;
    if ((size_t)tiles * batch <= 512) {
        hipLaunchKernelGGL(chol_update_kernel, dim3(tiles, 1, batch), dim3(256), 0, s, L, Npad, j0, j1, k0, k1, 0, lstride, P, iend);
        return;
    }
    int nsb = 0;
    for (int c = 0; c < nsc; c++) nsb += nsr - c;
    const int groups = (nsb + 7) / 8;                // every XCD gets `groups` super-blocks of 64 workgroups
    hipLaunchKernelGGL(chol_update_kernel, dim3(groups * 512, 1, batch), dim3(256), 0, s, L, Npad, j0, j1, k0, k1,
                       nsb, lstride, P, iend);
}

// 128 x 128 tiles (over the batch) from which the packed-operand kernel (update3.hip) takes a right-looking K = 256 update; a batch keeps
// it down to a quarter of that (its late, small updates are many short launches of the 64 x 64 kernel otherwise)
static const int kPackedUpdateMinTiles = 1024;

// The panel's (64 P)^2 DIAGONAL block in one launch, one workgroup per matrix: for each of its P block columns the diagonal
// factorisation (chol_diag_kernel), the row blocks below it inside the block (chol_trsm_kernel) and their K = 64 updates
// (chol_update_kernel) -- the same k4-steps on the same operands in the same order per element, identical bits -- without the 3 P - 2
// launches of one workgroup per matrix each (11 per panel, 1.5 ms of a 64-theta grid at N = 4096).  The blocks travel through
// global memory (the workgroup reads back what it stored: one CU, one L1) and LDS holds the chain's three stages (91.6 KB).
// Eight waves, the geometry of chol_panel_rows8r_kernel, the next product's operands fetched under the current one (round 5: 27.6 -> 27.3 ms
// per 64-theta grid against the four-wave form; the launch itself stays at ~90 us -- its products wait on their own stores at every barrier).
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void chol_panel_diag_kernel(double *L, int Npad, int p0, int P, double *__restrict__ diag64, int *info, size_t lstride, size_t dstride)
{
    __shared__ double S[64 * SD];
    __shared__ double V[64 * SD];
    __shared__ double T[64 * CHAIN_TD];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wr8 = wv >> 1, wc8 = wv & 1;
    L += blockIdx.z * lstride; diag64 += blockIdx.z * dstride; info += blockIdx.z;
    auto blk = [&](int r, int c) { return L + (size_t)(p0 + r) * 64 * Npad + (size_t)(p0 + c) * 64; };
    d2_t va[4], vb[4];
    for (int j = 0; j < P; j++) {
        const int jb = p0 + j;
        double *Djj = blk(j, j);
        {   // the diagonal block into the chain's layout (zeros in V, the identity rows in T)
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = Djj[(size_t)(8 * u + wv) * Npad + lane];
            if (j + 1 < P) pr8_fetch(blk(j + 1, j), Npad, va);       // the first row block: ready since the last column's updates
#pragma unroll
            for (int u = 0; u < 8; u++) { S[(8 * u + wv) * SD + lane] = v[u]; V[(8 * u + wv) * SD + lane] = 0.0; }
            if (t < 256) T[(t >> 4) * CHAIN_TD + (t & 15)] = ((t >> 4) == (t & 15)) ? 1.0 : 0.0;
        }
        __syncthreads();
        diag64_factor_invert<CHAIN_TD>(S, V, T, jb * 64, info);
        {
            double *Db = diag64 + (size_t)jb * 4096;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int r = 8 * u + wv, c = lane;
                Djj[(size_t)r * Npad + c] = (c <= r) ? S[r * SD + c] : 0.0;
                Db[r * 64 + c] = V[r * SD + c];
            }
        }
        __syncthreads();
        // X_rj = A_rj inv(L_jj)^T, r = j + 1 .. P - 1 (operand in S, the inverse in V)
        for (int r = j + 1; r < P; r++) {
            pr8_stash(S, va);
            __syncthreads();
            if (r + 1 < P) pr8_fetch(blk(r + 1, j), Npad, va);
            d4_t x[2] = {};
            if (wc8) pr8_mma_nt_tri_body<1, 2>(S, V, x, wr8, lane);
            else pr8_mma_nt_tri_body<0, 3>(S, V, x, wr8, lane);
            double *Arj = blk(r, j);
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int q = 0; q < 4; q++) Arj[(size_t)PR8_ROW(q) * Npad + PR8_COL_TRI(n)] = x[n][q];
            __syncthreads();
        }
        // A_rc -= X_rj X_cj^T, r >= c > j (operands in S and V: the inverse has been used for the last time)
        bool have = false;
        for (int c = j + 1; c < P; c++)
            for (int r = c; r < P; r++) {
                if (!have) { pr8_fetch(blk(r, j), Npad, va); pr8_fetch(blk(c, j), Npad, vb); }
                double *C = blk(r, c);
                d4_t acc[2];
#pragma unroll
                for (int n = 0; n < 2; n++)
#pragma unroll
                    for (int q = 0; q < 4; q++) acc[n][q] = C[(size_t)PR8_ROW(q) * Npad + PR8_COL(n)];
#pragma unroll
                for (int u = 0; u < 4; u++) va[u] = -va[u];
                pr8_stash(S, va);
                pr8_stash(V, vb);
                __syncthreads();
                {   // the next product's operands (the row blocks X were all stored before the barrier above)
                    int r2 = r + 1, c2 = c;
                    if (r2 >= P) { c2 = c + 1; r2 = c2; }
                    have = c2 < P;
                    if (have) { pr8_fetch(blk(r2, j), Npad, va); pr8_fetch(blk(c2, j), Npad, vb); }
                }
                pr8_mma_nt(S, V, acc, wr8, wc8, lane);
#pragma unroll
                for (int n = 0; n < 2; n++)
#pragma unroll
                    for (int q = 0; q < 4; q++) C[(size_t)PR8_ROW(q) * Npad + PR8_COL(n)] = acc[n][q];
                __syncthreads();
            }
    }
}

// ---- the panel's diagonal block AND the rows below it in ONE launch (round 5; the batched left-looking order).  As two launches the chain of the
// diagonal block (~90 us, one workgroup per matrix) runs alone on the chip before the row kernel may start.  Here workgroups [0, batch) take the diagonal
// blocks -- lowest indices: dispatched, hence resident, before any other -- and publish, column by column, that inv(L_jj) and the in-panel row blocks
// L_{j2,jj} are in memory: those blocks are stored and fetched with the sc1 cache policy (past the XCDs' non-coherent L2 lines), the publisher waits
// for its stores' acknowledgements and raises one flag word per matrix and column (fences instead -- L2 write-back and invalidation by a thousand
// workgroups -- made the launch five times slower); the row workgroups (two 64-row blocks each, the geometry and the
// arithmetic of chol_panel_rows8r_kernel per block: identical bits) load their own tiles, then wait for column jj's flag before they fetch its
// operands, so they work on column jj while the diagonal workgroup factors column jj + 1.  A wait is bounded: on a timeout the workgroup reports
// through the info word and leaves (a wrong factor flagged not-positive-definite instead of a hung GPU).
typedef unsigned pf_u2 __attribute__((ext_vector_type(2)));
typedef unsigned pf_u4 __attribute__((ext_vector_type(4)));
#define PF_SC1 16       // buffer instruction cache policy: past the non-coherent L2 lines, to where every XCD sees it
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pf_rsrc(const void *p, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)(bytes > 0x7fffffffu ? 0x7fffffffu : bytes), 0x00020000);
}
// pr8_fetch of the 64 x 64 block at byte offset `base` (row stride ld8 bytes) through the coherent path
__device__ __forceinline__ void pf_fetch(const __amdgpu_buffer_rsrc_t r, unsigned base, unsigned ld8, d2_t (&v)[4])
{
    const unsigned t = threadIdx.x, voff = (t >> 5) * ld8 + (t & 31) * 16u;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const pf_u4 w = __builtin_amdgcn_raw_buffer_load_b128(r, voff, base + (unsigned)(16 * u) * ld8, PF_SC1);
        v[u].x = __hiloint2double((int)w.y, (int)w.x); v[u].y = __hiloint2double((int)w.w, (int)w.z);
    }
}
__device__ __forceinline__ void pf_store(const __amdgpu_buffer_rsrc_t r, unsigned off, double x)
{
    pf_u2 w;
    w.x = (unsigned)__double2loint(x); w.y = (unsigned)__double2hiint(x);
    __builtin_amdgcn_raw_buffer_store_b64(w, r, off, 0, PF_SC1);
}
__device__ __forceinline__ bool panel_wait(const int *flag, int want, int *info)
{
    __shared__ int ok;
    if (threadIdx.x == 0) {
        int good = 1;
        const long long t0 = wall_clock64();                    // (100 MHz)
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > 200000000LL) { good = 0; atomicCAS(info, 0, kPanelWaitTimeout); break; }     // 2 s (the publisher needs ~20 us): something is wrong
        }
        ok = good;
    }
    __syncthreads();                                   // (the operands are then fetched through the coherent path: no cache invalidation)
    return ok != 0;
}
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void chol_panel_fused_kernel(double *L, int Npad, int p0, int pend, double *__restrict__ diag64, int *info, size_t lstride, size_t dstride,
                             double *__restrict__ Pk, size_t pstride, int rm_from, int batch, int nrows, int *flags)
{
    __shared__ double lds[3 * 64 * SD];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wr8 = wv >> 1, wc8 = wv & 1;
    const int P = pend - p0, want = p0 + 1;
    if ((int)blockIdx.x < batch) {
        // ---------------- a diagonal block (chol_panel_diag_kernel's body, publishing after each column's row blocks)
        const int z = blockIdx.x;
        double *S = lds, *V = lds + 64 * SD, *T = lds + 2 * 64 * SD;
        L += z * lstride; diag64 += z * dstride; info += z; flags += 4 * z;
        auto blk = [&](int r, int c) { return L + (size_t)(p0 + r) * 64 * Npad + (size_t)(p0 + c) * 64; };
        const __amdgpu_buffer_rsrc_t rL = pf_rsrc(L, (size_t)Npad * Npad * sizeof(double)), rD = pf_rsrc(diag64, (size_t)(Npad / 64) * 32768);
        const unsigned ld8 = (unsigned)Npad * 8u;
        auto boff = [&](int r, int c) { return (unsigned)((p0 + r) * 64) * ld8 + (unsigned)(p0 + c) * 512u; };
        d2_t va[4], vb[4];
        for (int j = 0; j < P; j++) {
            const int jb = p0 + j;
            double *Djj = blk(j, j);
            {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = Djj[(size_t)(8 * u + wv) * Npad + lane];
                if (j + 1 < P) pr8_fetch(blk(j + 1, j), Npad, va);
#pragma unroll
                for (int u = 0; u < 8; u++) { S[(8 * u + wv) * SD + lane] = v[u]; V[(8 * u + wv) * SD + lane] = 0.0; }
                if (t < 256) T[(t >> 4) * CHAIN_TD + (t & 15)] = ((t >> 4) == (t & 15)) ? 1.0 : 0.0;
            }
            __syncthreads();
            diag64_factor_invert<CHAIN_TD>(S, V, T, jb * 64, info);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int r = 8 * u + wv, c = lane;
                Djj[(size_t)r * Npad + c] = (c <= r) ? S[r * SD + c] : 0.0;
                pf_store(rD, (unsigned)jb * 32768u + (unsigned)(r * 64 + c) * 8u, V[r * SD + c]);      // (the inverse: what the row workgroups read)
            }
            __syncthreads();
            for (int r = j + 1; r < P; r++) {
                pr8_stash(S, va);
                __syncthreads();
                if (r + 1 < P) pr8_fetch(blk(r + 1, j), Npad, va);
                d4_t x[2] = {};
                if (wc8) pr8_mma_nt_tri_body<1, 2>(S, V, x, wr8, lane);
                else pr8_mma_nt_tri_body<0, 3>(S, V, x, wr8, lane);
#pragma unroll
                for (int n = 0; n < 2; n++)
#pragma unroll
                    for (int q = 0; q < 4; q++) pf_store(rL, boff(r, j) + (unsigned)PR8_ROW(q) * ld8 + (unsigned)PR8_COL_TRI(n) * 8u, x[n][q]);
                __syncthreads();
            }
            // column j's inverse and row blocks are on their way past the L2: when every wave's stores are acknowledged, publish
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) __hip_atomic_store(flags + j, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool have = false;
            for (int c = j + 1; c < P; c++)
                for (int r = c; r < P; r++) {
                    if (!have) { pf_fetch(rL, boff(r, j), ld8, va); pf_fetch(rL, boff(c, j), ld8, vb); }
                    double *C = blk(r, c);
                    d4_t acc[2];
#pragma unroll
                    for (int n = 0; n < 2; n++)
#pragma unroll
                        for (int q = 0; q < 4; q++) acc[n][q] = C[(size_t)PR8_ROW(q) * Npad + PR8_COL(n)];
#pragma unroll
                    for (int u = 0; u < 4; u++) va[u] = -va[u];
                    pr8_stash(S, va);
                    pr8_stash(V, vb);
                    __syncthreads();
                    {
                        int r2 = r + 1, c2 = c;
                        if (r2 >= P) { c2 = c + 1; r2 = c2; }
                        have = c2 < P;
                        if (have) { pf_fetch(rL, boff(r2, j), ld8, va); pf_fetch(rL, boff(c2, j), ld8, vb); }
                    }
                    pr8_mma_nt(S, V, acc, wr8, wc8, lane);
#pragma unroll
                    for (int n = 0; n < 2; n++)
#pragma unroll
                        for (int q = 0; q < 4; q++) C[(size_t)PR8_ROW(q) * Npad + PR8_COL(n)] = acc[n][q];
                    __syncthreads();
                }
        }
        return;
    }
    // ---------------- two row blocks below the panel
    const int nrw = (nrows + 1) / 2, rb = (int)blockIdx.x - batch, z = rb / nrw, pair = rb - z * nrw;
    double *Xc0 = lds, *Xc1 = lds + 64 * SD, *Bs = lds + 2 * 64 * SD;
    L += z * lstride; diag64 += z * dstride; info += z; flags += 4 * z;
    if (Pk) Pk += z * pstride;
    const int i0 = pend + 2 * pair;
    const bool two = 2 * pair + 1 < nrows;                    // (workgroup-uniform: the second block exists)
    const __amdgpu_buffer_rsrc_t rL = pf_rsrc(L, (size_t)Npad * Npad * sizeof(double)), rD = pf_rsrc(diag64, (size_t)(Npad / 64) * 32768);
    const unsigned ld8 = (unsigned)Npad * 8u;
    auto fetch_b = [&](int jj, int j2, d2_t (&vb)[4]) {
        if (j2 > jj) pf_fetch(rL, (unsigned)((p0 + j2) * 64) * ld8 + (unsigned)(p0 + jj) * 512u, ld8, vb);
        else pf_fetch(rD, (unsigned)(p0 + jj) * 32768u, 512u, vb);
    };
    auto pack_col = [&](int s, int jj, const double *Xm) {
        const int g = wv >> 1, nk8 = Npad >> 3;
        double *dst = Pk + (((size_t)((i0 + s) * 4 + g) * nk8 + (size_t)(p0 + jj) * 8 + 4 * (wv & 1)) * 64 + lane) * 2;
        const double *src = Xm + (16 * g + (lane & 15)) * SD + 32 * (wv & 1) + (lane >> 4);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            d2_t v;
            v.x = -src[8 * j]; v.y = -src[8 * j + 4];
            *(d2_t *)(dst + (size_t)j * 128) = v;
        }
    };
    d2_t vb[4];
    d4_t acc[2][4][2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const double *Ai = L + (size_t)(i0 + s) * 64 * Npad + (size_t)p0 * 64;
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (c < P && (s == 0 || two)) {
#pragma unroll
                for (int n = 0; n < 2; n++)
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[s][c][n][r] = Ai[(size_t)PR8_ROW(r) * Npad + c * 64 + PR8_COL(n)];
            }
    }
    if (!panel_wait(flags + 0, want, info)) return;
    fetch_b(0, 0, vb);
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
        if (jj >= P) break;
#pragma unroll
        for (int s = 0; s < 2; s++)
            if (s == 0 || two) {
                double *Xc = s ? Xc1 : Xc0;
#pragma unroll
                for (int n = 0; n < 2; n++)
#pragma unroll
                    for (int r = 0; r < 4; r++) Xc[PR8_ROW(r) * SD + PR8_COL(n)] = acc[s][jj][n][r];
            }
        pr8_stash(Bs, vb);
        __syncthreads();
        if (jj + 1 < P) fetch_b(jj, jj + 1, vb);
        d4_t x[2][2] = {};
#pragma unroll
        for (int s = 0; s < 2; s++)
            if (s == 0 || two) {
                double *Xc = s ? Xc1 : Xc0;
                if (wc8) pr8_mma_nt_tri_stream<1, 2>(Xc, Bs, x[s], wr8, lane);
                else pr8_mma_nt_tri_stream<0, 3>(Xc, Bs, x[s], wr8, lane);
                if (!Pk || i0 + s >= rm_from) {
                    double *Ai = L + (size_t)(i0 + s) * 64 * Npad + (size_t)p0 * 64;
#pragma unroll
                    for (int n = 0; n < 2; n++)
#pragma unroll
                        for (int r = 0; r < 4; r++) Ai[(size_t)PR8_ROW(r) * Npad + jj * 64 + PR8_COL_TRI(n)] = x[s][n][r];
                }
            }
        __syncthreads();
        if (jj + 1 < P || Pk) {
#pragma unroll
            for (int s = 0; s < 2; s++)
                if (s == 0 || two) {
                    double *Xc = s ? Xc1 : Xc0;
#pragma unroll
                    for (int n = 0; n < 2; n++)
#pragma unroll
                        for (int r = 0; r < 4; r++) Xc[PR8_ROW(r) * SD + PR8_COL_TRI(n)] = -x[s][n][r];
                }
        }
#pragma unroll
        for (int j2 = 1; j2 < 4; j2++) {
            if (j2 <= jj || j2 >= P) continue;
            pr8_stash(Bs, vb);
            __syncthreads();
            if (j2 + 1 < P) fetch_b(jj, j2 + 1, vb);
            else {
                if (!panel_wait(flags + jj + 1, want, info)) return;      // (the next column's inverse)
                fetch_b(jj + 1, jj + 1, vb);
            }
            if (Pk && j2 == jj + 1) { pack_col(0, jj, Xc0); if (two) pack_col(1, jj, Xc1); }
            pr8_mma_nt(Xc0, Bs, acc[0][j2], wr8, wc8, lane);
            if (two) pr8_mma_nt(Xc1, Bs, acc[1][j2], wr8, wc8, lane);
            __syncthreads();
        }
        if (Pk && jj + 1 == P) {
            __syncthreads();
            pack_col(0, jj, Xc0); if (two) pack_col(1, jj, Xc1);
        }
    }
}

// the block columns [p0, pend) of a panel whose columns are up to date: diagonal blocks, row blocks, K = 64 updates inside the panel.
// Returns true when the rows below the panel went to the packed store Pk (left-looking order) on the way.
static bool chol_inpanel(double *L, int Npad, int p0, int pend, double *diag64, int *info_dev, int batch, size_t lstride, hipStream_t s,
                         double *Pk = nullptr, size_t pstride = 0, int rm_from = 0, int *flags = nullptr)
{
    const int nb = Npad / 64;
    const size_t dstride = (size_t)nb * 4096;
    // With enough rows below to fill the chip the per-column launches stay inside the panel's diagonal block -- one launch,
    // chol_panel_diag_kernel -- and the rows below it take the whole panel in one more (chol_panel_rows8r_kernel): the same arithmetic
    // in the same order as the per-column sequence, which runs where the rows are few (its short launches finish sooner).
    const bool rows_fused = pend - p0 <= 4 && (size_t)(nb - pend) * batch >= 256;
    // (the fused launch reaches its blocks through buffer descriptors with 32-bit offsets: one matrix must lie inside 2^31 bytes -- 16 320 rows;
    // beyond, the two launches below, whose addresses are 64-bit)
    if (rows_fused && pend < nb && flags && (size_t)Npad * Npad * sizeof(double) <= 0x7fffffffu) {
        // (flags: four ints per matrix, zero at the start of the factorisation)
        const int nrows = nb - pend, nrw = (nrows + 1) / 2;
        hipLaunchKernelGGL(chol_panel_fused_kernel, dim3((unsigned)(batch + batch * nrw)), dim3(512), 0, s, L, Npad, p0, pend, diag64, info_dev, lstride,
                           dstride, Pk, pstride, rm_from, batch, nrows, flags);
        return Pk != nullptr;
    }
    if (rows_fused)
        hipLaunchKernelGGL(chol_panel_diag_kernel, dim3(1, 1, batch), dim3(512), 0, s, L, Npad, p0, pend - p0, diag64, info_dev, lstride, dstride);
    else
        for (int jb = p0; jb < pend; jb++) {
            hipLaunchKernelGGL(chol_diag_kernel, dim3(1, 1, batch), dim3(256), 0, s, L, Npad, jb, diag64, info_dev,
                               lstride, dstride, (double *)nullptr);
            const int m = nb - jb - 1;
            if (m > 0)
                hipLaunchKernelGGL(chol_trsm_kernel, dim3(m, 1, batch), dim3(256), 0, s, L, Npad, jb, diag64, lstride,
                                   dstride, (double *)nullptr);
            if (jb + 1 < pend) launch_update(L, Npad, jb, jb + 1, jb + 1, pend, batch, lstride, s, nullptr, 0);
        }
    if (rows_fused && pend < nb)
        hipLaunchKernelGGL(chol_panel_rows8r_kernel, dim3(nb - pend, 1, batch), dim3(512), 0, s, L, Npad, p0, pend, diag64,
                           lstride, dstride, Pk, pstride, rm_from);
    return rows_fused && pend < nb && Pk;
}

// `batch` matrices, `lstride` doubles apart (diag64: (Npad/64)*4096 apart, info: consecutive ints), are factored IN PLACE by the same
// launches (blockIdx.z).  panel = 1 is the plain right-looking order (shortest chain: one matrix, small N); panel = P > 1 keeps the
// per-column updates inside a P-block panel and applies the panel to the rest of the matrix once, with K = 64 P.
int launch_cholesky_batched(double *L, int Npad, double *diag64, int *info_dev, int batch, size_t lstride,
                            int panel, hipStream_t s, double *ws, size_t wstride)
{
    const int nb = Npad / 64, P = panel;
    HIPCHK(hipMemsetAsync(info_dev, 0, sizeof(int) * batch, s));
    for (int p0 = 0; p0 < nb; p0 += P) {
        const int pend = p0 + P < nb ? p0 + P : nb;
        chol_inpanel(L, Npad, p0, pend, diag64, info_dev, batch, lstride, s);
        if (pend < nb) {
            // the big update (K = 64 P): packed-operand kernel when the caller lent a workspace, bit-identical to the other
            const int nI2 = (Npad - 64 * pend + 127) / 128;
            if (ws && (size_t)nI2 * (nI2 + 1) / 2 * batch >= (size_t)(batch >= 8 ? kPackedUpdateMinTiles / 4 : kPackedUpdateMinTiles)) {
                int rc = launch_chol_update2(L, Npad, p0, pend, batch, lstride, ws, wstride, s);
                if (rc) return rc;
            } else launch_update(L, Npad, p0, pend, pend, nb, batch, lstride, s);
        }
    }
    return (int)hipGetLastError();
}

// The same factorisation in the LEFT-LOOKING outer order (update3.hip): a panel's block columns receive all their updates
// -- from every finished column, K = 64 p0 -- in one launch just before the panel is factored, from a packed copy of the
// finished columns (Pk: Npad^2 doubles per matrix, pstride apart) that grows by a panel per step.  Same sums in the same order
// as launch_cholesky_batched with the same panel width: identical bits.  nlive: rows >= nlive are identity pad (they are not
// touched); nfactor: block columns to factor (a caller that never reads the last block column -- the likelihood's y row
// alone in it -- passes nb - 1); rm_from: the factor's blocks BELOW a panel's diagonal block are stored row-major for block rows
// >= rm_from only (0: all of them, i.e. the whole factor; the likelihood reads the y row and the diagonal and passes N / 64).
// Several sub-batches (CholGroup: its matrices, its stream) advance panel by panel, their launches enqueued ALTERNATELY: with one
// sub-batch's whole chain (~70 launches) enqueued before the next one's first, the second stream starts that much host time late.
int launch_cholesky_batched_left(const CholGroup *groups, int ngroups, int Npad, size_t lstride, int panel, size_t pstride, int nlive,
                                 int nfactor, int rm_from)
{
    const int nb = Npad / 64;
    const int P = panel;
    if (nfactor <= 0 || nfactor > nb) nfactor = nb;
    for (int g = 0; g < ngroups; g++) {
        HIPCHK(hipMemsetAsync(groups[g].info, 0, sizeof(int) * groups[g].batch, groups[g].stream));
        if (groups[g].flags) HIPCHK(hipMemsetAsync(groups[g].flags, 0, sizeof(int) * 4 * groups[g].batch, groups[g].stream));
    }
    // THE TAIL.  Left-looking, the last panels have few tiles (17 .. 5 per matrix over the last 1024 columns) and a long K: the
    // chip runs half empty (46 .. 81 % of the update kernel's rate on panels 12 .. 15 of 16).  From block column `tail` on the
    // order changes: ONE update brings all remaining columns up to date with everything before `tail` (many tiles, long K), and
    // inside the tail the order is right-looking (after each panel a K = 256 update of the remaining tail columns: short K, but
    // the tail is small).  Every element still receives its terms in ascending k: identical bits.
    // (Round 5, measured and removed: that one update in K-chunks of 256 / 512 / 1024 columns, so that a matrix's ~780 packed rows stay in an
    // XCD's L2 from tile to tile -- 28.9 / 27.5 / 27.3 ms per 64-theta grid against 27.2 in one launch: its 42 TFLOP/s are not a traffic problem.
    // Tails of 20 / 28 / 36 block columns: 27.2 / 27.9 / 28.5 ms.)
    int tail = nfactor;
    const int kTail = 20;
    if (nfactor > kTail + P) tail = (nfactor - kTail) / P * P;
    for (int p0 = 0; p0 < nfactor; p0 += P) {
        const int pend = p0 + P < nb ? p0 + P : nb;
        for (int g = 0; g < ngroups; g++) {
            const CholGroup &G = groups[g];
            if (G.batch <= 0) continue;
            if (p0 > 0 && p0 <= tail) {
                const int width = p0 == tail ? 64 * (nfactor - p0) : 64 * (pend - p0);
                int rc = launch_chol_update3_range(G.L, Npad, 64 * p0, width, 0, 64 * p0, nlive, G.batch, lstride, G.Pk, pstride, G.stream);
                if (rc) return rc;
            }
            // the rows below the panel reach the packed store straight from chol_panel_rows8r_kernel's LDS where that kernel runs
            // (and then only the block rows >= rm_from are also stored row-major), through chol_pack3_kernel otherwise
            const bool packed = chol_inpanel(G.L, Npad, p0, pend, G.diag64, G.info, G.batch, lstride, G.stream, pend < nfactor ? G.Pk : nullptr, pstride, rm_from,
                                             G.flags);
            if (pend < nfactor && !packed) {
                int rc = launch_chol_pack3(G.L, Npad, 64 * pend, 64 * p0, 64 * (pend - p0), G.batch, lstride, G.Pk, pstride, G.stream);
                if (rc) return rc;
            }
            if (p0 >= tail && pend < nfactor) {              // inside the tail: this panel's update of the columns still to come
                int rc = launch_chol_update3_range(G.L, Npad, 64 * pend, 64 * (nfactor - pend), 64 * p0, 64 * pend, nlive, G.batch, lstride, G.Pk, pstride, G.stream);
                if (rc) return rc;
            }
        }
    }
    return (int)hipGetLastError();
}

// Plain right-looking order with one fused launch per block column (chol_step_kernel): `work` holds the matrix
// and is destroyed, the factor (lower blocks; the strict upper blocks are not touched) goes to `out`.
// Bit-identical to launch_cholesky with panel = 1.
// SOFTWARE-PIPELINED block columns (the fit path up to 2048 rows).  Launch jb holds
//   * the ROW workgroups of block column jb -- the matrix's row blocks jb + 1 .. and, with the ride-along, E's row blocks
//     0 .. jb: each first applies step jb - 1 to the two tiles it needs (its own block of column jb and the diagonal block:
//     C - X_i X_jb^T and C - X_jb X_jb^T with the row blocks X(jb - 1) the previous launch stored), then runs the chain on the
//     diagonal block and multiplies its block by inv(L_jj)^T -- chol_diag_trsm_kernel behind two products;
//   * the TILE workgroups with the rest of step jb - 1: tiles (i, k), k > jb, of the matrix and of E, one product each,
//     chol_update_step_kernel's body.  They touch nothing the row workgroups read or write in this launch.
// So the trailing update of a step runs BESIDE the next step's chain instead of before it: a step costs max(chain + three
// products, tiles) instead of their sum (25 -> 17 us at N = 2048).  With 133 KB of LDS every workgroup has a CU to itself, so
// no tile's MFMAs share a SIMD with a chain (which would slow the chain several times over: docs/notebook.md).  Each tile still
// receives the updates of steps 0, 1, .. in that order with the same operands: identical bits (tested).
// (The kernel is chol_pipe8_kernel below.)

// ---- the pipelined block column on EIGHT waves, with the row workgroups' own update UNDER the chain (round 4).
// chol_pipe_kernel's row workgroup brings two tiles up to date before its chain starts -- the diagonal block and its own block of column
// jb, one 64^3 product each on four waves -- and both sit on the critical path of every block column (21 us per column at N = 2048:
// gap 2.5 + loads 2.4 + two products 3 + restash 1.5 + chain 8.3 + product 1.5 + stores).  Only the diagonal block's update has to: the own
// block is needed after the chain.  Here it runs DURING the chain, on waves 5, 6, 7 -- which idle through it, on SIMDs 1-3 (wave 4 shares
// SIMD 0 and its fp64 pipe with the chain's wave and stays idle) -- in four slices of four k4-steps, one per panel of the chain, between the
// chain's own barriers (diag64_factor_invert's `side`); its A operand comes straight from memory in fragment form, its B operand (the row
// block X_jb of step jb - 1) from LDS.  The products before and after the chain run on eight waves.  Every element still receives steps
// 0, 1, .. in order, each as sixteen ascending k4-steps on the same operands: identical bits (tools/check_pipe.py, test_split_steps_equal_fused_steps).
// Tile workgroups: one product per tile, eight waves.
template <int SW>     // side wave SW = 0, 1, 2 (waves 5, 6, 7): tiles 0..5 / 6..10 / 11..15 of the 4 x 4 grid of 16 x 16 tiles, i.e. two row strips each
struct Pipe8Side {
    static constexpr int BASE = SW == 0 ? 0 : (SW == 1 ? 6 : 11), CNT = SW == 0 ? 6 : 5;
    static __device__ __forceinline__ int strip(int s) { return (BASE + s) >> 2; }
    static __device__ __forceinline__ int cblk(int s) { return (BASE + s) & 3; }
    static __device__ __forceinline__ void load(const double *Ap, int Npad, int lane, d4_t (&acc)[6])
    {
#pragma unroll
        for (int s = 0; s < CNT; s++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[s][r] = Ap[(size_t)(16 * strip(s) + (lane >> 4) + 4 * r) * Npad + 16 * cblk(s) + (lane & 15)];
    }
    // acc_s -= X_i[strip] X_d[cblk]^T over k4 in [4 b, 4 b + 4): A = -X_i fragments from memory, B = X_d fragments from LDS
    static __device__ __forceinline__ void slice(int b, const double *Xi, int Npad, const double *Us, int lane, d4_t (&acc)[6])
    {
        constexpr int S0 = BASE >> 2, S1 = (BASE + CNT - 1) >> 2;
        double a0[4], a1[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int k = (4 * b + kk) * 4 + (lane >> 4);
            a0[kk] = -Xi[(size_t)(16 * S0 + (lane & 15)) * Npad + k];
            a1[kk] = -Xi[(size_t)(16 * S1 + (lane & 15)) * Npad + k];
        }
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int k = (4 * b + kk) * 4 + (lane >> 4);
#pragma unroll
            for (int s = 0; s < CNT; s++) {
                const double bv = Us[(16 * cblk(s) + (lane & 15)) * SD + k];
                acc[s] = mfma_f64(strip(s) == S0 ? a0[kk] : a1[kk], bv, acc[s]);
            }
        }
    }
    static __device__ __forceinline__ void store(double *Us, int lane, const d4_t (&acc)[6])
    {
#pragma unroll
        for (int s = 0; s < CNT; s++)
#pragma unroll
            for (int r = 0; r < 4; r++) Us[(16 * strip(s) + (lane >> 4) + 4 * r) * SD + 16 * cblk(s) + (lane & 15)] = acc[s][r];
    }
};

// TMODE (which tiles ride in the launch; the row workgroups are the same in both).  0: step jp on every tile right of column jb -- one pass
// over the trailing matrix per block column.  Beyond ~2000 rows that pass is what a column costs (44 us at N = 4096 against the chain's 18:
// 128 KiB moved per 64^3 product), so there a tile gets TWO steps per pass, the accumulators staying in registers between them
// (the four operand blocks fill the four LDS arrays) -- 1: the launch carries `nsingle` tiles of column jb + 1 with step jp alone (odd jb:
// the column the next launch factors) and then the tiles of columns [c_lo, c_hi) with steps q - 1 and q; the host deals the columns of a
// pair of steps over the two launches that may carry it (launch_cholesky_fused).  A tile receives the same k4-steps in the same order on the
// same operands as with a store and a reload in between: identical bits.
template <int TMODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void chol_pipe8_kernel(double *__restrict__ A, double *__restrict__ Lout, int Npad, int jb, double *__restrict__ diag64, int *info,
                       int nrow, double *__restrict__ Ework, double *__restrict__ Eout, int kend, int pre, int nsingle, int q, int c_lo, int c_hi)
{
    __shared__ double S[64 * SD];
    __shared__ double V[64 * SD];
    __shared__ double T[64 * SD];
    __shared__ double U[64 * SD];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wr8 = wv >> 1, wc8 = wv & 1;
    const int nb = Npad / 64, m = nb - jb - 1, jp = jb - 1;
    if (TMODE != 0 && (int)blockIdx.x >= nrow) {
        int t = blockIdx.x - nrow;
        int i, k, sc;                                           // tile (i, k); sc: the block column of the (second) step's row blocks
        bool etile, two;
        if (t < nsingle) {                                      // column jb + 1: its m tiles of the matrix, then E's rows 0 .. jp
            k = jb + 1; sc = jp;
            etile = t >= m;
            i = etile ? t - m : k + t;
            two = false;
        } else {
            t -= nsingle;
            sc = q;
            int nchol = 0;
            for (int kk = c_lo; kk < c_hi; kk++) nchol += nb - kk;
            etile = t >= nchol;
            if (!etile) {
                k = c_lo;
                int rem = t;
                while (rem >= nb - k) { rem -= nb - k; k++; }
                i = k + rem;
            } else {
                const int e = t - nchol, w = c_hi - c_lo;
                i = e / w; k = c_lo + e % w;
            }
            two = !(etile && i == q);                           // (E's row q takes part from step q on)
        }
        const double *Xi = (etile ? Eout : Lout) + (size_t)i * 64 * Npad + sc * 64;
        const double *Xk = Lout + (size_t)k * 64 * Npad + sc * 64;
        double *C = (etile ? Ework : A) + (size_t)i * 64 * Npad + k * 64;
        d2_t va[4], vb[4], va2[4], vb2[4];
        if (two) {
            pr8_fetch(Xi - 64, Npad, va2);                      // step q - 1: the block column to the left
            pr8_fetch(Xk - 64, Npad, vb2);
        }
        pr8_fetch(Xi, Npad, va);
        pr8_fetch(Xk, Npad, vb);
        d4_t acc[2];
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[n][r] = C[(size_t)PR8_ROW(r) * Npad + PR8_COL(n)];
        if (two) {
#pragma unroll
            for (int u = 0; u < 4; u++) { va2[u] = -va2[u]; }
            pr8_stash(S, va2);
            pr8_stash(V, vb2);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) { va[u] = -va[u]; }
        pr8_stash(T, va);
        pr8_stash(U, vb);
        __syncthreads();
        if (two) pr8_mma_nt(S, V, acc, wr8, wc8, lane);
        pr8_mma_nt(T, U, acc, wr8, wc8, lane);
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) C[(size_t)PR8_ROW(r) * Npad + PR8_COL(n)] = acc[n][r];
        return;
    }
    if (TMODE == 0 && (int)blockIdx.x >= nrow) {
        // ---- a tile of step jp right of column jb (numbering as in chol_pipe_kernel)
        int nchol = 0;
        for (int kk = jb + 1; kk < kend; kk++) nchol += nb - kk;
        const int t = blockIdx.x - nrow;
        int i, k;
        const double *Xi;
        double *C;
        if (t < nchol) {
            k = jb + 1;
            int rem = t;
            while (rem >= nb - k) { rem -= nb - k; k++; }
            i = k + rem;
            Xi = Lout + (size_t)i * 64 * Npad + jp * 64;
            C = A + (size_t)i * 64 * Npad + k * 64;
        } else {
            const int e = t - nchol;
            i = e / m; k = jb + 1 + e % m;
            Xi = Eout + (size_t)i * 64 * Npad + jp * 64;
            C = Ework + (size_t)i * 64 * Npad + k * 64;
        }
        const double *Xk = Lout + (size_t)k * 64 * Npad + jp * 64;
        d2_t va[4], vb[4];
        pr8_fetch(Xi, Npad, va);
        pr8_fetch(Xk, Npad, vb);
        d4_t acc[2];
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[n][r] = C[(size_t)PR8_ROW(r) * Npad + PR8_COL(n)];
#pragma unroll
        for (int u = 0; u < 4; u++) { va[u] = -va[u]; }
        pr8_stash(S, va);
        pr8_stash(V, vb);
        __syncthreads();
        pr8_mma_nt(S, V, acc, wr8, wc8, lane);
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) C[(size_t)PR8_ROW(r) * Npad + PR8_COL(n)] = acc[n][r];
        return;
    }
    // ---- a row block of block column jb
    const size_t doff = (size_t)jb * 64 * Npad + jb * 64;
    const int nE = Ework ? jb + 1 : 0;
    const bool has_row = (int)blockIdx.x < m + nE;              // (a last column without ride-along: the diagonal block alone)
    const bool erow = (int)blockIdx.x >= m;
    const int ib = erow ? (int)blockIdx.x - m : jb + 1 + (int)blockIdx.x;
    const size_t roff = (size_t)ib * 64 * Npad + jb * 64;
    const bool upd_d = pre != 0, upd_a = pre != 0 && has_row && !(erow && ib == jb);     // E's block (jb, jb) is still the identity
    const double *Xi_glob = (erow ? Eout : Lout) + (size_t)ib * 64 * Npad + jp * 64;      // this row block's X of step jp (upd_a)
    const double *Ap = (erow ? Ework : A) + roff;
    // the diagonal block as eight-wave accumulators; this workgroup's own block as the side waves' 16 x 16 tiles
    // (the chain reads only the 16-blocks of the diagonal block on and below its diagonal -- diag64_panel, diag64_update_tile --, so
    // only those ten get the update; dealt so that the two waves of a SIMD hold three, three, two and two of them: 48 MFMAs on the
    // busiest fp64 pipe instead of 64 for the full 64^3 product, whose floor on one CU is 4.1 k cycles)
    d4_t ad[2], aa[6];
    d2_t vxd[4];
    const int dt_rb0 = wv == 0 ? 0 : (wv == 1 ? 1 : (wv < 4 ? 2 : 3)), dt_cb0 = wv == 0 ? 0 : (wv == 1 ? 1 : (wv == 2 ? 1 : (wv == 3 ? 2 : (wv == 4 ? 3 : (wv == 5 ? 2 : (wv == 6 ? 0 : 1))))));
    const int dt_rb1 = wv == 0 ? 1 : 2, dt_cb1 = 0;               // second tile: waves 0 and 1 only: (1, 0) and (2, 0)
    const bool dt_two = wv < 2;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        ad[0][r] = A[doff + (size_t)(16 * dt_rb0 + (lane >> 4) + 4 * r) * Npad + 16 * dt_cb0 + (lane & 15)];
        ad[1][r] = dt_two ? A[doff + (size_t)(16 * dt_rb1 + (lane >> 4) + 4 * r) * Npad + 16 * dt_cb1 + (lane & 15)] : 0.0;
    }
    if (upd_d) pr8_fetch(Lout + (size_t)jb * 64 * Npad + jp * 64, Npad, vxd);
    if (has_row) {
        if (wv == 5) Pipe8Side<0>::load(Ap, Npad, lane, aa);
        else if (wv == 6) Pipe8Side<1>::load(Ap, Npad, lane, aa);
        else if (wv == 7) Pipe8Side<2>::load(Ap, Npad, lane, aa);
    }
    // the chain's V (zeros) and T (identity rows) are laid out now, beside the operand: the diagonal block's update reads X_jb from U for
    // BOTH operands (the A side negated in the register: the same bits as a negated copy in LDS), so only S waits for the product
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
        for (int r = 0; r < 4; r++) V[PR8_ROW(r) * SD + PR8_COL(n)] = 0.0;
    if (threadIdx.x < 256) T[(threadIdx.x >> 4) * SD + (threadIdx.x & 15)] = ((threadIdx.x >> 4) == (threadIdx.x & 15)) ? 1.0 : 0.0;
    if (upd_d) {
        pr8_stash(U, vxd);                                      //  X_jb: both operands here, B operand of the side product
        __syncthreads();
        {   // every fragment first (one LDS latency), then the MFMAs
            double fa0[16], fb0[16], fa1[16];
#pragma unroll
            for (int k4 = 0; k4 < 16; k4++) {
                fa0[k4] = -U[(dt_rb0 * 16 + (lane & 15)) * SD + k4 * 4 + (lane >> 4)];
                fb0[k4] = U[(dt_cb0 * 16 + (lane & 15)) * SD + k4 * 4 + (lane >> 4)];
                fa1[k4] = -U[(dt_rb1 * 16 + (lane & 15)) * SD + k4 * 4 + (lane >> 4)];
            }
            if (dt_two) {                                       // (both of a wave's second tiles are in column block 0, as wave 0's first)
#pragma unroll
                for (int k4 = 0; k4 < 16; k4++) {
                    const double fb1 = U[(lane & 15) * SD + k4 * 4 + (lane >> 4)];
                    ad[0] = mfma_f64(fa0[k4], fb0[k4], ad[0]);
                    ad[1] = mfma_f64(fa1[k4], fb1, ad[1]);
                }
            } else {
#pragma unroll
                for (int k4 = 0; k4 < 16; k4++) ad[0] = mfma_f64(fa0[k4], fb0[k4], ad[0]);
            }
        }
    }
    // the diagonal block into the chain's layout
#pragma unroll
    for (int r = 0; r < 4; r++) {
        S[(16 * dt_rb0 + (lane >> 4) + 4 * r) * SD + 16 * dt_cb0 + (lane & 15)] = ad[0][r];
        if (dt_two) S[(16 * dt_rb1 + (lane >> 4) + 4 * r) * SD + 16 * dt_cb1 + (lane & 15)] = ad[1][r];
    }
    __syncthreads();
    auto side = [&](int b) {
        if (!upd_a) return;
        if (wv == 5) Pipe8Side<0>::slice(b, Xi_glob, Npad, U, lane, aa);
        else if (wv == 6) Pipe8Side<1>::slice(b, Xi_glob, Npad, U, lane, aa);
        else if (wv == 7) Pipe8Side<2>::slice(b, Xi_glob, Npad, U, lane, aa);
    };
    // (the LAST row-type workgroup has no row block -- the launch gives it none: it reports a failed pivot and stores the diagonal block and
    // its inverse, 1.3 us that sat on workgroup 0's path, hence on the launch's, while that workgroup still had its product to do)
    const bool keeper = (int)blockIdx.x == nrow - 1;
    diag64_factor_invert(S, V, T, jb * 64, keeper ? info : nullptr, side);      // (ends with a barrier)
    if (keeper) {
        const int tt = threadIdx.x;
        double *Lb = Lout + doff, *Db = diag64 + (size_t)jb * 4096;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int r = 8 * u + (tt >> 6), cc = tt & 63;
            Lb[(size_t)r * Npad + cc] = (cc <= r) ? S[r * SD + cc] : 0.0;
            Db[r * 64 + cc] = V[r * SD + cc];
        }
    }
    if (!has_row) return;
    // the own block, up to date, into U (X_jb there has been read for the last time before the chain's last barrier)
    if (wv == 5) Pipe8Side<0>::store(U, lane, aa);
    else if (wv == 6) Pipe8Side<1>::store(U, lane, aa);
    else if (wv == 7) Pipe8Side<2>::store(U, lane, aa);
    __syncthreads();
    d4_t acc[2] = {};
    if (wc8) pr8_mma_nt_tri_body<1, 2>(U, V, acc, wr8, lane);
    else pr8_mma_nt_tri_body<0, 3>(U, V, acc, wr8, lane);
    double *Ob = (erow ? Eout : Lout) + roff;
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
        for (int r = 0; r < 4; r++) Ob[(size_t)PR8_ROW(r) * Npad + PR8_COL_TRI(n)] = acc[n][r];
}

static const int kPipeFrom = 4;         // block columns from which the pipelined column takes the single-level order (below: fused steps)
static const int kPairsFrom = 12;       // block columns from which the trailing tiles get two steps per pass (768 rows)
static const int kStepSplit = 256;      // in-panel tiles of a two-level block column from which rows and updates are separate launches

// Plain right-looking order, one launch per block column, out of place: `work` holds the matrix and is destroyed, the factor (lower
// blocks; the strict upper blocks are not touched) goes to `out`; Ework (identity on entry) / Eout: the W = L^-1 ride-along.
int launch_cholesky_fused(double *work, double *out, int Npad, double *diag64, int *info_dev, hipStream_t s, double *Ework,
                          double *Eout, bool info_is_zero)
{
    const int nb = Npad / 64;
    if (!info_is_zero) HIPCHK(hipMemsetAsync(info_dev, 0, sizeof(int), s));
    if (nb >= kPipeFrom) {
        // (the eight-wave pipelined kernel, whose row workgroups update their own block under the chain, beats the fused step from four
        // block columns on: 0.320 -> 0.304 ms at N = 1024)
        const bool pairs = nb >= kPairsFrom;
        int split = nb;                                 // (pairs: first column whose pair of steps waits for the odd launch)
        for (int jb = 0; jb < nb; jb++) {
            const int m = nb - jb - 1, nE = Ework ? jb + 1 : 0;
            const int nrow = m + nE + 1;                                                  // (one more: the diagonal block's keeper)
            const int ntile = jb > 0 ? m * (m + 1) / 2 + (Ework ? jb * m : 0) : 0;        // step jb - 1 right of column jb
            if (pairs) {
                // two steps per pass.  The pair of steps (2 p, 2 p + 1) is due on every column right of 2 p + 2 and may ride in launch
                // 2 p + 2 or 2 p + 3: the columns up to `split` (at least the two that the next launches factor) take it in the even launch,
                // the rest in the odd one -- which also carries step jb - 1 for column jb + 1 alone -- so that both launches have about
                // the same number of tiles to hide under their chain.
                const int nE1 = Ework ? 1 : 0;
                int nsingle = 0, q = 0, c_lo = 0, c_hi = 0;
                if (jb & 1) {
                    nsingle = m > 0 ? m + nE1 * jb : 0;
                    if (jb >= 3) { q = jb - 2; c_lo = split < nb ? split : nb; c_hi = nb; }
                } else if (jb >= 2) {
                    q = jb - 1;
                    // tiles of column k: (nb - k) of the matrix + (q + 1) of E; half of them, but columns jb + 1 and jb + 2 in any case
                    long total = 0, run = 0;
                    for (int k = jb + 1; k < nb; k++) total += (nb - k) + nE1 * (q + 1);
                    const long later = nb - jb - 2 > 0 ? (nb - jb - 2) + nE1 * (jb + 1) : 0;        // the odd launch's own tiles (column jb + 2)
                    split = jb + 1;
                    while (split < nb && (split < jb + 3 || 2 * run < total + later)) { run += (nb - split) + nE1 * (q + 1); split++; }
                    c_lo = jb + 1; c_hi = split;
                }
                long npair = 0;
                for (int k = c_lo; k < c_hi; k++) npair += (nb - k) + nE1 * (q + 1);
                hipLaunchKernelGGL(chol_pipe8_kernel<1>, dim3(nrow + nsingle + (int)npair), dim3(512), 0, s, work, out, Npad, jb, diag64, info_dev,
                                   nrow, Ework, Eout, nb, jb > 0 ? 1 : 0, nsingle, q, c_lo, c_hi);
            } else {
                hipLaunchKernelGGL(chol_pipe8_kernel<0>, dim3(nrow + ntile), dim3(512), 0, s, work, out, Npad, jb, diag64, info_dev, nrow,
                                   Ework, Eout, nb, jb > 0 ? 1 : 0, 0, 0, 0, 0);
            }
        }
        return (int)hipGetLastError();
    }
    // up to three block columns: a step has at most three trailing tiles and two of the ride-along -- one fused launch each
    for (int jb = 0; jb < nb; jb++) {
        const int m = nb - jb - 1, nchol = m * (m + 1) / 2;
        const int nextra = (Ework && m > 0) ? (jb + 1) * m : 0;
        if (m > 0)
            hipLaunchKernelGGL(chol_step8_kernel, dim3(nchol + nextra), dim3(512), 0, s, work, out, Npad, jb, diag64, info_dev, nchol,
                               Ework, Eout, nchol + nextra);
        else if (Ework)          // last block column with the ride-along: E's row blocks only need the multiplication by inv(L_jj)^T
            hipLaunchKernelGGL(chol_diag_trsm_kernel, dim3(nb), dim3(256), 0, s, work, out, Npad, jb, diag64, info_dev, 0, Ework, Eout);
        else
            hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), 0, s, work, Npad, jb, diag64, info_dev, (size_t)0, (size_t)0, out);
    }
    return (int)hipGetLastError();
}

// ---- The single-level order in SUPER-PANELS (round 6; fits from kSuperFrom block columns on, with the ride-along).
// Plain right-looking, every trailing tile makes a round trip through the fabric per two steps: at N = 4096 the tiles, not the chain, are
// what a block column costs (8.8 us of CU time per pair tile, fabric-side traffic 13.7 x the algorithmic bytes).  Here the
// pipelined launches (chol_pipe8_kernel<1>, unchanged) keep their tiles inside a super-panel of kSuperPanel block columns -- the rows are
// all there, matrix and E alike --, and after a super-panel the columns beyond it take its sixteen steps in ONE deep pass (K = 1024) from
// packed operands on chol_update3_kernel: [A ; E] is one tall matrix (2 Npad rows), its packed store [out ; Eout] likewise, so E's tiles are
// simply the tall matrix's rows beyond Npad -- rows of E from which nothing can come yet (>= 64 c1) are not live, and Eout's never-written
// blocks left of a row's diagonal are packed as zeros.  A column still receives its steps in ascending order, each super-panel's as one
// ascending K loop with accumulators that start as -C (C = -acc): L and W are BIT-IDENTICAL to the step-by-step order (tested), so the
// switch is a matter of speed only -- 2.04 against 2.10 ms at N = 4096, 3.18 / 3.48 at 5000, 4.83 / 5.56 at 6144; slower below 4096 rows.
// Where the time goes at N = 4096 (profiles/r06_fit4096_super_timeline.txt): the 64 pipelined launches 1160 us (17-18 us each: the chain),
// three deep updates 263 + 254 + 140 us (58 TFLOP/s where the chip is full: 492 tiles; 392 tiles leave half the CUs with one workgroup and
// last as long; 228 are one round of one per CU), packing 48, covariance 39,
// transpose + alpha 125.  Measured and not kept: the later super-panels' part of an update on a second stream beside the next chain (its
// 64 KB workgroups hold the CUs the chain's 133 KB workgroups need: the next super-panel's first launch waits for them, 2.07 ms), panels
// of 8 / 24 / 32 block columns (2.21 / 2.08 / 2.17 ms).
static void pipe8_launch_in_panel(double *work, double *out, int Npad, int jb, int c0, int c1, int &split, double *diag64, int *info_dev, hipStream_t s,
                                  double *Ework, double *Eout)
{
    const int nb = Npad / 64;
    const int m = nb - jb - 1, nE = jb + 1;
    const int nrow = m + nE + 1;
    int nsingle = 0, q = 0, c_lo = 0, c_hi = 0;
    if (jb & 1) {
        if (jb + 1 < c1) nsingle = m + jb;                     // step jb - 1 on column jb + 1 (the next super-panel's first column takes it deep)
        if (jb >= c0 + 3) { q = jb - 2; c_lo = split < c1 ? split : c1; c_hi = c1; }
    } else if (jb >= c0 + 2) {
        q = jb - 1;
        long total = 0, run = 0;
        for (int k = jb + 1; k < c1; k++) total += (nb - k) + (q + 1);
        const long later = jb + 2 < c1 ? (nb - jb - 2) + (jb + 1) : 0;
        split = jb + 1;
        while (split < c1 && (split < jb + 3 || 2 * run < total + later)) { run += (nb - split) + (q + 1); split++; }
        c_lo = jb + 1; c_hi = split;
    } else split = c1;                                         // a super-panel's first launch: no pair is due yet
    long npair = 0;
    for (int k = c_lo; k < c_hi; k++) npair += (nb - k) + (q + 1);
    hipLaunchKernelGGL(chol_pipe8_kernel<1>, dim3(nrow + nsingle + (int)npair), dim3(512), 0, s, work, out, Npad, jb, diag64, info_dev,
                       nrow, Ework, Eout, nb, jb > c0 ? 1 : 0, nsingle, q, c_lo, c_hi);
}

int launch_cholesky_super(double *tall, double *out, int Npad, double *diag64, int *info_dev, hipStream_t s, double *Eout, double *Pk,
                          bool info_is_zero)
{
    const int S = kSuperPanel, nb = Npad / 64;
    const size_t nn = (size_t)Npad * Npad;
    double *work = tall, *Ework = tall + nn, *PkE = Pk + nn;
    if (!info_is_zero) HIPCHK(hipMemsetAsync(info_dev, 0, sizeof(int), s));
    for (int c0 = 0; c0 < nb; c0 += S) {
        const int c1 = c0 + S < nb ? c0 + S : nb;
        int split = c1;
        for (int jb = c0; jb < c1; jb++) pipe8_launch_in_panel(work, out, Npad, jb, c0, c1, split, diag64, info_dev, s, Ework, Eout);
        if (c1 >= nb) break;
        // the finished columns in fragment order (the matrix's rows below the super-panel, E's rows 0 .. 64 c1), then every later column's update
        int rc = launch_chol_pack3(out, Npad, 64 * c1, 64 * c0, 64 * (c1 - c0), 1, 0, Pk, 0, s);
        if (rc) return rc;
        rc = launch_chol_pack3e(Eout, Npad, 64 * c1, 64 * c0, 64 * (c1 - c0), PkE, s);
        if (rc) return rc;
        rc = launch_chol_update3_range(tall, Npad, 64 * c1, 64 * (nb - c1), 64 * c0, 64 * c1, Npad + 64 * c1, 1, 0, Pk, 0, s);
        if (rc) return rc;
    }
    return (int)hipGetLastError();
}

// The same out-of-place scheme in the TWO-LEVEL order (panels of P block columns; one matrix of >= 104 blocks): inside a panel every
// block column is one chol_step8_kernel launch over the tiles (i, k), jb < k < pend, k <= i < nb, instead of the diagonal / row-block /
// update launches (21.4 -> 18.5 us per column at N = 4096); a panel's last column has nothing to update inside the panel and keeps its
// one launch; then the K = 64 P update of the matrix right of the panel, its operands read from the finished columns in `out`.  The
// arithmetic and its order are those of launch_cholesky_batched with the same P: identical bits.
int launch_cholesky_fused2(double *work, double *out, int Npad, double *diag64, int *info_dev, int P, hipStream_t s, bool info_is_zero, double *ws)
{
    const int nb = Npad / 64;
    if (!info_is_zero) HIPCHK(hipMemsetAsync(info_dev, 0, sizeof(int), s));
    for (int p0 = 0; p0 < nb; p0 += P) {
        const int pend = p0 + P < nb ? p0 + P : nb;
        for (int jb = p0; jb < pend; jb++) {
            int nt = 0;
            for (int k = jb + 1; k < pend; k++) nt += nb - k;
            if (nt > 0 && nt <= kStepSplit) {
                hipLaunchKernelGGL(chol_step8_kernel, dim3(nt < 256 ? nt : 256), dim3(512), 0, s, work, out, Npad, jb, diag64,
                                   info_dev, nt, (double *)nullptr, (double *)nullptr, nt);
            } else if (nt > 0) {           // more in-panel tiles than CUs: row blocks first (once each, with the chain), then one product per tile
                hipLaunchKernelGGL(chol_diag_trsm_kernel, dim3(nb - jb - 1), dim3(256), 0, s, work, out, Npad, jb, diag64, info_dev, nb - jb - 1,
                                   (const double *)nullptr, (double *)nullptr);
                hipLaunchKernelGGL(chol_update_step_kernel, dim3(nt), dim3(256), 0, s, work, out, Npad, jb, nt, (double *)nullptr,
                                   (const double *)nullptr);
            } else if (jb + 1 < nb) {
                hipLaunchKernelGGL(chol_diag_trsm_kernel, dim3(nb - jb - 1), dim3(256), 0, s, work, out, Npad, jb, diag64, info_dev, nb - jb - 1,
                                   (const double *)nullptr, (double *)nullptr);
            } else {                       // the matrix's last block column
                hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), 0, s, work, Npad, jb, diag64, info_dev,
                                   (size_t)0, (size_t)0, out);
            }
        }
        if (pend < nb) {
            const int nI2 = (Npad - 64 * pend + 127) / 128;
            if (ws && nI2 * (nI2 + 1) / 2 >= kPackedUpdateMinTiles) {
                int rc = launch_chol_update2(work, Npad, p0, pend, 1, 0, ws, 0, s, out);
                if (rc) return rc;
            } else launch_update(work, Npad, p0, pend, pend, nb, 1, 0, s, out);
        }
    }
    return (int)hipGetLastError();
}

int launch_cholesky(double *L, int Npad, double *diag64, int *info_dev, hipStream_t s, double *ws)
{
    return launch_cholesky_batched(L, Npad, diag64, info_dev, 1, 0, Npad / 64 > 32 ? 4 : 1, s, ws, 0);
}

// ------------------------------------------------------------------------
// W = L^-1 by recursive doubling over 64-blocks:
//   [L11 0; L21 L22]^-1 = [W11 0; -W22 L21 W11, W22]
// level s: nodes of 2s blocks; T = L21 W11, then W21 = -W22 T.
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void trinv_place_diag_kernel(const double *__restrict__ diag64,
                                                               double *__restrict__ W, int Npad)
{
    int jb = blockIdx.x;
    const double *Db = diag64 + (size_t)jb * 4096;
    double *Wb = W + (size_t)jb * 64 * Npad + jb * 64;
    for (int e = threadIdx.x; e < 4096; e += 256) Wb[(size_t)(e >> 6) * Npad + (e & 63)] = Db[e];
}

// acc += sum over 64-blocks kb in [kb0, kb1) of A[:, kb] * B[kb, :]  (A, B row-major 64-row strips; the
// 64x64 tiles of stage kb+1 are in flight while stage kb is on the MFMAs).  Bs is 64 x TNN_LD: with a row
// stride of 80 doubles the four k-rows of a B fragment fall on disjoint bank halves.
// NW = 4: 2 x 2 waves of 32 x 32 (acc[2][2]).  NW = 8: 4 x 2 waves of 16 x 32 (acc[1][2]) -- a wave issues an fp64 MFMA
// every ~118 cycles at best, so a tile that has its CU to itself (the lower levels of the doubling: fewer tiles than
// CUs, and the tile with the longest K range IS the launch) takes half the time per stage on eight waves.  Every output
// element sees the same MFMAs in the same order either way: identical bits.
#define TNN_LD 80
template <int NW>
__device__ __forceinline__ void tile64_gemm_nn(const double *__restrict__ A, int lda, const double *__restrict__ B,
                                               int ldb, int kb0, int kb1, d4_t (&acc)[8 / NW][2], double *As, double *Bs)
{
    constexpr int MR = 8 / NW, NV = 32 / NW;            // row-blocks per wave; 16-byte loads per thread and tile
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wr = wv >> 1, wc = wv & 1;
    if (kb0 >= kb1) return;
    d2_t va[NV], vb[NV];
    auto fetch = [&](const double *P, int ld, d2_t (&v)[NV]) {
#pragma unroll
        for (int u = 0; u < NV; u++) v[u] = *(const d2_t *)(P + (size_t)(2 * NW * u + (t >> 5)) * ld + (t & 31) * 2);
    };
    auto stash_a = [&](const d2_t (&v)[NV]) {            // odd row stride: 8-byte stores
#pragma unroll
        for (int u = 0; u < NV; u++) {
            double *dst = As + (2 * NW * u + (t >> 5)) * T64_LD + (t & 31) * 2;
            dst[0] = v[u].x; dst[1] = v[u].y;
        }
    };
    auto stash_b = [&](const d2_t (&v)[NV]) {
#pragma unroll
        for (int u = 0; u < NV; u++) *(d2_t *)(Bs + (2 * NW * u + (t >> 5)) * TNN_LD + (t & 31) * 2) = v[u];
    };
    fetch(A + (size_t)kb0 * 64, lda, va);
    fetch(B + (size_t)kb0 * 64 * ldb, ldb, vb);
    for (int kb = kb0; kb < kb1; kb++) {
        stash_a(va);
        stash_b(vb);
        __syncthreads();
        if (kb + 1 < kb1) {
            fetch(A + (size_t)(kb + 1) * 64, lda, va);
            fetch(B + (size_t)(kb + 1) * 64 * ldb, ldb, vb);
        }
#pragma unroll
        for (int k4 = 0; k4 < 16; k4++) {
            double a[MR], b[2];
#pragma unroll
            for (int m = 0; m < MR; m++) a[m] = As[(wr * 16 * MR + m * 16 + (lane & 15)) * T64_LD + k4 * 4 + (lane >> 4)];
#pragma unroll
            for (int n = 0; n < 2; n++) b[n] = Bs[(k4 * 4 + (lane >> 4)) * TNN_LD + wc * 32 + n * 16 + (lane & 15)];
#pragma unroll
            for (int m = 0; m < MR; m++)
#pragma unroll
                for (int n = 0; n < 2; n++) acc[m][n] = mfma_f64(a[m], b[n], acc[m][n]);
        }
        if (kb + 1 < kb1) __syncthreads();
    }
}
// row of accumulator (m, q) / column of accumulator n in the 64 x 64 tile, NW-wave layout
#define TNN_ROW(m, q) (wr * 16 * (8 / NW) + (m) * 16 + (lane >> 4) + 4 * (q))
#define TNN_COL(n) (wc * 32 + (n) * 16 + (lane & 15))

template <int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, NW == 4 ? 2 : 4)))
void trinv_T_kernel(const double *__restrict__ L, const double *__restrict__ W, double *__restrict__ T, int Npad,
                    int s, int nb)
{
    __shared__ double As[64 * T64_LD];
    __shared__ double Bs[64 * TNN_LD];
    TILE_IDS;
    int o = blockIdx.y * 2 * s;
    int r = min(s, nb - o - s);
    int tj = blockIdx.x / s, ti = blockIdx.x % s;       // longest K ranges (small tj) first: the short ones fill the tail
    if (ti >= r) return;
    const double *A = L + (size_t)(o + s + ti) * 64 * Npad + (size_t)o * 64;
    const double *B = W + (size_t)o * 64 * Npad + (size_t)(o + tj) * 64;
    d4_t acc[8 / NW][2] = {};
    tile64_gemm_nn<NW>(A, Npad, B, Npad, tj, s, acc, As, Bs);           // W11 is lower triangular: k-blocks >= tj
    double *C = T + (size_t)(o + s + ti) * 64 * Npad + (size_t)(o + tj) * 64;
#pragma unroll
    for (int m = 0; m < 8 / NW; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int q = 0; q < 4; q++) C[(size_t)TNN_ROW(m, q) * Npad + TNN_COL(n)] = acc[m][n][q];
}

template <int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, NW == 4 ? 2 : 4)))
void trinv_W_kernel(double *__restrict__ W, const double *__restrict__ T, int Npad, int s, int nb)
{
    __shared__ double As[64 * T64_LD];
    __shared__ double Bs[64 * TNN_LD];
    TILE_IDS;
    int o = blockIdx.y * 2 * s;
    int r = min(s, nb - o - s);
    int ti = s - 1 - blockIdx.x / s, tj = blockIdx.x % s;   // longest K ranges (large ti) first
    if (ti >= r) return;
    const double *A = W + (size_t)(o + s + ti) * 64 * Npad + (size_t)(o + s) * 64;
    const double *B = T + (size_t)(o + s) * 64 * Npad + (size_t)(o + tj) * 64;
    d4_t acc[8 / NW][2] = {};
    tile64_gemm_nn<NW>(A, Npad, B, Npad, 0, ti + 1, acc, As, Bs);       // W22 is lower triangular: k-blocks <= ti
    double *C = W + (size_t)(o + s + ti) * 64 * Npad + (size_t)(o + tj) * 64;
#pragma unroll
    for (int m = 0; m < 8 / NW; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int q = 0; q < 4; q++) C[(size_t)TNN_ROW(m, q) * Npad + TNN_COL(n)] = -acc[m][n][q];
}

int launch_trinv(const double *L, int Npad, const double *diag64, double *W, double *T, hipStream_t s, bool zero_fill)
{
    int nb = Npad / 64;
    // the doubling only ever reads and writes blocks on or below the diagonal; the zeros above it are for
    // consumers that take W as a full matrix (the fit path re-writes all of W in pack_w_kernel instead)
    if (zero_fill) HIPCHK(hipMemsetAsync(W, 0, sizeof(double) * (size_t)Npad * Npad, s));
    hipLaunchKernelGGL(trinv_place_diag_kernel, dim3(nb), dim3(256), 0, s, diag64, W, Npad);
    for (int sz = 1; sz < nb; sz *= 2) {
        int nodes = (nb + 2 * sz - 1) / (2 * sz);
        dim3 grid(sz * sz, nodes);
        if (sz * sz * nodes <= 512) {       // a level whose longest tile has a CU to itself: eight waves (29 -> 25 us for 256 tiles)
            hipLaunchKernelGGL(trinv_T_kernel<8>, grid, dim3(512), 0, s, L, W, T, Npad, sz, nb);
            hipLaunchKernelGGL(trinv_W_kernel<8>, grid, dim3(512), 0, s, W, T, Npad, sz, nb);
        } else {
            hipLaunchKernelGGL(trinv_T_kernel<4>, grid, dim3(256), 0, s, L, W, T, Npad, sz, nb);
            hipLaunchKernelGGL(trinv_W_kernel<4>, grid, dim3(256), 0, s, W, T, Npad, sz, nb);
        }
    }
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------
// A^-1 = W^T W for A = L L^T (W = L^-1 lower triangular): transpose, then one tile GEMM
//   Ainv[i][j] = sum_{k >= max(i,j)} W[k][i] W[k][j]
// ------------------------------------------------------------------------
__global__ void transpose_kernel(const double *__restrict__ A, double *__restrict__ At, int Npad)
{
    __shared__ double tile[64][65];
    int bx = blockIdx.x * 64, by = blockIdx.y * 64;
    for (int e = threadIdx.x; e < 4096; e += 256) {
        int r = e >> 6, c = e & 63;
        tile[r][c] = A[(size_t)(by + r) * Npad + bx + c];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 4096; e += 256) {
        int r = e >> 6, c = e & 63;
        At[(size_t)(bx + r) * Npad + by + c] = tile[c][r];
    }
}

template <int NW>                                               // four or eight waves per 64 x 64 tile: the same MFMAs in the same order per element
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, NW == 4 ? 2 : 4)))
void wtw_kernel(const double *__restrict__ Wt, const double *__restrict__ W, double *__restrict__ C, int Npad, int lower_only, int nsb)
{
    __shared__ double As[64 * T64_LD];
    __shared__ double Bs[64 * TNN_LD];
    TILE_IDS;
    int ti, tj;
    if (nsb > 0) {
        // XCD-aware tile order (as chol_update_kernel): workgroup b runs on XCD b % 8 and that XCD's 64 consecutive workgroups take one 8 x 8
        // super-block of tiles -- 8 strips of W^T and 8 of W serve 64 tiles out of that XCD's L2 instead of every tile pulling its own 64 KiB per
        // stage through the fabric (at N = 4096: 2.9 GB in 0.8 ms, which is what the fabric gives).  Super-blocks by rows, the longest K ranges first.
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        const int sb = (q >> 6) * 8 + xcd, lt = q & 63;
        if (sb >= nsb) return;
        const int nsr = (Npad / 64 + 7) / 8;
        int SI, SJ;
        if (lower_only) { SI = 0; int rem = sb; while (rem > SI) { rem -= SI + 1; SI++; } SJ = rem; }       // (SI, SJ <= SI) row by row
        else { SI = sb / nsr; SJ = sb % nsr; }
        ti = 8 * SI + (lt >> 3); tj = 8 * SJ + (lt & 7);
        if (ti >= Npad / 64 || tj >= Npad / 64) return;
    } else { ti = blockIdx.y; tj = blockIdx.x; }
    if (lower_only && tj > ti) return;                          // the caller reads C[max(i,j)][min(i,j)] (C is symmetric, bit for bit)
    const double *A = Wt + (size_t)ti * 64 * Npad;              // rows i of W^T, all k
    const double *B = W + (size_t)tj * 64;                      // columns j of W
    d4_t acc[8 / NW][2] = {};
    tile64_gemm_nn<NW>(A, Npad, B, Npad, max(ti, tj), Npad / 64, acc, As, Bs);     // W is lower triangular: k >= max(i, j)
    double *Ct = C + (size_t)ti * 64 * Npad + (size_t)tj * 64;
#pragma unroll
    for (int m = 0; m < 8 / NW; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int q = 0; q < 4; q++) Ct[(size_t)TNN_ROW(m, q) * Npad + TNN_COL(n)] = acc[m][n][q];
}

// wt_ready: Wt already holds W^T on and right of the diagonal blocks (the ride-along's (L^-1)^T as the factorisation leaves it: the blocks
// left of the diagonal, which it never writes, are never read here) -- no transpose pass
int launch_wtw(const double *W, double *Wt, double *C, int Npad, hipStream_t s, int lower_only, int wt_ready)
{
    dim3 g(Npad / 64, Npad / 64);
    if (!wt_ready) hipLaunchKernelGGL(transpose_kernel, g, dim3(256), 0, s, W, Wt, Npad);
    int nsb = 0;
    if (Npad / 64 >= 32) {                                      // enough tiles that the operands do not stay in L2 by themselves
        const int nsr = (Npad / 64 + 7) / 8;
        nsb = lower_only ? nsr * (nsr + 1) / 2 : nsr * nsr;
        g = dim3((unsigned)((nsb + 7) / 8) * 512);
    }
    hipLaunchKernelGGL(wtw_kernel<8>, g, dim3(512), 0, s, Wt, W, C, Npad, lower_only, nsb);
    return (int)hipGetLastError();
}


void ibo_touch_linalg() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, (const void *)chol_diag_kernel); }     // (see small2.hip: ibo_touch_small2)
