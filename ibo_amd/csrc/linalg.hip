// linalg.hip -- fit-side kernels for gfx950: covariance assembly, blocked
// Cholesky, explicit triangular inverse, MFMA-fragment packing, alpha vectors.
//
// Replaces (reference, /root/reference):
//   GaussianProcess._computeCorrelations   ego/gaussianprocess/__init__.py:134-149
//   Kernel.covMatrix                       ego/gaussianprocess/kernel.py:46-53
//   linalg.cholesky(R)                     ego/gaussianprocess/__init__.py:299
//   linalg.inv(R) per maximize* call       ego/acquisition/__init__.py:385-388
//
// Layout: every N x N matrix lives row-major with leading dimension Npad
// (a multiple of 64) so that all tile kernels run without bounds checks; the
// pad is the identity for matrices that get factored and zero for W.
#include "ibo_common.h"
#include <atomic>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

// ------------------------------------------------------------------------
// covariance matrix  K[i][j] = k(A1_i, A2_j)
// ------------------------------------------------------------------------
// 64 x 64 entries of K per workgroup, 4 x 4 per thread: the tile's points are staged in LDS once and every value read
// from there serves four entries (per dimension 8 LDS reads and 48 fp64 instructions for 16 entries; the first
// version, one column point against four row points, was bound by its LDS reads: 55 us for the lower triangle of a
// 4096 x 4096 matrix in 16 dimensions against 25 now).  A thread's columns are 16 apart: a row's store instruction
// covers whole 128-byte segments.
// (z is accumulated in the reference's order: sum_d w_d (a_d - b_d)^2.)
// row stride of the staged points: odd (conflict-free column reads), >= the dimension: 33 up to 32 dimensions, 65 beyond (a
// template parameter: the wider stride halves the workgroups a CU holds)
#define COV_LD LD
// FAST (the marginal-likelihood grid, whose matrices never leave the device): coordinates scaled by sqrt(w_d) on their way
// into LDS (two instead of three fp64 instructions per dimension and entry) and the 16-instruction exp_fast / 7-instruction
// sqrt_fast of the sweep (relative error < 5e-16) instead of the library's -- 52 instead of 88 instructions per entry at
// D = 16.  GP.R and everything a caller can read back keep the reference's order of operations (FAST = false).
template <bool FAST, int LD>
__global__ __launch_bounds__(256) void cov_matrix_kernel(KParams kp1, int n1, const double *__restrict__ A1, int n2,
                                                         const double *__restrict__ A2, int ldp, int square,
                                                         int diag_rule, double noise, double *__restrict__ K, int ldk,
                                                         double *__restrict__ K2, int np2, int lower_only,
                                                         double *__restrict__ Eye, int *__restrict__ zero_word,
                                                         const KParams *__restrict__ kps, size_t kstride)
{
    // one matrix per blockIdx.z, its kernel parameters in kps[z] (device) and its output kstride doubles on: the
    // likelihood grid's matrices in ONE launch (2145 workgroups per matrix do not fill the chip for long; 64 launches
    // of 42 us each were 9 % of a grid)
    const KParams &kp = kps ? kps[blockIdx.z] : kp1;
    if (kps) K += blockIdx.z * kstride;
    __shared__ double AB[2 * 64 * COV_LD];           // the two tiles' points; afterwards the tile itself, transposed (64 x 65)
    double *As = AB, *Bs = AB + 64 * COV_LD;
    static_assert(2 * 64 * LD >= 64 * 65, "the transposed tile reuses the staging buffers");
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int j0 = blockIdx.x * 64, i0 = blockIdx.y * 64, D = kp.D;
    if (zero_word && t == 0 && blockIdx.x == 0 && blockIdx.y == 0) *zero_word = 0;
    // K(X, X) is symmetric bit for bit ((a - b)^2 = (b - a)^2): a tile below the diagonal also writes its mirror image,
    // the tiles above the diagonal compute nothing (half the fp64 exps; at N = 4096 the pass is then bound by its 200 MB of writes: 46 us)
    const bool mirror = square && !lower_only && K && j0 < i0;
    const bool skip = square && j0 > i0;             // (lower_only: a factorisation only reads the lower triangle)
    if (j0 > i0) K2 = nullptr;                       // ... so the working copy gets no blocks above the diagonal (67 MB less at N = 4096)
    if (Eye && j0 >= i0) {                           // an np2 x np2 identity in the same pass (the fit's ride-along rows: tile (i, k) of
                                                     // E is read by steps i <= j < k only -- its blocks left of the diagonal never)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int i = i0 + ty * 4 + r, j = j0 + tx + 16 * c;
                if (i < np2 && j < np2) Eye[(size_t)i * np2 + j] = (i == j) ? 1.0 : 0.0;
            }
    }
    if (skip) return;
    for (int e = t; e < 64 * D; e += 256) {
        const int r = e / D, d = e - r * D;
        const double sc = FAST ? kp.sw[d] : 1.0;
        As[r * COV_LD + d] = (i0 + r < n1) ? A1[(size_t)(i0 + r) * ldp + d] * sc : 0.0;
        Bs[r * COV_LD + d] = (j0 + r < n2) ? A2[(size_t)(j0 + r) * ldp + d] * sc : 0.0;
    }
    __syncthreads();
    double z[4][4] = {};
    for (int d = 0; d < D; d++) {
        const double w = kp.w[d];
        double a[4], b[4];
#pragma unroll
        for (int r = 0; r < 4; r++) a[r] = As[(ty * 4 + r) * COV_LD + d];
#pragma unroll
        for (int c = 0; c < 4; c++) b[c] = Bs[(tx + 16 * c) * COV_LD + d];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const double u = a[r] - b[c];
                if (FAST) z[r][c] = fma(u, u, z[r][c]);
                else z[r][c] += w * (u * u);
            }
    }
    const double log_sf2 = FAST ? log(kp.sf2) : 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int i = i0 + ty * 4 + r;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int j = j0 + tx + 16 * c;
            if (i < n1 && j < n2) {
                double v;
                if (FAST) {
                    v = kp.family == FAM_SE ? cov_from_z_fast<FAM_SE>(z[r][c], log_sf2, kp.sf2)
                        : (kp.family == FAM_M3 ? cov_from_z_fast<FAM_M3>(z[r][c], log_sf2, kp.sf2) : cov_from_z_fast<FAM_M5>(z[r][c], log_sf2, kp.sf2));
                    if (square && i == j) v = kp.sf2;            // k(x, x), exactly
                } else v = cov_from_z_rt(kp.family, z[r][c], kp.sf2);
                if (square && i == j) {
                    // diag_rule 0: the reference never calls the kernel on the diagonal and
                    // hard-wires 1+noise (ego/gaussianprocess/__init__.py:138)
                    v = (diag_rule == 0) ? (1.0 + noise) : (v + noise);
                }
                if (K) K[(size_t)i * ldk + j] = v;
                if (K2) K2[(size_t)i * np2 + j] = v;
                z[r][c] = v;
            } else if (K2 && i < np2 && j < np2) {
                K2[(size_t)i * np2 + j] = (i == j) ? 1.0 : 0.0;     // identity pad of the np2 x np2 working copy
            }
        }
    }
    if (mirror) {                                    // K[j][i] = K[i][j], written row by row from the transposed tile
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) AB[(tx + 16 * c) * 65 + ty * 4 + r] = z[r][c];
        __syncthreads();
        for (int e = t; e < 4096; e += 256) {
            const int rr = e >> 6, cc = e & 63;
            if (j0 + rr < n2 && i0 + cc < n1) K[(size_t)(j0 + rr) * ldk + i0 + cc] = AB[rr * 65 + cc];
        }
    }
}

// The FIT's covariance pass by itself (round 4): only what a factorisation reads -- the 64 x 64 blocks of K(X, X) + diag on and below the
// diagonal, padded with the identity to np2 x np2, into the working copy; the identity the W ride-along starts from (its blocks on and right of
// the diagonal) and the cleared info word in the same pass.  GP.R is formed on request (abi.hip ensure_R, cov_matrix_kernel).  Entry by entry
// the arithmetic of cov_matrix_kernel<false>: the same bits.  TS x TS entries per workgroup of 256 threads: with 64 x 64 tiles a 1024-point fit
// is 136 workgroups of four waves doing 16 entries per thread -- one wave per SIMD on half the chip, 14.7 us for 4 MB; 32 x 32 tiles put four
// times as many workgroups on it.
template <int LD, int TS>
__global__ __launch_bounds__(256) void cov_fit_kernel(KParams kp, int n, const double *__restrict__ X, int ldp, int diag_rule, double noise,
                                                      double *__restrict__ K2, int np2, double *__restrict__ Eye, int *__restrict__ zero_word)
{
    constexpr int R = TS / 16;
    __shared__ double As[TS * LD], Bs[TS * LD];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int j0 = blockIdx.x * TS, i0 = blockIdx.y * TS, D = kp.D;
    if (zero_word && t == 0 && blockIdx.x == 0 && blockIdx.y == 0) *zero_word = 0;
    const bool lower = j0 / 64 <= i0 / 64, upper = j0 / 64 >= i0 / 64;        // by 64 x 64 BLOCK: a diagonal block is written whole
    if (Eye && upper) {
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int c = 0; c < R; c++) {
                const int i = i0 + ty * R + r, j = j0 + tx + 16 * c;
                if (i < np2 && j < np2) Eye[(size_t)i * np2 + j] = (i == j) ? 1.0 : 0.0;
            }
    }
    if (!lower) return;
    for (int e = t; e < TS * D; e += 256) {
        const int r = e / D, d = e - r * D;
        As[r * LD + d] = (i0 + r < n) ? X[(size_t)(i0 + r) * ldp + d] : 0.0;
        Bs[r * LD + d] = (j0 + r < n) ? X[(size_t)(j0 + r) * ldp + d] : 0.0;
    }
    __syncthreads();
    double z[R][R] = {};
    for (int d = 0; d < D; d++) {
        const double w = kp.w[d];
        double a[R], b[R];
#pragma unroll
        for (int r = 0; r < R; r++) a[r] = As[(ty * R + r) * LD + d];
#pragma unroll
        for (int c = 0; c < R; c++) b[c] = Bs[(tx + 16 * c) * LD + d];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int c = 0; c < R; c++) {
                const double u = a[r] - b[c];
                z[r][c] += w * (u * u);
            }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = i0 + ty * R + r;
#pragma unroll
        for (int c = 0; c < R; c++) {
            const int j = j0 + tx + 16 * c;
            if (i < n && j < n) {
                double v = cov_from_z_rt(kp.family, z[r][c], kp.sf2);
                if (i == j) v = (diag_rule == 0) ? (1.0 + noise) : (v + noise);
                K2[(size_t)i * np2 + j] = v;
            } else if (i < np2 && j < np2) {
                K2[(size_t)i * np2 + j] = (i == j) ? 1.0 : 0.0;
            }
        }
    }
}
int launch_cov_fit(const KParams &kp, int n, const double *X, int ldp, int diag_rule, double noise, double *K2, int np2, double *Eye,
                   int *zero_word, hipStream_t s)
{
    if (np2 <= 2560) {
        dim3 grid(np2 / 32, np2 / 32);
        if (kp.D <= 32) hipLaunchKernelGGL((cov_fit_kernel<33, 32>), grid, dim3(256), 0, s, kp, n, X, ldp, diag_rule, noise, K2, np2, Eye, zero_word);
        else hipLaunchKernelGGL((cov_fit_kernel<65, 32>), grid, dim3(256), 0, s, kp, n, X, ldp, diag_rule, noise, K2, np2, Eye, zero_word);
    } else {
        dim3 grid(np2 / 64, np2 / 64);
        if (kp.D <= 32) hipLaunchKernelGGL((cov_fit_kernel<33, 64>), grid, dim3(256), 0, s, kp, n, X, ldp, diag_rule, noise, K2, np2, Eye, zero_word);
        else hipLaunchKernelGGL((cov_fit_kernel<65, 64>), grid, dim3(256), 0, s, kp, n, X, ldp, diag_rule, noise, K2, np2, Eye, zero_word);
    }
    return (int)hipGetLastError();
}

// K2 (optional, square case): a second, np2 x np2 copy of K padded with the identity -- the matrix the
// factorisation works on, written by the same kernel instead of a separate pad-and-copy pass: its blocks on and below
// the diagonal only.  K may be NULL when only the working copy is wanted.
#define COV_LAUNCH(FASTV, DIMS, GRID, STREAM, ...)                                                                         \
    do {                                                                                                              \
        if ((DIMS) <= 32) hipLaunchKernelGGL((cov_matrix_kernel<FASTV, 33>), GRID, dim3(256), 0, STREAM, __VA_ARGS__);        \
        else hipLaunchKernelGGL((cov_matrix_kernel<FASTV, 65>), GRID, dim3(256), 0, STREAM, __VA_ARGS__);                     \
    } while (0)

int launch_cov_matrix(const KParams &kp, int n1, const double *A1, int n2, const double *A2,
                      int ldp, int diag_rule, double noise, double *K, int ldk, hipStream_t s, double *K2, int np2, int lower_only,
                      double *Eye, int *zero_word, int fast)
{
    int square = (A2 == nullptr);
    if (square) { A2 = A1; n2 = n1; }
    const int c = K2 ? np2 : n2, r = K2 ? np2 : n1;
    dim3 grid((c + 63) / 64, (r + 63) / 64);
    if (fast)
        COV_LAUNCH(true, kp.D, grid, s, kp, n1, A1, n2, A2, ldp, square,
                           diag_rule, noise, K, ldk, K2, np2, square && !K2 ? lower_only : 0, K2 ? Eye : nullptr, zero_word,
                           (const KParams *)nullptr, (size_t)0);
    else
        COV_LAUNCH(false, kp.D, grid, s, kp, n1, A1, n2, A2, ldp, square,
                           diag_rule, noise, K, ldk, K2, np2, square && !K2 ? lower_only : 0, K2 ? Eye : nullptr, zero_word,
                           (const KParams *)nullptr, (size_t)0);
    return (int)hipGetLastError();
}

// The likelihood grid's covariance pass: lower part of K(X, X) + noise I for `batch` parameter sets, one launch.
// 64 x 128 entries per workgroup (4 x 8 per thread: 12 LDS reads per dimension for 32 entries), workgroups numbered over the
// tiles that touch the lower triangle only (a 2-D grid would dispatch as many dead workgroups as live ones), a row's 128
// columns leave in eight consecutive 128-byte stores -- 1 KiB per row and tile: with 64 x 64 tiles (512-byte row segments) the
// 4.3 GB of a 64-matrix grid went out at 1.7 TB/s.  Scaled coordinates, exp_fast / sqrt_fast as cov_matrix_kernel<true>.
template <int LD>
__global__ __launch_bounds__(256) void cov_grid_kernel(const KParams *__restrict__ kps, int n, const double *__restrict__ X, int ldp,
                                                       double noise, double *__restrict__ K, int ldk, size_t kstride)
{
    __shared__ double As[64 * COV_LD];
    __shared__ double Bs[128 * COV_LD];
    const KParams &kp = kps[blockIdx.z];
    K += blockIdx.z * kstride;
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4, D = kp.D;
    // tile (I, J): rows 64 I .., columns 128 J ..; row I has I / 2 + 1 tiles; rows 2 a, 2 a + 1 start at a (a + 1)
    const int q = blockIdx.x;
    int a = (int)((sqrt(1.0 + 4.0 * (double)q) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) <= q) a++;
    while (a * (a + 1) > q) a--;
    const int rem = q - a * (a + 1);
    const int I = 2 * a + rem / (a + 1), J = rem % (a + 1);
    const int i0 = 64 * I, j0 = 128 * J;
    if (i0 >= n) return;
    for (int e = t; e < 64 * D; e += 256) {
        const int r = e / D, d = e - r * D;
        As[r * COV_LD + d] = (i0 + r < n) ? X[(size_t)(i0 + r) * ldp + d] * kp.sw[d] : 0.0;
    }
    for (int e = t; e < 128 * D; e += 256) {
        const int r = e / D, d = e - r * D;
        Bs[r * COV_LD + d] = (j0 + r < n) ? X[(size_t)(j0 + r) * ldp + d] * kp.sw[d] : 0.0;
    }
    __syncthreads();
    double z[4][8] = {};
    for (int d = 0; d < D; d++) {
        double av[4], bv[8];
#pragma unroll
        for (int r = 0; r < 4; r++) av[r] = As[(ty * 4 + r) * COV_LD + d];
#pragma unroll
        for (int c = 0; c < 8; c++) bv[c] = Bs[(tx + 16 * c) * COV_LD + d];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const double u = av[r] - bv[c];
                z[r][c] = fma(u, u, z[r][c]);
            }
    }
    const double log_sf2 = log(kp.sf2);
    const int fam = kp.family;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int i = i0 + ty * 4 + r;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int j = j0 + tx + 16 * c;
            if (i < n && j < n && j <= (i | 63)) {         // the 64 x 64 blocks on and below the diagonal, as the factorisation reads them
                double v = fam == FAM_SE ? cov_from_z_fast<FAM_SE>(z[r][c], log_sf2, kp.sf2)
                           : (fam == FAM_M3 ? cov_from_z_fast<FAM_M3>(z[r][c], log_sf2, kp.sf2) : cov_from_z_fast<FAM_M5>(z[r][c], log_sf2, kp.sf2));
                if (i == j) v = kp.sf2 + noise;            // k(x, x) + noise, exactly
                K[(size_t)i * ldk + j] = v;
            }
        }
    }
}

// `batch` square covariance matrices K(A1, A1) (lower blocks only), parameters kps_dev[z] (device), outputs kstride doubles apart
int launch_cov_matrix_batched(const KParams *kps_dev, int batch, int n1, const double *A1, int ldp, int diag_rule, double noise,
                              double *K, int ldk, size_t kstride, hipStream_t s, int fast)
{
    dim3 grid((n1 + 63) / 64, (n1 + 63) / 64, batch);
    KParams dummy = KParams();
    if (fast && diag_rule == 1) {
        const int nI = (n1 + 63) / 64;                   // tiles: sum over rows I of I / 2 + 1
        long ntile = 0;
        for (int I = 0; I < nI; I++) ntile += I / 2 + 1;
        // (ldp = the dimension here: the points are handed over unpadded)
        if (ldp <= 32) hipLaunchKernelGGL(cov_grid_kernel<33>, dim3((unsigned)ntile, 1, batch), dim3(256), 0, s, kps_dev, n1, A1, ldp, noise, K, ldk, kstride);
        else hipLaunchKernelGGL(cov_grid_kernel<65>, dim3((unsigned)ntile, 1, batch), dim3(256), 0, s, kps_dev, n1, A1, ldp, noise, K, ldk, kstride);
        return (int)hipGetLastError();
    }
    if (fast)
        COV_LAUNCH(true, ldp, grid, s, dummy, n1, A1, n1, A1, ldp, 1, diag_rule, noise, K, ldk,
                           (double *)nullptr, 0, 1, (double *)nullptr, (int *)nullptr, kps_dev, kstride);
    else
        COV_LAUNCH(false, ldp, grid, s, dummy, n1, A1, n1, A1, ldp, 1, diag_rule, noise, K, ldk,
                           (double *)nullptr, 0, 1, (double *)nullptr, (int *)nullptr, kps_dev, kstride);
    return (int)hipGetLastError();
}

__global__ void scale_x_kernel(KParams kp, const double *__restrict__ Xp, int Npad, int DP,
                               double *__restrict__ Xs, double *__restrict__ ak)
{
    int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= Npad) return;
    double n2 = 0.0;
    for (int d = 0; d < DP; d++) {
        double v = (d < kp.D) ? Xp[(size_t)k * DP + d] * kp.sw[d] : 0.0;
        Xs[(size_t)k * DP + d] = v;
        n2 = fma(v, v, n2);
    }
    ak[k] = -0.5 * n2;
}

int launch_scale_x(const KParams &kp, const double *Xp, int Npad, int DP, double *Xs, double *ak, hipStream_t s)
{
    hipLaunchKernelGGL(scale_x_kernel, dim3((Npad + 255) / 256), dim3(256), 0, s, kp, Xp, Npad, DP, Xs, ak);
    return (int)hipGetLastError();
}

__global__ void pad_copy_kernel(const double *__restrict__ src, int N, int lds, double *__restrict__ dst,
                                int Npad, double pad_diag)
{
    int j = blockIdx.x * 64 + (threadIdx.x & 63);
    int i0 = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (j >= Npad || i0 >= Npad) return;
    int i = i0;
    double v;
    if (i < N && j < N) v = src[(size_t)i * lds + j];
    else v = (i == j) ? pad_diag : 0.0;
    dst[(size_t)i * Npad + j] = v;
}

int launch_pad_copy(const double *src, int N, int lds, double *dst, int Npad, double pad_diag, hipStream_t s)
{
    dim3 grid(Npad / 64, Npad / 4);
    hipLaunchKernelGGL(pad_copy_kernel, grid, dim3(256), 0, s, src, N, lds, dst, Npad, pad_diag);
    return (int)hipGetLastError();
}

__global__ void zero_upper_kernel(double *A, int Npad)
{
    int j = blockIdx.x * 64 + (threadIdx.x & 63);
    int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (j > i) A[(size_t)i * Npad + j] = 0.0;
}

int launch_zero_upper(double *A, int Npad, hipStream_t s)
{
    dim3 grid(Npad / 64, Npad / 4);
    hipLaunchKernelGGL(zero_upper_kernel, grid, dim3(256), 0, s, A, Npad);
    return (int)hipGetLastError();
}

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
// two such products with the same triangular B (X_i = A_i V^T and X_k = A_k V^T of a fused step): the B fragments are read
// once and the eight accumulators keep the MFMA pipe fed where one product's tail has only two
template <int LD, int CB0, int CB1>
__device__ __forceinline__ void tile64_mma_nt_tri2_body(const double *As1, const double *As2, const double *Bs, d4_t (&acc1)[2][2],
                                                        d4_t (&acc2)[2][2], int wr, int lane)
{
    double a1[16][2], a2[16][2], b0[4 * (CB0 + 1)], b1[4 * (CB1 + 1)];
#pragma unroll
    for (int k4 = 0; k4 < 4 * (CB1 + 1); k4++) {
#pragma unroll
        for (int m = 0; m < 2; m++) {
            a1[k4][m] = As1[(wr * 32 + m * 16 + (lane & 15)) * LD + k4 * 4 + (lane >> 4)];
            a2[k4][m] = As2[(wr * 32 + m * 16 + (lane & 15)) * LD + k4 * 4 + (lane >> 4)];
        }
        if (k4 < 4 * (CB0 + 1)) b0[k4] = Bs[(CB0 * 16 + (lane & 15)) * LD + k4 * 4 + (lane >> 4)];
        b1[k4] = Bs[(CB1 * 16 + (lane & 15)) * LD + k4 * 4 + (lane >> 4)];
    }
#pragma unroll
    for (int k4 = 0; k4 < 4 * (CB1 + 1); k4++) {
        if (k4 < 4 * (CB0 + 1)) {
#pragma unroll
            for (int m = 0; m < 2; m++) { acc1[m][0] = mfma_f64(a1[k4][m], b0[k4], acc1[m][0]); acc2[m][0] = mfma_f64(a2[k4][m], b0[k4], acc2[m][0]); }
        }
#pragma unroll
        for (int m = 0; m < 2; m++) { acc1[m][1] = mfma_f64(a1[k4][m], b1[k4], acc1[m][1]); acc2[m][1] = mfma_f64(a2[k4][m], b1[k4], acc2[m][1]); }
    }
}
template <int LD = T64_LD>
__device__ __forceinline__ void tile64_mma_nt_tri2(const double *As1, const double *As2, const double *Bs, d4_t (&acc1)[2][2],
                                                   d4_t (&acc2)[2][2])
{
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wr = wv >> 1, wc = wv & 1;
    if (wc) tile64_mma_nt_tri2_body<LD, 1, 2>(As1, As2, Bs, acc1, acc2, wr, lane);
    else tile64_mma_nt_tri2_body<LD, 0, 3>(As1, As2, Bs, acc1, acc2, wr, lane);
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
#ifdef IBO_STAMPS      // diagnostic build (tools/chol_diag_bench.hip): where does the diagonal block's time go?
__device__ unsigned long long g_chol_stamps[32];
#define CSTAMP(i) do { if (threadIdx.x == 0) g_chol_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ unsigned long long g_upd_stamps[4][32];      // update kernel: 4 sampled workgroups of batch member 0
#define USTAMP(i) do { if (threadIdx.x == 0 && blockIdx.z == 0 && (blockIdx.x & 1023) == 8 && (blockIdx.x >> 10) < 4) \
        g_upd_stamps[blockIdx.x >> 10][i] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ unsigned long long g_step_stamps[2][16];      // step kernel: workgroup 0 and workgroup 7
#define SSTAMP(i) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 7)) \
        g_step_stamps[blockIdx.x ? 1 : 0][i] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ unsigned long long g_pipe_stamps[2][16];      // chol_pipe8_kernel: row workgroups 0 and 1 of block column 8
#ifdef IBO_NO_PSTAMP
#define PSTAMP(i)
#else
#define PSTAMP(i) do { if (threadIdx.x == 0 && jb == 8 && blockIdx.x < 2) g_pipe_stamps[blockIdx.x][i] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#else
#define CSTAMP(i)
#define USTAMP(i)
#define SSTAMP(i)
#define PSTAMP(i)
#endif
__device__ __forceinline__ double lane_bcast(double x, int l)          // value of lane l, wave-uniform
{
    int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

// one wave: 16x16 (+)= A(16xK) * B(Kx16) with both operands in LDS (row stride SD).
// A element (r,k) = Am[r*SD + k];  B element (k,c) = TB ? Bm[c*SD + k] : Bm[k*SD + c].
// Result element r of the returned vector is row (lane>>4)+4r, column lane&15.
template <bool TB, int K>
__device__ __forceinline__ d4_t lds_mm16(const double *Am, const double *Bm)
{
    const int lane = threadIdx.x & 63, ar = lane & 15, q = lane >> 4;
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    double a[K / 4], b[K / 4];
#pragma unroll
    for (int s = 0; s < K / 4; s++) {               // all LDS reads in flight before the first MFMA
        a[s] = Am[ar * SD + 4 * s + q];
        b[s] = TB ? Bm[ar * SD + 4 * s + q] : Bm[(4 * s + q) * SD + ar];
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
__device__ __forceinline__ void diag64_stash(const double (&v)[16], double *S, double *V, double *T)
{
    const int t = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 16; u++) {
        S[(4 * u + (t >> 6)) * SD + (t & 63)] = v[u];
        V[(4 * u + (t >> 6)) * SD + (t & 63)] = 0.0;
    }
    T[(t >> 4) * SD + (t & 15)] = ((t >> 4) == (t & 15)) ? 1.0 : 0.0;       // diag64_dpp.h: the identity rows
}
__device__ __forceinline__ void diag64_load(const double *Lb, int Npad, double *S, double *V, double *T)
{
    double v[16];
    diag64_fetch(Lb, Npad, v);
    diag64_stash(v, S, V, T);
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

__global__ __launch_bounds__(256) void chol_diag_kernel(double *L, int Npad, int jb,
                                                        double *__restrict__ diag64, int *info,
                                                        size_t lstride, size_t dstride, double *Lout)
{
    __shared__ double S[64 * SD];          // the block; ends up holding L (lower)
    __shared__ double V[64 * SD];          // its inverse
    __shared__ double T[64 * SD];          // scratch (L21 * V11 products)
    L += blockIdx.z * lstride; diag64 += blockIdx.z * dstride; info += blockIdx.z;      // batch member
    if (!Lout) Lout = L; else Lout += blockIdx.z * lstride;
    const size_t off = (size_t)jb * 64 * Npad + jb * 64;
    CSTAMP(31);
    diag64_load(L + off, Npad, S, V, T);
    CSTAMP(0);
    __syncthreads();
    CSTAMP(1);
    diag64_factor_invert(S, V, T, jb * 64, info);
    CSTAMP(22);
    diag64_store(Lout + off, Npad, diag64 + (size_t)jb * 4096, S, V);
    CSTAMP(23);
}

// OUT OF PLACE: the matrix being reduced (A) is only read in its panel column and updated in its trailing
// tiles (each by exactly one workgroup); the factor goes to a second matrix (Lout).  Overwriting the panel in
// place would race with the workgroups that still have to read it.
// One launch per block column for the plain right-looking order (one matrix, the fit path): every workgroup
// of the trailing update first repeats the 64x64 factorisation of the diagonal block (the chain, ~15 us, the
// same on 120 workgroups as on one), then forms the two row blocks of L it needs (A_i V^T, A_k V^T: what
// chol_trsm_kernel does) and updates its tile.  Workgroup 0 stores the diagonal block and its inverse, the
// workgroups of the first trailing column store their row block of L.  Two launches and their gaps per
// column are gone; every product is rounded to fp64 at the same points as in the three-kernel sequence, so
// the result is bit-identical to it.
// W = L^-1 rides along: with E = I appended below the matrix, the factorisation's own "row block times inv(L_jj)^T,
// then update the trailing tiles" turns E into (L^-1)^T block column by block column -- L21 L11^T = I.  Tile (i, k) of E
// is touched by step jb only for i <= jb < k (row block i of E is zero left of its diagonal block until then), so a
// step carries (jb + 1)(nb - jb - 1) extra tiles, run by the same code on CUs the factorisation leaves idle; the
// recursive-doubling inversion (2 log2(nb) launches after the factorisation) disappears.
// Tiles >= nchol are extra tiles: number e -> (i = e / m, k = jb + 1 + e % m), A_i from Ework, X_i to Eout.
// HAVE_V: the diagonal block was factored by an earlier launch; its inverse is read from diag64 instead.
// A workgroup takes the tiles blockIdx.x, blockIdx.x + gridDim.x, ... < ntiles: with more tiles than CUs (N = 2048 with
// the ride-along: up to 496) the second tile of a workgroup reuses the inverse that is already in its LDS -- its operands
// are requested before the first tile's products start -- where a second launch would pay for launch, fetch and (without
// diag64) the chain again.
template <bool HAVE_V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void chol_step_kernel(double *__restrict__ L, double *__restrict__ Lout, int Npad, int jb,
                      double *__restrict__ diag64, int *info, int nchol, int extra0, double *__restrict__ Ework,
                      double *__restrict__ Eout, int ntiles)
{
    __shared__ double S[64 * SD];
    __shared__ double V[64 * SD];
    __shared__ double T[64 * SD];
    TILE_IDS;
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
            const int e = t - nchol + extra0;
            i = e / m; q.k = jb + 1 + e % m;
            q.Ai = Ework + (size_t)i * 64 * Npad + jb * 64;
            q.Xi = Eout + (size_t)i * 64 * Npad + jb * 64;
            q.C = Ework + (size_t)i * 64 * Npad + q.k * 64;
        }
        q.Ak = L + (size_t)q.k * 64 * Npad + jb * 64;
        return q;
    };
    auto fetch = [&](const Tile &q, d2_t (&va)[8], d2_t (&vb)[8], d4_t (&c)[2][2]) {
        tile64_fetch(q.Ai, Npad, va);
        tile64_fetch(q.Ak, Npad, vb);
#pragma unroll
        for (int mm = 0; mm < 2; mm++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) c[mm][n][r] = q.C[(size_t)TILE_ROW(mm, r) * Npad + TILE_COL(n)];
    };
    const size_t doff = (size_t)jb * 64 * Npad + jb * 64;
    int t = blockIdx.x;
    Tile cur = decode(t);
    // everything this workgroup will need from memory is requested before the chain starts
    SSTAMP(0);
    double vd[16];
    d2_t vv[8];
    if (HAVE_V) tile64_fetch(diag64 + (size_t)jb * 4096, 64, vv);
    else diag64_fetch(L + doff, Npad, vd);              // first: the chain starts when these are back
    d2_t va[8], vb[8];
    d4_t c[2][2];
    fetch(cur, va, vb, c);
    if (HAVE_V) {
        tile64_stash<false, SD>(V, vv);
    } else {
        diag64_stash(vd, S, V, T);
        __syncthreads();
        SSTAMP(1);
        diag64_factor_invert(S, V, T, jb * 64, blockIdx.x == 0 ? info : nullptr);
        SSTAMP(2);
        if (blockIdx.x == 0) diag64_store(Lout + doff, Npad, diag64 + (size_t)jb * 4096, S, V);
    }
    for (;;) {
        __syncthreads();                                   // S (and T) are about to be reused
        SSTAMP(3);
        // X_i = A_i V^T, X_k = A_k V^T  (V[c][k] row-major is the "B^T" operand as it stands)
        tile64_stash<false, SD>(S, va);
        tile64_stash<false, SD>(T, vb);
        const int tn = t + gridDim.x;
        const bool more = tn < ntiles;
        Tile nxt = cur;
        d4_t cn[2][2];
        if (more) {                                        // the next tile's operands travel during this tile's products
            nxt = decode(tn);
            fetch(nxt, va, vb, cn);
        }
        __syncthreads();
        SSTAMP(4);
        d4_t xi[2][2] = {}, xk[2][2] = {};
        tile64_mma_nt_tri2<SD>(S, T, V, xi, xk);
        SSTAMP(5);
        __syncthreads();
        SSTAMP(6);
        if (cur.k == jb + 1) {                             // first trailing column: this row block of L is final
#pragma unroll
            for (int mm = 0; mm < 2; mm++)
#pragma unroll
                for (int n = 0; n < 2; n++)
#pragma unroll
                    for (int r = 0; r < 4; r++) cur.Xi[(size_t)TILE_ROW(mm, r) * Npad + TILE_COL_TRI(n)] = xi[mm][n][r];
        }
#pragma unroll
        for (int mm = 0; mm < 2; mm++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    S[TILE_ROW(mm, r) * SD + TILE_COL_TRI(n)] = -xi[mm][n][r];
                    T[TILE_ROW(mm, r) * SD + TILE_COL_TRI(n)] = xk[mm][n][r];
                }
        __syncthreads();
        SSTAMP(7);
        tile64_mma_nt<SD>(S, T, c);
        SSTAMP(8);
#pragma unroll
        for (int mm = 0; mm < 2; mm++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) cur.C[(size_t)TILE_ROW(mm, r) * Npad + TILE_COL(n)] = c[mm][n][r];
        SSTAMP(9);
        if (!more) break;
        t = tn; cur = nxt;
#pragma unroll
        for (int mm = 0; mm < 2; mm++)
#pragma unroll
            for (int n = 0; n < 2; n++) c[mm][n] = cn[mm][n];
    }
}

// ---- the same step on EIGHT waves (round 4).  The chain is one wave's work whatever the workgroup's size, but the three
// products after it are MFMA-issue bound on four waves -- one wave per SIMD issues an fp64 MFMA every ~100-139 cycles where
// the pipe takes one per 64 (tools/mfma_f64_peak) -- so with two waves per SIMD the X_i / X_k products and the update take
// ~4.5 k + 3 k cycles instead of 8.0 k + 5.3 k of a 42 k-cycle step.  Wave w owns the 16 x 32 strip (row block w >> 1,
// column half w & 1) of every product: each output element is the same chain of MFMAs over ascending k as in the four-wave
// kernel -- identical bits (test_split_steps_equal_fused_steps runs both).  Only the HAVE_V = false form (the step that
// carries the chain); tiles that overflow a launch keep the four-wave kernel.
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
    USTAMP(0);
    for (int j = j0; j < j1; j++) {
        tile64_stash<true>(As, va);
        tile64_stash(Bs, vb);
        USTAMP(1 + 4 * (j - j0));
        __syncthreads();
        USTAMP(2 + 4 * (j - j0));
        if (j + 1 < j1) {
            tile64_fetch(Ai + (j + 1) * 64, Npad, va);
            tile64_fetch(Ak + (j + 1) * 64, Npad, vb);
        }
        tile64_mma_nt(As, Bs, acc);
        USTAMP(3 + 4 * (j - j0));
        if (j + 1 < j1) __syncthreads();
        USTAMP(4 + 4 * (j - j0));
    }
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) C[(size_t)TILE_ROW(m, r) * Npad + TILE_COL(n)] = acc[m][n][r];
    USTAMP(20);
}

// The rows below a panel of P <= 4 block columns [p0, pend) whose diagonal (64 P)^2 block is factored already: row block i
// (one workgroup) turns its P blocks A_i,j into the factor's blocks
//     X_i,j = (A_i,j - sum_{j' < j} X_i,j' L_j,j'^T) inv(L_jj)^T,   j = p0 .. pend-1,
// the updates of a block applied in ascending j', each as 16 k4-steps on accumulators that start as the block -- the
// arithmetic, in order, of the sequence "trsm of column j, K = 64 update of the panel's later columns" that it replaces
// (bit-identical), without that sequence's 2 P - 1 launches over all rows and without its traffic: there every update
// tile reads and writes 128 KB for half a megaflop; here a row block is read once and written once, and its operands
// (6 blocks of the diagonal block, 4 inverses) are shared by all row blocks through L2.
// LDS: -X_i,j' for the (at most 3) earlier columns, one stage for the L / inverse block of the product at hand.
// Pk (optional): the left-looking order's packed copy of the factor (update3.hip) -- the row block's finished columns go there
// straight from LDS, in fragment order (what chol_pack3_kernel would re-read them from L for), and only block rows
// >= rm_from are also stored row-major (a caller that reads nothing else of the rows below a panel: the likelihood's y row).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void chol_panel_rows_kernel(double *L, int Npad, int p0, int pend, const double *__restrict__ diag64, size_t lstride,
                            size_t dstride, double *__restrict__ Pk, size_t pstride, int rm_from)
{
    __shared__ double Xs[3][64 * SD];
    __shared__ double Bs[64 * SD];
    TILE_IDS;
    L += blockIdx.z * lstride; diag64 += blockIdx.z * dstride;
    if (Pk) Pk += blockIdx.z * pstride;
    const int i = pend + blockIdx.x, P = pend - p0;
    const bool rowmajor = !Pk || i >= rm_from;
    double *Ai = L + (size_t)i * 64 * Npad + (size_t)p0 * 64;
    // operand blocks in the order they are used: (jj, jp < jj): L_{p0+jj, p0+jp};  (jj, jj): inv(L_{p0+jj})
    auto fetch_b = [&](int jj, int jp, d2_t (&vb)[8]) {
        if (jp < jj) tile64_fetch(L + (size_t)(p0 + jj) * 64 * Npad + (size_t)(p0 + jp) * 64, Npad, vb);
        else tile64_fetch(diag64 + (size_t)(p0 + jj) * 4096, 64, vb);
    };
    auto load_acc = [&](int jj, d4_t (&acc)[2][2]) {
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[m][n][r] = Ai[(size_t)TILE_ROW(m, r) * Npad + jj * 64 + TILE_COL(n)];
    };
    // column jj of the row block, held as -X in Xm (64 x SD), into the packed store: wave g takes 16-row block g, lane l
    // the fragment element pair (row 16 g + (l & 15), columns 8 j + (l >> 4) and + 4) of every 8-column step j
    auto pack_col = [&](int jj, const double *Xm) {
        const int g = wv, nk8 = Npad >> 3;
        double *dst = Pk + (((size_t)(i * 4 + g) * nk8 + (size_t)(p0 + jj) * 8) * 64 + lane) * 2;
        const double *src = Xm + (16 * g + (lane & 15)) * SD + (lane >> 4);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            d2_t v;
            v.x = -src[8 * j]; v.y = -src[8 * j + 4];
            *(d2_t *)(dst + (size_t)j * 128) = v;
        }
    };
    d2_t vb[8];
    d4_t acc[2][2], accn[2][2];
    fetch_b(0, 0, vb);
    load_acc(0, acc);
    for (int jj = 0; jj < P; jj++) {
        for (int jp = 0; jp < jj; jp++) {
            tile64_stash<false, SD>(Bs, vb);
            __syncthreads();                            // also: -X of the previous column is in place
            fetch_b(jj, jp + 1, vb);
            if (Pk && jp == jj - 1) pack_col(jp, Xs[jp]);       // (jj <= 3 here: column jp's -X is never overwritten before)
            tile64_mma_nt<SD>(Xs[jp], Bs, acc);
            __syncthreads();
        }
        double *As = Xs[jj < 3 ? jj : 0];               // the last column's earlier blocks are not needed any more
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) As[TILE_ROW(m, r) * SD + TILE_COL(n)] = acc[m][n][r];
        tile64_stash<false, SD>(Bs, vb);
        __syncthreads();
        if (jj + 1 < P) {
            fetch_b(jj + 1, 0, vb);
            load_acc(jj + 1, accn);
        }
        d4_t x[2][2] = {};
        tile64_mma_nt_tri<SD>(As, Bs, x);
        if (rowmajor) {
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int n = 0; n < 2; n++)
#pragma unroll
                    for (int r = 0; r < 4; r++) Ai[(size_t)TILE_ROW(m, r) * Npad + jj * 64 + TILE_COL_TRI(n)] = x[m][n][r];
        }
        __syncthreads();
        if (jj + 1 < P || Pk) {
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int n = 0; n < 2; n++) {
#pragma unroll
                    for (int r = 0; r < 4; r++) As[TILE_ROW(m, r) * SD + TILE_COL_TRI(n)] = -x[m][n][r];
                    if (jj + 1 < P) acc[m][n] = accn[m][n];
                }
        }
        if (Pk && jj + 1 == P) {                        // the last column (or a one-column panel): nobody packs it later
            __syncthreads();
            pack_col(jj, As);
        }
    }
}

// The same kernel on EIGHT waves (4 x 2 waves of 16 x 32): a lone wave per SIMD issues an fp64 MFMA every ~139 cycles at best
// (tools/mfma_f64_peak), so the four-wave version's chain of ten dependent 64^3 products ran at 35 TFLOP/s -- 14 % of a
// likelihood grid.  Two waves per SIMD take turns.  Every element sees the same MFMAs in the same order: identical bits.
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
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void chol_panel_rows8_kernel(double *L, int Npad, int p0, int pend, const double *__restrict__ diag64, size_t lstride,
                             size_t dstride, double *__restrict__ Pk, size_t pstride, int rm_from)
{
    __shared__ double Xs[3][64 * SD];
    __shared__ double Bs[64 * SD];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wr8 = wv >> 1, wc8 = wv & 1;
    L += blockIdx.z * lstride; diag64 += blockIdx.z * dstride;
    if (Pk) Pk += blockIdx.z * pstride;
    const int i = pend + blockIdx.x, P = pend - p0;
    const bool rowmajor = !Pk || i >= rm_from;
    double *Ai = L + (size_t)i * 64 * Npad + (size_t)p0 * 64;
    auto fetch_b = [&](int jj, int jp, d2_t (&vb)[4]) {
        if (jp < jj) pr8_fetch(L + (size_t)(p0 + jj) * 64 * Npad + (size_t)(p0 + jp) * 64, Npad, vb);
        else pr8_fetch(diag64 + (size_t)(p0 + jj) * 4096, 64, vb);
    };
    auto load_acc = [&](int jj, d4_t (&acc)[2]) {
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[n][r] = Ai[(size_t)PR8_ROW(r) * Npad + jj * 64 + PR8_COL(n)];
    };
    // column jj (held as -X in Xm) into the packed store: waves 2 g and 2 g + 1 share 16-row block g, four 8-column steps each
    auto pack_col = [&](int jj, const double *Xm) {
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
    d4_t acc[2], accn[2];
    fetch_b(0, 0, vb);
    load_acc(0, acc);
    for (int jj = 0; jj < P; jj++) {
        for (int jp = 0; jp < jj; jp++) {
            pr8_stash(Bs, vb);
            __syncthreads();                            // also: -X of the previous column is in place
            fetch_b(jj, jp + 1, vb);
            if (Pk && jp == jj - 1) pack_col(jp, Xs[jp]);
            pr8_mma_nt(Xs[jp], Bs, acc, wr8, wc8, lane);
            __syncthreads();
        }
        double *As = Xs[jj < 3 ? jj : 0];
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) As[PR8_ROW(r) * SD + PR8_COL(n)] = acc[n][r];
        pr8_stash(Bs, vb);
        __syncthreads();
        if (jj + 1 < P) {
            fetch_b(jj + 1, 0, vb);
            load_acc(jj + 1, accn);
        }
        d4_t x[2] = {};
        if (wc8) pr8_mma_nt_tri_body<1, 2>(As, Bs, x, wr8, lane);
        else pr8_mma_nt_tri_body<0, 3>(As, Bs, x, wr8, lane);
        if (rowmajor) {
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) Ai[(size_t)PR8_ROW(r) * Npad + jj * 64 + PR8_COL_TRI(n)] = x[n][r];
        }
        __syncthreads();
        if (jj + 1 < P || Pk) {
#pragma unroll
            for (int n = 0; n < 2; n++) {
#pragma unroll
                for (int r = 0; r < 4; r++) As[PR8_ROW(r) * SD + PR8_COL_TRI(n)] = -x[n][r];
                if (jj + 1 < P) acc[n] = accn[n];
            }
        }
        if (Pk && jj + 1 == P) {
            __syncthreads();
            pack_col(jj, As);
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

// `batch` matrices, `lstride` doubles apart (diag64: (Npad/64)*4096 apart, info: consecutive ints), are
// factored by the same launches (blockIdx.z).  panel = 1 is the plain right-looking order (shortest chain:
// one matrix, small N); panel = P > 1 keeps the per-column updates inside a P-block panel and applies the
// panel to the rest of the matrix once, with K = 64 P (fewer passes over the trailing matrix: large N, batches).
static std::atomic<int> g_chol_panel{0};                         // 0 = choose; ibo_set_option("chol_panel", P)
void set_chol_panel(int p) { g_chol_panel = p; }
static std::atomic<int> g_panel_rows{3};                         // ibo_set_option("chol_panel_rows", 0/1/2): chol_panel_rows_kernel on four waves / chol_panel_rows8_kernel on eight / 3: chol_panel_rows8r_kernel, right-looking inside the workgroup, two workgroups per CU
void set_chol_panel_rows(int v) { g_panel_rows = v; }
static std::atomic<int> g_update2{1};                            // ibo_set_option("chol_update2", 0/1): packed-panel trailing update (update2.hip)
void set_chol_update2(int v) { g_update2 = v; }
static std::atomic<int> g_update2_min_tiles{1024};               // ibo_set_option("update2_min_tiles"): 128 x 128 tiles (over the batch) from which the packed-panel kernel takes the update
void set_chol_update2_min_tiles(int v) { g_update2_min_tiles = v; }

// The panel's (64 P)^2 DIAGONAL block in one launch, one workgroup per matrix: for each of its P block columns the diagonal
// factorisation (chol_diag_kernel), the row blocks below it inside the block (chol_trsm_kernel) and their K = 64 updates
// (chol_update_kernel) -- the same helpers on the same operands in the same order, identical bits -- without the 3 P - 2
// launches of one workgroup per matrix each (11 per panel, 1.5 ms of a 64-theta grid at N = 4096).  The blocks travel through
// global memory (the workgroup reads back what it stored: one CU, one L1) and LDS holds the chain's three 64 x 64 stages.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void chol_panel_diag_kernel(double *L, int Npad, int p0, int P, double *__restrict__ diag64, int *info, size_t lstride, size_t dstride)
{
    __shared__ double S[64 * SD];
    __shared__ double V[64 * SD];
    __shared__ double T[64 * SD];
    TILE_IDS;
    L += blockIdx.z * lstride; diag64 += blockIdx.z * dstride; info += blockIdx.z;
    for (int j = 0; j < P; j++) {
        const int jb = p0 + j;
        double *Djj = L + (size_t)jb * 64 * Npad + (size_t)jb * 64;
        diag64_load(Djj, Npad, S, V, T);
        __syncthreads();
        diag64_factor_invert(S, V, T, jb * 64, info);
        diag64_store(Djj, Npad, diag64 + (size_t)jb * 4096, S, V);
        __syncthreads();
        for (int r = j + 1; r < P; r++) {               // X_rj = A_rj inv(L_jj)^T
            double *Arj = L + (size_t)(p0 + r) * 64 * Npad + (size_t)jb * 64;
            d2_t va[8];
            tile64_fetch(Arj, Npad, va);
            tile64_stash<false, SD>(S, va);
            __syncthreads();
            d4_t acc[2][2] = {};
            tile64_mma_nt_tri<SD>(S, V, acc);
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int n = 0; n < 2; n++)
#pragma unroll
                    for (int q = 0; q < 4; q++) Arj[(size_t)TILE_ROW(m, q) * Npad + TILE_COL_TRI(n)] = acc[m][n][q];
            __syncthreads();
        }
        for (int c = j + 1; c < P; c++)                  // A_rc -= X_rj X_cj^T, r >= c
            for (int r = c; r < P; r++) {
                const double *Xr = L + (size_t)(p0 + r) * 64 * Npad + (size_t)jb * 64, *Xc = L + (size_t)(p0 + c) * 64 * Npad + (size_t)jb * 64;
                double *C = L + (size_t)(p0 + r) * 64 * Npad + (size_t)(p0 + c) * 64;
                d2_t va[8], vb[8];
                tile64_fetch(Xr, Npad, va);
                tile64_fetch(Xc, Npad, vb);
                d4_t acc[2][2];
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int n = 0; n < 2; n++)
#pragma unroll
                        for (int q = 0; q < 4; q++) acc[m][n][q] = C[(size_t)TILE_ROW(m, q) * Npad + TILE_COL(n)];
                tile64_stash<true, SD>(S, va);
                tile64_stash<false, SD>(T, vb);
                __syncthreads();
                tile64_mma_nt<SD>(S, T, acc);
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int n = 0; n < 2; n++)
#pragma unroll
                        for (int q = 0; q < 4; q++) C[(size_t)TILE_ROW(m, q) * Npad + TILE_COL(n)] = acc[m][n][q];
                __syncthreads();
            }
    }
}

static std::atomic<int> g_chol_tail{20};                         // ibo_set_option("chol_tail", blocks): block columns of the left-looking order's right-looking tail (0: none)
void set_chol_tail(int v) { g_chol_tail = v; }
static std::atomic<int> g_panel_diag{1};                         // ibo_set_option("chol_panel_diag", 0/1): chol_panel_diag_kernel
void set_chol_panel_diag(int v) { g_panel_diag = v; }

// the block columns [p0, pend) of a panel whose columns are up to date: diagonal blocks, row blocks, K = 64 updates inside the panel
// Returns true when the rows below the panel went to the packed store Pk (left-looking order) on the way.
static bool chol_inpanel(double *L, int Npad, int p0, int pend, double *diag64, int *info_dev, int batch, size_t lstride, hipStream_t s,
                         double *Pk = nullptr, size_t pstride = 0, int rm_from = 0)
{
    const int nb = Npad / 64;
    const size_t dstride = (size_t)nb * 4096;
    // P <= 4: the per-column launches stay inside the panel's diagonal block, the rows below it take the whole panel
    // in one launch (chol_panel_rows_kernel; the same arithmetic in the same order)
    // (taken when the rows fill the chip: with few of them the short launches it replaces finish sooner; either way
    // the bits are the same)
    const bool rows_fused = g_panel_rows && pend - p0 <= 4 && (size_t)(nb - pend) * batch >= 256;
    if (rows_fused && g_panel_diag)                  // the diagonal block's 3 P - 2 launches as one
        hipLaunchKernelGGL(chol_panel_diag_kernel, dim3(1, 1, batch), dim3(256), 0, s, L, Npad, p0, pend - p0, diag64, info_dev, lstride, dstride);
    else
    for (int jb = p0; jb < pend; jb++) {
        hipLaunchKernelGGL(chol_diag_kernel, dim3(1, 1, batch), dim3(256), 0, s, L, Npad, jb, diag64, info_dev,
                           lstride, dstride, (double *)nullptr);
        const int m = (rows_fused ? pend : nb) - jb - 1;
        if (m > 0)
            hipLaunchKernelGGL(chol_trsm_kernel, dim3(m, 1, batch), dim3(256), 0, s, L, Npad, jb, diag64, lstride,
                               dstride, (double *)nullptr);
        if (jb + 1 < pend) launch_update(L, Npad, jb, jb + 1, jb + 1, pend, batch, lstride, s, nullptr, rows_fused ? pend : 0);
    }
    if (rows_fused && pend < nb) {
        if (g_panel_rows >= 3)
            hipLaunchKernelGGL(chol_panel_rows8r_kernel, dim3(nb - pend, 1, batch), dim3(512), 0, s, L, Npad, p0, pend, diag64,
                               lstride, dstride, Pk, pstride, rm_from);
        else if (g_panel_rows >= 2)
            hipLaunchKernelGGL(chol_panel_rows8_kernel, dim3(nb - pend, 1, batch), dim3(512), 0, s, L, Npad, p0, pend, diag64,
                               lstride, dstride, Pk, pstride, rm_from);
        else
            hipLaunchKernelGGL(chol_panel_rows_kernel, dim3(nb - pend, 1, batch), dim3(256), 0, s, L, Npad, p0, pend, diag64,
                               lstride, dstride, Pk, pstride, rm_from);
    }
    return rows_fused && pend < nb && Pk;
}

int launch_cholesky_batched(double *L, int Npad, double *diag64, int *info_dev, int batch, size_t lstride,
                            int panel, hipStream_t s, double *ws, size_t wstride)
{
    const int nb = Npad / 64;
    // the panel width fixes the order of the floating-point sums, so it may depend on the matrix size and
    // on the entry point but never on how many matrices share the launches
    const int P = g_chol_panel > 0 ? g_chol_panel.load() : panel;
    HIPCHK(hipMemsetAsync(info_dev, 0, sizeof(int) * batch, s));
    for (int p0 = 0; p0 < nb; p0 += P) {
        const int pend = p0 + P < nb ? p0 + P : nb;
        chol_inpanel(L, Npad, p0, pend, diag64, info_dev, batch, lstride, s);
        if (pend < nb) {
            // the big update (K = 64 P): packed-panel kernel when the caller lent a workspace, bit-identical to the other
            const int nI2 = (Npad - 64 * pend + 127) / 128;
            // (a batch keeps the packed-panel kernel down to a quarter of the tiles one matrix needs: its late, small updates are many
            // short launches of the 64 x 64 kernel otherwise -- C5 0.626 -> 0.60 ms per theta)
            if (ws && g_update2 && (size_t)nI2 * (nI2 + 1) / 2 * batch >= (size_t)(batch >= 8 ? g_update2_min_tiles / 4 : g_update2_min_tiles.load())) {
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
int launch_cholesky_batched_left(double *L, int Npad, double *diag64, int *info_dev, int batch, size_t lstride, int panel,
                                 hipStream_t s, double *Pk, size_t pstride, int nlive, int nfactor, int rm_from)
{
    const int nb = Npad / 64;
    const int P = g_chol_panel > 0 ? g_chol_panel.load() : panel;
    if (nfactor <= 0 || nfactor > nb) nfactor = nb;
    HIPCHK(hipMemsetAsync(info_dev, 0, sizeof(int) * batch, s));
    // THE TAIL.  Left-looking, the last panels have few tiles (17 .. 5 per matrix over the last 1024 columns) and a long K: the
    // chip runs half empty (46 .. 81 % of the update kernel's rate on panels 12 .. 15 of 16).  From block column `tail` on the
    // order changes: ONE update brings all remaining columns up to date with everything before `tail` (many tiles, long K), and
    // inside the tail the order is right-looking (after each panel a K = 256 update of the remaining tail columns: short K, but
    // the tail is small).  Every element still receives its terms in ascending k: identical bits.
    int tail = nfactor;
    if (g_chol_tail > 0 && nfactor > g_chol_tail + P) tail = (nfactor - g_chol_tail) / P * P;
    for (int p0 = 0; p0 < nfactor; p0 += P) {
        const int pend = p0 + P < nb ? p0 + P : nb;
        if (p0 > 0 && p0 <= tail) {
            const int width = p0 == tail ? 64 * (nfactor - p0) : 64 * (pend - p0);
            int rc = launch_chol_update3_range(L, Npad, 64 * p0, width, 0, 64 * p0, nlive, batch, lstride, Pk, pstride, s);
            if (rc) return rc;
        }
        // the rows below the panel reach the packed store straight from chol_panel_rows_kernel's LDS where that kernel runs
        // (and then only the block rows >= rm_from are also stored row-major), through chol_pack3_kernel otherwise
        const bool packed = chol_inpanel(L, Npad, p0, pend, diag64, info_dev, batch, lstride, s, pend < nfactor ? Pk : nullptr, pstride, rm_from);
        if (pend < nfactor && !packed) {
            int rc = launch_chol_pack3(L, Npad, 64 * pend, 64 * p0, 64 * (pend - p0), batch, lstride, Pk, pstride, s);
            if (rc) return rc;
        }
        if (p0 >= tail && pend < nfactor) {              // inside the tail: this panel's update of the columns still to come
            int rc = launch_chol_update3_range(L, Npad, 64 * pend, 64 * (nfactor - pend), 64 * p0, 64 * pend, nlive, batch, lstride, Pk, pstride, s);
            if (rc) return rc;
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
// no tile's MFMAs share a SIMD with a chain (which would slow the chain several times over, DESIGN s9).  Each tile still
// receives the updates of steps 0, 1, .. in that order with the same operands: identical bits (tested).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void chol_pipe_kernel(double *__restrict__ A, double *__restrict__ Lout, int Npad, int jb, double *__restrict__ diag64, int *info,
                      int nrow, double *__restrict__ Ework, double *__restrict__ Eout, int kend, int pre)
{   // kend: the matrix's tiles of step jb - 1 stop before block column kend (the two-level order's panel end; nb otherwise);
    // pre = 0: column jb is up to date already (first column of the matrix or of a panel): no step jb - 1 to apply
    __shared__ double S[64 * SD];
    __shared__ double V[64 * SD];
    __shared__ double T[64 * SD];
    __shared__ double U[64 * SD];
    TILE_IDS;
    const int nb = Npad / 64, m = nb - jb - 1, jp = jb - 1;
    if ((int)blockIdx.x >= nrow) {
        // ---- a tile of step jp right of column jb
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
        tile64_stash<true, SD>(S, va);
        tile64_stash<false, SD>(V, vb);
        __syncthreads();
        tile64_mma_nt<SD>(S, V, acc);
#pragma unroll
        for (int mm = 0; mm < 2; mm++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) C[(size_t)TILE_ROW(mm, r) * Npad + TILE_COL(n)] = acc[mm][n][r];
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
    d4_t ad[2][2], aa[2][2];
    d2_t vxd[8], vxi[8];
    {
        const double *Dp = A + doff, *Ap = (erow ? Ework : A) + roff;
#pragma unroll
        for (int mm = 0; mm < 2; mm++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    ad[mm][n][r] = Dp[(size_t)TILE_ROW(mm, r) * Npad + TILE_COL(n)];
                    aa[mm][n][r] = has_row ? Ap[(size_t)TILE_ROW(mm, r) * Npad + TILE_COL(n)] : 0.0;
                }
    }
    if (upd_d) {
        tile64_fetch(Lout + (size_t)jb * 64 * Npad + jp * 64, Npad, vxd);
        if (upd_a) tile64_fetch((erow ? Eout : Lout) + (size_t)ib * 64 * Npad + jp * 64, Npad, vxi);
        tile64_stash<false, SD>(V, vxd);
        tile64_stash<true, SD>(T, vxd);
        if (upd_a) tile64_stash<true, SD>(U, vxi);
        __syncthreads();
        tile64_mma_nt<SD>(T, V, ad);
        if (upd_a) tile64_mma_nt<SD>(U, V, aa);             // (the two interleaved, B fragments shared: 1 % slower)
        __syncthreads();
    }
    // the diagonal block into the chain's layout (diag64_stash), this workgroup's block into U
#pragma unroll
    for (int mm = 0; mm < 2; mm++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                S[TILE_ROW(mm, r) * SD + TILE_COL(n)] = ad[mm][n][r];
                V[TILE_ROW(mm, r) * SD + TILE_COL(n)] = 0.0;
                U[TILE_ROW(mm, r) * SD + TILE_COL(n)] = aa[mm][n][r];
            }
    __syncthreads();                                            // T's old contents (-X_jb) are done with
    {
        const int t = threadIdx.x;
        T[(t >> 4) * SD + (t & 15)] = ((t >> 4) == (t & 15)) ? 1.0 : 0.0;
    }
    __syncthreads();
    diag64_factor_invert(S, V, T, jb * 64, blockIdx.x == 0 ? info : nullptr);
    if (blockIdx.x == 0) diag64_store(Lout + doff, Npad, diag64 + (size_t)jb * 4096, S, V);
    __syncthreads();
    if (!has_row) return;
    d4_t acc[2][2] = {};
    tile64_mma_nt_tri<SD>(U, V, acc);
    double *Ob = (erow ? Eout : Lout) + roff;
#pragma unroll
    for (int mm = 0; mm < 2; mm++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) Ob[(size_t)TILE_ROW(mm, r) * Npad + TILE_COL_TRI(n)] = acc[mm][n][r];
}

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
    PSTAMP(0);
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
    PSTAMP(1);
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
        PSTAMP(2);
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
        PSTAMP(3);
    }
    // the diagonal block into the chain's layout
#pragma unroll
    for (int r = 0; r < 4; r++) {
        S[(16 * dt_rb0 + (lane >> 4) + 4 * r) * SD + 16 * dt_cb0 + (lane & 15)] = ad[0][r];
        if (dt_two) S[(16 * dt_rb1 + (lane >> 4) + 4 * r) * SD + 16 * dt_cb1 + (lane & 15)] = ad[1][r];
    }
    __syncthreads();
    PSTAMP(4);
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
    PSTAMP(5);
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
    PSTAMP(6);
    d4_t acc[2] = {};
    if (wc8) pr8_mma_nt_tri_body<1, 2>(U, V, acc, wr8, lane);
    else pr8_mma_nt_tri_body<0, 3>(U, V, acc, wr8, lane);
    PSTAMP(7);
    double *Ob = (erow ? Eout : Lout) + roff;
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
        for (int r = 0; r < 4; r++) Ob[(size_t)PR8_ROW(r) * Npad + PR8_COL_TRI(n)] = acc[n][r];
    PSTAMP(8);
}

static std::atomic<int> g_chol_pipe{1};         // ibo_set_option("chol_pipe", 0/1)
void set_chol_pipe(int v) { g_chol_pipe = v; }

static std::atomic<int> g_step_waves{8};        // ibo_set_option("step_waves", 4/8): fused steps on four or eight waves (same bits)
void set_step_waves(int v) { g_step_waves = v; }
static std::atomic<int> g_pipe_pairs{12};       // ibo_set_option("pipe_pairs"): block columns from which the pipelined order applies two steps per pass (0: never)
void set_pipe_pairs(int v) { g_pipe_pairs = v; }
static std::atomic<int> g_step_split{256};      // ibo_set_option("step_split"): tiles of a block column from which rows and updates are separate launches
void set_step_split(int v) { g_step_split = v; }

int launch_cholesky_fused(double *work, double *out, int Npad, double *diag64, int *info_dev, hipStream_t s, double *Ework,
                          double *Eout, bool info_is_zero)
{
    const int nb = Npad / 64;
    const int CU = 256, MAXT = 2 * CU;                  // tiles one fused launch takes: two per workgroup
    if (!info_is_zero) HIPCHK(hipMemsetAsync(info_dev, 0, sizeof(int), s));
    // (four-wave kernels: from ~1300 rows -- below, a step has so few tiles that the fused step's shorter critical path wins by 1-2 %;
    // the eight-wave pipelined kernel, whose row workgroups update their own block under the chain, wins from four block columns on:
    // 0.320 -> 0.304 ms at N = 1024 against the eight-wave fused step, 0.725 -> 0.658 at N = 2048 against the four-wave pipeline)
    if (g_chol_pipe && (nb >= (g_step_waves == 8 ? 4 : 20) || g_chol_pipe > 1)) {
        const bool pairs = g_pipe_pairs > 0 && nb >= g_pipe_pairs;
        int split = nb;                                 // (pairs: first column whose pair of steps waits for the odd launch)
        for (int jb = 0; jb < nb; jb++) {
            const int m = nb - jb - 1, nE = Ework ? jb + 1 : 0;
            const int nrow = g_step_waves == 8 ? m + nE + 1 : (m + nE > 0 ? m + nE : 1);      // (eight waves: one more, the diagonal block's keeper)
            const int ntile = jb > 0 ? m * (m + 1) / 2 + (Ework ? jb * m : 0) : 0;        // step jb - 1 right of column jb
            if (g_step_waves == 8 && pairs) {
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
            } else if (g_step_waves == 8) {
                hipLaunchKernelGGL(chol_pipe8_kernel<0>, dim3(nrow + ntile), dim3(512), 0, s, work, out, Npad, jb, diag64, info_dev, nrow,
                                   Ework, Eout, nb, jb > 0 ? 1 : 0, 0, 0, 0, 0);
            } else
                hipLaunchKernelGGL(chol_pipe_kernel, dim3(nrow + ntile), dim3(256), 0, s, work, out, Npad, jb, diag64, info_dev, nrow,
                                   Ework, Eout, nb, jb > 0 ? 1 : 0);
#ifdef IBO_STAMPS
            if (jb == nb - 1 && getenv("IBO_PIPE_STAMPS")) {
                unsigned long long h[2][16];
                (void)hipStreamSynchronize(s);
                (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pipe_stamps), sizeof(h));
                for (int w = 0; w < 2; w++) {
                    fprintf(stderr, "[pipe8 stamps, column 8, workgroup %d] cycles from entry:", w);
                    for (int i = 1; i <= 8; i++) fprintf(stderr, " %llu", h[w][i] - h[w][0]);
                    fprintf(stderr, "\n");
                }
            }
#endif
        }
        return (int)hipGetLastError();
    }
    for (int jb = 0; jb < nb; jb++) {
        const int m = nb - jb - 1, nchol = m * (m + 1) / 2;
        const int nextra = (Ework && m > 0) ? (jb + 1) * m : 0;       // tiles of the W = L^-1 ride-along (chol_step_kernel)
        int ridden = 0;
        if (m > 0 && nchol + nextra > g_step_split) {
            // more tiles than CUs: a fused step would give a workgroup two tiles -- six products, four of them the row blocks
            // X = (block) inv(L_jj)^T that every tile of a row or column recomputes.  Row blocks once (with the chain,
            // nb workgroups), then one product per tile, two workgroups per CU: 29 -> 19 us per step at N = 2048.
            hipLaunchKernelGGL(chol_diag_trsm_kernel, dim3(m + (nextra ? jb + 1 : 0)), dim3(256), 0, s, work, out, Npad, jb, diag64,
                               info_dev, m, Ework, Eout);
            hipLaunchKernelGGL(chol_update_step_kernel, dim3(nchol + nextra), dim3(256), 0, s, work, out, Npad, jb, nchol, Ework, Eout);
            ridden = nextra;
        } else if (m > 0 && nchol <= MAXT) {
            // the trailing tiles fit on the chip (two rounds at most): repeating the diagonal factorisation in each
            // workgroup costs nothing and two launches disappear; extra tiles come along
            ridden = nextra < MAXT - nchol ? nextra : MAXT - nchol;
            const int nt = nchol + ridden;
            if (g_step_waves == 8)
                hipLaunchKernelGGL(chol_step8_kernel, dim3(nt < CU ? nt : CU), dim3(512), 0, s, work, out, Npad, jb, diag64,
                                   info_dev, nchol, Ework, Eout, nt);
            else
                hipLaunchKernelGGL(chol_step_kernel<false>, dim3(nt < CU ? nt : CU), dim3(256), 0, s, work, out, Npad, jb, diag64,
                                   info_dev, nchol, 0, Ework, Eout, nt);
        } else if (m == 0 && Ework) {
            // last block column with the ride-along: nothing trails it, E's row blocks only need the multiplication by inv(L_jj)^T
            hipLaunchKernelGGL(chol_diag_trsm_kernel, dim3(nb), dim3(256), 0, s, work, out, Npad, jb, diag64, info_dev, 0, Ework, Eout);
            continue;
        } else {
            hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), 0, s, work, Npad, jb, diag64, info_dev,
                               (size_t)0, (size_t)0, out);
            if (m > 0) {
                hipLaunchKernelGGL(chol_trsm_kernel, dim3(m), dim3(256), 0, s, work, Npad, jb, diag64, (size_t)0,
                                   (size_t)0, out, -1);
                launch_update(work, Npad, jb, jb + 1, jb + 1, nb, 1, 0, s, out);
            }
        }
        // extra tiles that found no room read the block's inverse from diag64 (no second factorisation)
        if (nextra > ridden) {
            const int nt = nextra - ridden;
            hipLaunchKernelGGL(chol_step_kernel<true>, dim3(nt < CU ? nt : CU), dim3(256), 0, s, work, out, Npad, jb,
                               diag64, info_dev, 0, ridden, Ework, Eout, nt);
        }
    }
    return (int)hipGetLastError();
}

// The same out-of-place scheme in the TWO-LEVEL order (panels of P block columns; one matrix of more than 32 blocks):
// inside a panel every block column is one chol_step_kernel launch over the tiles (i, k), jb < k < pend, k <= i < nb --
// at most 3 x 63, one per CU -- instead of the diagonal / row-block / update launches (21.4 -> 18.5 us per column at
// N = 4096); a panel's last column has nothing to update inside the panel and keeps its two launches; then the K = 64 P
// update of the matrix right of the panel, its operands read from the finished columns in `out`.  The arithmetic and its
// order are those of launch_cholesky_batched with the same P: identical bits (tested).
int launch_cholesky_fused2(double *work, double *out, int Npad, double *diag64, int *info_dev, int P, hipStream_t s, bool info_is_zero, double *ws)
{
    const int nb = Npad / 64;
    if (g_chol_panel > 0) P = g_chol_panel;
    if (!info_is_zero) HIPCHK(hipMemsetAsync(info_dev, 0, sizeof(int), s));
    for (int p0 = 0; p0 < nb; p0 += P) {
        const int pend = p0 + P < nb ? p0 + P : nb;
        // (the in-panel columns pipelined like the fused route's -- chol_pipe_kernel with kend = pend -- measured 2 % slower on four waves,
        // 1 % faster on eight (2.48 -> 2.45 ms at N = 4096): not worth a second order to keep bit-identical)
        for (int jb = p0; jb < pend; jb++) {
            int nt = 0;
            for (int k = jb + 1; k < pend; k++) nt += nb - k;
            if (nt > 0 && nt <= g_step_split) {
                if (g_step_waves == 8)
                    hipLaunchKernelGGL(chol_step8_kernel, dim3(nt < 256 ? nt : 256), dim3(512), 0, s, work, out, Npad, jb, diag64,
                                       info_dev, nt, (double *)nullptr, (double *)nullptr, nt);
                else
                    hipLaunchKernelGGL(chol_step_kernel<false>, dim3(nt < 256 ? nt : 256), dim3(256), 0, s, work, out, Npad, jb, diag64,
                                       info_dev, nt, 0, (double *)nullptr, (double *)nullptr, nt);
            } else if (nt > 0) {           // more in-panel tiles than CUs (beyond 5400 rows): row blocks first, then the updates
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
            if (ws && g_update2 && nI2 * (nI2 + 1) / 2 >= g_update2_min_tiles) {
                int rc = launch_chol_update2(work, Npad, p0, pend, 1, 0, ws, 0, s, out);
                if (rc) return rc;
            } else launch_update(work, Npad, p0, pend, pend, nb, 1, 0, s, out);
        }
    }
    return (int)hipGetLastError();
}

// W[r][c] = Et[c][r] for c <= r, 0 above the diagonal (Et = (L^-1)^T from the ride-along; its blocks below the
// diagonal were never written)
__global__ void transpose_lower_kernel(const double *__restrict__ Et, double *__restrict__ W, int Npad)
{
    __shared__ double tile[64][65];
    const int bx = blockIdx.x * 64, by = blockIdx.y * 64;       // W block (row block y, column block x)
    if (blockIdx.x > blockIdx.y) {
        for (int e = threadIdx.x; e < 4096; e += 256) W[(size_t)(by + (e >> 6)) * Npad + bx + (e & 63)] = 0.0;
        return;
    }
    for (int e = threadIdx.x; e < 4096; e += 256) {
        const int r = e >> 6, c = e & 63;
        tile[r][c] = Et[(size_t)(bx + r) * Npad + by + c];       // Et block (x, y)
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 4096; e += 256) {
        const int r = e >> 6, c = e & 63;
        W[(size_t)(by + r) * Npad + bx + c] = (bx + c <= by + r) ? tile[c][r] : 0.0;
    }
}

// The same with the result's rows >= N zeroed, and a second copy in MFMA fragment order (pack_w_kernel's layout, mode 0):
// the fit's transpose and packing passes in one.  Wp must not be Et's buffer.
__global__ void transpose_pack_kernel(const double *__restrict__ Et, int N, int Npad, double *__restrict__ W,
                                      double *__restrict__ Wp)
{
    __shared__ double tile[64][65];
    const int bx = blockIdx.x * 64, by = blockIdx.y * 64;       // W block (row block y, column block x)
    const int nk8 = Npad / 8;
    const bool lower = blockIdx.x <= blockIdx.y;
    if (lower) {
        for (int e = threadIdx.x; e < 4096; e += blockDim.x) {
            const int r = e >> 6, c = e & 63;
            tile[r][c] = Et[(size_t)(bx + r) * Npad + by + c];   // Et block (x, y)
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < 4096; e += blockDim.x) {
        const int r = e >> 6, c = e & 63;
        const int row = by + r, col = bx + c;
        W[(size_t)row * Npad + col] = (lower && row < N && col <= row) ? tile[c][r] : 0.0;
    }
    // packed copy (if wanted): the block's 4 row groups x 8 column steps, 128 consecutive doubles each
    if (Wp)
    for (int e = threadIdx.x; e < 4096; e += blockDim.x) {
        const int h = e & 1, lane = (e >> 1) & 63, chunk = e >> 7;       // chunk = g_local * 8 + j_local
        const int r = 16 * (chunk >> 3) + (lane & 15), c = 8 * (chunk & 7) + 4 * h + (lane >> 4);
        const int row = by + r, col = bx + c;
        const size_t dst = ((((size_t)(row >> 4) * nk8 + (col >> 3)) * 64 + lane) << 1) + h;
        Wp[dst] = (lower && row < N && col <= row) ? tile[c][r] : 0.0;
    }
}
int launch_transpose_pack(const double *Et, int N, int Npad, double *W, double *Wp, hipStream_t s)
{
    // (up to ~1500 rows the grid is at most two workgroups per CU and a workgroup's three passes over its 4096 elements are what the kernel lasts:
    // 1024 threads take four elements each instead of sixteen)
    hipLaunchKernelGGL(transpose_pack_kernel, dim3(Npad / 64, Npad / 64), dim3(Npad <= 1536 ? 1024 : 256), 0, s, Et, N, Npad, W, Wp);
    return (int)hipGetLastError();
}

int launch_transpose_lower(const double *Et, double *W, int Npad, hipStream_t s)
{
    hipLaunchKernelGGL(transpose_lower_kernel, dim3(Npad / 64, Npad / 64), dim3(256), 0, s, Et, W, Npad);
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

static std::atomic<int> g_trinv_wide{1};                          // ibo_set_option("trinv_wide", 0/1): eight-wave tiles where a level has at most 512 of them
void set_trinv_wide(int v) { g_trinv_wide = v; }

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
        if (g_trinv_wide && sz * sz * nodes <= 512) {
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
static std::atomic<int> g_wtw_waves{8};         // ibo_set_option("wtw_waves", 4/8)
void set_wtw_waves(int v) { g_wtw_waves = v; }
static std::atomic<int> g_wtw_xcd{32};          // ibo_set_option("wtw_xcd"): block rows from which W^T W's tiles are dealt to the XCDs in 8 x 8 super-blocks (0: never)
void set_wtw_xcd(int v) { g_wtw_xcd = v; }

// wt_ready: Wt already holds W^T on and right of the diagonal blocks (the ride-along's (L^-1)^T as the factorisation leaves it: the blocks
// left of the diagonal, which it never writes, are never read here) -- no transpose pass
int launch_wtw(const double *W, double *Wt, double *C, int Npad, hipStream_t s, int lower_only, int wt_ready)
{
    dim3 g(Npad / 64, Npad / 64);
    if (!wt_ready) hipLaunchKernelGGL(transpose_kernel, g, dim3(256), 0, s, W, Wt, Npad);
    int nsb = 0;
    if (g_wtw_xcd && Npad / 64 >= g_wtw_xcd) {                  // enough tiles that the operands do not stay in L2 by themselves
        const int nsr = (Npad / 64 + 7) / 8;
        nsb = lower_only ? nsr * (nsr + 1) / 2 : nsr * nsr;
        g = dim3((unsigned)((nsb + 7) / 8) * 512);
    }
    if (g_wtw_waves == 8) hipLaunchKernelGGL(wtw_kernel<8>, g, dim3(512), 0, s, Wt, W, C, Npad, lower_only, nsb);
    else hipLaunchKernelGGL(wtw_kernel<4>, g, dim3(256), 0, s, Wt, W, C, Npad, lower_only, nsb);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------
// gradient of the negative log marginal likelihood (ego/gaussianprocess/trainhyper.py:70-71):
//   dnlml_h = 1/2 sum_ab (K^-1 - alpha alpha^T)_ab * dK_h[a][b]
// with dK_h as the reference's Kernel.derivative(X, h) builds it (kernel.py:92-106,122-127,
// 152-166,183-188,212-227,251-266), quirks included (Matern-3/2 uses the unscaled distance).
// A 64 x 64 tile of (a, b) pairs per workgroup, 16 per thread, the tile's points staged in LDS; K_ab and every dK_h
// are recomputed from X, nothing N x N is stored besides K^-1.  A thread sums its pairs in a fixed order, a wave its
// lanes by shuffles, the four waves through LDS: one barrier per workgroup (the first version reduced a 16 x 16 tile
// through LDS once per hyper-parameter -- 128 barriers for 256 pairs: 181 us at N = 2048, D = 8; now 25).
// Per-workgroup partial sums, reduced in a fixed order by grad_reduce_kernel.
// ------------------------------------------------------------------------
template <int GM, int LD>  // GM >= gs.nh: accumulators held per thread; LD: row stride of the staged points (33 or 65)
__global__ __launch_bounds__(256) void nlml_grad_kernel(KParams kp, GradSpec gs, int N, const double *__restrict__ X,
                                                        int ldx, const double *__restrict__ Kinv, int ldk,
                                                        const double *__restrict__ alpha, double *__restrict__ partial)
{
    __shared__ double As[64 * COV_LD], Bs[64 * COV_LD], ala[64], alb[64];
    __shared__ double red[GM][4];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4, D = kp.D;
    const int b0 = blockIdx.x * 64, a0 = blockIdx.y * 64;
    for (int e = t; e < 64 * D; e += 256) {
        const int r = e / D, d = e - r * D;
        As[r * COV_LD + d] = (a0 + r < N) ? X[(size_t)(a0 + r) * ldx + d] : 0.0;
        Bs[r * COV_LD + d] = (b0 + r < N) ? X[(size_t)(b0 + r) * ldx + d] : 0.0;
    }
    if (t < 64) ala[t] = (a0 + t < N) ? alpha[a0 + t] : 0.0;
    else if (t < 128) alb[t - 64] = (b0 + t - 64 < N) ? alpha[b0 + t - 64] : 0.0;
    __syncthreads();
    double acc[GM];
#pragma unroll
    for (int h = 0; h < GM; h++) acc[h] = 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int la = ty * 4 + r, a = a0 + la;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int lb = tx + 16 * c, b = b0 + lb;
            if (a >= N || b >= N) continue;
            const double *xa = As + la * COV_LD, *xb = Bs + lb * COV_LD;
            double z = 0.0, d2 = 0.0;
            for (int d = 0; d < D; d++) { double u = xa[d] - xb[d]; z += kp.w[d] * (u * u); d2 += u * u; }
            const double kab = cov_from_z_rt(kp.family, z, kp.sf2);
            const double wm = Kinv[(size_t)(a > b ? a : b) * ldk + (a > b ? b : a)] - ala[la] * alb[lb];     // lower triangle only is formed
#pragma unroll
            for (int h = 0; h < GM; h++) {
                if (h >= gs.nh) continue;                  // (no break: the unrolled copies keep acc[] in registers)
                double dk;
                switch (gs.mode[h]) {
                case 0: { double u = xa[gs.dim[h]] - xb[gs.dim[h]]; dk = kab * kp.w[gs.dim[h]] * (u * u); break; }
                case 1: dk = kab * z; break;                                   // iso: w * |x_a - x_b|^2
                case 2: dk = 2.0 * kab; break;                                 // signal magnitude
                case 3: { double r3 = sqrt(d2); dk = (a == b) ? 0.0 : kp.sf2 * r3 * r3 * exp(-r3); break; }
                default: { double zz = 5.0 * z; dk = (a == b) ? 0.0 : kp.sf2 * (zz + sqrt(zz) * sqrt(zz) * sqrt(zz)) * exp(-sqrt(zz)) / 3.0; break; }
                }
                acc[h] = fma(wm, dk, acc[h]);
            }
        }
    }
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int h = 0; h < GM; h++) {
        if (h >= gs.nh) continue;
        double v = acc[h];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[h][wave] = v;
    }
    __syncthreads();
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    if (t < gs.nh) partial[(size_t)t * gridDim.x * gridDim.y + blk] = ((red[t][0] + red[t][1]) + red[t][2]) + red[t][3];
}

// The gradient kernel of round 4.  The first one (above) reads both points' coordinates from LDS once per pair and dimension and again per
// derivative, and walks a switch per pair and derivative -- ~1000 LDS reads per thread, 530 us at N = 4096, D = 16 for 0.8 GFLOP, two passes
// beyond 16 dimensions.  Here a thread's 4 x 4 pairs share their eight points' coordinates per dimension (8 LDS reads for 16 pairs): one pass
// over the dimensions gives z_ab (and the unscaled |x_a - x_b|^2 the Matern-3/2 derivative uses), then t_ab = (K^-1 - alpha alpha^T)_ab K_ab,
// then one short loop per derivative: a length scale of an ARD kernel is one more pass over ITS dimension (acc_h = sum_pairs t_ab w_h u_h^2),
// the others need only z, d2 and K.  The sum over (a, b) is symmetric: tiles above the diagonal contribute nothing, tiles below it count
// twice (an exact scaling).  Per-workgroup partial sums in a fixed order, as before.
template <int GM, int LD>
__global__ __launch_bounds__(256) void nlml_grad_fast_kernel(KParams kp, GradSpec gs, int N, const double *__restrict__ X, int ldx,
                                                             const double *__restrict__ Kinv, int ldk, const double *__restrict__ alpha,
                                                             double *__restrict__ partial)
{
    __shared__ double As[64 * LD], Bs[64 * LD], ala[64], alb[64];
    __shared__ double red[GM][4];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4, D = kp.D, nh = gs.nh;
    const int b0 = blockIdx.x * 64, a0 = blockIdx.y * 64;
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    if (blockIdx.x > blockIdx.y) {                              // (b-block > a-block: its mirror image carries the weight)
        if (t < nh) partial[(size_t)t * gridDim.x * gridDim.y + blk] = 0.0;
        return;
    }
    for (int e = t; e < 64 * D; e += 256) {
        const int r = e / D, d = e - r * D;
        As[r * LD + d] = (a0 + r < N) ? X[(size_t)(a0 + r) * ldx + d] : 0.0;
        Bs[r * LD + d] = (b0 + r < N) ? X[(size_t)(b0 + r) * ldx + d] : 0.0;
    }
    if (t < 64) ala[t] = (a0 + t < N) ? alpha[a0 + t] : 0.0;
    else if (t < 128) alb[t - 64] = (b0 + t - 64 < N) ? alpha[b0 + t - 64] : 0.0;
    // (K^-1 - alpha alpha^T): requested now, used after the first pass
    double wm[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int a = a0 + ty * 4 + r, b = b0 + tx + 16 * c;
            wm[r][c] = (a < N && b < N) ? Kinv[(size_t)(a > b ? a : b) * ldk + (a > b ? b : a)] : 0.0;
        }
    __syncthreads();
    double z[4][4] = {}, d2[4][4] = {};
    for (int d = 0; d < D; d++) {
        const double w = kp.w[d];
        double av[4], bv[4];
#pragma unroll
        for (int r = 0; r < 4; r++) av[r] = As[(ty * 4 + r) * LD + d];
#pragma unroll
        for (int c = 0; c < 4; c++) bv[c] = Bs[(tx + 16 * c) * LD + d];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) { const double u = av[r] - bv[c]; z[r][c] += w * (u * u); d2[r][c] += u * u; }
    }
    double tt[4][4], kk[4][4];                                  // t_ab = (K^-1 - alpha alpha^T)_ab K_ab and (K^-1 - alpha alpha^T)_ab (0 off the matrix)
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int a = a0 + ty * 4 + r, b = b0 + tx + 16 * c;
            const double kab = cov_from_z_rt(kp.family, z[r][c], kp.sf2);
            const bool in = a < N && b < N;
            kk[r][c] = in ? wm[r][c] - ala[ty * 4 + r] * alb[tx + 16 * c] : 0.0;
            tt[r][c] = kk[r][c] * kab;
        }
    const int lane = t & 63, wave = t >> 6;
    const double scale = blockIdx.x < blockIdx.y ? 2.0 : 1.0;
    for (int h = 0; h < nh; h++) {
        const int mode = gs.mode[h];
        double s = 0.0;
        if (mode == 0) {                                        // SE-ARD length scale of dimension dim[h]: dK = K w u^2
            const int d = gs.dim[h];
            const double w = kp.w[d];
            double av[4], bv[4];
#pragma unroll
            for (int r = 0; r < 4; r++) av[r] = As[(ty * 4 + r) * LD + d];
#pragma unroll
            for (int c = 0; c < 4; c++) bv[c] = Bs[(tx + 16 * c) * LD + d];
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int c = 0; c < 4; c++) { const double u = av[r] - bv[c]; s = fma(tt[r][c], w * (u * u), s); }
        } else if (mode == 1) {                                 // SE-iso length scale: dK = K z
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int c = 0; c < 4; c++) s = fma(tt[r][c], z[r][c], s);
        } else if (mode == 2) {                                 // signal magnitude: dK = 2 K
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int c = 0; c < 4; c++) s += tt[r][c];
            s *= 2.0;
        } else {                                                // Matern length scales, as the reference's derivative() has them (quirks included)
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int a = a0 + ty * 4 + r, b = b0 + tx + 16 * c;
                    double dk;
                    if (mode == 3) { const double r3 = sqrt(d2[r][c]); dk = kp.sf2 * r3 * r3 * exp(-r3); }
                    else { const double zz = 5.0 * z[r][c], q = sqrt(zz); dk = kp.sf2 * (zz + q * q * q) * exp(-q) / 3.0; }
                    s = fma(kk[r][c], (a == b) ? 0.0 : dk, s);
                }
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) red[h][wave] = s;
    }
    __syncthreads();
    if (t < nh) partial[(size_t)t * gridDim.x * gridDim.y + blk] = scale * (((red[t][0] + red[t][1]) + red[t][2]) + red[t][3]);
}

__global__ __launch_bounds__(256) void grad_reduce_kernel(const double *__restrict__ partial, int nblk, double *__restrict__ out)
{
    __shared__ double red[256];
    const int h = blockIdx.x, t = threadIdx.x;
    double s = 0.0;
    for (int i = t; i < nblk; i += 256) s += partial[(size_t)h * nblk + i];
    red[t] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
    if (t == 0) out[h] = 0.5 * red[0];
}

static std::atomic<int> g_grad_ard{1};         // ibo_set_option("grad_ard", 0/1): the SE-ARD gradient kernel (0: the general one)
void set_grad_ard(int v) { g_grad_ard = v; }
int launch_nlml_grad(const KParams &kp, const GradSpec &gs, int N, const double *X, int ldx, const double *Kinv, int ldk,
                     const double *alpha, double *partial, double *out, hipStream_t s)
{
    dim3 grid((N + 63) / 64, (N + 63) / 64);
    // at most 17 derivatives per pass (17 accumulators per thread: 64 VGPRs, no spills): beyond 16 dimensions the components go
    // in two passes that each rebuild K_ab -- the 33-accumulator instantiation needed 256 VGPRs, 232 spilled SGPRs, occupancy 1
    const int nblk = (int)(grid.x * grid.y);
    if (g_grad_ard && gs.nh <= 33) {                    // the round-4 kernel ("grad_ard" = 0: the first one, below)
        if (kp.D <= 32) hipLaunchKernelGGL((nlml_grad_fast_kernel<33, 33>), grid, dim3(256), 0, s, kp, gs, N, X, ldx, Kinv, ldk, alpha, partial);
        else hipLaunchKernelGGL((nlml_grad_fast_kernel<33, 65>), grid, dim3(256), 0, s, kp, gs, N, X, ldx, Kinv, ldk, alpha, partial);
        hipLaunchKernelGGL(grad_reduce_kernel, dim3(gs.nh), dim3(256), 0, s, partial, nblk, out);
        return (int)hipGetLastError();
    }
    for (int h0 = 0; h0 < gs.nh; h0 += 17) {
        GradSpec part;
        part.nh = gs.nh - h0 < 17 ? gs.nh - h0 : 17;
        for (int h = 0; h < part.nh; h++) { part.mode[h] = gs.mode[h0 + h]; part.dim[h] = gs.dim[h0 + h]; }
        if (kp.D <= 32) hipLaunchKernelGGL((nlml_grad_kernel<17, 33>), grid, dim3(256), 0, s, kp, part, N, X, ldx, Kinv, ldk, alpha, partial + (size_t)h0 * nblk);
        else hipLaunchKernelGGL((nlml_grad_kernel<17, 65>), grid, dim3(256), 0, s, kp, part, N, X, ldx, Kinv, ldk, alpha, partial + (size_t)h0 * nblk);
    }
    hipLaunchKernelGGL(grad_reduce_kernel, dim3(gs.nh), dim3(256), 0, s, partial, (int)(grid.x * grid.y), out);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------
// pack W into MFMA A-fragment order for the sweep:
//   Wp[((g*nk8 + j)*64 + lane)*2 + h] = W[16g + (lane&15)][8j + 4h + (lane>>4)]
// so that one 16-byte load per lane yields the A operands of two consecutive
// k4-steps of row-block g.  mode 1 applies W[i][j] = S[N-1-j][N-1-i]
// (turns the upper factor G^T of the legacy invR = G G^T into a lower one).
// ------------------------------------------------------------------------
__global__ void pack_w_kernel(const double *S, int N, int Npad, int mode, double *Wout,
                              double *__restrict__ Wp)
{
    size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t total = (size_t)Npad * Npad;
    if (e >= total) return;
    int h = (int)(e & 1);
    int lane = (int)((e >> 1) & 63);
    size_t gj = e >> 7;
    int nk8 = Npad / 8;
    int j = (int)(gj % nk8), g = (int)(gj / nk8);
    int row = 16 * g + (lane & 15), col = 8 * j + 4 * h + (lane >> 4);
    double v = 0.0;
    if (row < N && col <= row) {
        v = (mode == 0) ? S[(size_t)row * Npad + col] : S[(size_t)(N - 1 - col) * Npad + (N - 1 - row)];
    }
    Wp[e] = v;
    if (Wout) Wout[(size_t)row * Npad + col] = v;
}

int launch_pack_w(const double *S, int N, int Npad, int mode, double *Wout, double *Wp, hipStream_t s)
{
    size_t total = (size_t)Npad * Npad;
    hipLaunchKernelGGL(pack_w_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, S, N, Npad, mode,
                       Wout, Wp);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------
// alpha = W^T (W y), for y and for the all-ones vector (prior-mean term)
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemv_lower2_kernel(const double *__restrict__ W, int N, int Npad,
                                                          const double *__restrict__ y, double *__restrict__ t2)
{
    int lane = threadIdx.x & 63;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Npad) return;
    const double *w = W + (size_t)row * Npad;
    double s0 = 0.0, s1 = 0.0;
    const int kend = row < N - 1 ? row : N - 1;                 // last column of this row
    for (int k0 = lane; k0 <= kend; k0 += 8 * 64) {             // a lane's terms in index order, eight loads in flight
        double v[8], yy[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = k0 + 64 * u;
            v[u] = k <= kend ? w[k] : 0.0;
            yy[u] = k <= kend ? y[k] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (k0 + 64 * u <= kend) { s0 += v[u] * yy[u]; s1 += v[u]; }
    }
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
    if (lane == 0) { t2[row] = s0; t2[Npad + row] = s1; }
}

// partial[c][j] = sum_{i in chunk c, i >= j} W[i][j] t[i]; 256 columns x 64 rows per block
__global__ __launch_bounds__(256) void gemvT_lower2_kernel(const double *__restrict__ W, int Npad,
                                                           const double *__restrict__ t2,
                                                           double *__restrict__ partial)
{
    int j = blockIdx.x * 256 + threadIdx.x;
    int c = blockIdx.y;
    if (j >= Npad) return;
    double s0 = 0.0, s1 = 0.0;
    int i0 = c * 64;
    // W is stored with explicit zeros above the diagonal: a fixed trip count lets the loads be batched
    // (a dependent loop from max(i0, j) exposes the memory latency 64 times)
    if (i0 + 63 >= blockIdx.x * 256) {
#pragma unroll 16
        for (int i = i0; i < i0 + 64; i++) {
            double v = W[(size_t)i * Npad + j];
            s0 = fma(v, t2[i], s0);
            s1 = fma(v, t2[Npad + i], s1);
        }
    }
    int nch = Npad / 64;
    partial[(size_t)c * Npad + j] = s0;
    partial[(size_t)(nch + c) * Npad + j] = s1;
}

__global__ void alpha_reduce_kernel(const double *__restrict__ partial, int Npad, double *__restrict__ aY,
                                    double *__restrict__ a1)
{
    int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= Npad) return;
    int nch = Npad / 64;
    double s0 = 0.0, s1 = 0.0;
    // index order, eight terms' loads in flight at a time (one by one the L2 round trip of every term is on the chain:
    // 64 terms at N = 4096 took 20 us)
    for (int c0 = 0; c0 < nch; c0 += 8) {
        double v0[8], v1[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            v0[u] = c0 + u < nch ? partial[(size_t)(c0 + u) * Npad + j] : 0.0;
            v1[u] = c0 + u < nch ? partial[(size_t)(nch + c0 + u) * Npad + j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) if (c0 + u < nch) { s0 += v0[u]; s1 += v1[u]; }
    }
    aY[j] = s0; a1[j] = s1;
}

// tmp2: 2*Npad (t vectors) + 2*(Npad/64)*Npad (partials) doubles
int launch_alpha(const double *W, int N, int Npad, const double *y, double *tmp2, double *alphaY,
                 double *alpha1, hipStream_t s)
{
    double *t2 = tmp2, *partial = tmp2 + 2 * (size_t)Npad;
    hipLaunchKernelGGL(gemv_lower2_kernel, dim3((Npad + 3) / 4), dim3(256), 0, s, W, N, Npad, y, t2);
    dim3 grid((Npad + 255) / 256, Npad / 64);
    hipLaunchKernelGGL(gemvT_lower2_kernel, grid, dim3(256), 0, s, W, Npad, t2, partial);
    hipLaunchKernelGGL(alpha_reduce_kernel, dim3((Npad + 255) / 256), dim3(256), 0, s, partial, Npad, alphaY, alpha1);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------
// One-point block extension of a fitted model (ibo_gp_extend; ego/gaussianprocess/__init__.py:301-308)
// ------------------------------------------------------------------------
// the new point is row N of Xp.  R's entries come out of the same expression, in the same order, as
// cov_matrix_kernel's, so an extended R equals a rebuilt one bit for bit.
__global__ __launch_bounds__(256) void extend_kvec_kernel(KParams kp, const double *__restrict__ Xp, int ldp, int N, int Npad,
                                                          double noise, double *__restrict__ R, double *__restrict__ kvec)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Npad) return;
    double v = 0.0;
    if (i < N) {
        double z = 0.0;
        for (int d = 0; d < kp.D; d++) {
            const double u = Xp[(size_t)i * ldp + d] - Xp[(size_t)N * ldp + d];
            z += kp.w[d] * (u * u);
        }
        v = cov_from_z_rt(kp.family, z, kp.sf2);
        if (R) { R[(size_t)N * Npad + i] = v; R[(size_t)i * Npad + N] = v; }
    } else if (i == N && R) R[(size_t)N * Npad + N] = 1.0 + noise;
    kvec[i] = v;
}

int launch_extend_kvec(const KParams &kp, const double *Xp, int ldp, int N, int Npad, double noise, double *R, double *kvec,
                       hipStream_t s)
{
    hipLaunchKernelGGL(extend_kvec_kernel, dim3((Npad + 255) / 256), dim3(256), 0, s, kp, Xp, ldp, N, Npad, noise, R, kvec);
    return (int)hipGetLastError();
}

// one workgroup: the pivot (fixed-order reduction of |z|^2), then the new rows of L and W and the row-block of
// W's fragment copy that contains row N
__global__ __launch_bounds__(1024) void extend_rows_kernel(int N, int Npad, double noise, const double *__restrict__ z,
                                                           const double *__restrict__ u, double *__restrict__ L,
                                                           double *__restrict__ W, double *__restrict__ Wp, int *info)
{
    __shared__ double red[1024];
    const int t = threadIdx.x;
    double s = 0.0;
    for (int k = t; k < N; k += 1024) s = fma(z[k], z[k], s);
    red[t] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
    const double d2 = (1.0 + noise) - red[0];
    if (!(d2 > 0.0)) { if (t == 0) atomicCAS(info, 0, N + 1); return; }
    const double d = sqrt(d2), id = 1.0 / d;
    for (int k = t; k < N; k += 1024) {
        L[(size_t)N * Npad + k] = z[k];
        W[(size_t)N * Npad + k] = -u[k] * id;
    }
    if (t == 0) { L[(size_t)N * Npad + N] = d; W[(size_t)N * Npad + N] = id; }
    // fragment copy of row-block N/16: its earlier rows come from W (written by earlier launches), row N from u
    const int g = N >> 4, nk8 = Npad >> 3;
    for (int e = t; e < nk8 * 128; e += 1024) {
        const int h = e & 1, lane = (e >> 1) & 63, j = e >> 7;
        const int row = 16 * g + (lane & 15), col = 8 * j + 4 * h + (lane >> 4);
        double v = 0.0;
        if (row <= N && col <= row) v = (row == N) ? (col == N ? id : -u[col] * id) : W[(size_t)row * Npad + col];
        Wp[((size_t)g * nk8 + j) * 128 + e % 128] = v;
    }
}

int launch_extend_rows(int N, int Npad, double noise, const double *z, const double *u, double *L, double *W, double *Wp,
                       int *info, hipStream_t s)
{
    hipLaunchKernelGGL(extend_rows_kernel, dim3(1), dim3(1024), 0, s, N, Npad, noise, z, u, L, W, Wp, info);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------
// Preference GP (ego/gaussianprocess/__init__.py:351-498): the matrices of its Newton steps and of L = chol(R + C^-1)
// are assembled where they are factored.  out (Npad x Npad) = base (or 0) + diag I on [0, N)^2, the identity on the pad;
// a matrix that is a sum of per-pair terms w (e_v - e_u)(e_v - e_u)^T arrives as its distinct entries (row * N + col,
// value), summed on the host in the order the reference's scatter-adds take.
// ------------------------------------------------------------------------
__global__ void pref_build_kernel(const double *__restrict__ base, int N, int Npad, double diag, double *__restrict__ out)
{
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= Npad) return;
    double v;
    if (i < N && j < N) v = (base ? base[(size_t)i * Npad + j] : 0.0) + (i == j ? diag : 0.0);
    else v = (i == j) ? 1.0 : 0.0;
    out[(size_t)i * Npad + j] = v;
}
__global__ void pref_scatter_kernel(int nnz, const long long *__restrict__ lin, const double *__restrict__ val, int N,
                                    int Npad, double *__restrict__ out)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz) return;
    const long long i = lin[e] / N, j = lin[e] - i * N;
    out[(size_t)i * Npad + j] += val[e];                 // entries are distinct
}
// A (N x N, dense) = R + Cinv (both with row stride Npad)
__global__ void pref_sum_kernel(const double *__restrict__ R, const double *__restrict__ Cinv, int N, int Npad,
                                double *__restrict__ A)
{
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j < N) A[(size_t)i * N + j] = R[(size_t)i * Npad + j] + Cinv[(size_t)i * Npad + j];
}
int launch_pref_build(const double *base, int N, int Npad, double diag, int nnz, const long long *lin, const double *val,
                      double *out, hipStream_t s)
{
    hipLaunchKernelGGL(pref_build_kernel, dim3((Npad + 255) / 256, Npad), dim3(256), 0, s, base, N, Npad, diag, out);
    if (nnz > 0) hipLaunchKernelGGL(pref_scatter_kernel, dim3((nnz + 255) / 256), dim3(256), 0, s, nnz, lin, val, N, Npad, out);
    return (int)hipGetLastError();
}
int launch_pref_sum(const double *R, const double *Cinv, int N, int Npad, double *A, hipStream_t s)
{
    hipLaunchKernelGGL(pref_sum_kernel, dim3((N + 255) / 256, N), dim3(256), 0, s, R, Cinv, N, Npad, A);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------
// Marginal likelihood scalars |L^-1 y|^2 and sum log L_ii (ego/gaussianprocess/trainhyper.py:60-68)
// without a separate triangular solve: append y as row N of the matrix
// being factored ([[K, y],[y^T, c]]); after the Cholesky that row IS z = L^-1 y, produced by
// the factorisation's own trsm/syrk kernels.  c is huge so the extra pivot never fails.
// ------------------------------------------------------------------------
// The pad rows below the y row are rewritten as identity rows every time: a factorisation that failed (not positive
// definite) leaves NaNs in them, and the matrix slot is used again.
__global__ void aug_row_kernel(double *__restrict__ L, int Npad, int N, const double *__restrict__ y, size_t lstride)
{
    const int k = blockIdx.x * 256 + threadIdx.x, r = N + blockIdx.y;
    if (k >= Npad) return;
    L += blockIdx.z * lstride;
    if (blockIdx.y == 0) {
        if (k < N) L[(size_t)N * Npad + k] = y[k];
        else if (k == N) L[(size_t)N * Npad + N] = 1e300;
    } else {
        L[(size_t)r * Npad + k] = (k == r) ? 1.0 : 0.0;
    }
}

__global__ __launch_bounds__(256) void nlml_reduce_kernel(const double *__restrict__ L, int Npad, int N,
                                                          double *__restrict__ out2, size_t lstride)
{
    __shared__ double rq[256], rl[256];
    const int t = threadIdx.x;
    L += blockIdx.x * lstride; out2 += 2 * blockIdx.x;       // one workgroup per matrix of the batch
    double q = 0.0, ld = 0.0;
    for (int k0 = t; k0 < N; k0 += 8 * 256) {         // same order of the sums; eight diagonal entries' loads in flight
        double z[8], dg[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = k0 + 256 * u;
            z[u] = k < N ? L[(size_t)N * Npad + k] : 0.0;
            dg[u] = k < N ? L[(size_t)k * Npad + k] : 1.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (k0 + 256 * u < N) { q = fma(z[u], z[u], q); ld += log(dg[u]); }
    }
    rq[t] = q; rl[t] = ld;
    __syncthreads();
    if (t == 0) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < 256; i++) { a += rq[i]; b += rl[i]; }
        out2[0] = a; out2[1] = b;
    }
}

// out2 = (y . alpha, sum_i log L_ii): the two scalars of the marginal likelihood when alpha is at hand (ibo_nlml_grad)
__global__ __launch_bounds__(256) void nlml_scalars_kernel(const double *__restrict__ L, int Npad, int N, const double *__restrict__ y,
                                                           const double *__restrict__ alpha, double *__restrict__ out2)
{
    __shared__ double rq[256], rl[256];
    const int t = threadIdx.x;
    double q = 0.0, ld = 0.0;
    for (int k0 = t; k0 < N; k0 += 8 * 256) {
        double yy[8], al[8], dg[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = k0 + 256 * u;
            yy[u] = k < N ? y[k] : 0.0;
            al[u] = k < N ? alpha[k] : 0.0;
            dg[u] = k < N ? L[(size_t)k * Npad + k] : 1.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (k0 + 256 * u < N) { q = fma(yy[u], al[u], q); ld += log(dg[u]); }
    }
    rq[t] = q; rl[t] = ld;
    __syncthreads();
    if (t == 0) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < 256; i++) { a += rq[i]; b += rl[i]; }
        out2[0] = a; out2[1] = b;
    }
}
int launch_nlml_scalars(const double *L, int Npad, int N, const double *y, const double *alpha, double *out2, hipStream_t s)
{
    hipLaunchKernelGGL(nlml_scalars_kernel, dim3(1), dim3(256), 0, s, L, Npad, N, y, alpha, out2);
    return (int)hipGetLastError();
}

int launch_nlml_aug(double *L, int Npad, int N, const double *y, hipStream_t s, int batch, size_t lstride)
{
    hipLaunchKernelGGL(aug_row_kernel, dim3((Npad + 255) / 256, Npad - N, batch), dim3(256), 0, s, L, Npad, N, y, lstride);
    return (int)hipGetLastError();
}

int launch_nlml_reduce(const double *L, int Npad, int N, double *out2, hipStream_t s, int batch, size_t lstride)
{
    hipLaunchKernelGGL(nlml_reduce_kernel, dim3(batch), dim3(256), 0, s, L, Npad, N, out2, lstride);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------
// fp64 MFMA fragment-layout self test: asymmetric integer operands, exact.
// ------------------------------------------------------------------------
__global__ void mfma_selftest_kernel(double *out_err)
{
    __shared__ double Cm[256];
    int l = threadIdx.x;
    // A[i][k] = 3 i + 7 k + 1 ; B[k][j] = 5 k - 2 j + (k == 1 ? 11 : 0)
    int ai = l & 15, ak = l >> 4;
    double a = 3.0 * ai + 7.0 * ak + 1.0;
    int bk = l >> 4, bj = l & 15;
    double b = 5.0 * bk - 2.0 * bj + (bk == 1 ? 11.0 : 0.0);
    d4_t acc = {0, 0, 0, 0};
    acc = mfma_f64(a, b, acc);
    for (int r = 0; r < 4; r++) Cm[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
    __syncthreads();
    double err = 0.0;
    for (int e = l; e < 256; e += 64) {
        int i = e >> 4, j = e & 15;
        double ref = 0.0;
        for (int k = 0; k < 4; k++) ref += (3.0 * i + 7.0 * k + 1.0) * (5.0 * k - 2.0 * j + (k == 1 ? 11.0 : 0.0));
        err = fmax(err, fabs(ref - Cm[e]));
    }
    for (int o = 32; o > 0; o >>= 1) err = fmax(err, __shfl_xor(err, o));
    if (l == 0) out_err[0] = err;
}

int launch_mfma_selftest(double *out_err, hipStream_t s)
{
    hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, s, out_err);
    return (int)hipGetLastError();
}
