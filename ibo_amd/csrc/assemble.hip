// assemble.hip -- everything of a fit that is not the factorisation, on gfx950: covariance assembly, coordinate scaling and padding,
// transposes and MFMA-fragment packing, the alpha vectors, the one-point block extension, the preference GP's matrix assembly, the
// marginal likelihood's scalars and its gradient contraction, the fp64-MFMA layout self test.
//
// Replaces (reference, /root/reference):
//   GaussianProcess._computeCorrelations   ego/gaussianprocess/__init__.py:134-149
//   Kernel.covMatrix                       ego/gaussianprocess/kernel.py:46-53
//   addData's block extension              ego/gaussianprocess/__init__.py:301-308
//   marginalLikelihood's gradient          ego/gaussianprocess/trainhyper.py:70-75
#include "ibo_common.h"

// ------------------------------------------------------------------------
// covariance matrix  K[i][j] = k(A1_i, A2_j)
// ------------------------------------------------------------------------
// 64 x 64 entries of K per workgroup, 4 x 4 per thread: the tile's points are staged in LDS once and every value read
// from there serves four entries (per dimension 8 LDS reads and 48 fp64 instructions for 16 entries).  A thread's columns
// are 16 apart: a row's store instruction covers whole 128-byte segments.
// (z is accumulated in the reference's order: sum_d w_d (a_d - b_d)^2 -- what GP.R, ibo_cov_matrix and everything else a caller
// can read back are made of; the fit's working copy comes from cov_fit_kernel, the likelihood grid's from cov_grid_kernel.)
// row stride of the staged points: odd (conflict-free column reads), >= the dimension: 33 up to 32 dimensions, 65 beyond (a
// template parameter: the wider stride halves the workgroups a CU holds)
#define COV_LD LD
template <int LD>
__global__ __launch_bounds__(256) void cov_matrix_kernel(KParams kp, int n1, const double *__restrict__ A1, int n2,
                                                         const double *__restrict__ A2, int ldp, int square,
                                                         int diag_rule, double noise, double *__restrict__ K, int ldk, int lower_only)
{
    __shared__ double AB[2 * 64 * COV_LD];           // the two tiles' points; afterwards the tile itself, transposed (64 x 65)
    double *As = AB, *Bs = AB + 64 * COV_LD;
    static_assert(2 * 64 * LD >= 64 * 65, "the transposed tile reuses the staging buffers");
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int j0 = blockIdx.x * 64, i0 = blockIdx.y * 64, D = kp.D;
    // K(X, X) is symmetric bit for bit ((a - b)^2 = (b - a)^2): a tile below the diagonal also writes its mirror image,
    // the tiles above the diagonal compute nothing (half the fp64 exps)
    const bool mirror = square && !lower_only && j0 < i0;
    if (square && j0 > i0) return;
    for (int e = t; e < 64 * D; e += 256) {
        const int r = e / D, d = e - r * D;
        As[r * COV_LD + d] = (i0 + r < n1) ? A1[(size_t)(i0 + r) * ldp + d] : 0.0;
        Bs[r * COV_LD + d] = (j0 + r < n2) ? A2[(size_t)(j0 + r) * ldp + d] : 0.0;
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
                z[r][c] += w * (u * u);
            }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int i = i0 + ty * 4 + r;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int j = j0 + tx + 16 * c;
            if (i < n1 && j < n2) {
                double v = cov_from_z_rt(kp.family, z[r][c], kp.sf2);
                if (square && i == j) {
                    // diag_rule 0: the reference never calls the kernel on the diagonal and
                    // hard-wires 1+noise (ego/gaussianprocess/__init__.py:138)
                    v = (diag_rule == 0) ? (1.0 + noise) : (v + noise);
                }
                K[(size_t)i * ldk + j] = v;
                z[r][c] = v;
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

int launch_cov_matrix(const KParams &kp, int n1, const double *A1, int n2, const double *A2,
                      int ldp, int diag_rule, double noise, double *K, int ldk, hipStream_t s, int lower_only)
{
    const int square = (A2 == nullptr);
    if (square) { A2 = A1; n2 = n1; }
    dim3 grid((n2 + 63) / 64, (n1 + 63) / 64);
    if (kp.D <= 32) hipLaunchKernelGGL(cov_matrix_kernel<33>, grid, dim3(256), 0, s, kp, n1, A1, n2, A2, ldp, square, diag_rule, noise, K, ldk, square ? lower_only : 0);
    else hipLaunchKernelGGL(cov_matrix_kernel<65>, grid, dim3(256), 0, s, kp, n1, A1, n2, A2, ldp, square, diag_rule, noise, K, ldk, square ? lower_only : 0);
    return (int)hipGetLastError();
}

// The FIT's covariance pass by itself (round 4): only what a factorisation reads -- the 64 x 64 blocks of K(X, X) + diag on and below the
// diagonal, padded with the identity to np2 x np2, into the working copy; the identity the W ride-along starts from (its blocks on and right of
// the diagonal) and the cleared info word in the same pass.  GP.R is formed on request (abi.hip ensure_R, cov_matrix_kernel).  Entry by entry
// the arithmetic of cov_matrix_kernel: the same bits.  TS x TS entries per workgroup of 256 threads: with 64 x 64 tiles a 1024-point fit
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

// The likelihood grid's covariance pass: lower part of K(X, X) + noise I for `batch` parameter sets, one launch.
// 64 x 128 entries per workgroup (4 x 8 per thread: 12 LDS reads per dimension for 32 entries), workgroups numbered over the
// tiles that touch the lower triangle only (a 2-D grid would dispatch as many dead workgroups as live ones), a row's 128
// columns leave in eight consecutive 128-byte stores -- 1 KiB per row and tile: with 64 x 64 tiles (512-byte row segments) the
// 4.3 GB of a 64-matrix grid went out at 1.7 TB/s.  Coordinates scaled by sqrt(w_d) on their way into LDS (two instead of three fp64 instructions per
// dimension and entry) and the 16-instruction exp_fast / 7-instruction sqrt_fast of the sweep (relative error < 5e-16) instead of the library's:
// these matrices never leave the device.
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

// The same pass with the exponent on the MFMA unit (round 6; north_star: "MFMA only for the dense K(X,X) / K(X,X*) GEMM-shaped blocks";
// maths: ego/gaussianprocess/kernel.py:46-53,147-149, trainhyper.py:55).  As in the candidate sweep (sweep2_kernels.h), -z/2 of a pair is
//     y_ij = a_i + a_j + x~_i . x~_j,   a_k = -|x~_k|^2 / 2,   x~ = x sqrt(w)  (the theta-point's scaling),
// i.e. the product [x~_i | a_i | 1] [x~_j | 1 | a_j]^T: ceil((D + 2) / 4) fp64 MFMAs per 16 x 16 entries where the difference form spends
// 2 D VALU instructions per entry (52 with its exp at D = 16: the kernel was bound by them, 1.45 ms per 64-theta grid against 0.9 ms for
// its 4.3 GB of stores).  Same tiles and numbering as cov_grid_kernel (64 rows x 128 columns, four waves: wave w takes rows 16 w .. of all
// eight column blocks, so a row's 1 KiB still leaves the workgroup together); the augmented rows are built in LDS with an odd stride
// (conflict-free fragment reads); exp_fast / sqrt_fast as before.  |y|'s absolute error is ~|x~|^2 2^-52, hence the caller's guard
// (|x~|^2 <= 1e5 for every row and every theta-point of the call: launch_cov_matrix_batched's `dot_ok`), beyond which the difference form runs.
// The 64 x 64 diagonal blocks are written whole, as before; inside them K_ij and K_ji may differ in the last bit (the two a's enter the
// sum in the other order) -- the factorisation reads the lower triangle only.
template <int KA4>
__global__ __launch_bounds__(256) void cov_grid_mfma_kernel(const KParams *__restrict__ kps, int n, const double *__restrict__ X, int ldp,
                                                            double noise, double *__restrict__ K, int ldk, size_t kstride)
{
    constexpr int KA = 4 * KA4, LD = KA + 1;
    __shared__ double As[64 * LD];                   // [x~ | a | 1 | 0 ..]
    __shared__ double Bs[128 * LD];                  // [x~ | 1 | a | 0 ..]
    const KParams &kp = kps[blockIdx.z];
    K += blockIdx.z * kstride;
    const int t = threadIdx.x, lane = t & 63, D = kp.D;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int q = blockIdx.x;
    int a = (int)((sqrt(1.0 + 4.0 * (double)q) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) <= q) a++;
    while (a * (a + 1) > q) a--;
    const int rem = q - a * (a + 1);
    const int I = 2 * a + rem / (a + 1), J = rem % (a + 1);
    const int i0 = 64 * I, j0 = 128 * J;
    if (i0 >= n) return;
    for (int e = t; e < 64 * KA; e += 256) {
        const int r = e / KA, d = e - r * KA;
        As[r * LD + d] = (d < D && i0 + r < n) ? X[(size_t)(i0 + r) * ldp + d] * kp.sw[d] : 0.0;
    }
    for (int e = t; e < 128 * KA; e += 256) {
        const int r = e / KA, d = e - r * KA;
        Bs[r * LD + d] = (d < D && j0 + r < n) ? X[(size_t)(j0 + r) * ldp + d] * kp.sw[d] : 0.0;
    }
    __syncthreads();
    if (t < 192) {                                   // the rows' -|x~|^2 / 2 and the constant 1, each where its operand wants it
        double *row = t < 64 ? As + t * LD : Bs + (t - 64) * LD;
        double n2 = 0.0;
        for (int d = 0; d < D; d++) n2 = fma(row[d], row[d], n2);
        row[D + (t < 64 ? 0 : 1)] = -0.5 * n2;
        row[D + (t < 64 ? 1 : 0)] = 1.0;
    }
    __syncthreads();
    const int fam = kp.family;
    const double log_sf2 = log(kp.sf2), sf2 = kp.sf2;
    double af[KA4];
    {
        const double *ap = As + (16 * wave + (lane & 15)) * LD + (lane >> 4);
#pragma unroll
        for (int s = 0; s < KA4; s++) af[s] = ap[4 * s];
    }
    const int ib = i0 + 16 * wave;                   // this wave's first row
#pragma unroll 2
    for (int cb = 0; cb < 8; cb++) {
        const int jb = j0 + 16 * cb;
        if (jb >= n || (jb >> 6) > (ib >> 6)) continue;          // beyond the data, or a 64 x 64 block right of the diagonal one (wave-uniform)
        const double *bp = Bs + (16 * cb + (lane & 15)) * LD + (lane >> 4);
        d4_t y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KA4; s++) y = mfma_f64(af[s], bp[4 * s], y);
        const int j = jb + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = ib + (lane >> 4) + 4 * r;
            double v;
            if (fam == FAM_SE) v = exp_fast(y[r] + log_sf2);
            else {
                const double z = fmax(-2.0 * y[r], 0.0);
                const double rr = sqrt_fast((fam == FAM_M3 ? 3.0 : 5.0) * z);
                const double poly = fam == FAM_M3 ? 1.0 + rr : fma(rr, fma(rr, 1.0 / 3.0, 1.0), 1.0);
                v = sf2 * poly * exp_fast(-rr);
            }
            if (i == j) v = sf2 + noise;             // k(x, x) + noise, exactly
            if (i < n && j < n) K[(size_t)i * ldk + j] = v;
        }
    }
}

// `batch` covariance matrices K(A1, A1) + noise I (the blocks on and below the diagonal), parameters kps_dev[z] (device), outputs kstride
// doubles apart (ldp = the dimension: the points are handed over unpadded).  dot_ok: every scaled point of every parameter set lies within
// |x~|^2 <= 1e5 (the caller's bound) -- the exponent then comes from the MFMA unit, else from coordinate differences.
int launch_cov_matrix_batched(const KParams *kps_dev, int batch, int n1, const double *A1, int ldp, int diag_rule, double noise,
                              double *K, int ldk, size_t kstride, hipStream_t s, int dot_ok)
{
    if (diag_rule != 1) return (int)hipErrorInvalidValue;      // (IBO_DIAG_KERNEL_PLUS_NOISE: the likelihood's rule is the only one a grid has)
    const int nI = (n1 + 63) / 64;                   // tiles: sum over rows I of I / 2 + 1
    long ntile = 0;
    for (int I = 0; I < nI; I++) ntile += I / 2 + 1;
    const dim3 grid((unsigned)ntile, 1, batch);
    if (dot_ok && ldp <= IBO_DDOT) {
#define COV_MFMA(KA4) hipLaunchKernelGGL(cov_grid_mfma_kernel<KA4>, grid, dim3(256), 0, s, kps_dev, n1, A1, ldp, noise, K, ldk, kstride)
        switch ((ldp + 2 + 3) / 4) {
        case 1: COV_MFMA(1); break; case 2: COV_MFMA(2); break; case 3: COV_MFMA(3); break; case 4: COV_MFMA(4); break;
        case 5: COV_MFMA(5); break; case 6: COV_MFMA(6); break; case 7: COV_MFMA(7); break; case 8: COV_MFMA(8); break;
        default: COV_MFMA(9); break;
        }
#undef COV_MFMA
    } else if (ldp <= 32) hipLaunchKernelGGL(cov_grid_kernel<33>, grid, dim3(256), 0, s, kps_dev, n1, A1, ldp, noise, K, ldk, kstride);
    else hipLaunchKernelGGL(cov_grid_kernel<65>, grid, dim3(256), 0, s, kps_dev, n1, A1, ldp, noise, K, ldk, kstride);
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
// the wave's 64 values summed in a fixed order without a trip through the LDS crossbar: four DPP steps inside the rows of 16 lanes (xor 1, xor 2,
// half-row mirror, row mirror), then the four row sums, first to last
__device__ __forceinline__ double wave_sum_rows(double v)
{
#define WS_STEP(ctrl) { const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, true), \
                                  hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, true); v += __hiloint2double(hi, lo); }
    WS_STEP(0xB1) WS_STEP(0x4E) WS_STEP(0x141) WS_STEP(0x140)
#undef WS_STEP
    auto row = [&](int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); };
    return ((row(0) + row(16)) + row(32)) + row(48);
}

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
        s = wave_sum_rows(s);
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

int launch_nlml_grad(const KParams &kp, const GradSpec &gs, int N, const double *X, int ldx, const double *Kinv, int ldk,
                     const double *alpha, double *partial, double *out, hipStream_t s)
{
    dim3 grid((N + 63) / 64, (N + 63) / 64);
    // at most 17 derivatives per pass (17 accumulators per thread: 64 VGPRs, no spills): beyond 16 dimensions the components go
    // in two passes that each rebuild K_ab -- the 33-accumulator instantiation needed 256 VGPRs, 232 spilled SGPRs, occupancy 1
    const int nblk = (int)(grid.x * grid.y);
    if (gs.nh <= 33) {                                  // up to 33 derivatives: the shared-coordinate kernel; beyond (33 .. 64 dimensions): the general one
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

// ---- do two streams run side by side?  HIP maps its streams onto a handful of hardware queues (four unless GPU_MAX_HW_QUEUES says
// otherwise) and a stream created when all are taken shares one -- possibly with the very stream it is meant to run beside: the
// likelihood grid's two sub-batch streams then take turns (measured: 29.7 instead of 27.5 ms per 64-theta grid, depending on nothing
// but how many streams the process had created before).  No API names a stream's queue, so the pair is probed: a wave that sleeps
// ~40 us on each stream, the span from the first's start to the second's end: one sleep when they overlap, two when they share a queue.
__global__ void stream_probe_kernel(long long ticks)
{
    const long long t0 = wall_clock64();                   // (100 MHz)
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
int streams_run_side_by_side(hipStream_t a, hipStream_t b, bool *yes)
{
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 2 && e == hipSuccess; rep++) {      // (the first launch of a kernel pays its load)
        e = hipEventRecord(e0, a);
        hipLaunchKernelGGL(stream_probe_kernel, dim3(1), dim3(64), 0, a, 4000LL);
        hipLaunchKernelGGL(stream_probe_kernel, dim3(1), dim3(64), 0, b, 4000LL);
        if (e == hipSuccess) e = hipEventRecord(e1, b);
        if (e == hipSuccess) e = hipStreamSynchronize(a);
        if (e == hipSuccess) e = hipStreamSynchronize(b);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    *yes = best < 0.065f;                                  // 40 us side by side, 80 one after the other
    return (int)e;
}

void ibo_touch_assemble() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, (const void *)pad_copy_kernel); }     // (see small2.hip: ibo_touch_small2)
