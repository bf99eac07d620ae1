// ibo_common.h -- shared declarations for the gfx950 implementation of the
// GP-posterior + acquisition path (see include/ibo_abi.h for the boundary).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#define IBO_DMAX 16            // largest input dimensionality handled on device

// covariance families after normalising the reference's four kernel types to
// "weighted squared distance z = sum_d w_d (x_d - c_d)^2, then a scalar map":
//   SE  (ARD: w_d = 1/theta_d^2, ISO: w_d = 1/theta^2)  k = sf2 exp(-z/2)
//   M3  (w_d = 1/theta^2)  r = sqrt(3 z)  k = sf2 (1 + r) exp(-r)
//   M5  (w_d = 1/theta^2)  r = sqrt(5 z)  k = sf2 (1 + r + r^2/3) exp(-r)
// (cpp/optimizeGP.cpp:67-113, ego/gaussianprocess/kernel.py:87-89,147-149,207-210,246-249)
enum { FAM_SE = 0, FAM_M3 = 1, FAM_M5 = 2 };

struct KParams {
    int family;
    int D;                      // true dimensionality (<= IBO_DMAX)
    double sf2;
    double w[IBO_DMAX];         // zero beyond D
};

template <int FAM>
__device__ __forceinline__ double cov_from_z(double z, double sf2)
{
    if (FAM == FAM_SE) return sf2 * exp(-0.5 * z);
    if (FAM == FAM_M3) { double r = sqrt(3.0 * z); return sf2 * (1.0 + r) * exp(-r); }
    double r = sqrt(5.0 * z);
    return sf2 * (1.0 + r + r * r * (1.0 / 3.0)) * exp(-r);
}

__device__ __forceinline__ double cov_from_z_rt(int fam, double z, double sf2)
{
    if (fam == FAM_SE) return cov_from_z<FAM_SE>(z, sf2);
    if (fam == FAM_M3) return cov_from_z<FAM_M3>(z, sf2);
    return cov_from_z<FAM_M5>(z, sf2);
}

typedef double d4_t __attribute__((ext_vector_type(4)));

// fp64 MFMA 16x16x4: D(16x16) += A(16x4) * B(4x16).  Lane l supplies
// A[row = l&15][k = l>>4] and B[k = l>>4][col = l&15]; it receives
// D[row = (l>>4) + 4*r][col = l&15] in element r (cdna_hip_programming.md s3).
__device__ __forceinline__ d4_t mfma_f64(double a, double b, d4_t c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// RBF-network mean prior parameters (device pointers)
struct PriorDev {
    int nb;
    double theta;
    const double *means;    // nb x D
    const double *beta;     // nb
    const double *lowerb;   // D
    const double *width;    // D
};

// acquisition value (positive) from (mu, sigma); mirrors
// cpp/optimizeGP.cpp:194-236 (libm) and ego/acquisition/__init__.py:68-71,107-110,150-164
// with CDF/PDF of ego/gaussianprocess/__init__.py:55-77 (NR).
__device__ __forceinline__ double erf_nr_dev(double z)
{
    double t = 1.0 / (1.0 + 0.5 * fabs(z));
    double p = 0.17087277;
    p = -0.82215223 + t * p;
    p = 1.48851587 + t * p;
    p = -1.13520398 + t * p;
    p = 0.27886807 + t * p;
    p = -0.18628806 + t * p;
    p = 0.09678418 + t * p;
    p = 0.37409196 + t * p;
    p = 1.00002368 + t * p;
    double ans = 1.0 - t * exp(-z * z - 1.26551223 + t * p);
    return z >= 0.0 ? ans : -ans;
}

__device__ __forceinline__ double acq_value_dev(int acq, int erf_mode, double mu, double sigma,
                                                double ymax, double parm)
{
    if (acq == 2) return mu + parm * sigma;
    double ydiff = mu - ymax - parm;
    double Z = ydiff / sigma;
    double cdf, pdf;
    if (erf_mode == 0) {
        cdf = 0.5 * (1.0 + erf(Z / sqrt(2.0)));
        pdf = exp(-(Z * Z / 2.0)) / sqrt(2.0 * M_PI);
    } else {
        cdf = 0.5 * (1.0 + erf_nr_dev(Z * 0.707106));
        pdf = exp(-(Z * Z / 2.0)) * 0.398942;
    }
    if (acq == 1) return cdf;
    return ydiff * cdf + sigma * pdf;
}

// ---- host-side launch API of the kernels (defined in linalg.hip / sweep.hip)
struct SweepArgs {
    KParams kp;
    int N, Npad, DP;
    int64_t M;
    const double *Xp;        // Npad x DP, zero padded
    const double *W;         // Npad x Npad row-major lower-triangular, q = |W k*|^2
    const double *Wp;        // same matrix in MFMA fragment order (see pack_w_kernel)
    const double *alphaY;    // Npad
    const double *alpha1;    // Npad
    const double *cand;      // M x D
    PriorDev prior;
    double noise, clamp_lo, ymax, parm;
    int acq, erf_mode;
    int n_excl; const double *excl; double excl_radius;   // n_excl x D (device)
    int64_t index_base;
    double *out_mu, *out_s2, *out_acq;   // optional
    double *part_val; int64_t *part_idx; // one per 64-candidate tile
    double *qpart;                       // gemv path scratch: rowchunks x M
    double *mupart;                      // gemv path scratch: 2 x M
    double *result_val; int64_t *result_idx;   // device, single element each
};

int launch_sweep_mfma(const SweepArgs &a, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
int launch_sweep_gemv(const SweepArgs &a, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
int launch_argmax_final(const SweepArgs &a, int64_t ntiles, hipStream_t s);

int launch_cov_matrix(const KParams &kp, int n1, const double *A1, int n2, const double *A2,
                      int lda_pts, int diag_rule, double noise, double *K, int ldk, hipStream_t s);
// factor the Npad x Npad matrix in L (lower part, ld = Npad) in place; diag64 receives the
// inverses of the 64x64 diagonal blocks; info (device int) gets the 1-based failing pivot or 0
int launch_cholesky(double *L, int Npad, double *diag64, int *info_dev, hipStream_t s);
// W = L^-1 (row-major, ld = Npad) using diag64 from launch_cholesky and a scratch T (Npad x Npad)
int launch_trinv(const double *L, int Npad, const double *diag64, double *W, double *T, hipStream_t s);
// zero the strict upper triangle (ld = Npad)
int launch_zero_upper(double *A, int Npad, hipStream_t s);
// Wout/Wp from S; mode 0: W[i][j] = S[i][j]; mode 1: W[i][j] = S[N-1-j][N-1-i] (i,j < N);
// rows/cols >= N are zero.  Wout may be NULL or == S only for mode 0.
int launch_pack_w(const double *S, int N, int Npad, int mode, double *Wout, double *Wp, hipStream_t s);
// alpha = W^T (W y) for two right-hand sides at once (y and the all-ones vector)
int launch_alpha(const double *W, int N, int Npad, const double *y, double *tmp2, double *alphaY,
                 double *alpha1, hipStream_t s);
// z = L^-1 y (blocked forward substitution), returns |z|^2 and sum(log diag L) in out2[0..1]
int launch_fwd_quad_logdet(const double *L, int N, int Npad, const double *diag64, const double *y,
                           double *z, double *out2, hipStream_t s);
int launch_pad_copy(const double *src, int N, int lds, double *dst, int Npad, double pad_diag, hipStream_t s);
int launch_mfma_selftest(double *out_err, hipStream_t s);
