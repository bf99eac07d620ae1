// legacy.hip -- libego's arithmetic, bit for bit, for the legacy symbol acqmaxGP (include/ibo_abi.h, section A).
//
// acqmaxGP is handed the caller's inv(R) and evaluates, per sample point of DIRECT (cpp/optimizeGP.cpp:57-170):
//     r_i    = k(x, X_i)                                     (:67-113, libm pow / exp / sqrt)
//     ypred  = m + aMb(r, invR, Y - m)                       (:116-146;  m = RBF-network prior mean, 0 without one)
//     sig2   = clamp(1 + noise - aMb(r, invR, r), 1e-8, 10)  (:149-157)
//     aMb(a, M, b) = sum_i (sum_j M[i][j] b[j]) a[i]          (:173-190, both sums sequential, every product and sum rounded)
// On badly conditioned data (near-duplicate observations, noise 1e-4) the entries of inv(R) reach 1e4 and cancel down to
// O(1): the result carries ~1e-9 of rounding noise against a variance of 1e-4, and that noise is a deterministic function
// of the operation ORDER.  Any other evaluation of the same formula -- a Cholesky factor of inv(R), a blocked or tree
// summation, a fused multiply-add, a k* that differs in its last bit -- lands 1e-5 .. 1e-2 (relative) away from libego's
// number (tools/legacy_probe.py; round 3's legacy path did), and DIRECT's trajectory follows the values.  A drop-in for
// libego under the reference's own ctypes call must return libego's numbers, so this file reproduces the order:
//   * k*, the prior mean and the acquisition formula (O(N D) and O(1) per point) run on the HOST with the host's libm --
//     the same libm libego.so calls, hence the same bits (no device exp is bit-compatible with glibc's);
//   * the O(N^2) contractions run on the DEVICE in libego's order: one thread per row i walks j = 0 .. N-1 with separately
//     rounded multiply and add (inv(R) is read through an exact transposed copy so that the walk is coalesced across
//     rows), then one thread adds the N products Mb[i] a[i] in order.  IEEE-754 multiply and add are the same function on
//     both machines: identical bits.
// Cost: N sequential steps per row instead of a tree -- ~10 us per batch at N = 1000 -- and the host's k* (30 us per point);
// a whole acqmaxGP call (3000 samples) takes ~0.1 s against libego's ~10 s.  The fast path (Cholesky of inv(R), MFMA sweep
// kernels: 2 ms) remains behind ibo_set_option("legacy_exact", 0); the handle-based API never comes here.
#include "ibo_common.h"
#include "legacy.h"

#include <cmath>
#include <cstdlib>
#include <vector>

#pragma clang fp contract(off)     // host and device code of this file: no fused multiply-add, anywhere

// Mb[v][i] = sum_j M[i][j] B[v][j], j ascending, product and sum rounded separately (cpp/optimizeGP.cpp:176-183).
// MT[j * N + i] = M[i][j].  grid (ceil(N / 256), nvec), 256 threads; B's vector goes through LDS in chunks of 2048.
__global__ __launch_bounds__(256) void legacy_matvec_kernel(const double *__restrict__ MT, const double *__restrict__ B,
                                                            double *__restrict__ Mb, int N)
{
    __shared__ double b[2048];
    const int v = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const double *bv = B + (size_t)v * N;
    double acc = 0.0;
    for (int j0 = 0; j0 < N; j0 += 2048) {
        const int n = min(2048, N - j0);
        for (int e = threadIdx.x; e < n; e += 256) b[e] = bv[j0 + e];
        __syncthreads();
        if (i < N) {
            const double *col = MT + (size_t)j0 * N + i;
#pragma unroll 8
            for (int j = 0; j < n; j++) { const double p = col[(size_t)j * N] * b[j]; acc = acc + p; }      // (plain operators under the pragma: HIP's
                                                                                                   // __dmul_rn / __dadd_rn are inline x * y / x + y parsed under the header's contraction mode, and get fused)
        }
        __syncthreads();
    }
    if (i < N) Mb[(size_t)v * N + i] = acc;
}

// out[v] = sum_i Mb[v][i] A[v][i], i ascending (cpp/optimizeGP.cpp:185-186).  One workgroup per vector; a chunk of both
// operands is staged in LDS by all threads, then thread 0 walks it.
// (mb_stride = 0: one Mb for every vector -- inv(R) Y, which is the same for every sample point when there is no prior)
__global__ __launch_bounds__(256) void legacy_dot_kernel(const double *__restrict__ Mb, size_t mb_stride, const double *__restrict__ A, double *__restrict__ out, int N)
{
    __shared__ double m[2048], a[2048];
    const int v = blockIdx.x;
    double x = 0.0;
    for (int i0 = 0; i0 < N; i0 += 2048) {
        const int n = min(2048, N - i0);
        for (int e = threadIdx.x; e < n; e += 256) { m[e] = Mb[(size_t)v * mb_stride + i0 + e]; a[e] = A[(size_t)v * N + i0 + e]; }
        __syncthreads();
        if (threadIdx.x == 0)
            for (int i = 0; i < n; i++) { const double p = m[i] * a[i]; x = x + p; }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[v] = x;
}

__global__ void legacy_transpose_kernel(const double *__restrict__ M, double *__restrict__ MT, int N)
{
    __shared__ double t[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += 8)
        if (by + r < N && bx + threadIdx.x < N) t[r][threadIdx.x] = M[(size_t)(by + r) * N + bx + threadIdx.x];
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8)
        if (bx + r < N && by + threadIdx.x < N) MT[(size_t)(bx + r) * N + by + threadIdx.x] = t[threadIdx.x][r];
}

int launch_legacy_transpose(const double *M, double *MT, int N, hipStream_t s)
{
    hipLaunchKernelGGL(legacy_transpose_kernel, dim3((N + 31) / 32, (N + 31) / 32), dim3(32, 8), 0, s, M, MT, N);
    return (int)hipGetLastError();
}

// aMb for nvec (b, a) pairs: B, A: nvec x N (device); Mb: nvec x N scratch; out: nvec
int launch_legacy_aMb(const double *MT, const double *B, const double *A, double *Mb, double *out, int N, int nvec, hipStream_t s)
{
    hipLaunchKernelGGL(legacy_matvec_kernel, dim3((N + 255) / 256, nvec), dim3(256), 0, s, MT, B, Mb, N);
    hipLaunchKernelGGL(legacy_dot_kernel, dim3(nvec), dim3(256), 0, s, (const double *)Mb, (size_t)N, A, out, N);
    return (int)hipGetLastError();
}

// out[v] = sum_i Mb[i] A[v][i] for one shared Mb
int launch_legacy_dots(const double *Mb, const double *A, double *out, int N, int nvec, hipStream_t s)
{
    hipLaunchKernelGGL(legacy_dot_kernel, dim3(nvec), dim3(256), 0, s, Mb, (size_t)0, A, out, N);
    return (int)hipGetLastError();
}

// ---- host side: the statements of cpp/optimizeGP.cpp in their order, with the host's libm --------------------------------
// r[i] = k(x, X_i) (cpp/optimizeGP.cpp:67-113).  Kernel type 3 reads its magnitude from hyperparams[1] (the reference reads
// hyperparams[ndim], out of bounds unless ndim = 1, and prints every value: DESIGN 7 -- both deviations are kept out).
static inline double sq(double v) { return v * v; }     // pow(v, 2): libstdc++'s pow(double, int) is __builtin_powi = v * v

void legacy_kstar(int kerneltype, int NA, int NX, const double *X, const double *hyperparams, double sf2, const double *x, double *r)
{
    for (int i = 0; i < NX; i++) {
        double z = 0;
        switch (kerneltype) {
        case 0:
            for (int j = 0; j < NA; j++) z += 1 / sq(hyperparams[j]) * sq(X[NA * i + j] - x[j]);
            r[i] = sf2 * exp(-.5 * z);
            break;
        case 1:
            for (int j = 0; j < NA; j++) z += sq((X[NA * i + j] - x[j]) / hyperparams[0]);
            r[i] = sf2 * exp(-.5 * z);
            break;
        case 2:
            for (int j = 0; j < NA; j++) z += sq((X[NA * i + j] - x[j]) / hyperparams[0]);
            z = sqrt(3) * sqrt(z);
            r[i] = sf2 * (1.0 + z) * exp(-z);
            break;
        default:
            for (int j = 0; j < NA; j++) z += sq(X[NA * i + j] - x[j]);
            z = sqrt(z);
            r[i] = sf2 * (1.0 + sqrt(5) * z / hyperparams[0] + 5 * z * z / (3 * hyperparams[0] * hyperparams[0])) * exp(-(sqrt(5) * z / hyperparams[0]));
            break;
        }
    }
}

// the prior mean at x (cpp/optimizeGP.cpp:118-136)
double legacy_prior_mean(int NA, const double *x, int npbases, const double *pbasismeans, const double *pbasisbeta, double pbasistheta,
                         const double *pbasislowerb, const double *pbasiswidth)
{
    double mu = 0.0;
    for (int i = 0; i < npbases; i++) {
        double d = 0;
        for (int j = 0; j < NA; j++) d += sq((x[j] - pbasislowerb[j]) / pbasiswidth[j] - pbasismeans[i * NA + j]);
        mu += pbasisbeta[i] * exp(-pbasistheta * d);
    }
    return mu;
}

// -EI / -PI / -UCB from the two contractions (cpp/optimizeGP.cpp:139-236): ypred = m + x1, sig2 = clamp(1 + noise - x2)
double legacy_neg_acq(int acqfunc, double prior_mu, double x1, double x2, double noise, double maxY, double parm)
{
    const double ypred = prior_mu + x1;               // (without a prior the reference has no addition: prior_mu is then +0.0 -- exact)
    double sig2 = 1. + noise - x2;
    if (sig2 < 1e-8) sig2 = 1e-8;
    else if (sig2 > 10.) sig2 = 10.;
    const double sigma = sqrt(sig2), mu = ypred;
    if (acqfunc == 2) return -(mu + parm * sigma);
    const double ydiff = mu - maxY - parm;
    const double Z = ydiff / sigma;
    const double cdf = 0.5 * (1. + erf(Z / sqrt(2.)));
    if (acqfunc == 1) return -cdf;
    const double pdf = exp(-(Z * Z / 2.)) / (sqrt(2. * M_PI));
    const double EI = ydiff * cdf + sigma * pdf;
    return -EI;
}
