// ibo_common.h -- shared declarations for the gfx950 implementation of the
// GP-posterior + acquisition path (see include/ibo_abi.h for the boundary).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <vector>

#define IBO_DMAX 64            // largest input dimensionality handled on device (rows of X are padded to DP = 4, 8, 16, 32 or 64)
#define IBO_DDOT 32            // ... and the largest for which the dot-form kernels (sweep2.hip, small2.hip: exponent GEMM of up to
                               // nine k4-steps) are instantiated; beyond, the difference-form kernels of sweep.hip take every batch

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
    double sw[IBO_DMAX];        // sqrt(w): coordinates are pre-scaled for the sweep's k* generation
};

template <int FAM>
__device__ __forceinline__ double cov_from_z(double z, double sf2)
{
    if (FAM == FAM_SE) return sf2 * exp(-0.5 * z);
    if (FAM == FAM_M3) { double r = sqrt(3.0 * z); return sf2 * (1.0 + r) * exp(-r); }
    double r = sqrt(5.0 * z);
    return sf2 * (1.0 + r + r * r * (1.0 / 3.0)) * exp(-r);
}

// exp(y) for the k* generation in 16 VALU instructions (the library's takes ~30).
// Next to a saturated fp64 MFMA pipe EVERY VALU instruction -- fp64, fp32 or integer --
// costs 7-13 pipe cycles (tools/mfma_f64_peak), so instruction count is sweep time.
//   t = y log2(e) + 1.5*2^52 puts n = rint(y log2 e) in the low mantissa bits of t;
//   r = y - n ln2 (two-term Cody-Waite);  degree-10 near-minimax polynomial on
//   |r| <= ln2/2 (max relative error 4.4e-16, fitted at Chebyshev nodes);
//   2^n by an integer add into the exponent field.  Valid for -708 <= y <= 709;
//   smaller y is clamped (result 3e-308 instead of a denormal or 0).
__device__ __forceinline__ double exp_fast(double y)
{
    y = fmax(y, -708.0);
    const double magic = 6755399441055744.0;               // 1.5 * 2^52
    const double t = fma(y, 1.4426950408889634074, magic);
    const double n = t - magic;
    double r = fma(n, -6.93147180369123816490e-01, y);
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = 2.76263763215172631873e-07;
    p = fma(p, r, 2.76401811512398433028e-06);
    p = fma(p, r, 2.48015043005816596198e-05);
    p = fma(p, r, 1.98411702694117908236e-04);
    p = fma(p, r, 1.38888889325245227201e-03);
    p = fma(p, r, 8.33333338566888577603e-03);
    p = fma(p, r, 4.16666666665730586749e-02);
    p = fma(p, r, 1.66666666665543999892e-01);
    p = fma(p, r, 5.00000000000000555112e-01);
    p = fma(p, r, 1.00000000000000666134e+00);
    p = fma(p, r, 1.0);
    const int hi = __double2hiint(p) + (__double2loint(t) << 20);
    return __hiloint2double(hi, __double2loint(p));
}

// sqrt(x), x >= 0, for the sweep's hot loop: v_rsq_f64 and one third-order correction (7 instructions; the
// library sqrt is a ~20-instruction sequence with scaling for denormals and exceptional inputs, and every VALU
// instruction costs MFMA issue slots here).  x is floored at 1e-300 so that x = 0 (a candidate on top of an
// observation) gives 1e-150 instead of 0 * inf; relative error < 2e-16 elsewhere.
__device__ __forceinline__ double sqrt_fast(double x)
{
    x = fmax(x, 1e-300);
    const double y = __builtin_amdgcn_rsq(x);
    const double r = x * y;
    const double e = fma(-r, y, 1.0);                 // 1 - x y^2
    return fma(r * e, fma(0.375, e, 0.5), r);
}

// k* from pre-scaled coordinates.  SE: y = log sf2 - z/2 fed to exp_fast.
template <int FAM>
__device__ __forceinline__ double cov_from_z_fast(double z, double log_sf2, double sf2)
{
    if (FAM == FAM_SE) return exp_fast(fma(-0.5, z, log_sf2));
    if (FAM == FAM_M3) { double r = sqrt_fast(3.0 * z); return sf2 * (1.0 + r) * exp_fast(-r); }
    double r = sqrt_fast(5.0 * z);
    return sf2 * fma(r, fma(r, 1.0 / 3.0, 1.0), 1.0) * exp_fast(-r);
}

__device__ __forceinline__ double cov_from_z_rt(int fam, double z, double sf2)
{
    if (fam == FAM_SE) return cov_from_z<FAM_SE>(z, sf2);
    if (fam == FAM_M3) return cov_from_z<FAM_M3>(z, sf2);
    return cov_from_z<FAM_M5>(z, sf2);
}

typedef double d4_t __attribute__((ext_vector_type(4)));

// which derivative each hyper-parameter index asks for (nlml_grad_kernel)
#define IBO_GRAD_MAX 65          // D length scales + the signal magnitude
struct GradSpec {
    int nh;
    int mode[IBO_GRAD_MAX];     // 0 SE-ARD length scale of dimension dim[h]; 1 SE-iso length scale; 2 signal
                                // magnitude (2K); 3 Matern-3/2 length scale; 4 Matern-5/2 length scale
    int dim[IBO_GRAD_MAX];
};

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

// ---- host-side launch API of the kernels (defined in linalg.hip / assemble.hip / update3.hip / sweep*.hip / small2.hip)
#define IBO_SPLIT_PANEL 64      // rows per workgroup of the small-batch (SPLIT) sweep
#define IBO_S2_TCAND 32         // candidates per workgroup of sweep2_kernel (sweep2.hip)

struct SweepArgs {
    KParams kp;
    int N, Npad, DP;
    int64_t M;
    const double *Xp;        // Npad x DP, zero padded
    const double *Xs;        // Npad x DP, coordinates scaled by sqrt(w_d)  (x~)
    const double *ak;        // Npad: -|x~_k|^2 / 2
    const double *XA;        // [x~ | a_k | 1] in MFMA A-fragment order (pack_xa_kernel), rows padded to a multiple of 128
    const double *exp_tab;   // 2^(j/2048), j < 2048 (device; sweep2's exp)
    double log_sf2;          // log of the k* signal variance
    int dot_form;            // y = ak + bc + x~.c~ (D+1 FMAs) instead of the difference form (2D)
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
    int rank1_row;                             // sweep2_rank1_kernel: the appended row of W being folded into the state (< 0: only the means are formed)
    // Kept state of ibo_acq_sweep_incremental (state5 != 0): qpart = [q_a, aY.k*, a1.k*, zsum, q_b][M]; q = (q_a + q_b) + zsum.
    // q_a: rows [0, h) of W; q_b: rows [h, Npad) -- 0 until the candidate's tile is complete; zsum: the squares the appended rows added.
    // With q_b missing the candidate's variance is an UPPER bound (q can only grow), so EI / UCB computed from it bound its value.
    int state5;
    int part_lo, part_hi;                      // sweep2_kernel<.., PART>: the rows of W this launch covers (part_lo > 0: the second part)
    int part_rows;                             // the model's rows when the state was formed: later rows never enter q_a / q_b (zsum has them)
    int *tile_done;                            // per 32-candidate tile: the last LEVEL of W's rows folded into (q_a, q_b): 0 after the first part,
                                               // part_nlev - 1 once the tile is complete (round 4: up to four levels, see sweep2_part_levels)
    double *tile_ub;                           // per tile: largest value (exact or bound) of its candidates (acq_bound_kernel)
    unsigned long long *part_thresh;           // order-preserving bits of the value a tile's bound must reach to be completed (0: any finite bound)
    int part_all;                              // complete every incomplete tile, whatever its bound
    int part_means;                            // the first part also forms aY.k* and a1.k* (its launch carries the alpha vectors in LDS)
    double part_slack;                         // absolute part of the slack a bound is given against the threshold (s2_part_limit)
    // lazy refresh of such a state: per tile the appended rows already folded into zsum (and the means' age), the selection flags of
    // the launch at hand, the model's rows now, W y (the drift margin's source), and whether tiles may be left stale at all
    int *tile_rows; int *tile_sel; const double *wy; int rank_hi; int part_lazy;
    int part_level, part_nlev;                 // this launch's level (>= 1: only tiles whose tile_done is part_level - 1 run), and how many levels there are
    double nu_max;                             // bound on |(W k*)_i| for any candidate: sf2_k / sqrt(sf2_fit) (abi.hip: run_sweep); the drift margin's factor
    unsigned long long *part_best;             // acq_bound_kernel: running maximum over the COMPLETE tiles, same encoding
    // small2.hip: when set, the last workgroup of the last kernel stores done_seq there (host-visible memory) after all
    // results are out -- the host spins on that word instead of going through an event
    unsigned long long *done_flag; unsigned long long done_seq; unsigned *done_count;
    const double *cand_host;                   // the same candidates where the HOST can read them (pinned staging), or NULL
};

int launch_sweep_mfma(const SweepArgs &a, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
int launch_sweep_gemv(const SweepArgs &a, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
// large batches, dot form: 32-candidate tiles, 1024-row panels, exponent GEMM on the MFMA unit (sweep2.hip)
int launch_sweep2(const SweepArgs &a, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
int launch_sweep2_refresh(const SweepArgs &a, int row_first, int row_last, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
// first sweep of a kept state with the second part of W only where a tile's bound can still win (sweep2.hip); prune = false:
// every tile is completed (same launches, same arithmetic -- the reference the pruned run is held to)
int launch_sweep2_pruned(const SweepArgs &a, bool prune, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
// after launch_sweep2_refresh's rank-1 launches on such a state: complete the tiles whose bound has risen to the best complete value
int launch_sweep2_complete(const SweepArgs &a, hipStream_t s);
int launch_sweep2_pruned_finish_all(const SweepArgs &a, hipStream_t s);
bool sweep2_part_fits(int Npad, int D);
int sweep2_part_nlev(int Npad);                   // levels a new kept state gets (<= ibo_set_option("part_levels"))
void set_part_levels(int v);
int sweep2_part_levels(int Npad, int *h);        // the row splits h[0] < h[1] < .. (multiples of 128, at most 3): level l covers rows [h[l-1], h[l]); returns the number of levels
bool sweep2_fits(int Npad);
bool sweep2_rank1_fits(int Npad, int D);
// small batches (16 < M <= 8192), dot form: k* to HBM, one workgroup per 16-row block of W, fixed-order sums (small2.hip)
int launch_sweep_small(const SweepArgs &a, double *ws, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
size_t small_sweep_workspace(int Npad, int64_t M);
// the resident evaluation server of ibo_direct_max (small2.hip): mailbox layout in doubles -- [0] seq, [1] M, [2] done, [3] state, [8 ..) candidates
#define IBO_SRV_BOX_CAND 8
#define IBO_SRV_EXIT 0xffffffffull
enum { IBO_SRV_READY = 1, IBO_SRV_NOT_RESIDENT = 2, IBO_SRV_LEFT_ON_DEADLINE = 3, IBO_SRV_LEFT = 4 };
#define IBO_SRV_CTL_BYTES 64
bool direct_server_takes(const SweepArgs &a);
int launch_direct_server(const SweepArgs &a, void *ctl, double *box, double *ws, int Mmax, int G, double idle_ms, hipStream_t s, unsigned long long *stamps = nullptr);      // its LDS budget holds both alpha vectors (N <= ~5000)
int launch_pack_xa(const double *Xs, const double *ak, int N, int Npad, int DP, int D, double *XA, hipStream_t s);
int launch_argmax_final(const SweepArgs &a, int64_t ntiles, hipStream_t s);

// K[i][j] = k(A1_i, A2_j) (A2 = NULL: the square matrix K(A1, A1) with the diagonal rule applied; lower_only: only its blocks on and below the diagonal)
int launch_cov_matrix(const KParams &kp, int n1, const double *A1, int n2, const double *A2,
                      int lda_pts, int diag_rule, double noise, double *K, int ldk, hipStream_t s, int lower_only = 0);
int launch_cov_fit(const KParams &kp, int n, const double *X, int ldp, int diag_rule, double noise, double *K2, int np2, double *Eye,
                   int *zero_word, hipStream_t s);          // the fit's own pass: working copy (lower blocks, identity pad), ride-along identity, info word
int launch_cov_matrix_batched(const KParams *kps_dev, int batch, int n1, const double *A1, int ldp, int diag_rule, double noise,
                              double *K, int ldk, size_t kstride, hipStream_t s, int dot_ok = 0);
// factor the Npad x Npad matrix in L (lower part, ld = Npad) in place; diag64 receives the
// inverses of the 64x64 diagonal blocks; info (device int) gets the 1-based failing pivot or 0
// ws (optional): Npad * Npad doubles per matrix (wstride apart), the packed store of the packed-operand trailing update
// (update3.hip: launch_chol_update2); without it the 64 x 64-tile update kernel runs.  Results are bit-identical either way.
int launch_cholesky(double *L, int Npad, double *diag64, int *info_dev, hipStream_t s, double *ws = nullptr);
int launch_cholesky_batched(double *L, int Npad, double *diag64, int *info_dev, int batch, size_t lstride,
                            int panel, hipStream_t s, double *ws = nullptr, size_t wstride = 0);
int launch_chol_update2(double *L, int Npad, int p0, int pend, int batch, size_t lstride, double *ws, size_t wstride,
                        hipStream_t s, const double *Lpanel = nullptr);
// left-looking outer order from one packed copy of the finished block columns (update3.hip); bit-identical to the above
// what a row workgroup of chol_panel_fused_kernel leaves in its matrix's info word when a flag wait runs out (not a pivot: the caller
// runs the batch again through the launches that have no waits)
static const int kPanelWaitTimeout = 0x7ffffff0;
struct CholGroup { double *L, *diag64, *Pk; int *info; int batch; hipStream_t stream; int *flags; };   // flags: 4 ints per matrix (the fused panel kernel's hand-overs) or null      // a sub-batch of matrices (lstride / pstride apart) on its stream
int launch_cholesky_batched_left(const CholGroup *groups, int ngroups, int Npad, size_t lstride, int panel, size_t pstride, int nlive,
                                 int nfactor = 0, int rm_from = 0);
int launch_chol_pack3(const double *L, int Npad, int r0, int c0, int K, int batch, size_t lstride, double *Pk, size_t pstride,
                      hipStream_t s);
int launch_chol_pack3e(const double *E, int Npad, int rend, int c0, int K, double *PkE, hipStream_t s);
int launch_chol_update3(double *L, int Npad, int c0, int width, int nlive, int batch, size_t lstride, const double *Pk,
                        size_t pstride, hipStream_t s);
// the same kernel on any region that starts on the diagonal (rows >= c0, columns [c0, c0 + width)) and any range [kbeg, kend) of packed columns
// C = P P^T (lower 128-tiles) of an upper triangular P on the packed-operand kernel, long K ranges in pieces (update3.hip)
void syrk3_plan(int Npad, int piece, std::vector<int> &tasks, std::vector<int> &sums, int *nslots);
int launch_syrk3(const double *P, double *Pk, double *C, int Npad, const int *tasks_dev, int ntasks, const int *sums_dev, int nsums, double *part,
                 hipStream_t s);
int launch_chol_update3_range(double *L, int Npad, int c0, int width, int kbeg, int kend, int nlive, int batch, size_t lstride,
                              const double *Pk, size_t pstride, hipStream_t s);
// out-of-place, one fused launch per block column (plain right-looking order; `work` is destroyed)
// Ework (identity on entry) / Eout, optional: W = L^-1 rides along -- Eout receives (L^-1)^T, blocks on and above the diagonal
int launch_cholesky_fused(double *work, double *out, int Npad, double *diag64, int *info_dev, hipStream_t s,
                          double *Ework = nullptr, double *Eout = nullptr, bool info_is_zero = false);
// the two-level order (panels of P block columns) out of place, one fused launch per block column inside a panel
int launch_cholesky_fused2(double *work, double *out, int Npad, double *diag64, int *info_dev, int P, hipStream_t s,
                           bool info_is_zero = false, double *ws = nullptr);
// the single-level order in SUPER-PANELS (linalg.hip): the pipelined launches keep their trailing tiles inside a super-panel of 16 block
// columns; the columns beyond take a finished super-panel's steps as ONE deep update from packed operands (update3.hip).  tall: [A ; E] in
// one buffer (2 Npad x Npad: the matrix being reduced, then the ride-along's identity); Pk: 2 Npad^2 doubles (the packed store of
// [out ; Eout]).  Same bits as launch_cholesky_fused.
static const int kSuperFrom = 64;       // block columns (4096 rows) from which it is the faster one (3840 rows: 1.85 against 1.77 ms; 4096: 2.04 / 2.10; 5000: 3.18 / 3.48)
static const int kSuperPanel = 16;
int launch_cholesky_super(double *tall, double *out, int Npad, double *diag64, int *info_dev, hipStream_t s, double *Eout, double *Pk,
                          bool info_is_zero = false);
int launch_transpose_lower(const double *Et, double *W, int Npad, hipStream_t s);
// W = Et^T (lower, rows >= N zero) and its MFMA-fragment-order copy Wp (another buffer than Et) in one pass
int launch_transpose_pack(const double *Et, int N, int Npad, double *W, double *Wp, hipStream_t s);
// W = L^-1 (row-major, ld = Npad) using diag64 from launch_cholesky and a scratch T (Npad x Npad)
int launch_trinv(const double *L, int Npad, const double *diag64, double *W, double *T, hipStream_t s,
                 bool zero_fill = true);
// zero the strict upper triangle (ld = Npad)
int launch_zero_upper(double *A, int Npad, hipStream_t s);
// Wout/Wp from S; mode 0: W[i][j] = S[i][j]; mode 1: W[i][j] = S[N-1-j][N-1-i] (i,j < N);
// rows/cols >= N are zero.  Wout may be NULL or == S only for mode 0.
int launch_pack_w(const double *S, int N, int Npad, int mode, double *Wout, double *Wp, hipStream_t s);
// alpha = W^T (W y) for two right-hand sides at once (y and the all-ones vector)
int launch_alpha(const double *W, int N, int Npad, const double *y, double *tmp2, double *alphaY,
                 double *alpha1, hipStream_t s);
// Xs = Xp * sqrt(w), ak = -|Xs_k|^2/2; maxnorm2 (device double) receives max_k |Xs_k|^2
int launch_scale_x(const KParams &kp, const double *Xp, int Npad, int DP, double *Xs, double *ak, hipStream_t s);
// C = W^T W (Wt: scratch for the transpose), all Npad x Npad
int launch_wtw(const double *W, double *Wt, double *C, int Npad, hipStream_t s, int lower_only = 0, int wt_ready = 0);     // lower_only: blocks on and below the diagonal
int launch_nlml_grad(const KParams &kp, const GradSpec &gs, int N, const double *X, int ldx, const double *Kinv, int ldk,
                     const double *alpha, double *partial, double *out, hipStream_t s);
// one-point block extension (ibo_gp_extend): kvec[i] = k(x_i, x_N) for i < N (zero beyond) and row / column N of R
int launch_extend_kvec(const KParams &kp, const double *Xp, int ldp, int N, int Npad, double noise, double *R, double *kvec,
                       hipStream_t s);
// d = sqrt(1 + noise - |z|^2); L[N][:N] = z, L[N][N] = d; W[N][:N] = -u/d, W[N][N] = 1/d; row-block N/16 of Wp repacked
int launch_extend_rows(int N, int Npad, double noise, const double *z, const double *u, double *L, double *W, double *Wp,
                       int *info, hipStream_t s);
int launch_nlml_aug(double *L, int Npad, int N, const double *y, hipStream_t s, int batch = 1, size_t lstride = 0);
int launch_nlml_scalars(const double *L, int Npad, int N, const double *y, const double *alpha, double *out2, hipStream_t s);
int launch_nlml_reduce(const double *L, int Npad, int N, double *out2, hipStream_t s, int batch = 1, size_t lstride = 0);
int launch_pad_copy(const double *src, int N, int lds, double *dst, int Npad, double pad_diag, hipStream_t s);
// preference GP: out = base (or 0) + diag I + sparse entries (lin = row * N + col, distinct), identity pad;  A = R + Cinv
int launch_pref_build(const double *base, int N, int Npad, double diag, int nnz, const long long *lin, const double *val,
                      double *out, hipStream_t s);
int launch_pref_sum(const double *R, const double *Cinv, int N, int Npad, double *A, hipStream_t s);
int launch_mfma_selftest(double *out_err, hipStream_t s);
int streams_run_side_by_side(hipStream_t a, hipStream_t b, bool *yes);      // a timing probe: do the two streams sit on different hardware queues?
