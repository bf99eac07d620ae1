/*
 * ibo_abi.h -- C ABI of libibo_hip.so, the MI355X (gfx950) implementation of the
 * GP-posterior + acquisition hot path of misterwindupbird/IBO.
 *
 * Two groups of entry points:
 *
 *  (A) LEGACY symbols -- byte-for-byte the signatures the reference's Python
 *      binds with ctypes today, so the .so is a drop-in for cpp/libs/libego:
 *        acqmaxGP   replaces  cpp/optimizeGP.cpp:262-283
 *        logCDFs    replaces  cpp/helpers.cpp:30-56
 *                   bound at  ego/acquisition/__init__.py:343-364
 *        direct     replaces  cpp/direct.cpp:329 (cpp/direct.h:76)
 *                   bound at  ego/utils/optimize.py:320-333
 *
 *  (B) HANDLE-BASED symbols (ibo_*) -- re-entrant, explicit status codes,
 *      batch/candidate-array aware (the legacy ABI has no notion of a
 *      candidate array).  These are what ibo_amd's Python host code calls and
 *      what a maintainer would bind to move GaussianProcess.addData /
 *      posterior(s) / maximize* / fastUCBGallery onto the GPU.
 *
 * Conventions: all matrices row-major fp64; "host" pointers are ordinary
 * process memory borrowed for the duration of the call; "dev" pointers are HIP
 * device memory on the handle's device (from ibo_dev_alloc or any hipMalloc).
 * Every ibo_* function returns an IBO_* status; ibo_last_error() describes the
 * last failure on the calling thread.  No torch / C++ types cross this line.
 */
#ifndef IBO_ABI_H
#define IBO_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IBO_ABI_VERSION 7   /* 2: + ibo_gp_extend, ibo_comm_count; 3: + ibo_pref_*; 4: + ibo_dev_generation; 5: + ibo_sweep_state_info; 6: + ibo_sweep_state_levels;
                             * 7: + ibo_gpu_time_ms, ibo_acq_sweep_exchange, ibo_direct_server_info; options "super_min_nb", "direct_resident", "direct_idle_ms", "arena_mb" -- ibo_set_option knows the keys listed below and nothing else: the experiment switches of rounds 2-4
                             * (nlml_groups, cov_fast, chol_fused, small_local, zero_copy, gallery_lazy, pipe_fit, .. -- about 35 keys) were removed in
                             * round 5 and now return IBO_ERR_ARG "unknown option", as does a NULL key; ibo_nlml_grid's covariance pass is the fast one */

/* status codes */
#define IBO_OK              0
#define IBO_ERR_ARG         1   /* bad argument (null pointer, size, enum)            */
#define IBO_ERR_HIP         2   /* HIP runtime error (see ibo_last_error)              */
#define IBO_ERR_NOT_PD      3   /* matrix not positive definite (numpy LinAlgError)    */
#define IBO_ERR_STATE       4   /* call order (e.g. sweep before fit)                  */
#define IBO_ERR_NO_DEVICE   5   /* no gfx950 device visible -- there is NO CPU fallback */
#define IBO_ERR_COMM        6   /* RCCL error                                          */

/* kernel type codes == the reference's (ego/acquisition/__init__.py:323-333) */
#define IBO_K_SE_ARD   0        /* hyper = D length scales                             */
#define IBO_K_SE_ISO   1        /* hyper = [theta]                                     */
#define IBO_K_MATERN3  2        /* hyper = [theta]  (magnitude goes in sf2)            */
#define IBO_K_MATERN5  3        /* hyper = [theta]  (magnitude goes in sf2)            */

/* acquisition codes == the reference's (ego/acquisition/__init__.py:309-321) */
#define IBO_ACQ_EI   0
#define IBO_ACQ_PI   1
#define IBO_ACQ_UCB  2
#define IBO_ACQ_NONE 3          /* posterior only                                      */

/* erf flavour: libm as cpp/optimizeGP.cpp:200-204, or the
 * Numerical-Recipes fit with truncated constants of
 * ego/gaussianprocess/__init__.py:55-77 */
#define IBO_ERF_LIBM 0
#define IBO_ERF_NR   1

/* diagonal rule for the covariance matrix */
#define IBO_DIAG_UNIT_PLUS_NOISE 0   /* 1+noise : GaussianProcess._computeCorrelations,
                                        ego/gaussianprocess/__init__.py:138            */
#define IBO_DIAG_KERNEL_PLUS_NOISE 1 /* k(x,x)+noise : marginalLikelihood,
                                        ego/gaussianprocess/trainhyper.py:55           */

typedef struct ibo_gp ibo_gp_t;       /* a fitted GP resident on one GPU */
typedef struct ibo_comm ibo_comm_t;   /* an RCCL communicator (one rank per GPU) */

/* ---------------------------------------------------------------- library */
int         ibo_abi_version(void);
const char *ibo_last_error(void);
int         ibo_device_count(int *count);
/* name/arch string of a device ("gfx950...") into buf */
int         ibo_device_name(int device, char *buf, size_t buflen);
/* self-test of the fp64 MFMA fragment layout on the device (returns IBO_OK or
 * IBO_ERR_HIP with a message); cheap, used by smoke() */
int         ibo_selftest_mfma(int device, double *max_abs_err);
/* milliseconds of device time this process has measured with HIP events on `device` so far: fits and block extensions, candidate
 * sweeps (the dominant kernel's span, as ibo_last_sweep_kernel_ms), likelihood grids (a batch's longest sub-batch span) and
 * gradients.  DIRECT's small batches and the copies are not event-timed and not in it.  bench.py reports it as gpu_kernel_s_total
 * so that a line can be related to an outside observer's busy-GPU samples. */
int         ibo_gpu_time_ms(int device, double *ms);
/* The fourteen option keys (everything else is decided by the data: sizes, dimensions, what the caller asks for).
 * Functional:  "legacy_exact" 1/0 -- acqmaxGP in libego's operation order (default) or on the MFMA sweep kernels (see acqmaxGP);
 *   "nlml_batch" B -- matrices per batched factorisation in ibo_nlml_grid (0: as many as 12 GB hold; the values do not depend on it);
 *   "pool_limit_mb" n -- the per-device free list of recycled buffers (ibo_trim);
 *   "arena_mb" n -- MiB per slab of the buffer arena (1024; 0: none): device buffers of up to half a slab are sub-allocated from slabs
 *   taken from the device once, the first when the library first allocates there, so a new model finds warm memory (ibo_trim keeps the first);
 *   "super_min_nb" nb -- block columns from which a single-level factorisation runs in super-panels of 16 (64; linalg.hip launch_cholesky_super:
 *   the columns beyond a super-panel take its steps as one deep update from packed operands -- the same bits, a matter of speed only);
 *   "direct_resident" 0/1, "direct_idle_ms" n -- ibo_direct_max's resident evaluation server (see ibo_direct_max; off);
 *   "fused2_min_nb" nb -- block columns (of 64 rows) from which a single matrix is factored in the two-level order (104; the order fixes the
 *   last bits of L and W -- one rule for ibo_gp_fit, the preference GP and ibo_nlml_grad).
 * Comparators kept for the tests (a second route to the same numbers):  "sweep_path" 0 auto (small2.hip's three kernels up to 4096
 *   candidates, sweep2_kernel above; GEMV / panel-split / first-generation tile kernels where the dot form is not admissible) / 1 GEMV /
 *   2 MFMA tile / 3 panel-split;  "dot_form" -1 auto / 0 / 1 (k* by differences or by the exponent GEMM);  "gallery_prune" 0/1/2 and
 *   "part_levels" 2..4 (see ibo_acq_sweep_incremental);  "host_pipeline" 1/0 (large host batches in overlapped chunks or in one shot);
 *   "chol_left" 1/0 (ibo_nlml_grid's left-looking order or the right-looking one: identical bits).
 * Env: IBO_SWEEP_IMPL=gemv|mfma, IBO_DOT_FORM, IBO_POOL_LIMIT_MB, IBO_HOST_THREADS (the legacy symbol's host crew), IBO_DEVICE (legacy symbols),
 *   IBO_NLML_GROUPS=1..4 (sub-batches of an ibo_nlml_grid batch, each on its own stream; 2; the values do not depend on it).
 * Threading (the reference's library keeps its whole model in process-wide statics, cpp/optimizeGP.cpp:36-55,240-259, and is not
 * re-entrant; this one is): handles are independent of each other -- each has its own stream, events, staging and buffers --
 * so several threads may drive several handles on one device at the same time (one handle belongs to one thread at a time);
 * the buffer pool and the allocation table are mutexed; the per-device workspaces behind ibo_nlml_grid / ibo_nlml_grad and
 * ibo_trim are serialised by a per-device mutex (concurrent grids on one device take turns; a sweep on another handle never waits for them); the last
 * error is thread-local.  The option switches are process-wide CONFIGURATION held in atomics: changing one while another
 * thread computes is defined but takes effect at an unspecified call boundary -- set them before the threads start.
 * (tests/test_gpu_gallery_oracle.py::test_two_threads_two_handles_one_device.) */
int         ibo_set_option(const char *key, int value);

/* ---------------------------------------------------------------- device memory */
int ibo_dev_alloc(int device, size_t bytes, void **dev_ptr);
int ibo_dev_free(int device, void *dev_ptr);
int ibo_memcpy_h2d(int device, void *dev_dst, const void *host_src, size_t bytes);
int ibo_memcpy_d2h(int device, void *host_dst, const void *dev_src, size_t bytes);
int ibo_device_synchronize(int device);
/* Every ibo_dev_alloc allocation carries a generation: a process-wide counter value taken at allocation and again by each
 * ibo_memcpy_h2d into it (0 for memory this library did not allocate).  State kept per candidate array
 * (ibo_acq_sweep_incremental) is keyed on it, never on the address, which hipFree / hipMalloc recycle. */
int ibo_dev_generation(int device, const void *dev_ptr, uint64_t *generation);

/* ---------------------------------------------------------------- model (fit) */
int ibo_gp_create(int device, ibo_gp_t **out);
int ibo_gp_destroy(ibo_gp_t *gp);

/*
 * Fit: replaces GaussianProcess._computeCorrelations + linalg.cholesky
 * (ego/gaussianprocess/__init__.py:134-149,294-299) and the per-call
 * linalg.inv(R) of cdirectGP (ego/acquisition/__init__.py:385-388).
 *   R = K(X,X) with diagonal 1+noise;  L = chol(R);  W = L^-1 (explicit,
 *   kept in an MFMA-fragment layout);  alphaY = R^-1 Y, alpha1 = R^-1 1.
 * hyper: nhyper doubles per kernel type (see IBO_K_*); sf2 multiplies the
 * kernel (1 for SE kernels, magnitude^2 for SV / Matern in the Python model).
 * On IBO_ERR_NOT_PD *info (optional) receives the 1-based failing pivot.
 * 1 <= D <= 64 (IBO_ERR_ARG otherwise; the reference has no limit, its tests and demos stay below 7).  Up to 32 dimensions
 * the exponent-GEMM kernels take the sweeps; 33 .. 64 run through the difference-form kernels only (and without the kept
 * state of ibo_acq_sweep_incremental: every call is a full sweep).
 */
int ibo_gp_fit(ibo_gp_t *gp, int ktype, int N, int D,
               const double *X_host, const double *Y_host,
               const double *hyper_host, int nhyper, double sf2, double noise,
               int *info);

/*
 * Same, but factor a caller-supplied symmetric matrix A (N x N, host) in place
 * of R: PrefGaussianProcess uses A = R + C^-1
 * (ego/gaussianprocess/__init__.py:487-498, ego/acquisition/__init__.py:385-386).
 * R itself stays available (ibo_gp_get_R) because callers read GP.R.
 */
int ibo_gp_fit_with_matrix(ibo_gp_t *gp, int ktype, int N, int D,
                           const double *X_host, const double *Y_host,
                           const double *hyper_host, int nhyper, double sf2, double noise,
                           const double *A_host, int *info);

/*
 * Append n observations (Xnew_host: n x D) to a fitted model WITHOUT refactoring: the block extension of
 * GaussianProcess.addData (ego/gaussianprocess/__init__.py:301-308: z = solve(L, m), d = chol(r - z^T z)),
 * one point at a time, O(N^2) per point.  Y_all_host holds all N + n targets.  R, L, W and both alpha vectors
 * are updated in place on the device.  Returns IBO_ERR_STATE -- and changes nothing -- when the handle cannot
 * be extended (never fitted, fitted from a caller-supplied matrix or from an inverse, or N + n exceeds the
 * row padding, a multiple of 64): the caller then calls ibo_gp_fit with all the data.  IBO_ERR_NOT_PD as
 * ibo_gp_fit, and any other error, leave the handle UNFITTED (every later call returns IBO_ERR_STATE until a refit).
 */
int ibo_gp_extend(ibo_gp_t *gp, int n, const double *Xnew_host, const double *Y_all_host, int *info);
/* head-room: later fits of this handle pad the matrices to a multiple of 64 that leaves at least `rows` free rows, so
 * that many observations can be appended by ibo_gp_extend before a refit is due (a gallery of n points on a model
 * whose size is a multiple of 64 would otherwise refit, and sweep in full, in its very first round) */
int ibo_gp_reserve(ibo_gp_t *gp, int rows);

/*
 * Preference GP on the device (PrefGaussianProcess.addPreferences, ego/gaussianprocess/__init__.py:347-498: the MAP of
 *     S(y) = -sum_pairs (d+1) log Phi((y_v - y_u)/sqrt 2) + y^T R^-1 y / 2     (:351-385)
 * and then L = chol(R + C^-1), :459-498).  The host keeps what is O(pairs) -- Phi, its derivatives, the line search --
 * and the device everything that is N x N: only vectors and the distinct entries of the pair sums cross the bus
 * (round 1 shipped an N x N Hessian per Newton step through ibo_spd_solve, and C and C^-1 through ibo_spd_inverse).
 *   ibo_pref_begin        after a plain ibo_gp_fit of the points: R^-1 = W^T W is formed on the handle
 *   ibo_pref_rinv_mul     out = R^-1 y
 *   ibo_pref_newton_step  H = R^-1 + sum of the sparse term (lin[e] = row * N + col, distinct entries: the host sums
 *                         the per-pair contributions rho (e_v - e_u)(e_v - e_u)^T); delta = -H^-1 grad; rdelta = R^-1 delta
 *   ibo_pref_finish       C = diag I + sparse term; the handle is refactored from R + C^-1 exactly as
 *                         ibo_gp_fit_with_matrix would (Y as set by ibo_gp_set_y).  IBO_ERR_NOT_PD: call again with a
 *                         larger diag (the reference's regulariser loop, :489-497) or refit.
 */
int ibo_pref_begin(ibo_gp_t *gp);
int ibo_pref_rinv_mul(ibo_gp_t *gp, const double *y_host, double *out_host);
int ibo_pref_newton_step(ibo_gp_t *gp, int nnz, const int64_t *lin_host, const double *val_host,
                         const double *grad_host, double *delta_host, double *rdelta_host, int *info);
int ibo_pref_finish(ibo_gp_t *gp, int nnz, const int64_t *lin_host, const double *val_host, double diag, int *info);

/* replace Y (and the alpha vectors) without refactoring: the preference GP's
 * C-matrix loop re-reads mu with L fixed (ego/gaussianprocess/__init__.py:476) */
int ibo_gp_set_y(ibo_gp_t *gp, const double *Y_host);

/* signal variance used for the CROSS-covariances k(x_i, c) of later sweeps only.
 * libego evaluates k* with sf2 = 1 for kernel types 0-2 whatever the Python
 * kernel's magnitude (cpp/optimizeGP.cpp:303-310) while R was built with it; a
 * drop-in maximize* sets this to reproduce that, then restores it. */
int ibo_gp_set_kstar_sf2(ibo_gp_t *gp, double sf2);

/* RBF-network mean prior m(x) = sum_i beta_i exp(-theta |(x-lowerb)/width - mean_i|^2)
 * (ego/gaussianprocess/prior.py:60-66, cpp/optimizeGP.cpp:116-133); nb = 0 clears it */
int ibo_gp_set_prior(ibo_gp_t *gp, int nb, const double *means_host, const double *beta_host,
                     double theta, const double *lowerb_host, const double *width_host);

/* copy the public attributes back (N x N row-major each).  R = K(X, X) with the reference's diagonal 1 + noise is formed on the
 * first request after a fit (a fit itself only needs its factor; ibo_gp_extend keeps a formed R up to date) */
int ibo_gp_get_R(ibo_gp_t *gp, double *R_host);
int ibo_gp_get_L(ibo_gp_t *gp, double *L_host);
/* W = L^-1 (N x N, lower triangular), and R^-1 = W^T W if wanted by a caller */
int ibo_gp_get_W(ibo_gp_t *gp, double *W_host);
int ibo_gp_info(ibo_gp_t *gp, int *N, int *D, int *device, double *max_y);
/* milliseconds of the last fit, device-side (hipEvent) */
int ibo_gp_last_fit_ms(ibo_gp_t *gp, float *ms);

/* covariance matrix only (no factorisation): Kernel.covMatrix / _computeCorrelations.
 * A2 may be NULL (square K(A1,A1) with the chosen diagonal rule) or a second
 * point set (cross-covariance K(A1,A2), n1 x n2, no diagonal rule). */
int ibo_cov_matrix(int device, int ktype, int D, const double *hyper_host, int nhyper, double sf2,
                   int n1, const double *A1_host, int n2, const double *A2_host,
                   int diag_rule, double noise, double *K_host);

/* X = A^-1 B for a symmetric positive-definite A (N x N) and nrhs right-hand sides (B, X:
 * nrhs x N row-major), all host buffers: blocked Cholesky + explicit L^-1 on the GPU.  The
 * preference GP's MAP (ego/gaussianprocess/__init__.py:442) runs Newton steps through this:
 * the Hessian of its functional is R^-1 plus the preference terms.  IBO_ERR_NOT_PD / *info
 * as ibo_gp_fit. */
int ibo_spd_solve(int device, int N, const double *A_host, int nrhs, const double *B_host,
                  double *X_host, int *info);

/* A^-1 of a symmetric positive-definite A (N x N, host in / host out) = W^T W with W = chol(A)^-1.
 * Replaces linalg.inv(self.C) of the preference GP (ego/gaussianprocess/__init__.py:488,514). */
int ibo_spd_inverse(int device, int N, const double *A_host, double *Ainv_host, int *info);

/* ---------------------------------------------------------------- posterior / sweep */
/*
 * Batched posterior: replaces GaussianProcess.posterior / posteriors / mu
 * (ego/gaussianprocess/__init__.py:169-254).  clamp_lo = 1e-7 reproduces the
 * Python clip(.., 10e-8, 10), 1e-8 the native clamp (cpp/optimizeGP.cpp:150-157).
 * Host buffers in and out (PCIe-inclusive).  s2_host may be NULL.
 */
int ibo_posterior_batch(ibo_gp_t *gp, int64_t M, const double *Q_host, double clamp_lo,
                        double *mu_host, double *s2_host);

/*
 * The same evaluation for points that live on the HOST (Q_host: M x D), results into host arrays (any of mu_host,
 * s2_host, acq_host may be NULL): EI / PI / UCB.negf(x) and their vectorised forms (ego/acquisition/__init__.py:47-166).
 * ymax NaN = max(Y).  Small batches cost no allocation and no copy launch; from 2^18 points on upload, sweep and download
 * are pipelined in chunks.  ibo_posterior_batch is this with acq = IBO_ACQ_NONE.
 */
int ibo_acq_batch(ibo_gp_t *gp, int64_t M, const double *Q_host, int acq, double parm, int erf_mode,
                  double clamp_lo, double ymax, double *mu_host, double *s2_host, double *acq_host);

/*
 * Fused candidate sweep: the batched equivalent of M calls of
 * GP_Maximizer::negei/negpi/negucb (cpp/optimizeGP.cpp:57-236), i.e. what
 * maximizeEI/PI/UCB evaluate inside DIRECT and what fastUCBGallery's
 * latin-hypercube step evaluates (ego/acquisition/gallery.py:111-116).
 *
 *   cand_dev     M x D candidates, DEVICE memory, row-major
 *   acq          IBO_ACQ_*;  parm = xi (EI/PI) or the sigma multiplier (UCB)
 *   ymax         incumbent; pass NAN to use max(Y) as acqmaxGP does (:316-321)
 *   excl_host    n_excl x D points (host) -- candidates with
 *                min_j |c - excl_j|_2 <= excl_radius are left out of the argmax
 *                (the gallery's 0.5-distance rule, gallery.py:102,113); n_excl=0: none
 *   index_base   added to the local row index to form the reported index
 *                (global index of this rank's shard)
 *   mu_dev, s2_dev, acq_dev   optional DEVICE outputs (M doubles each) or NULL
 *   best_val, best_idx        HOST outputs: maximum of the (positive) acquisition
 *                and the FIRST index attaining it (numpy.argmax order, and the
 *                strict '<' of cpp/direct.cpp:124).  best_idx = -1 if every
 *                candidate is excluded.
 * Blocking.  The posterior part costs N^2 + 3ND + 4N flops per candidate.
 */
int ibo_acq_sweep(ibo_gp_t *gp, int64_t M, const double *cand_dev,
                  int acq, double parm, int erf_mode, double clamp_lo, double ymax,
                  int n_excl, const double *excl_host, double excl_radius,
                  int64_t index_base,
                  double *mu_dev, double *s2_dev, double *acq_dev,
                  double *best_val, int64_t *best_idx);

/*
 * ibo_acq_sweep for a caller that sweeps the SAME device candidate array again and again while the model grows by
 * ibo_gp_extend -- fastUCBGallery's rounds (ego/acquisition/gallery.py:92-134: one hallucinated observation per
 * round, the same sample set).  The first call is a full sweep and leaves q = |W k*|^2 (and the two mean terms) per
 * candidate on the handle, 24 bytes each.  A later call with the same array, after at most 8 rows were appended and
 * nothing else changed, folds the new rows of W into q -- (w_new . k*)^2, O(N) per candidate instead of O(N^2) --
 * re-forms the means from the current alpha vectors and evaluates the acquisition as usual.  Anything else (other
 * array or size, a refit, ibo_gp_set_y, another k* variance, batches small enough for the other kernels) is a full sweep.
 * "The same array" means the same ibo_dev_alloc allocation at the same GENERATION (ibo_dev_generation) and offset: an
 * array that was freed and reallocated at the same address, or overwritten through ibo_memcpy_h2d, is a different one,
 * and memory the library did not allocate is swept in full every time.  Contents changed behind the library's back (the
 * caller's own kernels or hipMemcpy) are the one thing it cannot see.
 *
 * When only the arg-max is asked for (mu_dev, s2_dev and acq_dev all NULL) and the acquisition grows with the variance
 * (IBO_ACQ_EI, IBO_ACQ_UCB with parm >= 0), the state is formed in LEVELS of W's rows, split at about N/8, N/4 and N/2 (multiples
 * of 128): level 0, rows [0, N/8) -- 1/64 of the work: W is triangular -- for every candidate, together with the means; each later
 * level only for the 32-candidate tiles whose BOUND -- the acquisition at the variance 1 + noise - q(rows so far), which can only
 * shrink as rows are added -- reaches a value that a complete candidate attains (the top 3 % of the level-0 ranking are completed
 * first to supply it).  The returned (best_val, best_idx) are those of the full sweep: a tile left incomplete cannot hold the
 * maximum.  ibo_set_option("part_levels", 2 | 3 | 4) caps the levels of states formed afterwards (2: one split at N/2, rounds 2-3).
 * Later calls are lazy too: the complete tiles fold in the rows appended since, the best value they reach is the threshold,
 * and only tiles whose bound -- from their stale state, the means widened by nu_max sum |(W y)_i| over the appended rows (nu_max =
 * sf2_k / sqrt(sf2_fit) bounds |W k*|; no lazy mode where the fitted matrix admits no such bound):
 * nothing for observations on the posterior mean (the gallery's), everything for real ones -- reaches it are refreshed and
 * completed.  A call that wants per-candidate outputs (or IBO_ACQ_PI / IBO_ACQ_NONE), or a model with a mean prior, refreshes
 * and completes every tile first.
 * 512 <= padded rows <= 4096; 40 bytes of state per candidate.  ibo_set_option("gallery_prune", 0) restores the one-kernel
 * first sweep, 2 runs the same launches with every tile completed (what the pruned run is tested against, bit for bit).
 */
int ibo_acq_sweep_incremental(ibo_gp_t *gp, int64_t M, const double *cand_dev,
                              int acq, double parm, int erf_mode, double clamp_lo, double ymax,
                              int n_excl, const double *excl_host, double excl_radius,
                              int64_t index_base,
                              double *mu_dev, double *s2_dev, double *acq_dev,
                              double *best_val, int64_t *best_idx);

/* the kept state of ibo_acq_sweep_incremental: its 32-candidate tiles and how many of them carry their full variance
 * (equal unless the state was formed in two parts); both 0 when the handle keeps no state */
int ibo_sweep_state_info(ibo_gp_t *gp, int64_t *tiles, int64_t *complete);
/* the same in detail: the state's number of levels (1: formed by the one-kernel sweep or none), splits[3] = the rows where
 * levels 1, 2, 3 begin (0 beyond nlev - 1), tiles_at_level[4] = how many tiles stand at each level (a tile at level
 * nlev - 1 is complete) */
int ibo_sweep_state_levels(ibo_gp_t *gp, int *nlev, int *splits, int64_t *tiles_at_level);

/* device-side duration (hipEvent, ms) of the dominant kernel of the last
 * ibo_acq_sweep / ibo_posterior_batch on this handle, and its name */
int ibo_last_sweep_kernel_ms(ibo_gp_t *gp, float *ms, const char **kernel_name);

/* ---------------------------------------------------------------- DIRECT on the GPU objective */
/*
 * maximise an acquisition over a box with the reference's DIRECT
 * (cpp/direct.cpp:329-581) -- tree logic on the host, every batch of new
 * sample points evaluated by the sweep kernel.  compat != 0 reproduces the
 * reference's trajectory quirks incl. the dimension-0 stall;
 * compat == 0 applies the fixed-dimension test to dimension 0 as well.
 * opt = maximum of the acquisition, optx[D] its location, nsamples optional.
 */
int ibo_direct_max(ibo_gp_t *gp, int D, const double *lb, const double *ub,
                   int acq, double parm, int erf_mode, double clamp_lo,
                   int maxiter, int maxtime, int maxsample, int compat,
                   double *opt, double *optx, int64_t *nsamples);
/* ibo_set_option("direct_resident", 1): for the lifetime of one ibo_direct_max call a kernel stays resident on the chip (one workgroup per CU)
 * and takes DIRECT's batches from a mailbox in pinned host memory instead of three launches per batch -- the same items on the same operands,
 * the same values bit for bit.  Every wait is bounded on both sides ("direct_idle_ms", 20: the kernel leaves when its mailbox stays silent
 * that long; not all workgroups resident within ~2 ms: it never starts; the host finishes the call by launches whenever the server is not there),
 * so there is no hung-GPU mode.  OFF by default: measured slower than the launches (2.26 against 1.48 ms per maximizeEI at N = 1024;
 * profiles/r06_direct_server_breakdown.txt).  ibo_direct_server_info: batches the server evaluated in the last call on this handle, and why not all. */
int ibo_direct_server_info(ibo_gp_t *gp, int *batches, const char **why);

/* DIRECT minimisation of a HOST callback with the reference's semantics
 * (cpp/direct.cpp:329; what ego.utils.optimize.cdirect wraps), plus the sample
 * counter and the compat switch (bit 0).  Bit 1 of `compat` selects the batched schedule ibo_direct_max runs the GPU
 * objective under -- one evaluation batch per iteration: every potentially-optimal rectangle's probes plus its child
 * centres, guessed before the probe values are known and verified bit for bit afterwards -- with the same
 * (fmin, xmin, nsamples) as the sequential call order.  Host-side only: no GPU is touched. */
int ibo_direct_host(double (*objective)(int, double *), int ndim, const double *lb, const double *ub,
                    int maxiter, int maxtime, int maxsample, int compat,
                    double *fmin, double *xmin, int64_t *nsamples);

/* ---------------------------------------------------------------- marginal likelihood grid */
/*
 * nlml[t] for n_theta hyper-parameter rows (each nhyper doubles):
 * marginalLikelihood(..., computeGradient=False) of
 * ego/gaussianprocess/trainhyper.py:47-75 (K = covMatrix + noise I).
 * A non-positive-definite K yields NAN in that slot (the reference's nlml()
 * wrapper maps the LinAlgError to 100, trainhyper.py:111-114 -- done host side).
 */
int ibo_nlml_grid(int device, int ktype, int N, int D,
                  const double *X_host, const double *Y_host,
                  int n_theta, const double *thetas_host, int nhyper,
                  const double *sf2_host /* n_theta or NULL (=1) */, double noise,
                  double *nlml_host);
/* Device memory is recycled: ibo_nlml_grid and ibo_nlml_grad keep their workspaces (the batch of factor
 * matrices; the N x N buffers of the gradient) between calls, and the buffers of destroyed handles go to a
 * per-device free list (at most 2 GiB; ibo_set_option("pool_limit_mb", n) or env IBO_POOL_LIMIT_MB) for the next handle; buffers of up
 * to half a slab live in the arena (see "arena_mb").  This releases all of it EXCEPT the library's standing reservation on the device: the
 * arena's first slab and four stream / event / staging sets, which the next model would otherwise pay milliseconds to make again. */
int ibo_trim(int device);

/*
 * NLML and its gradient w.r.t. each LOG hyper-parameter for one theta: marginalLikelihood(...,
 * computeGradient=True), ego/gaussianprocess/trainhyper.py:47-75, with dK/dtheta_h as
 * Kernel.derivative(X, h) builds it (ego/gaussianprocess/kernel.py).  modes[h]: 0 SE-ARD length
 * scale of dimension dims[h]; 1 SE-iso length scale; 2 signal magnitude (2K); 3 Matern-3/2 and
 * 4 Matern-5/2 length scale.  grad_host receives ngrad values (1 <= ngrad <= 65; 17 per pass of the gradient kernel).
 * A learning loop calls this dozens of times with one data set and another theta: X and Y stay on the device between calls and go up again
 * only when their CONTENT differs from the last call's (compared on the host; ibo_trim forgets them).
 */
int ibo_nlml_grad(int device, int ktype, int N, int D, const double *X_host, const double *Y_host,
                  const double *hyper_host, int nhyper, double sf2, double noise,
                  int ngrad, const int *modes, const int *dims, double *nlml_host, double *grad_host);

/* ---------------------------------------------------------------- multi-GPU arg-max exchange (RCCL) */
#define IBO_COMM_ID_BYTES 128
int ibo_comm_get_unique_id(unsigned char id[IBO_COMM_ID_BYTES]);
int ibo_comm_init(int device, int world_size, int rank,
                  const unsigned char id[IBO_COMM_ID_BYTES], ibo_comm_t **out);
int ibo_comm_destroy(ibo_comm_t *comm);
/* ranks in the communicator as RCCL reports them (ncclCommCount) */
int ibo_comm_count(ibo_comm_t *comm, int *nranks);
/*
 * One all-reduce(sum) over a world_size x (3+npayload) slot buffer (value,
 * index, valid flag, payload) in which each rank fills only its own slot (RCCL has no MAXLOC), followed by the same
 * deterministic local reduction on every rank: maximum value, ties to the
 * lowest global index.  payload (npayload doubles, e.g. the winner's
 * coordinates) travels in the same buffer.  All outputs are identical on
 * every rank.
 */
int ibo_comm_argmax(ibo_comm_t *comm, double val, int64_t idx,
                    const double *payload, int npayload,
                    double *best_val, int64_t *best_idx, double *best_payload, int *best_rank);
/*
 * The sharded sweep's step in ONE call: ibo_acq_sweep (incremental != 0: ibo_acq_sweep_incremental) over this rank's block of the
 * candidate array (rows index_base ..), then the exchange above with the winner's D coordinates as payload -- the sweep's (value,
 * index) never visit the host on the way: a kernel writes them and the coordinates into the rank's slot of the all-reduce buffer,
 * ncclAllReduce runs on the sweep's stream, one copy into pinned memory brings back every rank's slot, one synchronisation.
 * local_val / local_idx: this rank's own maximum (index -1 and -inf without an admissible candidate); best_* as ibo_comm_argmax,
 * best_x: D doubles.  Replaces, per round, the candidate loop of ego/acquisition/gallery.py:93-134 cut over the ranks.
 */
int ibo_acq_sweep_exchange(ibo_gp_t *gp, ibo_comm_t *comm, int incremental, int64_t M, const double *cand_dev,
                           int acq, double parm, int erf_mode, double clamp_lo, double ymax,
                           int n_excl, const double *excl_host, double excl_radius, int64_t index_base,
                           double *local_val, int64_t *local_idx,
                           double *best_val, int64_t *best_idx, double *best_x, int *best_rank);
/* in-place ncclAllReduce(sum) of a host buffer: gathers the sharded NLML grid (each rank fills
 * its own theta slots of a zero buffer) */
int ibo_comm_allreduce_sum(ibo_comm_t *comm, double *host_buf, int64_t n);
int ibo_comm_barrier(ibo_comm_t *comm);

/* ---------------------------------------------------------------- (A) legacy libego symbols */
typedef double (*objective_t)(int, double *);

/* cpp/optimizeGP.cpp:262-283.  Returns malloc'd [fmin, xmin[0..ndim)] with
 * fmin = minimum of the NEGATED acquisition; caller frees with free(). NULL on
 * unknown acqfunc (as the reference) or on any failure (message on stderr).
 * The objective is evaluated in the reference's own operation order (cpp/optimizeGP.cpp:57-236: k*, prior mean and the
 * acquisition with the host's libm; aMb's two sequential sums per contraction on the device, products and sums rounded
 * separately), so fmin and xmin equal libego's BIT FOR BIT, whatever the conditioning of invR (csrc/legacy.hip).
 * Differences kept on purpose: kerneltype 3 takes its magnitude from hyperparams[1] (the reference reads hyperparams[ndim],
 * out of bounds for ndim > 1) and prints nothing; a kerneltype outside 0..3 (the reference's switch leaves k* uninitialised) returns NULL.
 * The host half (k*, prior mean, acquisition: O(N D) per sample point) runs on a crew of host threads over the batch's points
 * (IBO_HOST_THREADS, default min(16, the cores the process may use)).  ibo_set_option("legacy_exact", 0): the fast route (invR factored on the
 * device, MFMA sweep kernels; within 1e-6 of libego on well-conditioned data only).  Re-entrant: own handle per call. */
const double *acqmaxGP(int ndim, double *lb, double *ub, double *invR, double *X, double *Y,
                       int nx, int acqfunc, int kerneltype, double *hyperparams,
                       int npbases, double *pbasismeans, double *pbasisbeta, double pbasistheta,
                       double *pbasislowerb, double *pbasiswidth, double parm, double noise,
                       int maxiter, int maxtime, int maxsample);

/* cpp/direct.cpp:329: DIRECT minimisation of a host callback (host-side only;
 * kept so ego.utils.optimize.cdirect keeps working against this library). */
const double *direct(objective_t objective, int ndim, double *lb, double *ub,
                     int maxiter, int maxtime, int maxsample);

/* cpp/helpers.cpp:30-56: sum of log(Phi((x[p[i]] - x[p[i+1]]) / sqrt 2) / sqrt 2) over i = 0, 2, 4, ... < n
 * (terms whose argument of log is exactly 0 are skipped).  Host-side only.  The reference's only caller
 * (ego/gaussianprocess/__init__.py:362-371, off by default) passes flattened (v, u, degree) triples with
 * n = 3 P and so also reads one int past the end for odd P; this entry point keeps the pair-stride loop
 * but never reads p[n]. */
double logCDFs(int nprefinds, int *prefinds, double *x);

#ifdef __cplusplus
}
#endif
#endif /* IBO_ABI_H */
