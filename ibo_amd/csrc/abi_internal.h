// abi_internal.h -- what the translation units behind the C ABI (include/ibo_abi.h) share: the error channel, the option switches, the
// per-device memory pool and its buffer type, the handle (struct ibo_gp) and the helpers one unit lends another.
//   abi_core.hip    library / options / device memory / pools / handle life cycle
//   abi_fit.hip     fit, block extension, preference GP, accessors, ibo_cov_matrix, ibo_spd_*
//   abi_sweep.hip   candidate sweeps, host batches, DIRECT on the GPU objective
//   abi_nlml.hip    marginal-likelihood grid and gradient, ibo_trim
//   abi_legacy.hip  libego's symbols (acqmaxGP, direct, logCDFs) and ibo_direct_host
// There is no CPU fallback anywhere behind this header: without a gfx950 device every compute entry point returns IBO_ERR_NO_DEVICE.
#pragma once
#include "../../include/ibo_abi.h"
#include "ibo_common.h"
#include "direct_host.h"
#include "legacy.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <atomic>
#include <mutex>
#include <vector>

// the thread's last error message (ibo_last_error); returns `code`
int ibo_fail(int code, const char *fmt, ...);
#define fail ibo_fail

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(IBO_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define KERNEL_TRY(expr)                                                                      \
    do {                                                                                      \
        int e_ = (expr);                                                                      \
        if (e_ != 0)                                                                          \
            return fail(IBO_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString((hipError_t)e_), __FILE__, __LINE__); \
    } while (0)
#define IBO_TRY(expr) do { int s_ = (expr); if (s_ != IBO_OK) return s_; } while (0)

// ---- option switches (abi_core.hip: ibo_set_option)
extern std::atomic<int> g_super_min_nb;        // ibo_set_option("super_min_nb", nb): block columns from which a fit runs in super-panels
extern std::atomic<int> g_direct_resident, g_direct_idle_ms;      // ibo_set_option("direct_resident", 0/1), ("direct_idle_ms", n)
extern std::atomic<int> g_host_pipeline, g_fused2_min_nb, g_gallery_prune, g_nlml_batch, g_chol_left, g_dot_override, g_legacy_exact, g_force_path, g_nlml_groups;
extern std::mutex g_dev_mu[16];             // serialises the per-device workspaces of ibo_nlml_grid / ibo_nlml_grad / ibo_trim
extern std::atomic<size_t> g_pool_limit;

int use_device(int device);
void gpu_time_add(int device, double ms);    // accumulates what ibo_gpu_time_ms reports

// ---- recycled device memory (abi_core.hip)
void *pool_get(size_t bytes, size_t *got);
void pool_put(void *p, size_t bytes);
void pool_trim(int dev);
void pool_warm(int dev);
extern thread_local bool g_pool_quiet;       // the caller has synchronised the device already (ibo_gp_destroy: once for all its buffers)

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    size_t bytes = 0;              // what the pool handed out (a multiple of its grain, not of sizeof(T)): what goes back to it
    int ensure(size_t n)
    {
        if (n <= cap) return IBO_OK;
        release();
        size_t got = 0;
        void *q = pool_get(n * sizeof(T), &got);
        if (!q) {
            got = n * sizeof(T);
            hipError_t e = hipMalloc(&q, got);
            if (e != hipSuccess) {
                int dev = 0;
                (void)hipGetDevice(&dev);
                pool_trim(dev);                       // give the cached blocks back and try once more
                e = hipMalloc(&q, got);
            }
            if (e != hipSuccess) return fail(IBO_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", got, hipGetErrorString(e));
        }
        p = (T *)q;
        cap = got / sizeof(T);
        bytes = got;
        return IBO_OK;
    }
    void release() { if (p) pool_put(p, bytes); p = nullptr; cap = 0; bytes = 0; }
};
// function-local buffers: handed back on every exit path (the members of handles and of the static
// workspaces are released explicitly -- a static object must not call into HIP at process exit)
template <typename T>
struct ScopedBuf : DevBuf<T> {
    ~ScopedBuf() { this->release(); }
};

struct ibo_gp {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t h2d_stream = nullptr, d2h_stream = nullptr;      // host-array batches: copies overlap the sweep
    hipEvent_t pe_in[2] = {nullptr, nullptr}, pe_k[2] = {nullptr, nullptr}, pe_out[2] = {nullptr, nullptr};
    hipEvent_t ev0 = nullptr, ev1 = nullptr, fit0 = nullptr, fit1 = nullptr;
    bool fitted = false;
    int N = 0, D = 0, Npad = 0, DP = 0;
    bool reversed = false;          // legacy invR path stores the observations in reverse order
    bool plain_fit = false;         // L = chol(R) of the model's own kernel matrix: ibo_gp_extend may append rows
    bool L_upper_dirty = false;     // zero_upper is deferred to ibo_gp_get_L
    bool R_valid = false;           // R = K(X, X) + diag is formed when someone asks for it (ibo_gp_get_R, ibo_pref_finish): ensure_R
    int dot_form = 1;               // SE k* via a_k + b_c + x~.c~; off when |x~|^2 is so large that the
                                    // cancellation would cost more than 1e-10 (pathological length scales)
    KParams kp;
    KParams kp_fit;                 // kp as fitted (kp.sf2 is overridden per sweep by ibo_gp_set_kstar_sf2)
    double noise = 0.0, maxY = 0.0;
    float fit_ms = 0.f, sweep_ms = 0.f;
    const char *sweep_kernel = "";
    std::vector<double> Yhost;
    double *pin = nullptr; size_t pin_cap = 0;      // pinned host staging for small host-in/host-out batches
    DevBuf<double> Xp, Xs, ak, XA, Y, R, A, L, W, T, Wp, diag64, alphaY, alpha1, tmp, cand, outs, excl, qpart, mupart, partv, res_v;
    DevBuf<int64_t> parti, res_i;
    // kept sweep state (ibo_acq_sweep_incremental): (q, aY.k*, a1.k*) per candidate of ONE device candidate array
    DevBuf<double> state;
    DevBuf<int> tile_done; DevBuf<double> tile_ub; DevBuf<unsigned long long> part_words; DevBuf<int> tile_rows, tile_sel;   // kept state with incomplete tiles (st_pruned)
    bool st_pruned = false; int st_N0 = 0; int st_nlev = 2;   // st_N0: the model's rows when the state was formed; st_nlev: its levels of W's rows
    DevBuf<double> small_ws;        // small2.hip: k* in fragment order + partial sums of a small batch
    uint64_t st_gen = 0; size_t st_off = 0; int64_t st_M = 0; int st_N = 0; double st_sf2 = 0.0; unsigned st_epoch = 0;   // st_gen: generation of the candidate array's allocation (0: no state)
    unsigned fit_epoch = 0;         // bumped by every full fit: a kept state never survives one
    int reserve = 0;                // rows of head-room the next fit leaves for ibo_gp_extend (ibo_gp_reserve)
    DevBuf<int> info;
    unsigned long long *done_flag = nullptr;   // pinned host word small2.hip's last kernel writes; done_seq: last value asked for
    unsigned long long done_seq = 0;
    bool signal_pending = false;
    const double *alpha_tail_Y = nullptr, *alpha_tail_1 = nullptr; int alpha_tail_Np = 0;   // where the alpha vectors' zero tails are
    DevBuf<unsigned> done_count;
    // fits from g_super_min_nb block columns on (launch_cholesky_super): [A ; E] in one tall buffer, the packed store of [L ; E^T-in-progress]
    DevBuf<double> tall, Pk2;
    DevBuf<unsigned> srv_ctl;       // the resident evaluation server's control words (small2.hip: ServerCtl)
    int srv_batches = 0; const char *srv_why = "";      // the last ibo_direct_max on this handle: batches the server evaluated; why it did not (all of them)
    // preference GP (ibo_pref_*): R^-1, the matrix being factored and its factors, vectors, sparse terms
    struct PrefWork {
        DevBuf<double> Rinv, A, Lh, E, Et, d64, vec, tmp, val;
        DevBuf<long long> lin;
        DevBuf<int> info;
        bool ready = false;
        int epoch = -1;
    } pw;
    // prior
    int nb = 0; double ptheta = 0.0;
    DevBuf<double> pmeans, pbeta, plowerb, pwidth;
};

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
// Which order factors an Np-row matrix: the single-level right-looking order with pipelined block columns and W = L^-1 riding along
// (launch_cholesky_fused) below g_fused2_min_nb block columns, the two-level order (panels of four, K = 256 updates, recursive-doubling
// inversion) from there on: 104 block columns (6656 rows) by default -- with the eight-wave pipelined column and two steps per pass the
// single-level order wins up to there (N = 4096: 2.49 -> 2.07 ms; 6400 rows: 6.20 against 6.50; 7040: 8.18 against 7.70).  ONE predicate for
// ibo_gp_fit, the preference GP's factorisations and ibo_nlml_grad: the order fixes the last bits of L and W.
static inline bool single_level_order(int Np) { return Np / 64 < g_fused2_min_nb; }

static inline bool super_order(int Np) { return single_level_order(Np) && Np / 64 >= g_super_min_nb; }

// ---- helpers one unit lends another
int exp_table(int device, const double **out);                                                            // abi_sweep.hip: 2^(j/2048), one per device
int ibo_comm_exchange_dev(ibo_comm_t *c, hipStream_t s, const double *res_v, const int64_t *res_i, const double *cand_dev, int D, int64_t index_base,
                          double *local_val, int64_t *local_idx, double *best_val, int64_t *best_idx, double *best_x, int *best_rank);   // comm.hip
uint64_t alloc_generation(int device, const void *p, size_t bytes, size_t *offset);                       // abi_core.hip
int ensure_pinned(ibo_gp *g, size_t need);                                                                // abi_core.hip
int make_kparams(int ktype, int D, const double *hyper, int nhyper, double sf2, KParams *kp);             // abi_fit.hip
int fit_from_inverse(ibo_gp *g, int ktype, int N, int D, const double *X, const double *Y,
                     const double *hyper, int nhyper, double sf2, double noise, const double *invR);      // abi_fit.hip
int direct_on_gp(ibo_gp *g, int D, const double *lb, const double *ub, int acq, double parm, int erf_mode,
                 double clamp_lo, int maxiter, int maxtime, int maxsample, int compat,
                 double *opt, double *optx, int64_t *nsamples);                                           // abi_sweep.hip
