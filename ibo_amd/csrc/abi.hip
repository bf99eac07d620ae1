// abi.hip -- the C ABI of libibo_hip.so (include/ibo_abi.h): handle management,
// fit orchestration, sweep / posterior entry points, DIRECT on the GPU
// objective, the marginal-likelihood grid and the legacy libego symbols.
// There is no CPU fallback anywhere in this file: without a gfx950 device every
// compute entry point returns IBO_ERR_NO_DEVICE.
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

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// the other translation units (comm.hip) report through the same buffer, so ibo_last_error() always
// describes the call that failed
void ibo_internal_set_error(const char *msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
}

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

static std::atomic<int> g_host_pipeline{1};  // ibo_set_option("host_pipeline", 0/1): chunked, overlapped host batches (0: one shot -- the test's comparator)
static std::atomic<int> g_fused2_min_nb{104};  // ibo_set_option("fused2_min_nb"): block columns from which a single matrix takes the two-level order
static std::atomic<int> g_gallery_prune{1};  // ibo_set_option("gallery_prune", 0/1/2): kept-state sweeps in levels of W's rows, the later ones only where a
                                 // tile's bound can still win (1); the same launches with every tile completed (2); the one-kernel sweep (0)
static std::atomic<int> g_nlml_batch{0};     // ibo_set_option("nlml_batch", b): matrices per batched factorisation of ibo_nlml_grid (0: as many as 12 GB hold)
static std::atomic<int> g_chol_left{1};      // ibo_set_option("chol_left", 0/1): ibo_nlml_grid factors in the left-looking outer order (update3.hip); 0: the
                                 // right-looking order of launch_cholesky_batched -- the same bits, the test's comparator
static std::atomic<int> g_dot_override{-1};  // ibo_set_option("dot_form", -1/0/1): -1 auto, 0/1 force the difference / dot form of k* (tests)
static std::atomic<int> g_legacy_exact{1};   // ibo_set_option("legacy_exact", 0/1): acqmaxGP evaluates libego's formulas in libego's operation order (legacy.hip)
static std::atomic<int> g_force_path{0};     // ibo_set_option("sweep_path"): 0 auto, 1 gemv, 2 mfma, 3 panel-split (IBO_SWEEP_IMPL env / tests)
static std::atomic<int> g_nlml_groups{2};    // IBO_NLML_GROUPS=1..4 (env): a batch of theta-points runs as that many sub-batches, each on its own stream(s); values do not depend on it
// The option switches above are process-wide configuration (atomics: setting one while another thread computes is a defined,
// if unspecified-moment, change); the per-device workspaces of ibo_nlml_grid / ibo_nlml_grad and ibo_trim are serialised by
// g_dev_mu (the exp table has its own lock, held only while it is created); handles are independent of each other (own stream,
// events, buffers) -- two threads may drive two handles on one device at once.  ONE handle is for one thread at a time.
static std::mutex g_dev_mu[16];
static std::atomic<size_t> g_pool_limit{(size_t)2 << 30};    // ibo_set_option("pool_limit_mb", n) / env IBO_POOL_LIMIT_MB

static int use_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(IBO_ERR_NO_DEVICE, "no HIP device visible (%s); libibo_hip has no CPU fallback",
                    e == hipSuccess ? "count=0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(IBO_ERR_ARG, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    static std::once_flag env_read;
    std::call_once(env_read, [] {
        const char *s = getenv("IBO_SWEEP_IMPL");
        if (s && !strcmp(s, "gemv")) g_force_path = 1;
        if (s && !strcmp(s, "mfma")) g_force_path = 2;
        if (const char *a = getenv("IBO_NLML_GROUPS")) { const int v = atoi(a); if (v >= 1 && v <= 4) g_nlml_groups = v; }
        const char *pl = getenv("IBO_POOL_LIMIT_MB");
        if (pl && atoll(pl) >= 0) g_pool_limit = (size_t)atoll(pl) << 20;
    });
    return IBO_OK;
}

// Device allocations are recycled: a Bayesian-optimisation loop builds a new model (a new handle, five N x N
// buffers) every round, and hipMalloc/hipFree of tens of megabytes cost more than the fit itself.  Freed
// blocks go to a per-device free list (up to g_pool_limit bytes, 2 GiB unless configured; ibo_trim() empties it) and are handed out
// again to requests of up to half their size less.
struct PoolBlock { void *p; size_t bytes; };
static std::vector<PoolBlock> g_pool[16];
static size_t g_pool_bytes[16];
static std::mutex g_pool_mu;

static void *pool_get(size_t bytes, size_t *got)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_pool_mu);
    std::vector<PoolBlock> &v = g_pool[dev & 15];
    size_t best = v.size();
    for (size_t i = 0; i < v.size(); i++)
        if (v[i].bytes >= bytes && v[i].bytes <= 2 * bytes + 4096 && (best == v.size() || v[i].bytes < v[best].bytes)) best = i;
    if (best == v.size()) return nullptr;
    void *p = v[best].p;
    *got = v[best].bytes;
    g_pool_bytes[dev & 15] -= v[best].bytes;
    v.erase(v.begin() + best);
    return p;
}

static thread_local bool g_pool_quiet = false;       // the caller has synchronised the device already (ibo_gp_destroy: once for all its buffers)
static void pool_put(void *p, size_t bytes)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!g_pool_quiet) (void)hipDeviceSynchronize(); // what hipFree would have waited for: nothing in flight uses p
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_pool_bytes[dev & 15] + bytes > g_pool_limit) { (void)hipFree(p); return; }
    g_pool[dev & 15].push_back({p, bytes});
    g_pool_bytes[dev & 15] += bytes;
}

// A handle's stream, events and pinned staging are recycled the same way: creating them costs 1.5-2 ms and destroying
// them 1.2 ms -- several times the 0.4 ms fit of the model the handle is created for.
struct ExecSet {
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, fit0 = nullptr, fit1 = nullptr;
    double *pin = nullptr; size_t pin_cap = 0;
    unsigned long long *done_flag = nullptr;
};
static std::vector<ExecSet> g_exec_pool[16];
static void exec_set_free(ExecSet &x)
{
    if (x.ev0) (void)hipEventDestroy(x.ev0);
    if (x.ev1) (void)hipEventDestroy(x.ev1);
    if (x.fit0) (void)hipEventDestroy(x.fit0);
    if (x.fit1) (void)hipEventDestroy(x.fit1);
    if (x.pin) (void)hipHostFree(x.pin);
    if (x.done_flag) (void)hipHostFree(x.done_flag);
    if (x.stream) (void)hipStreamDestroy(x.stream);
    x = ExecSet();
}
static bool exec_set_get(int dev, ExecSet *x)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    std::vector<ExecSet> &v = g_exec_pool[dev & 15];
    if (v.empty()) return false;
    *x = v.back();
    v.pop_back();
    return true;
}
static void exec_set_put(int dev, ExecSet x)
{
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        std::vector<ExecSet> &v = g_exec_pool[dev & 15];
        if (v.size() < 8 && x.pin_cap * sizeof(double) <= ((size_t)64 << 20)) { v.push_back(x); return; }
    }
    exec_set_free(x);
}

static void pool_trim(int dev)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (PoolBlock &b : g_pool[dev & 15]) (void)hipFree(b.p);
    g_pool[dev & 15].clear();
    g_pool_bytes[dev & 15] = 0;
    for (ExecSet &x : g_exec_pool[dev & 15]) exec_set_free(x);
    g_exec_pool[dev & 15].clear();
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
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
        return IBO_OK;
    }
    void release() { if (p) pool_put(p, cap * sizeof(T)); p = nullptr; cap = 0; }
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

// ------------------------------------------------------------------------ library
extern "C" int ibo_abi_version(void) { return IBO_ABI_VERSION; }
extern "C" const char *ibo_last_error(void) { return g_err; }

extern "C" int ibo_device_count(int *count)
{
    if (!count) return fail(IBO_ERR_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return IBO_OK;
}

extern "C" int ibo_device_name(int device, char *buf, size_t buflen)
{
    if (!buf || !buflen) return fail(IBO_ERR_ARG, "buf is NULL");
    IBO_TRY(use_device(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s %s cu=%d clk=%dMHz", prop.name, prop.gcnArchName, prop.multiProcessorCount,
             prop.clockRate / 1000);
    return IBO_OK;
}

extern "C" int ibo_set_option(const char *key, int value)
{
    if (!key) return fail(IBO_ERR_ARG, "key is NULL");
    if (!strcmp(key, "sweep_path")) { g_force_path = value; return IBO_OK; }
    if (!strcmp(key, "dot_form")) { g_dot_override = value; return IBO_OK; }
    if (!strcmp(key, "gallery_prune")) { g_gallery_prune = value; return IBO_OK; }
    if (!strcmp(key, "part_levels")) { set_part_levels(value); return IBO_OK; }
    if (!strcmp(key, "legacy_exact")) { g_legacy_exact = value; return IBO_OK; }
    if (!strcmp(key, "host_pipeline")) { g_host_pipeline = value; return IBO_OK; }
    if (!strcmp(key, "nlml_batch")) { g_nlml_batch = value; return IBO_OK; }
    if (!strcmp(key, "chol_left")) { g_chol_left = value; return IBO_OK; }
    if (!strcmp(key, "fused2_min_nb")) { if (value < 1) return fail(IBO_ERR_ARG, "fused2_min_nb < 1"); g_fused2_min_nb = value; return IBO_OK; }
    if (!strcmp(key, "pool_limit_mb")) { if (value < 0) return fail(IBO_ERR_ARG, "pool_limit_mb < 0"); g_pool_limit = (size_t)value << 20; return IBO_OK; }
    return fail(IBO_ERR_ARG, "unknown option");
}

extern "C" int ibo_selftest_mfma(int device, double *max_abs_err)
{
    IBO_TRY(use_device(device));
    double *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, sizeof(double)));
    KERNEL_TRY(launch_mfma_selftest(d, nullptr));
    double h = -1.0;
    HIP_TRY(hipMemcpy(&h, d, sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    if (max_abs_err) *max_abs_err = h;
    if (h != 0.0) return fail(IBO_ERR_HIP, "fp64 MFMA fragment layout self-test failed: max |err| = %g", h);
    return IBO_OK;
}

// ------------------------------------------------------------------------ device memory
// Every allocation handed out by ibo_dev_alloc carries a GENERATION: a process-wide counter value taken when it is
// allocated and again whenever ibo_memcpy_h2d writes into it.  hipFree / hipMalloc routinely hand the same address to
// the next array of the same size, so state that is kept "per candidate array" (ibo_acq_sweep_incremental) is keyed
// on the generation, never on the raw pointer: a freed-and-reallocated or overwritten array can not be mistaken for
// the one the state was formed from.  Memory the library did not allocate has no generation (0) and never qualifies.
struct DevAlloc { char *base; size_t bytes; int device; uint64_t gen; };
static std::vector<DevAlloc> g_allocs;
static uint64_t g_gen_counter = 0;
static std::mutex g_alloc_mu;

// generation of the allocation that contains [p, p + bytes) on `device`, 0 if none; *offset = p - base
static uint64_t alloc_generation(int device, const void *p, size_t bytes, size_t *offset)
{
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    const char *c = (const char *)p;
    for (const DevAlloc &a : g_allocs)
        if (a.device == device && c >= a.base && c + bytes <= a.base + a.bytes) {
            if (offset) *offset = (size_t)(c - a.base);
            return a.gen;
        }
    return 0;
}

extern "C" int ibo_dev_alloc(int device, size_t bytes, void **dev_ptr)
{
    if (!dev_ptr) return fail(IBO_ERR_ARG, "dev_ptr is NULL");
    IBO_TRY(use_device(device));
    HIP_TRY(hipMalloc(dev_ptr, bytes ? bytes : 8));
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    g_allocs.push_back({(char *)*dev_ptr, bytes ? bytes : 8, device, ++g_gen_counter});
    return IBO_OK;
}
extern "C" int ibo_dev_free(int device, void *dev_ptr)
{
    IBO_TRY(use_device(device));
    if (dev_ptr) {
        {
            std::lock_guard<std::mutex> lk(g_alloc_mu);
            for (size_t i = 0; i < g_allocs.size(); i++)
                if (g_allocs[i].base == (char *)dev_ptr && g_allocs[i].device == device) { g_allocs.erase(g_allocs.begin() + i); break; }
        }
        HIP_TRY(hipFree(dev_ptr));
    }
    return IBO_OK;
}
extern "C" int ibo_memcpy_h2d(int device, void *dev_dst, const void *host_src, size_t bytes)
{
    IBO_TRY(use_device(device));
    {
        std::lock_guard<std::mutex> lk(g_alloc_mu);       // new contents: a new generation for the allocation written into
        const char *c = (const char *)dev_dst;
        for (DevAlloc &a : g_allocs)
            if (a.device == device && c < a.base + a.bytes && c + bytes > a.base) a.gen = ++g_gen_counter;
    }
    HIP_TRY(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
    return IBO_OK;
}
extern "C" int ibo_dev_generation(int device, const void *dev_ptr, uint64_t *generation)
{
    if (!generation) return fail(IBO_ERR_ARG, "generation is NULL");
    *generation = alloc_generation(device, dev_ptr, 1, nullptr);
    return IBO_OK;
}
extern "C" int ibo_memcpy_d2h(int device, void *host_dst, const void *dev_src, size_t bytes)
{
    IBO_TRY(use_device(device));
    HIP_TRY(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
    return IBO_OK;
}
extern "C" int ibo_device_synchronize(int device)
{
    IBO_TRY(use_device(device));
    HIP_TRY(hipDeviceSynchronize());
    return IBO_OK;
}

// ------------------------------------------------------------------------ model
extern "C" int ibo_gp_create(int device, ibo_gp_t **out)
{
    if (!out) return fail(IBO_ERR_ARG, "out is NULL");
    IBO_TRY(use_device(device));
    ibo_gp *g = new ibo_gp();
    g->device = device;
    memset(&g->kp, 0, sizeof(g->kp));
    ExecSet x;
    if (exec_set_get(device, &x)) {
        g->stream = x.stream; g->ev0 = x.ev0; g->ev1 = x.ev1; g->fit0 = x.fit0; g->fit1 = x.fit1;
        g->pin = x.pin; g->pin_cap = x.pin_cap; g->done_flag = x.done_flag;
        if (g->done_flag) *g->done_flag = 0;
        *out = g;
        return IBO_OK;
    }
    hipError_t e = hipStreamCreate(&g->stream);
    if (e == hipSuccess) e = hipEventCreate(&g->ev0);
    if (e == hipSuccess) e = hipEventCreate(&g->ev1);
    if (e == hipSuccess) e = hipEventCreate(&g->fit0);
    if (e == hipSuccess) e = hipEventCreate(&g->fit1);
    if (e != hipSuccess) {                            // hand back whatever was created
        if (g->ev0) (void)hipEventDestroy(g->ev0);
        if (g->ev1) (void)hipEventDestroy(g->ev1);
        if (g->fit0) (void)hipEventDestroy(g->fit0);
        if (g->fit1) (void)hipEventDestroy(g->fit1);
        if (g->stream) (void)hipStreamDestroy(g->stream);
        delete g;
        return fail(IBO_ERR_HIP, "creating the handle's stream/events failed: %s", hipGetErrorString(e));
    }
    *out = g;
    return IBO_OK;
}

extern "C" int ibo_gp_destroy(ibo_gp_t *g)
{
    if (!g) return IBO_OK;
    (void)hipSetDevice(g->device);
    (void)hipDeviceSynchronize();                    // once, for every buffer handed back below
    g_pool_quiet = true;
    g->Xp.release(); g->Xs.release(); g->ak.release(); g->XA.release(); g->Y.release(); g->R.release(); g->A.release(); g->L.release(); g->W.release();
    g->T.release(); g->Wp.release(); g->diag64.release(); g->alphaY.release(); g->alpha1.release();
    g->tmp.release(); g->cand.release(); g->outs.release(); g->excl.release(); g->qpart.release();
    g->mupart.release(); g->partv.release(); g->res_v.release(); g->parti.release(); g->res_i.release(); g->state.release(); g->small_ws.release();
    g->tile_done.release(); g->tile_ub.release(); g->part_words.release(); g->tile_rows.release(); g->tile_sel.release();
    g->done_count.release();
    g->pw.Rinv.release(); g->pw.A.release(); g->pw.Lh.release(); g->pw.E.release(); g->pw.Et.release(); g->pw.d64.release();
    g->pw.vec.release(); g->pw.tmp.release(); g->pw.val.release(); g->pw.lin.release(); g->pw.info.release();
    g->info.release(); g->pmeans.release(); g->pbeta.release(); g->plowerb.release(); g->pwidth.release();
    g_pool_quiet = false;
    if (g->h2d_stream) {
        for (int b = 0; b < 2; b++) { (void)hipEventDestroy(g->pe_in[b]); (void)hipEventDestroy(g->pe_k[b]); (void)hipEventDestroy(g->pe_out[b]); }
        (void)hipStreamDestroy(g->h2d_stream); (void)hipStreamDestroy(g->d2h_stream);
    }
    ExecSet x;
    x.stream = g->stream; x.ev0 = g->ev0; x.ev1 = g->ev1; x.fit0 = g->fit0; x.fit1 = g->fit1;
    x.pin = g->pin; x.pin_cap = g->pin_cap; x.done_flag = g->done_flag;
    exec_set_put(g->device, x);
    delete g;
    return IBO_OK;
}

static int make_kparams(int ktype, int D, const double *hyper, int nhyper, double sf2, KParams *kp)
{
    if (D < 1 || D > IBO_DMAX) return fail(IBO_ERR_ARG, "D=%d unsupported (1..%d)", D, IBO_DMAX);
    if (!hyper) return fail(IBO_ERR_ARG, "hyper is NULL");
    memset(kp, 0, sizeof(*kp));
    kp->D = D; kp->sf2 = sf2;
    switch (ktype) {
    case IBO_K_SE_ARD:
        if (nhyper < D) return fail(IBO_ERR_ARG, "SE-ARD needs %d length scales, got %d", D, nhyper);
        kp->family = FAM_SE;
        for (int d = 0; d < D; d++) { kp->w[d] = 1.0 / (hyper[d] * hyper[d]); kp->sw[d] = 1.0 / fabs(hyper[d]); }
        break;
    case IBO_K_SE_ISO:
    case IBO_K_MATERN3:
    case IBO_K_MATERN5:
        if (nhyper < 1) return fail(IBO_ERR_ARG, "kernel needs a length scale");
        kp->family = ktype == IBO_K_SE_ISO ? FAM_SE : (ktype == IBO_K_MATERN3 ? FAM_M3 : FAM_M5);
        for (int d = 0; d < D; d++) { kp->w[d] = 1.0 / (hyper[0] * hyper[0]); kp->sw[d] = 1.0 / fabs(hyper[0]); }
        break;
    default:
        return fail(IBO_ERR_ARG, "unknown kernel type %d", ktype);
    }
    return IBO_OK;
}

// |x~|^2 bounds the absolute error of y = a_k + b_c + x~.c~ by ~|x~|^2 * 2^-52
static int dot_form_ok(const KParams &kp, const double *X, int N, int D)
{
    if (D > IBO_DDOT) return 0;                      // 33 .. 64 dimensions: difference-form kernels only
    double mx = 0.0;
    for (int i = 0; i < N; i++) {
        double n2 = 0.0;
        for (int d = 0; d < D; d++) { double v = X[(size_t)i * D + d] * kp.sw[d]; n2 += v * v; }
        if (n2 > mx) mx = n2;
    }
    const char *e = getenv("IBO_DOT_FORM");
    if (e) return atoi(e);
    return mx <= 1e5;
}

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
// Which order factors an Np-row matrix: the single-level right-looking order with pipelined block columns and W = L^-1 riding along
// (launch_cholesky_fused) below g_fused2_min_nb block columns, the two-level order (panels of four, K = 256 updates, recursive-doubling
// inversion) from there on: 104 block columns (6656 rows) by default -- with the eight-wave pipelined column and two steps per pass the
// single-level order wins up to there (N = 4096: 2.49 -> 2.07 ms; 6400 rows: 6.20 against 6.50; 7040: 8.18 against 7.70).  ONE predicate for
// ibo_gp_fit, the preference GP's factorisations and ibo_nlml_grad: the order fixes the last bits of L and W.
static inline bool single_level_order(int Np) { return Np / 64 < g_fused2_min_nb; }
static int ensure_pinned(ibo_gp *g, size_t need);

// stage observations (optionally in reverse order) and size every buffer
static int stage_data(ibo_gp *g, int N, int D, const double *X, const double *Y, bool reverse)
{
    if (N < 1) return fail(IBO_ERR_ARG, "N=%d", N);
    if (!X || !Y) return fail(IBO_ERR_ARG, "X/Y is NULL");
    g->N = N; g->D = D; g->Npad = round_up(N + (reverse ? 0 : g->reserve), 64); g->DP = D <= 4 ? 4 : (D <= 8 ? 8 : (D <= 16 ? 16 : (D <= 32 ? 32 : 64)));
    g->reversed = reverse;
    g->R_valid = false;             // new points: R is formed again when someone asks (ensure_R)
    const int Np = g->Npad, DP = g->DP;
    size_t nn = (size_t)Np * Np;
    IBO_TRY(g->Xp.ensure((size_t)Np * DP)); IBO_TRY(g->Xs.ensure((size_t)Np * DP)); IBO_TRY(g->ak.ensure(Np));
    IBO_TRY(g->XA.ensure((size_t)((Np + 127) / 128 * 8) * ((D + 5) / 4) * 64));
    IBO_TRY(g->Y.ensure(Np));
    IBO_TRY(g->L.ensure(nn)); IBO_TRY(g->W.ensure(nn));
    IBO_TRY(g->T.ensure(nn)); IBO_TRY(g->Wp.ensure(nn)); IBO_TRY(g->diag64.ensure((size_t)(Np / 64) * 4096));
    // sweep2's stages cover rows up to the next multiple of 128: the tail of both alpha vectors stays zero
    IBO_TRY(g->alphaY.ensure((size_t)Np + 128)); IBO_TRY(g->alpha1.ensure((size_t)Np + 128));
    if (g->alpha_tail_Y != g->alphaY.p || g->alpha_tail_1 != g->alpha1.p || g->alpha_tail_Np != Np) {     // (nothing writes there)
        HIP_TRY(hipMemsetAsync(g->alphaY.p + Np, 0, 128 * sizeof(double), g->stream));
        HIP_TRY(hipMemsetAsync(g->alpha1.p + Np, 0, 128 * sizeof(double), g->stream));
        g->alpha_tail_Y = g->alphaY.p; g->alpha_tail_1 = g->alpha1.p; g->alpha_tail_Np = Np;
    }
    IBO_TRY(g->tmp.ensure(3 * (size_t)Np + 2 * (size_t)(Np / 64) * Np + 64));     // launch_alpha's scratch + one vector (ibo_gp_extend)
    IBO_TRY(g->info.ensure(1));
    // staged through the handle's pinned buffer: the copies are truly asynchronous and nothing has to be waited for before the fit's
    // kernels are queued (a pageable source is staged by the runtime and had to be kept alive by a stream synchronise: ~25 us of a 0.37 ms fit)
    IBO_TRY(ensure_pinned(g, (size_t)Np * DP + Np));
    double *xp = g->pin, *yp = g->pin + (size_t)Np * DP;
    memset(g->pin, 0, sizeof(double) * ((size_t)Np * DP + Np));
    g->Yhost.assign(N, 0.0);
    double my = Y[0];
    for (int i = 0; i < N; i++) {
        int s = reverse ? N - 1 - i : i;
        for (int d = 0; d < D; d++) xp[(size_t)i * DP + d] = X[(size_t)s * D + d];
        yp[i] = Y[s];
        g->Yhost[i] = Y[s];
        if (Y[i] > my) my = Y[i];      // acqmaxGP's maxY scan, cpp/optimizeGP.cpp:316-321
    }
    g->maxY = my;
    HIP_TRY(hipMemcpyAsync(g->Xp.p, xp, sizeof(double) * (size_t)Np * DP, hipMemcpyHostToDevice, g->stream));
    HIP_TRY(hipMemcpyAsync(g->Y.p, yp, sizeof(double) * Np, hipMemcpyHostToDevice, g->stream));
    return IBO_OK;                                  // (every caller ends with a stream synchronise before the pinned buffer is used again)
}

static int check_info(ibo_gp *g, int *info)
{
    int h = 0;
    HIP_TRY(hipMemcpyAsync(&h, g->info.p, sizeof(int), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    if (info) *info = h;
    if (h != 0) {
        g->fitted = false;
        return fail(IBO_ERR_NOT_PD, "matrix is not positive definite (pivot %d)", h);
    }
    return IBO_OK;
}

// R = K(X, X) with the reference's hard-wired diagonal 1 + noise (ego/gaussianprocess/__init__.py:138), over the rows the
// model holds now, by the kernel and in the order of operations the fit's own covariance pass uses: what a fit, or a fit
// and its extensions, would have written had they kept R up to date.
static int ensure_R(ibo_gp *g)
{
    if (g->R_valid) return IBO_OK;
    IBO_TRY(g->R.ensure((size_t)g->Npad * g->Npad));             // N x N with row stride Npad (room to extend)
    KERNEL_TRY(launch_cov_matrix(g->kp_fit, g->N, g->Xp.p, 0, nullptr, g->DP, IBO_DIAG_UNIT_PLUS_NOISE, g->noise, g->R.p, g->Npad, g->stream));
    g->R_valid = true;
    return IBO_OK;
}

// Everything of a fit after the data are staged: R, L = chol(R) -- or chol(A) for a matrix already in g->A (N x N) --,
// W = L^-1 and its packed copy, both alpha vectors.
static int fit_factor(ibo_gp *g, const KParams &kp, int N, double noise, bool have_A, int *info)
{
    const int Np = g->Npad;
    hipStream_t s = g->stream;
    const double *A_host = have_A ? g->A.p : nullptr;      // (only tested for presence below)
    HIP_TRY(hipEventRecord(g->fit0, s));
    // R, and in the same pass the identity-padded copy the factorisation works on
    const bool fused = single_level_order(Np);                   // (else the two-level order; both out of place: the matrix in T, the factor into L)
    double *work = g->T.p;                                       // T is free until launch_trinv uses it as scratch
    // (with the working copy the same pass writes the identity the ride-along starts from and clears the info word)
    const bool one_pass = fused && !A_host;
    // (GP.R itself is not written here: 33 MB of stores at N = 2048 that only ibo_gp_get_R and ibo_pref_finish read -- ensure_R;
    // stage_data marked it stale)
    if (!A_host)
        KERNEL_TRY(launch_cov_fit(kp, N, g->Xp.p, g->DP, IBO_DIAG_UNIT_PLUS_NOISE, noise, work, Np, one_pass ? g->W.p : nullptr, g->info.p, s));
    else {
        HIP_TRY(hipMemsetAsync(g->info.p, 0, sizeof(int), s));
        KERNEL_TRY(launch_pad_copy(g->A.p, N, N, work, Np, 1.0, s));
    }
    if (fused) {
        // the plain right-looking order: one launch per block column, out of place, with W = L^-1 riding along (E = I in W's buffer
        // turns into (L^-1)^T in Wp's, which is transposed into W and packed into T's buffer -- free by then -- in one pass; T and
        // Wp then trade places)
        if (!one_pass) KERNEL_TRY(launch_pad_copy(g->Xp.p, 0, 1, g->W.p, Np, 1.0, s));       // identity
        KERNEL_TRY(launch_cholesky_fused(g->T.p, g->L.p, Np, g->diag64.p, g->info.p, s, g->W.p, g->Wp.p, true));
        KERNEL_TRY(launch_transpose_pack(g->Wp.p, N, Np, g->W.p, g->T.p, s));
        std::swap(g->T, g->Wp);
    } else {
        KERNEL_TRY(launch_cholesky_fused2(g->T.p, g->L.p, Np, g->diag64.p, g->info.p, 4, s, true, g->W.p));    // W: free until launch_trinv
        KERNEL_TRY(launch_trinv(g->L.p, Np, g->diag64.p, g->W.p, g->T.p, s, false));
        KERNEL_TRY(launch_pack_w(g->W.p, N, Np, 0, g->W.p, g->Wp.p, s));
    }
    g->L_upper_dirty = true;        // the strict upper blocks of L are scratch until someone asks for L
    KERNEL_TRY(launch_alpha(g->W.p, N, Np, g->Y.p, g->tmp.p, g->alphaY.p, g->alpha1.p, s));
    HIP_TRY(hipEventRecord(g->fit1, s));
    IBO_TRY(check_info(g, info));
    HIP_TRY(hipEventElapsedTime(&g->fit_ms, g->fit0, g->fit1));
    g->fitted = true;
    g->plain_fit = !have_A;
    g->fit_epoch++;
    return IBO_OK;
}

static int fit_impl(ibo_gp *g, int ktype, int N, int D, const double *X, const double *Y,
                    const double *hyper, int nhyper, double sf2, double noise, const double *A_host, int *info)
{
    if (!g) return fail(IBO_ERR_ARG, "gp is NULL");
    IBO_TRY(use_device(g->device));
    KParams kp;
    IBO_TRY(make_kparams(ktype, D, hyper, nhyper, sf2, &kp));
    g->fitted = false;
    IBO_TRY(stage_data(g, N, D, X, Y, false));
    g->kp = kp; g->kp_fit = kp; g->noise = noise;
    const int Np = g->Npad;
    hipStream_t s = g->stream;
    KERNEL_TRY(launch_scale_x(kp, g->Xp.p, Np, g->DP, g->Xs.p, g->ak.p, s));
    KERNEL_TRY(launch_pack_xa(g->Xs.p, g->ak.p, N, Np, g->DP, D, g->XA.p, s));
    g->dot_form = dot_form_ok(kp, X, N, D);
    if (A_host) {
        IBO_TRY(g->A.ensure((size_t)N * N));
        HIP_TRY(hipMemcpyAsync(g->A.p, A_host, sizeof(double) * (size_t)N * N, hipMemcpyHostToDevice, s));
    }
    return fit_factor(g, kp, N, noise, A_host != nullptr, info);
}

// Append observations to a fitted model without refactoring: the block extension of
// GaussianProcess.addData (ego/gaussianprocess/__init__.py:301-308), z = L^-1 m, d = chol(r - z^T z), one
// point at a time.  With W = L^-1 already on the device, z = W k and the new row of W is -(W^T z)/d: two
// triangular matrix-vector products (the same kernels that form alpha), O(N^2) instead of the O(N^3) refit.
extern "C" int ibo_gp_extend(ibo_gp_t *g, int n, const double *Xnew, const double *Yall, int *info)
{
    if (!g || !Xnew || !Yall || n < 1) return fail(IBO_ERR_ARG, "bad argument");
    if (!g->fitted || !g->plain_fit || g->reversed) return fail(IBO_ERR_STATE, "model cannot be extended in place");
    if (g->N + n > g->Npad) return fail(IBO_ERR_STATE, "no room in the current padding (%d + %d > %d)", g->N, n, g->Npad);
    IBO_TRY(use_device(g->device));
    hipStream_t s = g->stream;
    const int Np = g->Npad, DP = g->DP, D = g->D, N0 = g->N;
    if (info) *info = 0;
    // stage the new rows of X (padded to DP) behind the old ones; sizes do not change
    std::vector<double> xp((size_t)n * DP, 0.0);
    for (int i = 0; i < n; i++)
        for (int d = 0; d < D; d++) xp[(size_t)i * DP + d] = Xnew[(size_t)i * D + d];
    HIP_TRY(hipMemcpyAsync(g->Xp.p + (size_t)N0 * DP, xp.data(), xp.size() * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(g->info.p, 0, sizeof(int), s));
    HIP_TRY(hipEventRecord(g->fit0, s));
    // from here on the handle's rows are being rewritten: any early return (a HIP or launch error) must leave it marked
    // unfitted -- the caller then refits -- rather than "fitted" with rows N0.. of L / W / Wp half-written
    g->fitted = false;
    for (int i = 0; i < n; i++) {
        const int N = N0 + i;                       // rows present before this point
        // k = K(X, x_new) (also the new row / column of R), z = W k and u = W^T z, then the new rows of L and W
        KERNEL_TRY(launch_extend_kvec(g->kp_fit, g->Xp.p, DP, N, Np, g->noise, g->R_valid ? g->R.p : nullptr, g->tmp.p + 2 * (size_t)Np + 2 * (size_t)(Np / 64) * Np, s));
        double *kvec = g->tmp.p + 2 * (size_t)Np + 2 * (size_t)(Np / 64) * Np;
        KERNEL_TRY(launch_alpha(g->W.p, N, Np, kvec, g->tmp.p, g->T.p, g->T.p + Np, s));      // t2[0..Np) = z, T[0..Np) = W^T z
        KERNEL_TRY(launch_extend_rows(N, Np, g->noise, g->tmp.p, g->T.p, g->L.p, g->W.p, g->Wp.p, g->info.p, s));
    }
    const int N1 = N0 + n;
    std::vector<double> yp(Np, 0.0);
    double my = Yall[0];
    for (int i = 0; i < N1; i++) { yp[i] = Yall[i]; if (Yall[i] > my) my = Yall[i]; }
    HIP_TRY(hipMemcpyAsync(g->Y.p, yp.data(), yp.size() * sizeof(double), hipMemcpyHostToDevice, s));
    KERNEL_TRY(launch_scale_x(g->kp_fit, g->Xp.p, Np, DP, g->Xs.p, g->ak.p, s));
    KERNEL_TRY(launch_pack_xa(g->Xs.p, g->ak.p, N1, Np, DP, D, g->XA.p, s));
    KERNEL_TRY(launch_alpha(g->W.p, N1, Np, g->Y.p, g->tmp.p, g->alphaY.p, g->alpha1.p, s));
    HIP_TRY(hipEventRecord(g->fit1, s));
    int h = 0;
    HIP_TRY(hipMemcpyAsync(&h, g->info.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));               // also: xp / yp go out of scope
    if (h != 0) {
        // the rows written so far belong to a matrix that is not positive definite: the handle needs a refit
        g->fitted = false;
        if (info) *info = h;
        return fail(IBO_ERR_NOT_PD, "matrix is not positive definite (pivot %d)", h);
    }
    HIP_TRY(hipEventElapsedTime(&g->fit_ms, g->fit0, g->fit1));
    if (g->dot_form) {                              // |x~|^2 of the new points still admits the dot form?
        for (int i = 0; i < n && g->dot_form; i++) {
            double n2 = 0.0;
            for (int d = 0; d < D; d++) { const double v = Xnew[(size_t)i * D + d] * g->kp_fit.sw[d]; n2 += v * v; }
            if (n2 > 1e5) g->dot_form = 0;
        }
    }
    // the kept sweep state's stale tiles carry means formed with the OLD alpha vectors, and the lazy refresh's drift margin only
    // covers the appended rows' (W y)_i: a caller that changed an earlier target along the way (GaussianProcess.Y is a public
    // attribute) gets a full sweep next time, as after ibo_gp_set_y
    for (int i = 0; i < N0; i++)
        if (!(Yall[i] == g->Yhost[i])) { g->st_gen = 0; break; }
    g->N = N1; g->maxY = my;
    g->Yhost.assign(Yall, Yall + N1);
    g->L_upper_dirty = true;
    g->fitted = true;
    return IBO_OK;
}

extern "C" int ibo_gp_reserve(ibo_gp_t *g, int rows)
{
    if (!g || rows < 0) return fail(IBO_ERR_ARG, "bad argument");
    g->reserve = rows;
    return IBO_OK;
}

extern "C" int ibo_gp_fit(ibo_gp_t *g, int ktype, int N, int D, const double *X, const double *Y,
                          const double *hyper, int nhyper, double sf2, double noise, int *info)
{
    return fit_impl(g, ktype, N, D, X, Y, hyper, nhyper, sf2, noise, nullptr, info);
}

extern "C" int ibo_gp_fit_with_matrix(ibo_gp_t *g, int ktype, int N, int D, const double *X, const double *Y,
                                      const double *hyper, int nhyper, double sf2, double noise,
                                      const double *A_host, int *info)
{
    if (!A_host) return fail(IBO_ERR_ARG, "A_host is NULL");
    return fit_impl(g, ktype, N, D, X, Y, hyper, nhyper, sf2, noise, A_host, info);
}

// legacy entry: the caller hands over invR (ego/acquisition/__init__.py:385-388).
// invR = G G^T; q = |G^T k*|^2; reversing the index order makes G^T lower
// triangular so the same sweep kernel applies (see pack_w_kernel, mode 1).
static int fit_from_inverse(ibo_gp *g, int ktype, int N, int D, const double *X, const double *Y,
                            const double *hyper, int nhyper, double sf2, double noise, const double *invR)
{
    IBO_TRY(use_device(g->device));
    KParams kp;
    IBO_TRY(make_kparams(ktype, D, hyper, nhyper, sf2, &kp));
    g->fitted = false;
    IBO_TRY(stage_data(g, N, D, X, Y, true));
    g->kp = kp; g->noise = noise;
    const int Np = g->Npad;
    hipStream_t s = g->stream;
    KERNEL_TRY(launch_scale_x(kp, g->Xp.p, Np, g->DP, g->Xs.p, g->ak.p, s));
    KERNEL_TRY(launch_pack_xa(g->Xs.p, g->ak.p, N, Np, g->DP, D, g->XA.p, s));
    g->dot_form = dot_form_ok(kp, X, N, D);
    IBO_TRY(g->A.ensure((size_t)N * N));
    HIP_TRY(hipMemcpyAsync(g->A.p, invR, sizeof(double) * (size_t)N * N, hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(g->fit0, s));
    KERNEL_TRY(launch_pad_copy(g->A.p, N, N, g->L.p, Np, 1.0, s));
    KERNEL_TRY(launch_cholesky(g->L.p, Np, g->diag64.p, g->info.p, s));
    KERNEL_TRY(launch_pack_w(g->L.p, N, Np, 1, g->W.p, g->Wp.p, s));
    KERNEL_TRY(launch_alpha(g->W.p, N, Np, g->Y.p, g->tmp.p, g->alphaY.p, g->alpha1.p, s));
    HIP_TRY(hipEventRecord(g->fit1, s));
    IBO_TRY(check_info(g, nullptr));
    HIP_TRY(hipEventElapsedTime(&g->fit_ms, g->fit0, g->fit1));
    g->fitted = true;
    g->plain_fit = false;
    g->fit_epoch++;
    return IBO_OK;
}

extern "C" int ibo_gp_set_y(ibo_gp_t *g, const double *Y_host)
{
    if (!g || !Y_host) return fail(IBO_ERR_ARG, "NULL argument");
    if (!g->fitted) return fail(IBO_ERR_STATE, "set_y before fit");
    IBO_TRY(use_device(g->device));
    std::vector<double> yp(g->Npad, 0.0);
    double my = Y_host[0];
    for (int i = 0; i < g->N; i++) {
        yp[i] = Y_host[g->reversed ? g->N - 1 - i : i];
        g->Yhost[i] = yp[i];
        if (Y_host[i] > my) my = Y_host[i];
    }
    g->maxY = my;
    g->st_gen = 0;                                  // the kept per-candidate means were formed with the old alpha vectors
    HIP_TRY(hipMemcpyAsync(g->Y.p, yp.data(), yp.size() * sizeof(double), hipMemcpyHostToDevice, g->stream));
    KERNEL_TRY(launch_alpha(g->W.p, g->N, g->Npad, g->Y.p, g->tmp.p, g->alphaY.p, g->alpha1.p, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return IBO_OK;
}

// ------------------------------------------------------------------------ preference GP on the device
// PrefGaussianProcess.addPreferences (ego/gaussianprocess/__init__.py:347-498) minimises
//     S(y) = -sum_pairs (d + 1) log Phi((y_v - y_u)/sqrt 2) + y^T R^-1 y / 2
// and then factors R + C^-1.  The O(pairs) terms (Phi, its derivatives, the line search) stay with the host; every
// N x N object -- R^-1 = W^T W, the Hessian R^-1 + sum rho (e_v - e_u)(e_v - e_u)^T and its factorisation, C, C^-1,
// R + C^-1 -- lives on the device, and only vectors and the pairs' distinct matrix entries cross the bus.
static int pref_alloc(ibo_gp *g)
{
    const int Np = g->Npad;
    const size_t nn = (size_t)Np * Np;
    auto &pw = g->pw;
    IBO_TRY(pw.Rinv.ensure(nn)); IBO_TRY(pw.A.ensure(nn)); IBO_TRY(pw.Lh.ensure(nn)); IBO_TRY(pw.E.ensure(nn));
    IBO_TRY(pw.Et.ensure(nn)); IBO_TRY(pw.d64.ensure((size_t)(Np / 64) * 4096)); IBO_TRY(pw.vec.ensure(4 * (size_t)Np));
    IBO_TRY(pw.tmp.ensure(2 * (size_t)Np + 2 * (size_t)(Np / 64) * Np + 64)); IBO_TRY(pw.info.ensure(1));
    return IBO_OK;
}
// pw.A (N x N in an identity-padded Npad x Npad frame; destroyed) -> pw.E = the inverse of its Cholesky factor, pad rows zero
static int pref_factor(ibo_gp *g, int *info)
{
    auto &pw = g->pw;
    const int N = g->N, Np = g->Npad;
    hipStream_t s = g->stream;
    if (single_level_order(Np)) {
        KERNEL_TRY(launch_pad_copy(g->Xp.p, 0, 1, pw.E.p, Np, 1.0, s));                  // identity
        KERNEL_TRY(launch_cholesky_fused(pw.A.p, pw.Lh.p, Np, pw.d64.p, pw.info.p, s, pw.E.p, pw.Et.p));
        KERNEL_TRY(launch_transpose_lower(pw.Et.p, pw.E.p, Np, s));
    } else {
        KERNEL_TRY(launch_cholesky(pw.A.p, Np, pw.d64.p, pw.info.p, s, pw.Lh.p));
        KERNEL_TRY(launch_trinv(pw.A.p, Np, pw.d64.p, pw.E.p, pw.Et.p, s, false));
    }
    KERNEL_TRY(launch_pack_w(pw.E.p, N, Np, 0, pw.E.p, pw.Et.p, s));                     // zero the pad rows (Et: scratch)
    int h = 0;
    HIP_TRY(hipMemcpyAsync(&h, pw.info.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (info) *info = h;
    if (h != 0) return fail(IBO_ERR_NOT_PD, "matrix is not positive definite (pivot %d)", h);
    return IBO_OK;
}
static int pref_sparse(ibo_gp *g, int nnz, const int64_t *lin_host, const double *val_host)
{
    auto &pw = g->pw;
    if (nnz < 0 || (nnz > 0 && (!lin_host || !val_host))) return fail(IBO_ERR_ARG, "bad sparse term");
    for (int e = 0; e < nnz; e++)
        if (lin_host[e] < 0 || lin_host[e] >= (int64_t)g->N * g->N) return fail(IBO_ERR_ARG, "matrix entry %d out of range", e);
    if (nnz == 0) return IBO_OK;
    IBO_TRY(pw.lin.ensure(nnz)); IBO_TRY(pw.val.ensure(nnz));
    HIP_TRY(hipMemcpyAsync(pw.lin.p, lin_host, sizeof(int64_t) * nnz, hipMemcpyHostToDevice, g->stream));
    HIP_TRY(hipMemcpyAsync(pw.val.p, val_host, sizeof(double) * nnz, hipMemcpyHostToDevice, g->stream));
    return IBO_OK;
}

extern "C" int ibo_pref_begin(ibo_gp_t *g)
{
    if (!g) return fail(IBO_ERR_ARG, "gp is NULL");
    if (!g->fitted || !g->plain_fit || g->reversed) return fail(IBO_ERR_STATE, "ibo_pref_begin needs a plain fitted model (L = chol(R))");
    IBO_TRY(use_device(g->device));
    IBO_TRY(pref_alloc(g));
    KERNEL_TRY(launch_wtw(g->W.p, g->pw.Et.p, g->pw.Rinv.p, g->Npad, g->stream));       // R^-1 = W^T W (zero on the pad)
    g->pw.ready = true; g->pw.epoch = g->fit_epoch;
    return IBO_OK;
}

static int pref_check(ibo_gp *g)
{
    if (!g) return fail(IBO_ERR_ARG, "gp is NULL");
    if (!g->pw.ready || g->pw.epoch != g->fit_epoch || !g->fitted || !g->plain_fit)
        return fail(IBO_ERR_STATE, "no ibo_pref_begin since the last plain fit of this model");
    return use_device(g->device);
}

extern "C" int ibo_pref_rinv_mul(ibo_gp_t *g, const double *y_host, double *out_host)
{
    IBO_TRY(pref_check(g));
    if (!y_host || !out_host) return fail(IBO_ERR_ARG, "NULL argument");
    auto &pw = g->pw;
    const int N = g->N, Np = g->Npad;
    hipStream_t s = g->stream;
    std::vector<double> yp(Np, 0.0);
    for (int i = 0; i < N; i++) yp[i] = y_host[i];
    HIP_TRY(hipMemcpyAsync(pw.vec.p, yp.data(), sizeof(double) * Np, hipMemcpyHostToDevice, s));
    KERNEL_TRY(launch_alpha(g->W.p, N, Np, pw.vec.p, pw.tmp.p, pw.vec.p + Np, pw.vec.p + 2 * (size_t)Np, s));
    HIP_TRY(hipMemcpyAsync(out_host, pw.vec.p + Np, sizeof(double) * N, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return IBO_OK;
}

extern "C" int ibo_pref_newton_step(ibo_gp_t *g, int nnz, const int64_t *lin_host, const double *val_host,
                                    const double *grad_host, double *delta_host, double *rdelta_host, int *info)
{
    IBO_TRY(pref_check(g));
    if (!grad_host || !delta_host || !rdelta_host) return fail(IBO_ERR_ARG, "NULL argument");
    auto &pw = g->pw;
    const int N = g->N, Np = g->Npad;
    hipStream_t s = g->stream;
    IBO_TRY(pref_sparse(g, nnz, lin_host, val_host));
    std::vector<double> bp(Np, 0.0);
    for (int i = 0; i < N; i++) bp[i] = -grad_host[i];
    HIP_TRY(hipMemcpyAsync(pw.vec.p, bp.data(), sizeof(double) * Np, hipMemcpyHostToDevice, s));
    KERNEL_TRY(launch_pref_build(pw.Rinv.p, N, Np, 0.0, nnz, pw.lin.p, pw.val.p, pw.A.p, s));
    IBO_TRY(pref_factor(g, info));                      // synchronises: bp may go
    double *delta = pw.vec.p + Np, *rdelta = pw.vec.p + 2 * (size_t)Np, *junk = pw.vec.p + 3 * (size_t)Np;
    KERNEL_TRY(launch_alpha(pw.E.p, N, Np, pw.vec.p, pw.tmp.p, delta, junk, s));        // delta = H^-1 (-g)
    KERNEL_TRY(launch_alpha(g->W.p, N, Np, delta, pw.tmp.p, rdelta, junk, s));           // R^-1 delta, for the line search
    HIP_TRY(hipMemcpyAsync(delta_host, delta, sizeof(double) * N, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(rdelta_host, rdelta, sizeof(double) * N, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return IBO_OK;
}

// C = diag I + the pairs' entries; the handle's factor becomes chol(R + C^-1) (W, alpha vectors with it), as
// ibo_gp_fit_with_matrix(R + C^-1) would leave it.  IBO_ERR_NOT_PD (from C or from the sum): nothing usable is left
// but the data; the caller adds to `diag` and calls again, or refits.
extern "C" int ibo_pref_finish(ibo_gp_t *g, int nnz, const int64_t *lin_host, const double *val_host, double diag, int *info)
{
    if (!g) return fail(IBO_ERR_ARG, "gp is NULL");
    if (!g->pw.ready || g->reversed || g->N < 1) return fail(IBO_ERR_STATE, "no ibo_pref_begin on this model");
    IBO_TRY(use_device(g->device));
    auto &pw = g->pw;
    const int N = g->N, Np = g->Npad;
    hipStream_t s = g->stream;
    IBO_TRY(pref_sparse(g, nnz, lin_host, val_host));
    KERNEL_TRY(launch_pref_build(nullptr, N, Np, diag, nnz, pw.lin.p, pw.val.p, pw.A.p, s));
    g->fitted = false;                                   // from here on the old factor is not to be trusted
    IBO_TRY(pref_factor(g, info));
    KERNEL_TRY(launch_wtw(pw.E.p, pw.Et.p, pw.A.p, Np, s));                              // C^-1
    IBO_TRY(g->A.ensure((size_t)N * N));
    IBO_TRY(ensure_R(g));
    KERNEL_TRY(launch_pref_sum(g->R.p, pw.A.p, N, Np, g->A.p, s));
    return fit_factor(g, g->kp_fit, N, g->noise, true, info);
}

extern "C" int ibo_gp_set_kstar_sf2(ibo_gp_t *g, double sf2)
{
    if (!g) return fail(IBO_ERR_ARG, "gp is NULL");
    g->kp.sf2 = sf2;
    return IBO_OK;
}

extern "C" int ibo_gp_set_prior(ibo_gp_t *g, int nb, const double *means, const double *beta, double theta,
                                const double *lowerb, const double *width)
{
    if (!g) return fail(IBO_ERR_ARG, "gp is NULL");
    if (nb <= 0) { g->nb = 0; return IBO_OK; }
    if (g->D <= 0) return fail(IBO_ERR_STATE, "set_prior before fit (dimension unknown)");
    if (!means || !beta || !lowerb || !width) return fail(IBO_ERR_ARG, "NULL prior array");
    IBO_TRY(use_device(g->device));
    const int D = g->D;
    IBO_TRY(g->pmeans.ensure((size_t)nb * D)); IBO_TRY(g->pbeta.ensure(nb));
    IBO_TRY(g->plowerb.ensure(D)); IBO_TRY(g->pwidth.ensure(D));
    HIP_TRY(hipMemcpy(g->pmeans.p, means, sizeof(double) * nb * D, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(g->pbeta.p, beta, sizeof(double) * nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(g->plowerb.p, lowerb, sizeof(double) * D, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(g->pwidth.p, width, sizeof(double) * D, hipMemcpyHostToDevice));
    g->nb = nb; g->ptheta = theta;
    return IBO_OK;
}

static int copy_square(ibo_gp *g, const double *src, int ld, double *dst_host)
{
    IBO_TRY(use_device(g->device));
    HIP_TRY(hipMemcpy2D(dst_host, sizeof(double) * g->N, src, sizeof(double) * ld, sizeof(double) * g->N, g->N,
                        hipMemcpyDeviceToHost));
    return IBO_OK;
}

extern "C" int ibo_gp_get_R(ibo_gp_t *g, double *R_host)
{
    if (!g || !R_host) return fail(IBO_ERR_ARG, "NULL argument");
    if (!g->fitted || g->reversed) return fail(IBO_ERR_STATE, "R not available");
    IBO_TRY(use_device(g->device));
    IBO_TRY(ensure_R(g));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return copy_square(g, g->R.p, g->Npad, R_host);
}
extern "C" int ibo_gp_get_L(ibo_gp_t *g, double *L_host)
{
    if (!g || !L_host) return fail(IBO_ERR_ARG, "NULL argument");
    if (!g->fitted || g->reversed) return fail(IBO_ERR_STATE, "L not available");
    if (g->L_upper_dirty) {
        IBO_TRY(use_device(g->device));
        KERNEL_TRY(launch_zero_upper(g->L.p, g->Npad, g->stream));
        HIP_TRY(hipStreamSynchronize(g->stream));
        g->L_upper_dirty = false;
    }
    return copy_square(g, g->L.p, g->Npad, L_host);
}
extern "C" int ibo_gp_get_W(ibo_gp_t *g, double *W_host)
{
    if (!g || !W_host) return fail(IBO_ERR_ARG, "NULL argument");
    if (!g->fitted || g->reversed) return fail(IBO_ERR_STATE, "W not available");
    return copy_square(g, g->W.p, g->Npad, W_host);
}
extern "C" int ibo_gp_info(ibo_gp_t *g, int *N, int *D, int *device, double *max_y)
{
    if (!g) return fail(IBO_ERR_ARG, "gp is NULL");
    if (N) *N = g->N;
    if (D) *D = g->D;
    if (device) *device = g->device;
    if (max_y) *max_y = g->maxY;
    return IBO_OK;
}
extern "C" int ibo_gp_last_fit_ms(ibo_gp_t *g, float *ms)
{
    if (!g || !ms) return fail(IBO_ERR_ARG, "NULL argument");
    *ms = g->fit_ms;
    return IBO_OK;
}

extern "C" int ibo_cov_matrix(int device, int ktype, int D, const double *hyper, int nhyper, double sf2,
                              int n1, const double *A1, int n2, const double *A2, int diag_rule, double noise,
                              double *K_host)
{
    if (!A1 || !K_host || n1 < 1) return fail(IBO_ERR_ARG, "bad argument");
    IBO_TRY(use_device(device));
    KParams kp;
    IBO_TRY(make_kparams(ktype, D, hyper, nhyper, sf2, &kp));
    int m2 = A2 ? n2 : n1;
    ScopedBuf<double> a1, a2, k;
    IBO_TRY(a1.ensure((size_t)n1 * D)); IBO_TRY(k.ensure((size_t)n1 * m2));
    HIP_TRY(hipMemcpy(a1.p, A1, sizeof(double) * n1 * D, hipMemcpyHostToDevice));
    if (A2) {
        IBO_TRY(a2.ensure((size_t)n2 * D));
        HIP_TRY(hipMemcpy(a2.p, A2, sizeof(double) * n2 * D, hipMemcpyHostToDevice));
    }
    KERNEL_TRY(launch_cov_matrix(kp, n1, a1.p, n2, A2 ? a2.p : nullptr, D, diag_rule, noise, k.p, m2, nullptr, 0));
    HIP_TRY(hipMemcpy(K_host, k.p, sizeof(double) * (size_t)n1 * m2, hipMemcpyDeviceToHost));
    return IBO_OK;
}

// Solve A X = B for a symmetric positive-definite A (N x N, host) and nrhs right-hand sides
// (B, X: nrhs x N row-major, host) on the GPU: blocked Cholesky, explicit L^-1, X = L^-T (L^-1 B).
// Used by the preference GP's Newton iterations (the Hessian of the MAP functional).
extern "C" int ibo_spd_solve(int device, int N, const double *A_host, int nrhs, const double *B_host,
                             double *X_host, int *info)
{
    if (!A_host || !B_host || !X_host || N < 1 || nrhs < 1) return fail(IBO_ERR_ARG, "bad argument");
    IBO_TRY(use_device(device));
    const int Np = round_up(N, 64);
    const size_t nn = (size_t)Np * Np;
    ScopedBuf<double> dA, dL, dW, dT, d64, db, dx, d1, tmp;
    ScopedBuf<int> dinfo;
    IBO_TRY(dA.ensure((size_t)N * N)); IBO_TRY(dL.ensure(nn)); IBO_TRY(dW.ensure(nn)); IBO_TRY(dT.ensure(nn));
    IBO_TRY(d64.ensure((size_t)(Np / 64) * 4096)); IBO_TRY(db.ensure(Np)); IBO_TRY(dx.ensure(Np)); IBO_TRY(d1.ensure(Np));
    IBO_TRY(tmp.ensure(2 * (size_t)Np + 2 * (size_t)(Np / 64) * Np + 64)); IBO_TRY(dinfo.ensure(1));
    hipStream_t s = nullptr;
    HIP_TRY(hipMemcpy(dA.p, A_host, sizeof(double) * (size_t)N * N, hipMemcpyHostToDevice));
    KERNEL_TRY(launch_pad_copy(dA.p, N, N, dL.p, Np, 1.0, s));
    KERNEL_TRY(launch_cholesky(dL.p, Np, d64.p, dinfo.p, s));
    int h = 0;
    HIP_TRY(hipMemcpy(&h, dinfo.p, sizeof(int), hipMemcpyDeviceToHost));
    if (info) *info = h;
    int rc = IBO_OK;
    if (h != 0) rc = fail(IBO_ERR_NOT_PD, "matrix is not positive definite (pivot %d)", h);
    else {
        KERNEL_TRY(launch_zero_upper(dL.p, Np, s));
        KERNEL_TRY(launch_trinv(dL.p, Np, d64.p, dW.p, dT.p, s));
        KERNEL_TRY(launch_pack_w(dW.p, N, Np, 0, dW.p, dT.p, s));      // zero the pad rows (dT reused as scratch)
        std::vector<double> bp(Np, 0.0);
        for (int r = 0; r < nrhs; r++) {
            for (int i = 0; i < N; i++) bp[i] = B_host[(size_t)r * N + i];
            HIP_TRY(hipMemcpy(db.p, bp.data(), sizeof(double) * Np, hipMemcpyHostToDevice));
            KERNEL_TRY(launch_alpha(dW.p, N, Np, db.p, tmp.p, dx.p, d1.p, s));
            HIP_TRY(hipMemcpy(X_host + (size_t)r * N, dx.p, sizeof(double) * N, hipMemcpyDeviceToHost));
        }
    }
    return rc;
}

// inverse of a symmetric positive-definite matrix (N x N host in / out): Cholesky, L^-1, W^T W.
// The preference GP needs C^-1 for L = chol(R + C^-1) (ego/gaussianprocess/__init__.py:488).
extern "C" int ibo_spd_inverse(int device, int N, const double *A_host, double *Ainv_host, int *info)
{
    if (!A_host || !Ainv_host || N < 1) return fail(IBO_ERR_ARG, "bad argument");
    IBO_TRY(use_device(device));
    const int Np = round_up(N, 64);
    const size_t nn = (size_t)Np * Np;
    ScopedBuf<double> dA, dL, dW, dT, d64;
    ScopedBuf<int> dinfo;
    IBO_TRY(dA.ensure(nn)); IBO_TRY(dL.ensure(nn)); IBO_TRY(dW.ensure(nn)); IBO_TRY(dT.ensure(nn));
    IBO_TRY(d64.ensure((size_t)(Np / 64) * 4096)); IBO_TRY(dinfo.ensure(1));
    hipStream_t s = nullptr;
    HIP_TRY(hipMemcpy(dA.p, A_host, sizeof(double) * (size_t)N * N, hipMemcpyHostToDevice));
    KERNEL_TRY(launch_pad_copy(dA.p, N, N, dL.p, Np, 1.0, s));
    KERNEL_TRY(launch_cholesky(dL.p, Np, d64.p, dinfo.p, s));
    int h = 0;
    HIP_TRY(hipMemcpy(&h, dinfo.p, sizeof(int), hipMemcpyDeviceToHost));
    if (info) *info = h;
    int rc = IBO_OK;
    if (h != 0) rc = fail(IBO_ERR_NOT_PD, "matrix is not positive definite (pivot %d)", h);
    else {
        KERNEL_TRY(launch_zero_upper(dL.p, Np, s));
        KERNEL_TRY(launch_trinv(dL.p, Np, d64.p, dW.p, dT.p, s));
        KERNEL_TRY(launch_pack_w(dW.p, N, Np, 0, dW.p, dT.p, s));      // zero the pad rows (dT reused as scratch)
        KERNEL_TRY(launch_wtw(dW.p, dT.p, dA.p, Np, s));
        HIP_TRY(hipMemcpy2D(Ainv_host, sizeof(double) * N, dA.p, sizeof(double) * Np, sizeof(double) * N, N,
                            hipMemcpyDeviceToHost));
    }
    return rc;
}

// 2^(j/2048), j < 2048: the table behind sweep2's exp (one per device, created on first use)
static std::atomic<double *> g_exp_tab[16];
static std::mutex g_exp_mu;                          // held only while a device's table is being created (never across a grid or a gradient)
static int exp_table(int device, const double **out)
{
    double *p = g_exp_tab[device & 15].load(std::memory_order_acquire);
    if (!p) {
        std::lock_guard<std::mutex> lk(g_exp_mu);     // created once per device, by whichever handle sweeps first
        p = g_exp_tab[device & 15].load(std::memory_order_relaxed);
        if (!p) {
            std::vector<double> h(2048);
            for (int j = 0; j < 2048; j++) h[j] = exp2((double)j / 2048.0);
            HIP_TRY(hipMalloc((void **)&p, sizeof(double) * 2048));
            HIP_TRY(hipMemcpy(p, h.data(), sizeof(double) * 2048, hipMemcpyHostToDevice));
            g_exp_tab[device & 15].store(p, std::memory_order_release);
        }
    }
    *out = p;
    return IBO_OK;
}

// ------------------------------------------------------------------------ sweep
static int run_sweep(ibo_gp *g, int64_t M, const double *cand_dev, int acq, double parm, int erf_mode,
                     double clamp_lo, double ymax, int n_excl, const double *excl_host, double excl_radius,
                     int64_t index_base, double *mu_dev, double *s2_dev, double *acq_dev,
                     double *best_val, int64_t *best_idx, bool incremental = false, bool timed = true, bool signal = false,
                     const double *cand_host = nullptr)
{
    if (!g->fitted) return fail(IBO_ERR_STATE, "sweep before a successful fit");
    if (M < 1 || !cand_dev) return fail(IBO_ERR_ARG, "empty candidate set");
    if (acq < 0 || acq > 3) return fail(IBO_ERR_ARG, "unknown acquisition %d", acq);
    hipStream_t s = g->stream;
    SweepArgs a;
    memset(&a, 0, sizeof(a));
    a.kp = g->kp; a.N = g->N; a.Npad = g->Npad; a.DP = g->DP; a.M = M;
    a.Xs = g->Xs.p; a.ak = g->ak.p; a.XA = g->XA.p; a.log_sf2 = log(g->kp.sf2); a.dot_form = (g_dot_override >= 0 && g->D <= IBO_DDOT) ? g_dot_override.load() : g->dot_form;
    a.Xp = g->Xp.p; a.W = g->W.p; a.Wp = g->Wp.p; a.alphaY = g->alphaY.p; a.alpha1 = g->alpha1.p;
    a.cand = cand_dev; a.cand_host = cand_host;
    a.prior.nb = g->nb; a.prior.theta = g->ptheta; a.prior.means = g->pmeans.p; a.prior.beta = g->pbeta.p;
    a.prior.lowerb = g->plowerb.p; a.prior.width = g->pwidth.p;
    a.noise = g->noise; a.clamp_lo = clamp_lo; a.ymax = (ymax == ymax) ? ymax : g->maxY; a.parm = parm;
    a.acq = acq; a.erf_mode = erf_mode;
    a.n_excl = 0; a.excl_radius = excl_radius;
    if (n_excl > 0) {
        if (!excl_host) return fail(IBO_ERR_ARG, "excl_host is NULL");
        IBO_TRY(g->excl.ensure((size_t)n_excl * g->D));
        HIP_TRY(hipMemcpyAsync(g->excl.p, excl_host, sizeof(double) * n_excl * g->D, hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        a.n_excl = n_excl; a.excl = g->excl.p;
    }
    a.index_base = index_base;
    a.out_mu = mu_dev; a.out_s2 = s2_dev; a.out_acq = acq_dev;
    int64_t ntiles = (M + 63) / 64;
    IBO_TRY(g->partv.ensure(2 * ntiles)); IBO_TRY(g->parti.ensure(2 * ntiles));     // sweep2 has 32-candidate tiles
    IBO_TRY(g->res_v.ensure(1)); IBO_TRY(g->res_i.ensure(1));
    a.part_val = g->partv.p; a.part_idx = g->parti.p;
    const bool want_best = best_val || best_idx;
    a.result_val = want_best ? g->res_v.p : nullptr; a.result_idx = want_best ? g->res_i.p : nullptr;
    // batches up to 4096 candidates where the dot form holds: three short kernels spread over the chip (small2.hip;
    // from ~8192 candidates on the panel-split kernel's tiles fill the chip by themselves and it is the faster one).
    // They beat the GEMV kernel down to a single candidate (N = 2048: 22 us against 87; N = 1024: 16 against 38), which
    // is left with the models they do not take (no dot form, rows beyond sweep2's LDS budget).
    const bool small2_ok = g_force_path == 0 && M <= 4096 && a.dot_form && sweep2_fits(a.Npad);
    bool gemv = (g_force_path == 1) || (g_force_path == 0 && M <= 16 && !small2_ok);
    // small batches: spread the IBO_SPLIT_PANEL-row panels over the grid too (one tile per 64 candidates alone
    // would leave most of the 256 CUs idle); above ~128 tiles the plain kernel fills the chip
    // (4097 .. 8192 candidates are at most 256 tiles of the large-batch kernel -- one round of the chip, 134 us at N = 1024 and
    // 495 us at N = 2048 whatever their number, where the panel-split kernel takes 142 .. 221 and 478 .. 842 us)
    const bool sweep2_ok = a.dot_form && sweep2_fits(a.Npad);
    bool split = !gemv && (g_force_path == 3 || (g_force_path == 0 && ntiles * 2 <= 256 && !(sweep2_ok && M > 4096)));
    const bool small2 = split && small2_ok;
    if (small2) {
        IBO_TRY(exp_table(g->device, &a.exp_tab));
        IBO_TRY(g->small_ws.ensure(small_sweep_workspace(g->Npad, M)));
        if (signal && !want_best) {                  // the caller will spin on a host-visible word the last kernel writes
            if (!g->done_flag) {                     // (a recycled handle brings its flag along)
                HIP_TRY(hipHostMalloc((void **)&g->done_flag, 64, hipHostMallocDefault));
                *g->done_flag = 0;
            }
            if (!g->done_count.p) {
                IBO_TRY(g->done_count.ensure(1));
                HIP_TRY(hipMemsetAsync(g->done_count.p, 0, sizeof(unsigned), s));
            }
            a.done_flag = g->done_flag; a.done_seq = ++g->done_seq; a.done_count = g->done_count.p;
            g->signal_pending = true;
        }
        KERNEL_TRY(launch_sweep_small(a, g->small_ws.p, s, timed ? g->ev0 : nullptr, timed ? g->ev1 : nullptr));
        g->sweep_kernel = "wk_small_kernel";
    } else if (split) {
        IBO_TRY(g->qpart.ensure((size_t)((g->Npad + IBO_SPLIT_PANEL - 1) / IBO_SPLIT_PANEL) * M)); IBO_TRY(g->mupart.ensure(2 * (size_t)M));
        a.qpart = g->qpart.p; a.mupart = g->mupart.p;
        KERNEL_TRY(launch_sweep_mfma(a, s, g->ev0, g->ev1));
        g->sweep_kernel = "sweep_mfma_kernel<split>";
    } else if (gemv) {
        IBO_TRY(g->qpart.ensure((size_t)(g->Npad / 64) * M)); IBO_TRY(g->mupart.ensure(2 * (size_t)M));
        a.qpart = g->qpart.p; a.mupart = g->mupart.p;
        KERNEL_TRY(launch_sweep_gemv(a, s, g->ev0, g->ev1));
        g->sweep_kernel = "sweep_gemv_kernel";
    } else if (sweep2_ok) {
        IBO_TRY(exp_table(g->device, &a.exp_tab));
        if (incremental) {
            // the state of this candidate array is kept on the handle; if the model has only grown by a few rows
            // (ibo_gp_extend) since it was formed, those rows are folded in -- O(N) per candidate, not O(N^2)
            // (keyed on the array's GENERATION, not its address: see ibo_dev_alloc.  An array the library did not allocate
            // has none, and is swept in full every time)
            size_t off = 0;
            const uint64_t gen = alloc_generation(g->device, cand_dev, sizeof(double) * (size_t)M * g->D, &off);
            const bool usable = gen != 0 && g->st_gen == gen && g->st_off == off && g->st_M == M && g->st_epoch == g->fit_epoch && g->st_sf2 == g->kp.sf2 &&
                                g->st_N >= 1 && g->st_N <= g->N && g->N - g->st_N <= 8 && (!g->st_pruned || g->N - g->st_N0 <= 16) && g->state.cap >= 5 * (size_t)M &&
                                sweep2_rank1_fits(a.Npad, a.kp.D);
            IBO_TRY(g->state.ensure(5 * (size_t)M));     // [q_a, aY.k*, a1.k*, zsum, q_b]: q = (q_a + q_b) + zsum
            a.qpart = g->state.p;
            a.state5 = 1;
            // EI and UCB grow with the variance, and the variance computed from PART of W's rows bounds it from above: where only
            // the arg-max is wanted, the second half of W's rows (three quarters of the work) runs only for tiles whose bound can
            // still reach the best complete value (sweep2.hip: launch_sweep2_pruned).  PI and the plain mean, per-candidate
            // outputs, or a model the part kernels do not take: every tile complete, as before.
            // (UCB = mu + parm sigma grows with sigma only for parm >= 0: a caller's negative coefficient -- a lower confidence bound --
            // takes the complete-every-tile route)
            const bool monotone = (acq == IBO_ACQ_EI || (acq == IBO_ACQ_UCB && parm >= 0.0)) && !mu_dev && !s2_dev && !acq_dev;
            const int64_t nt32 = (M + IBO_S2_TCAND - 1) / IBO_S2_TCAND;
            a.part_rows = usable ? g->st_N0 : g->N;
            a.part_slack = 1e-13 * (1.0 + fabs(a.ymax) + fabs(a.parm));
            a.rank_hi = g->N; a.wy = g->tmp.p;               // (g->tmp[0 .. Npad) is W y after every fit, extension and ibo_gp_set_y)
            // Drift margin of the lazy refresh: an appended row i moves a stale candidate's mean by nu_i (W y)_i, nu = W k*.  With
            // R = sf2_fit P + (1 + noise - sf2_fit) I (P: the correlation matrix, unit diagonal -- the reference's diagonal rule) and
            // k* = sf2_k p*, R >= sf2_fit P whenever sf2_fit <= 1 + noise, hence |nu_i|^2 <= q = k*^T R^-1 k* <= sf2_k^2 / sf2_fit
            // (p*^T P^-1 p* <= 1 for a valid kernel).  1 for the squared exponentials, magnitude^2-dependent for the SV / Matern
            // kernels and under ibo_gp_set_kstar_sf2.  A model fitted with sf2_fit > 1 + noise has no such bound: never lazy.
            const bool nu_bounded = g->kp_fit.sf2 > 0.0 && g->kp_fit.sf2 <= 1.0 + g->noise;
            a.nu_max = nu_bounded ? (g->kp.sf2 / sqrt(g->kp_fit.sf2)) * (1.0 + 1e-9) : INFINITY;
            if (usable && g->st_pruned) {
                // a two-part state: its tiles fold the appended rows in lazily (launch_sweep2_refresh); a caller that needs every
                // candidate's own numbers (outputs, PI, the plain mean), the A/B switch, or a mean prior (whose second vector W 1
                // moves the means of stale tiles by more than any margin allows) has every tile refreshed and completed instead
                a.tile_done = g->tile_done.p; a.tile_ub = g->tile_ub.p; a.part_best = g->part_words.p; a.part_thresh = g->part_words.p + 1;
                a.tile_rows = g->tile_rows.p; a.tile_sel = g->tile_sel.p; a.part_nlev = g->st_nlev;
                a.part_lazy = monotone && g_gallery_prune == 1 && g->nb == 0 && nu_bounded;
            }
            if (usable) {
                KERNEL_TRY(launch_sweep2_refresh(a, g->st_N, g->N - 1, s, g->ev0, g->ev1));
                g->sweep_kernel = g->N > g->st_N ? "sweep2_rank1_kernel" : "acq_finish_kernel";
            } else if (g_gallery_prune && monotone && sweep2_part_fits(a.Npad, a.kp.D)) {
                IBO_TRY(g->tile_done.ensure((size_t)nt32)); IBO_TRY(g->tile_ub.ensure((size_t)nt32)); IBO_TRY(g->part_words.ensure(2));
                IBO_TRY(g->tile_rows.ensure((size_t)nt32)); IBO_TRY(g->tile_sel.ensure(2 * (size_t)nt32 + 16));     // flags | compact list | counters
                HIP_TRY(hipMemsetAsync(g->tile_done.p, 0, sizeof(int) * (size_t)nt32, s));
                HIP_TRY(hipMemsetAsync(g->tile_rows.p, 0, sizeof(int) * (size_t)nt32, s));
                a.tile_rows = g->tile_rows.p; a.tile_sel = g->tile_sel.p;
                HIP_TRY(hipMemsetAsync(g->state.p + 3 * (size_t)M, 0, sizeof(double) * 2 * (size_t)M, s));
                a.tile_done = g->tile_done.p; a.tile_ub = g->tile_ub.p; a.part_best = g->part_words.p; a.part_thresh = g->part_words.p + 1;
                a.part_nlev = g->st_nlev = sweep2_part_nlev(a.Npad);
                KERNEL_TRY(launch_sweep2_pruned(a, g_gallery_prune == 1, s, g->ev0, g->ev1));
                g->st_pruned = true;
                g->sweep_kernel = "sweep2_kernel<part>";
            } else {
                HIP_TRY(hipMemsetAsync(g->state.p + 3 * (size_t)M, 0, sizeof(double) * 2 * (size_t)M, s));
                KERNEL_TRY(launch_sweep2(a, s, g->ev0, g->ev1));
                g->st_pruned = false;
                g->sweep_kernel = "sweep2_kernel";
            }
            if (!usable) g->st_N0 = g->N;
            g->st_gen = gen; g->st_off = off; g->st_M = M; g->st_N = g->N; g->st_sf2 = g->kp.sf2; g->st_epoch = g->fit_epoch;
        } else {
            IBO_TRY(g->qpart.ensure(3 * (size_t)M));    // (q, aY.k*, a1.k*) per candidate, finished by acq_finish_kernel
            a.qpart = g->qpart.p;
#ifdef IBO_STAMPS
            const size_t nt32s = (size_t)((M + IBO_S2_TCAND - 1) / IBO_S2_TCAND);
            IBO_TRY(g->mupart.ensure(nt32s * 8 + 16));
            a.mupart = g->mupart.p;
#endif
            KERNEL_TRY(launch_sweep2(a, s, g->ev0, g->ev1));
            g->sweep_kernel = "sweep2_kernel";
#ifdef IBO_STAMPS
            if (getenv("IBO_STAMP_FILE")) {
                std::vector<unsigned long long> h(nt32s * 8);
                HIP_TRY(hipStreamSynchronize(s));
                HIP_TRY(hipMemcpy(h.data(), g->mupart.p, h.size() * 8, hipMemcpyDeviceToHost));
                FILE *f = fopen(getenv("IBO_STAMP_FILE"), "wb");
                if (f) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
            }
#endif
        }
    } else {
#ifdef IBO_STAMPS
        IBO_TRY(g->mupart.ensure((size_t)ntiles * 16 + 16));
        a.mupart = g->mupart.p;
#endif
        a.dot_form = 0;                              // the first-generation tile kernel is kept in its difference form only
        KERNEL_TRY(launch_sweep_mfma(a, s, g->ev0, g->ev1));
        g->sweep_kernel = "sweep_mfma_kernel";
#ifdef IBO_STAMPS
        if (getenv("IBO_STAMP_FILE")) {
            std::vector<unsigned long long> h((size_t)ntiles * 16);
            HIP_TRY(hipStreamSynchronize(s));
            HIP_TRY(hipMemcpy(h.data(), g->mupart.p, h.size() * 8, hipMemcpyDeviceToHost));
            FILE *f = fopen(getenv("IBO_STAMP_FILE"), "wb");
            if (f) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
        }
#endif
    }
    if (!best_val && !best_idx) return IBO_OK;        // internal callers that only want the per-point outputs
    double hv; int64_t hi;
    HIP_TRY(hipMemcpyAsync(&hv, g->res_v.p, sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&hi, g->res_i.p, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipEventElapsedTime(&g->sweep_ms, g->ev0, g->ev1));
    if (best_val) *best_val = hv;
    if (best_idx) *best_idx = hi;
    return IBO_OK;
}

extern "C" int ibo_acq_sweep(ibo_gp_t *g, int64_t M, const double *cand_dev, int acq, double parm, int erf_mode,
                             double clamp_lo, double ymax, int n_excl, const double *excl_host,
                             double excl_radius, int64_t index_base, double *mu_dev, double *s2_dev,
                             double *acq_dev, double *best_val, int64_t *best_idx)
{
    if (!g) return fail(IBO_ERR_ARG, "gp is NULL");
    IBO_TRY(use_device(g->device));
    return run_sweep(g, M, cand_dev, acq, parm, erf_mode, clamp_lo, ymax, n_excl, excl_host, excl_radius,
                     index_base, mu_dev, s2_dev, acq_dev, best_val, best_idx);
}

extern "C" int ibo_acq_sweep_incremental(ibo_gp_t *g, int64_t M, const double *cand_dev, int acq, double parm, int erf_mode,
                                         double clamp_lo, double ymax, int n_excl, const double *excl_host,
                                         double excl_radius, int64_t index_base, double *mu_dev, double *s2_dev,
                                         double *acq_dev, double *best_val, int64_t *best_idx)
{
    if (!g) return fail(IBO_ERR_ARG, "gp is NULL");
    IBO_TRY(use_device(g->device));
    return run_sweep(g, M, cand_dev, acq, parm, erf_mode, clamp_lo, ymax, n_excl, excl_host, excl_radius,
                     index_base, mu_dev, s2_dev, acq_dev, best_val, best_idx, true);
}

extern "C" int ibo_sweep_state_info(ibo_gp_t *g, int64_t *tiles, int64_t *complete)
{
    if (!g) return fail(IBO_ERR_ARG, "gp is NULL");
    IBO_TRY(use_device(g->device));
    const int64_t nt = g->st_gen ? (g->st_M + IBO_S2_TCAND - 1) / IBO_S2_TCAND : 0;
    int64_t done = nt;
    if (nt && g->st_pruned) {
        std::vector<int> h((size_t)nt);
        HIP_TRY(hipStreamSynchronize(g->stream));
        HIP_TRY(hipMemcpy(h.data(), g->tile_done.p, sizeof(int) * (size_t)nt, hipMemcpyDeviceToHost));
        done = 0;
        for (int v : h) done += v == g->st_nlev - 1;
    }
    if (tiles) *tiles = nt;
    if (complete) *complete = done;
    return IBO_OK;
}

extern "C" int ibo_sweep_state_levels(ibo_gp_t *g, int *nlev, int *splits, int64_t *tiles_at_level)
{
    if (!g) return fail(IBO_ERR_ARG, "gp is NULL");
    IBO_TRY(use_device(g->device));
    const int64_t nt = g->st_gen ? (g->st_M + IBO_S2_TCAND - 1) / IBO_S2_TCAND : 0;
    const int nl = (nt && g->st_pruned) ? g->st_nlev : 1;
    if (nlev) *nlev = nl;
    if (splits) {
        int all[3];
        const int n = sweep2_part_levels(g->Npad, all) - 1;
        for (int i = 0; i < 3; i++) splits[i] = 0;
        for (int i = 0; i < nl - 1; i++) splits[i] = all[n - (nl - 1) + i];
    }
    if (tiles_at_level) {
        for (int i = 0; i < 4; i++) tiles_at_level[i] = 0;
        if (nl == 1) tiles_at_level[0] = nt;
        else {
            std::vector<int> h((size_t)nt);
            HIP_TRY(hipStreamSynchronize(g->stream));
            HIP_TRY(hipMemcpy(h.data(), g->tile_done.p, sizeof(int) * (size_t)nt, hipMemcpyDeviceToHost));
            for (int v : h) if (v >= 0 && v < 4) tiles_at_level[v]++;
        }
    }
    return IBO_OK;
}

extern "C" int ibo_last_sweep_kernel_ms(ibo_gp_t *g, float *ms, const char **kernel_name)
{
    if (!g) return fail(IBO_ERR_ARG, "gp is NULL");
    if (ms) *ms = g->sweep_ms;
    if (kernel_name) *kernel_name = g->sweep_kernel;
    return IBO_OK;
}

// host-in / host-out evaluation of M points: values of one acquisition (or the
// posterior) -- used by posterior_batch and by DIRECT's batches
static int ensure_pinned(ibo_gp *g, size_t need)
{
    if (need <= g->pin_cap) return IBO_OK;
    if (g->pin) (void)hipHostFree(g->pin);
    g->pin = nullptr; g->pin_cap = 0;
    // head-room for the small, growing batches of DIRECT; exact for large requests (pinning costs ~1 ms/MB)
    size_t cap = need < 4096 ? 4096 : (need < ((size_t)1 << 20) ? need * 2 : need);
    HIP_TRY(hipHostMalloc((void **)&g->pin, cap * sizeof(double), hipHostMallocDefault));
    g->pin_cap = cap;
    return IBO_OK;
}

// Large host-in / host-out batches (GP.posteriors(X) on 10^5..10^7 NumPy rows): chunks of 2^17 points go through
// two sets of pinned + device buffers; the upload of chunk c+1 and the download of chunk c-1 run on their own
// streams while chunk c is in the sweep kernel, so the call costs about the kernel time, not kernel + PCIe +
// pageable staging.
static int eval_host_points_pipelined(ibo_gp *g, int64_t M, const double *Q_host, int acq, double parm, int erf_mode,
                                      double clamp_lo, double *mu_host, double *s2_host, double *acq_host, double ymax)
{
    const int64_t CH = (int64_t)1 << 17;
    const int D = g->D;
    if (!g->h2d_stream) {                             // copy streams and their events: created on first use
        HIP_TRY(hipStreamCreateWithFlags(&g->h2d_stream, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&g->d2h_stream, hipStreamNonBlocking));
        for (int b = 0; b < 2; b++) {
            HIP_TRY(hipEventCreateWithFlags(&g->pe_in[b], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&g->pe_k[b], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&g->pe_out[b], hipEventDisableTiming));
        }
    }
    const int nout = (mu_host ? 1 : 0) + (s2_host ? 1 : 0) + (acq_host ? 1 : 0);
    IBO_TRY(g->cand.ensure((size_t)(2 * CH) * D));
    IBO_TRY(g->outs.ensure((size_t)(2 * CH) * 3));
    IBO_TRY(ensure_pinned(g, (size_t)(2 * CH) * (D + 3)));
    double *pin_in[2] = {g->pin, g->pin + CH * D};
    double *pin_out[2] = {g->pin + 2 * CH * D, g->pin + 2 * CH * D + 3 * CH};
    double *dev_in[2] = {g->cand.p, g->cand.p + CH * D};
    double *dev_out[2] = {g->outs.p, g->outs.p + 3 * CH};
    const int64_t nch = (M + CH - 1) / CH;
    auto drain = [&](int64_t c) -> int {              // results of chunk c: pinned -> caller's arrays
        const int b = (int)(c & 1);
        const int64_t m = (c + 1 < nch) ? CH : M - c * CH;
        HIP_TRY(hipEventSynchronize(g->pe_out[b]));
        int k = 0;
        if (mu_host) memcpy(mu_host + c * CH, pin_out[b] + m * k++, sizeof(double) * m);
        if (s2_host) memcpy(s2_host + c * CH, pin_out[b] + m * k++, sizeof(double) * m);
        if (acq_host) memcpy(acq_host + c * CH, pin_out[b] + m * k++, sizeof(double) * m);
        return IBO_OK;
    };
    for (int64_t c = 0; c < nch; c++) {
        const int b = (int)(c & 1);
        const int64_t m = (c + 1 < nch) ? CH : M - c * CH;
        if (c >= 2) IBO_TRY(drain(c - 2));           // frees buffer set b (its download has finished)
        memcpy(pin_in[b], Q_host + c * CH * D, sizeof(double) * m * D);
        HIP_TRY(hipMemcpyAsync(dev_in[b], pin_in[b], sizeof(double) * m * D, hipMemcpyHostToDevice, g->h2d_stream));
        HIP_TRY(hipEventRecord(g->pe_in[b], g->h2d_stream));
        HIP_TRY(hipStreamWaitEvent(g->stream, g->pe_in[b], 0));
        int k = 0;
        double *dmu = mu_host ? dev_out[b] + m * k++ : nullptr;
        double *ds2 = s2_host ? dev_out[b] + m * k++ : nullptr;
        double *dacq = acq_host ? dev_out[b] + m * k++ : nullptr;
        IBO_TRY(run_sweep(g, m, dev_in[b], acq, parm, erf_mode, clamp_lo, ymax, 0, nullptr, 0.0, 0, dmu, ds2, dacq,
                          nullptr, nullptr));
        HIP_TRY(hipEventRecord(g->pe_k[b], g->stream));
        HIP_TRY(hipStreamWaitEvent(g->d2h_stream, g->pe_k[b], 0));
        HIP_TRY(hipMemcpyAsync(pin_out[b], dev_out[b], sizeof(double) * m * nout, hipMemcpyDeviceToHost, g->d2h_stream));
        HIP_TRY(hipEventRecord(g->pe_out[b], g->d2h_stream));
    }
    if (nch >= 2) IBO_TRY(drain(nch - 2));
    IBO_TRY(drain(nch - 1));
    return IBO_OK;
}

static int eval_host_points(ibo_gp *g, int64_t M, const double *Q_host, int acq, double parm, int erf_mode,
                            double clamp_lo, double *mu_host, double *s2_host, double *acq_host, double ymax = NAN)
{
    if (M >= ((int64_t)1 << 18) && g_host_pipeline)
        return eval_host_points_pipelined(g, M, Q_host, acq, parm, erf_mode, clamp_lo, mu_host, s2_host, acq_host, ymax);
    IBO_TRY(g->cand.ensure((size_t)M * g->D));
    IBO_TRY(g->outs.ensure(3 * (size_t)M));
    // pinned staging (input points + up to 3 output arrays): pageable copies cost ~15 us each and
    // DIRECT issues ~100 small batches per maximisation
    IBO_TRY(ensure_pinned(g, (size_t)M * (g->D + 3)));
    hipStream_t s = g->stream;
    double *pin_in = g->pin, *pin_out = g->pin + (size_t)M * g->D;
    memcpy(pin_in, Q_host, sizeof(double) * M * g->D);
    // Batches of at most 8192 points skip the copy launches altogether: pinned host memory is device-visible, the
    // kernels read the few KB of candidates from it and store the results into it (two ~10 us launches per batch).
    const bool zero_copy = M <= 8192;
    if (!zero_copy) HIP_TRY(hipMemcpyAsync(g->cand.p, pin_in, sizeof(double) * M * g->D, hipMemcpyHostToDevice, s));
    // outputs are contiguous in the order (mu, s2, acq) restricted to the wanted ones
    int nout = 0;
    double *obase = zero_copy ? pin_out : g->outs.p;
    double *dmu = nullptr, *ds2 = nullptr, *dacq = nullptr;
    if (mu_host) dmu = obase + (size_t)M * nout++;
    if (s2_host) ds2 = obase + (size_t)M * nout++;
    if (acq_host) dacq = obase + (size_t)M * nout++;
    g->signal_pending = false;
    IBO_TRY(run_sweep(g, M, zero_copy ? pin_in : g->cand.p, acq, parm, erf_mode, clamp_lo, ymax, 0, nullptr, 0.0, 0, dmu, ds2, dacq,
                      nullptr, nullptr, false, !zero_copy, zero_copy, zero_copy ? pin_in : nullptr));    // small batches: no kernel-time events either
    if (!zero_copy) HIP_TRY(hipMemcpyAsync(pin_out, g->outs.p, sizeof(double) * M * nout, hipMemcpyDeviceToHost, s));
    if (zero_copy) {
        // a batch of this size is back in tens of microseconds: spin for a moment before handing the thread to the runtime's
        // blocking wait (whose wake-up alone costs about as much as the batch) -- on the word small2.hip's last kernel stores
        // behind its results (no event to record, signal and query), or on a completion event for the other kernels
        const bool flag = g->signal_pending;
        if (!flag) HIP_TRY(hipEventRecord(g->fit1, s));
        struct timespec w0, w1;
        clock_gettime(CLOCK_MONOTONIC, &w0);
        for (int spin = 0;; spin++) {
            if (flag) {
                if (*(volatile unsigned long long *)g->done_flag == g->done_seq) break;
                if (spin & 63) continue;
            } else {
                hipError_t q = hipEventQuery(g->fit1);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) HIP_TRY(q);
            }
            clock_gettime(CLOCK_MONOTONIC, &w1);
            if ((w1.tv_sec - w0.tv_sec) * 1e6 + (w1.tv_nsec - w0.tv_nsec) * 1e-3 > 300.0) { HIP_TRY(hipStreamSynchronize(s)); break; }
        }
    } else HIP_TRY(hipStreamSynchronize(s));
    nout = 0;
    if (mu_host) memcpy(mu_host, pin_out + (size_t)M * nout++, sizeof(double) * M);
    if (s2_host) memcpy(s2_host, pin_out + (size_t)M * nout++, sizeof(double) * M);
    if (acq_host) memcpy(acq_host, pin_out + (size_t)M * nout++, sizeof(double) * M);
    return IBO_OK;
}

extern "C" int ibo_posterior_batch(ibo_gp_t *g, int64_t M, const double *Q_host, double clamp_lo,
                                   double *mu_host, double *s2_host)
{
    if (!g || !Q_host || !mu_host) return fail(IBO_ERR_ARG, "NULL argument");
    if (M < 1) return fail(IBO_ERR_ARG, "M=%lld", (long long)M);
    IBO_TRY(use_device(g->device));
    if (!g->fitted) return fail(IBO_ERR_STATE, "posterior before a successful fit");
    return eval_host_points(g, M, Q_host, IBO_ACQ_NONE, 0.0, IBO_ERF_LIBM, clamp_lo, mu_host, s2_host, nullptr);
}

// host points in, host arrays out (any of mu / s2 / acq may be NULL): what EI(GP).negf(x), PI, UCB and their vectorised
// forms ask for -- small batches cost no allocation and no copy launch (pinned staging read and written by the kernels)
extern "C" int ibo_acq_batch(ibo_gp_t *g, int64_t M, const double *Q_host, int acq, double parm, int erf_mode,
                             double clamp_lo, double ymax, double *mu_host, double *s2_host, double *acq_host)
{
    if (!g || !Q_host || (!mu_host && !s2_host && !acq_host)) return fail(IBO_ERR_ARG, "NULL argument");
    if (M < 1) return fail(IBO_ERR_ARG, "M=%lld", (long long)M);
    if (acq < 0 || acq > 3) return fail(IBO_ERR_ARG, "unknown acquisition %d", acq);
    IBO_TRY(use_device(g->device));
    if (!g->fitted) return fail(IBO_ERR_STATE, "evaluation before a successful fit");
    return eval_host_points(g, M, Q_host, acq, parm, erf_mode, clamp_lo, mu_host, s2_host, acq_host, ymax);
}

// ------------------------------------------------------------------------ DIRECT on the GPU objective
static int direct_on_gp(ibo_gp *g, int D, const double *lb, const double *ub, int acq, double parm, int erf_mode,
                        double clamp_lo, int maxiter, int maxtime, int maxsample, int compat,
                        double *opt, double *optx, int64_t *nsamples)
{
    if (D != g->D) return fail(IBO_ERR_ARG, "bounds have %d dimensions, model has %d", D, g->D);
    const bool dbg = getenv("IBO_DEBUG") != nullptr;
    double t_eval = 0.0; int n_batches = 0; int64_t n_pts = 0;
    ibo::batch_eval_t ev = [&](const double *pts, int n, double *vals) -> int {
        struct timespec a0, a1;
        if (dbg) clock_gettime(CLOCK_MONOTONIC, &a0);
        int rc = eval_host_points(g, n, pts, acq, parm, erf_mode, clamp_lo, nullptr, nullptr, vals);
        if (dbg) { clock_gettime(CLOCK_MONOTONIC, &a1); t_eval += (a1.tv_sec - a0.tv_sec) * 1e3 + (a1.tv_nsec - a0.tv_nsec) * 1e-6; n_batches++; n_pts += n; }
        if (rc) return rc;
        for (int i = 0; i < n; i++) vals[i] = -vals[i];     // DIRECT minimises the negated acquisition
        return 0;
    };
    ibo::DirectOptions o;
    o.maxiter = maxiter; o.maxtime = maxtime; o.maxsample = maxsample; o.compat = compat != 0;
    o.per_rectangle = false;
    struct timespec w0, w1;
    clock_gettime(CLOCK_MONOTONIC, &w0);
    ibo::DirectResult r = ibo::direct_minimize(ev, D, lb, ub, o);
    clock_gettime(CLOCK_MONOTONIC, &w1);
    if (dbg) fprintf(stderr, "[libibo_hip] DIRECT: %d iterations, %lld samples, %d batches (%lld points): %.2f ms total, %.2f ms in GPU evaluation\n",
                     r.iterations, (long long)r.nsamples, n_batches, (long long)n_pts,
                     (w1.tv_sec - w0.tv_sec) * 1e3 + (w1.tv_nsec - w0.tv_nsec) * 1e-6, t_eval);
    if (r.status) return r.status;
    if (opt) *opt = -r.fmin;
    if (optx) for (int i = 0; i < D; i++) optx[i] = r.xmin[i];
    if (nsamples) *nsamples = r.nsamples;
    return IBO_OK;
}

extern "C" int ibo_direct_max(ibo_gp_t *g, int D, const double *lb, const double *ub, int acq, double parm,
                              int erf_mode, double clamp_lo, int maxiter, int maxtime, int maxsample,
                              int compat, double *opt, double *optx, int64_t *nsamples)
{
    if (!g || !lb || !ub) return fail(IBO_ERR_ARG, "NULL argument");
    if (acq < 0 || acq > 2) return fail(IBO_ERR_ARG, "unknown acquisition %d", acq);
    IBO_TRY(use_device(g->device));
    if (!g->fitted) return fail(IBO_ERR_STATE, "direct before a successful fit");
    return direct_on_gp(g, D, lb, ub, acq, parm, erf_mode, clamp_lo, maxiter, maxtime, maxsample, compat,
                        opt, optx, nsamples);
}

// ------------------------------------------------------------------------ marginal-likelihood grid
struct NlmlWorkspace {
    DevBuf<double> dX, dY, dout, dL, d64, dP;       // dP: packed store of the trailing updates (update3.hip)
    DevBuf<KParams> dkp;                            // the theta-points' kernel parameters (one covariance launch per sub-batch)
    DevBuf<int> dinfo, dflags;                      // dflags: four hand-over words per matrix (chol_panel_fused_kernel)
    const double *padded = nullptr;                 // dL as it was when its matrices got their identity pad,
    int pad_Np = 0, pad_N = 0, pad_B = 0;           // and for which geometry
    hipStream_t streams[4] = {nullptr, nullptr, nullptr, nullptr};     // sub-batches of a grid run side by side (created on first use, kept)
};
static NlmlWorkspace g_nlml_ws[16];
static const int kSyrk3From = 2560;          // rows from which ibo_nlml_grad forms K^-1 = W^T W on the packed-operand kernel (launch_syrk3)
struct GradWorkspace {
    DevBuf<double> dX, dY, dL, dW, dT, dKi, d64, dal, da1, tmp, dpart, dout, dpiece;
    DevBuf<int> dinfo, dtasks, dsums;
    int plan_Np = 0, ntasks = 0, nsums = 0;         // launch_syrk3's lists on the device, for this Npad
};
static GradWorkspace g_grad_ws[16];

extern "C" int ibo_trim(int device)
{
    IBO_TRY(use_device(device));
    std::lock_guard<std::mutex> lk(g_dev_mu[device & 15]);
    NlmlWorkspace &ws = g_nlml_ws[device & 15];
    ws.dX.release(); ws.dY.release(); ws.dout.release(); ws.dL.release(); ws.d64.release(); ws.dP.release(); ws.dinfo.release(); ws.dflags.release(); ws.dkp.release();
    ws.padded = nullptr;
    for (int g = 0; g < 4; g++) if (ws.streams[g]) { (void)hipStreamDestroy(ws.streams[g]); ws.streams[g] = nullptr; }
    GradWorkspace &gw = g_grad_ws[device & 15];
    gw.dX.release(); gw.dY.release(); gw.dL.release(); gw.dW.release(); gw.dT.release(); gw.dKi.release(); gw.d64.release();
    gw.dal.release(); gw.da1.release(); gw.tmp.release(); gw.dpart.release(); gw.dout.release(); gw.dinfo.release();
    gw.dpiece.release(); gw.dtasks.release(); gw.dsums.release(); gw.plan_Np = 0;
    pool_trim(device);
    return IBO_OK;
}

extern "C" int ibo_nlml_grid(int device, int ktype, int N, int D, const double *X, const double *Y,
                             int n_theta, const double *thetas, int nhyper, const double *sf2s, double noise,
                             double *nlml_host)
{
    if (!X || !Y || !thetas || !nlml_host || N < 1 || n_theta < 1) return fail(IBO_ERR_ARG, "bad argument");
    IBO_TRY(use_device(device));
    std::lock_guard<std::mutex> lk(g_dev_mu[device & 15]);      // the batch workspace is per device: concurrent grids take turns
    const int Np = round_up(N + 1, 64);            // room for the appended y row (see aug_row_kernel)
    // theta-points are independent and one factorisation is a latency-bound chain of small kernels:
    // B matrices sit side by side in HBM (B x 8 Np^2 bytes -- 4.4 GB for 32 x N=4096, nothing on a 288 GB
    // part) and every launch of the chain works on all of them (blockIdx.z), so the chain's latency is
    // paid once per batch and the update kernels fill the chip.
    const size_t nn = (size_t)Np * Np;
    int B;
    {
        const size_t budget = (size_t)12 << 30;    // bytes of factor storage per batch (of 288 GB)
        size_t fit = budget / (nn * sizeof(double));
        if (fit < 1) fit = 1;
        if (fit > 256) fit = 256;                   // N = 4096: 64 matrices side by side 0.617 ms per theta, 32: 0.655, 16: 0.72; N = 1024: 256: 45 us, 32: 69 us
        B = g_nlml_batch > 0 ? g_nlml_batch.load() : (int)fit;
        if (B > n_theta) B = n_theta;
    }
    // the workspace is kept between calls (hyper-parameter learning calls this in a loop and allocating and
    // freeing gigabytes costs more than the factorisations); ibo_trim() gives it back
    NlmlWorkspace &ws = g_nlml_ws[device & 15];
    DevBuf<double> &dX = ws.dX, &dY = ws.dY, &dout = ws.dout, &dL = ws.dL, &d64 = ws.d64;
    DevBuf<int> &dinfo = ws.dinfo;
    hipStream_t s = nullptr;
    IBO_TRY(dX.ensure((size_t)N * D)); IBO_TRY(dY.ensure(N));
    IBO_TRY(dout.ensure(2 * (size_t)n_theta)); IBO_TRY(dinfo.ensure(n_theta));
    IBO_TRY(dL.ensure(nn * B)); IBO_TRY(d64.ensure((size_t)(Np / 64) * 4096 * B));
    // packed operands of the trailing updates, per matrix: the factor's finished columns in fragment order (update3.hip; the
    // right-looking A/B order writes and reads one panel of it at a time)
    const bool left = g_chol_left != 0;
    const size_t pws = nn;
    IBO_TRY(ws.dP.ensure(pws * B));
    IBO_TRY(ws.dflags.ensure((size_t)4 * B));
    HIP_TRY(hipMemcpy(dX.p, X, sizeof(double) * N * D, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dY.p, Y, sizeof(double) * N, hipMemcpyHostToDevice));
    // identity pad once: the factorisation leaves the pad rows/columns as it found them, so the workspace of an
    // earlier call with the same geometry still has them (a learning loop calls this again and again)
    if (ws.padded != dL.p || ws.pad_Np != Np || ws.pad_N != N || ws.pad_B < B) {
        for (int k = 0; k < B; k++) KERNEL_TRY(launch_pad_copy(dX.p, 0, 1, dL.p + nn * k, Np, 1.0, s));
        ws.padded = dL.p; ws.pad_Np = Np; ws.pad_N = N; ws.pad_B = B;
    }
    // every theta-point's kernel parameters go up once; a sub-batch's covariance matrices are one launch
    std::vector<KParams> kps(n_theta);
    for (int t = 0; t < n_theta; t++)
        IBO_TRY(make_kparams(ktype, D, thetas + (size_t)t * nhyper, nhyper, sf2s ? sf2s[t] : 1.0, &kps[t]));
    IBO_TRY(ws.dkp.ensure(n_theta));
    HIP_TRY(hipMemcpy(ws.dkp.p, kps.data(), sizeof(KParams) * n_theta, hipMemcpyHostToDevice));
    HIP_TRY(hipStreamSynchronize(s));               // the identity pad is in place before the sub-batches' streams start
    // Whatever way this function is left, nothing of it stays in flight on the sub-batch streams: they are non-blocking, so the next
    // call's blocking copies into dX / dY / dkp would not wait for them (an error return inside the batch loop used to leave them running).
    struct StreamDrain {
        NlmlWorkspace &w;
        ~StreamDrain() { for (int g = 0; g < 4; g++) if (w.streams[g]) (void)hipStreamSynchronize(w.streams[g]); }
    } drain{ws};
    std::vector<int> info(n_theta);
    for (int t0 = 0; t0 < n_theta; t0 += B) {
        const int nb = n_theta - t0 < B ? n_theta - t0 : B;
        // One batch.  with_flags: the panels' diagonal blocks and the rows below them in one launch whose row workgroups wait on flags the
        // diagonal workgroups raise (chol_panel_fused_kernel).  Such a wait is bounded; if one ever runs out (the launch's forward progress
        // rests on the dispatch order of its workgroups) the workgroup leaves the code kPanelWaitTimeout in the matrix's info word -- which
        // says nothing about the matrix: the batch is then run again through the two-launch form of the panels (no waits, the same bits).
        auto run_batch = [&](bool with_flags) -> int {
            // sub-batches of at least 8 matrices, each on its own stream: one's in-panel chain (64 workgroups at a time, latency)
            // and launch tails run beside the other's long-K updates
            int G = left ? g_nlml_groups.load() : 1;
            while (G > 1 && nb / G < 8) G--;
            CholGroup grp[4];
            for (int g = 0; g < G; g++) {
                if (!ws.streams[g]) HIP_TRY(hipStreamCreateWithFlags(&ws.streams[g], hipStreamNonBlocking));
                hipStream_t sg = ws.streams[g];
                const int k0 = (int)((long long)nb * g / G), k1 = (int)((long long)nb * (g + 1) / G), ng = k1 - k0;
                grp[g] = CholGroup{dL.p + nn * k0, d64.p + (size_t)(Np / 64) * 4096 * k0, ws.dP.p + pws * k0, dinfo.p + t0 + k0, ng, sg,
                                   with_flags ? ws.dflags.p + 4 * k0 : nullptr};
                KERNEL_TRY(launch_cov_matrix_batched(ws.dkp.p + t0 + k0, ng, N, dX.p, D, IBO_DIAG_KERNEL_PLUS_NOISE, noise, dL.p + nn * k0, Np, nn, sg));
                KERNEL_TRY(launch_nlml_aug(dL.p + nn * k0, Np, N, dY.p, sg, ng, nn));
            }
            // (N a multiple of 64: the y row sits alone in the last block column, whose factor nobody reads -- it is left out)
            if (left) KERNEL_TRY(launch_cholesky_batched_left(grp, G, Np, nn, 4, pws, N + 1, N % 64 == 0 ? Np / 64 - 1 : Np / 64, N / 64));
            else KERNEL_TRY(launch_cholesky_batched(grp[0].L, Np, grp[0].diag64, grp[0].info, grp[0].batch, nn, 4, grp[0].stream, grp[0].Pk, pws));      // (G = 1)
            for (int g = 0; g < G; g++) KERNEL_TRY(launch_nlml_reduce(grp[g].L, Np, N, dout.p + 2 * (size_t)(grp[g].info - dinfo.p), grp[g].stream, grp[g].batch, nn));
            for (int g = 0; g < G; g++) HIP_TRY(hipStreamSynchronize(ws.streams[g]));      // the next batch reuses the matrix slots
            HIP_TRY(hipMemcpy(info.data() + t0, dinfo.p + t0, sizeof(int) * nb, hipMemcpyDeviceToHost));
            return IBO_OK;
        };
        IBO_TRY(run_batch(true));
        bool timed_out = false;
        for (int k = 0; k < nb; k++) timed_out |= info[t0 + k] == kPanelWaitTimeout;
        if (timed_out) {
            // (the matrices were overwritten by the failed attempt: run_batch forms them again; the identity pad is untouched by a factorisation)
            IBO_TRY(run_batch(false));
            for (int k = 0; k < nb; k++)
                if (info[t0 + k] == kPanelWaitTimeout) return fail(IBO_ERR_HIP, "a panel launch reported a wait that ran out on the path that has no waits");
        }
    }
    HIP_TRY(hipStreamSynchronize(s));
    std::vector<double> out(2 * (size_t)n_theta);
    HIP_TRY(hipMemcpy(out.data(), dout.p, sizeof(double) * out.size(), hipMemcpyDeviceToHost));
    const double half_log_2pi_n = 0.5 * N * log(2.0 * M_PI);
    for (int t = 0; t < n_theta; t++)
        nlml_host[t] = info[t] ? NAN : 0.5 * out[2 * t] + out[2 * t + 1] + half_log_2pi_n;
    return IBO_OK;
}

// NLML and its gradient w.r.t. the log hyper-parameters for ONE theta: marginalLikelihood(...,
// computeGradient=True) of ego/gaussianprocess/trainhyper.py:47-75.  modes/dims describe
// Kernel.derivative(X, h) for h < ngrad (see GradSpec).
extern "C" int ibo_nlml_grad(int device, int ktype, int N, int D, const double *X, const double *Y,
                             const double *hyper, int nhyper, double sf2, double noise,
                             int ngrad, const int *modes, const int *dims, double *nlml_host, double *grad_host)
{
    if (!X || !Y || !hyper || !modes || !dims || !nlml_host || !grad_host || N < 1) return fail(IBO_ERR_ARG, "bad argument");
    if (ngrad < 1 || ngrad > IBO_GRAD_MAX) return fail(IBO_ERR_ARG, "ngrad=%d unsupported (1..%d)", ngrad, IBO_GRAD_MAX);
    IBO_TRY(use_device(device));
    std::lock_guard<std::mutex> lk(g_dev_mu[device & 15]);      // (its workspace too)
    KParams kp;
    IBO_TRY(make_kparams(ktype, D, hyper, nhyper, sf2, &kp));
    GradSpec gs;
    gs.nh = ngrad;
    for (int h = 0; h < ngrad; h++) {
        if (modes[h] < 0 || modes[h] > 4 || dims[h] < 0 || dims[h] >= D) return fail(IBO_ERR_ARG, "bad derivative spec");
        gs.mode[h] = modes[h]; gs.dim[h] = dims[h];
    }
    const int Np = round_up(N, 64);
    const size_t nn = (size_t)Np * Np;
    const int nblk = ((N + 15) / 16) * ((N + 15) / 16);
    // workspace kept between calls (BFGS calls this dozens of times; five N^2 buffers allocated and freed per
    // call cost as much as the arithmetic); ibo_trim() releases it
    GradWorkspace &ws = g_grad_ws[device & 15];
    DevBuf<double> &dX = ws.dX, &dY = ws.dY, &dL = ws.dL, &dW = ws.dW, &dT = ws.dT, &dKi = ws.dKi, &d64 = ws.d64,
                   &dal = ws.dal, &da1 = ws.da1, &tmp = ws.tmp, &dpart = ws.dpart, &dout = ws.dout;
    DevBuf<int> &dinfo = ws.dinfo;
    IBO_TRY(dX.ensure((size_t)N * D)); IBO_TRY(dY.ensure(Np)); IBO_TRY(dL.ensure(nn)); IBO_TRY(dW.ensure(nn));
    IBO_TRY(dT.ensure(nn)); IBO_TRY(dKi.ensure(nn)); IBO_TRY(d64.ensure((size_t)(Np / 64) * 4096));
    IBO_TRY(dal.ensure(Np)); IBO_TRY(da1.ensure(Np)); IBO_TRY(tmp.ensure(2 * (size_t)Np + 2 * (size_t)(Np / 64) * Np + 64));
    IBO_TRY(dpart.ensure((size_t)ngrad * nblk)); IBO_TRY(dout.ensure(ngrad + 2)); IBO_TRY(dinfo.ensure(1));
    hipStream_t s = nullptr;
    std::vector<double> yp(Np, 0.0);
    for (int i = 0; i < N; i++) yp[i] = Y[i];
    HIP_TRY(hipMemcpy(dX.p, X, sizeof(double) * N * D, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dY.p, yp.data(), sizeof(double) * Np, hipMemcpyHostToDevice));
    // up to 2048 rows: the fit's route -- fused steps with W = L^-1 riding along (dT: the matrix being reduced, dKi: (L^-1)^T
    // until the transpose) -- instead of the three-kernel columns and the recursive-doubling inversion
    const bool fused = single_level_order(Np);
    if (fused) {
        KERNEL_TRY(launch_cov_fit(kp, N, dX.p, D, IBO_DIAG_KERNEL_PLUS_NOISE, noise, dT.p, Np, dW.p, dinfo.p, s));
        KERNEL_TRY(launch_cholesky_fused(dT.p, dL.p, Np, d64.p, dinfo.p, s, dW.p, dKi.p, true));
    } else {
        // beyond: the two-level order with fused in-panel columns, out of place
        KERNEL_TRY(launch_cov_fit(kp, N, dX.p, D, IBO_DIAG_KERNEL_PLUS_NOISE, noise, dT.p, Np, nullptr, dinfo.p, s));
        KERNEL_TRY(launch_cholesky_fused2(dT.p, dL.p, Np, d64.p, dinfo.p, 4, s, true));
    }
    // no look at the info word until everything is queued: a failed factorisation only turns the rest into NaNs
    if (fused) KERNEL_TRY(launch_transpose_pack(dKi.p, N, Np, dW.p, nullptr, s));      // W, pad rows zero (no packed copy: nothing sweeps here)
    else {
        KERNEL_TRY(launch_zero_upper(dL.p, Np, s));
        KERNEL_TRY(launch_trinv(dL.p, Np, d64.p, dW.p, dT.p, s));
        KERNEL_TRY(launch_pack_w(dW.p, N, Np, 0, dW.p, dT.p, s));                      // zero the pad rows
    }
    KERNEL_TRY(launch_alpha(dW.p, N, Np, dY.p, tmp.p, dal.p, da1.p, s));
    // K^-1 = W^T W: with the ride-along, W^T is what the factorisation left in dKi -- no transpose; the result goes to dT, free by now
    const double *Kinv = fused ? dT.p : dKi.p;
    KERNEL_TRY(launch_nlml_scalars(dL.p, Np, N, dY.p, dal.p, dout.p + ngrad, s));        // (y . alpha, sum log L_ii): L has been read for the last time
    if (fused && Np >= kSyrk3From) {
        // from 2560 rows the product runs on the packed-operand kernel (128 x 128 tiles, A fragments straight from L2), its long K ranges in pieces;
        // the packed copy of W^T goes where L was
        if (ws.plan_Np != Np) {
            std::vector<int> tasks, sums;
            int nslots = 0;
            syrk3_plan(Np, 1024, tasks, sums, &nslots);
            IBO_TRY(ws.dtasks.ensure(tasks.size())); IBO_TRY(ws.dsums.ensure(sums.size() + 4)); IBO_TRY(ws.dpiece.ensure((size_t)(nslots + 1) * 16384));
            HIP_TRY(hipMemcpy(ws.dtasks.p, tasks.data(), sizeof(int) * tasks.size(), hipMemcpyHostToDevice));
            if (!sums.empty()) HIP_TRY(hipMemcpy(ws.dsums.p, sums.data(), sizeof(int) * sums.size(), hipMemcpyHostToDevice));
            ws.plan_Np = Np; ws.ntasks = (int)tasks.size() / 4; ws.nsums = (int)sums.size() / 4;
        }
        KERNEL_TRY(launch_syrk3(dKi.p, dL.p, dT.p, Np, ws.dtasks.p, ws.ntasks, ws.dsums.p, ws.nsums, ws.dpiece.p, s));
    } else if (fused) KERNEL_TRY(launch_wtw(dW.p, dKi.p, dT.p, Np, s, 1, 1));
    else KERNEL_TRY(launch_wtw(dW.p, dT.p, dKi.p, Np, s, 1));
    KERNEL_TRY(launch_nlml_grad(kp, gs, N, dX.p, D, Kinv, Np, dal.p, dpart.p, dout.p, s));
    std::vector<double> res(ngrad + 2);
    int h = 0;
    HIP_TRY(hipMemcpy(res.data(), dout.p, sizeof(double) * (ngrad + 2), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&h, dinfo.p, sizeof(int), hipMemcpyDeviceToHost));
    if (h != 0) return fail(IBO_ERR_NOT_PD, "covariance matrix is not positive definite (pivot %d)", h);
    for (int i = 0; i < ngrad; i++) grad_host[i] = res[i];
    *nlml_host = 0.5 * res[ngrad] + res[ngrad + 1] + 0.5 * N * log(2.0 * M_PI);
    return IBO_OK;
}

// ------------------------------------------------------------------------ libego's arithmetic in libego's order (legacy.hip)
// DIRECT (same host search as every other entry point, libego's dimension-0 quirk on) over an objective whose every number is
// libego's: k*, the prior mean and the acquisition on the host's libm (LegacyHost, a crew of host threads over the batch's
// points), the two N^2 contractions per point on the device in libego's summation order.  Without a prior the first
// contraction's inner vector inv(R) Y is the same for every point: formed once.  Buffers: the handle's (MT in g->W, vectors in
// g->cand / g->outs / g->tmp, pinned staging).
static int legacy_direct(ibo_gp *g, const LegacySpec &m, const double *invR_host, const double *lb, const double *ub,
                         int maxiter, int maxtime, int maxsample, double *fmin, double *xmin)
{
    IBO_TRY(use_device(g->device));
    const int N = m.rows, D = m.dim;
    if (N < 1 || D < 1 || !invR_host || !m.obs || !m.targets || !m.hyper) return fail(IBO_ERR_ARG, "bad argument");
    hipStream_t s = g->stream;
    const size_t nn = (size_t)N * N;
    IBO_TRY(g->A.ensure(nn)); IBO_TRY(g->W.ensure(nn)); IBO_TRY(g->Y.ensure(2 * (size_t)N));
    HIP_TRY(hipMemcpyAsync(g->A.p, invR_host, sizeof(double) * nn, hipMemcpyHostToDevice, s));
    KERNEL_TRY(launch_legacy_transpose(g->A.p, g->W.p, N, s));
    const bool prior = m.nbasis > 0;
    double *MbY = g->Y.p + N;                                                            // inv(R) Y in libego's order (no prior)
    if (!prior) {
        HIP_TRY(hipMemcpyAsync(g->Y.p, m.targets, sizeof(double) * N, hipMemcpyHostToDevice, s));
        IBO_TRY(g->outs.ensure(1));
        // (the matvec half of aMb; its dot half runs per point against that point's r)
        KERNEL_TRY(launch_legacy_aMb(g->W.p, g->Y.p, g->Y.p, MbY, g->outs.p, N, 1, s));
    }
    LegacyHost host(m);
    std::vector<double> pmu;
    ibo::batch_eval_t ev = [&](const double *pts, int n, double *vals) -> int {
        // per point: r (and Y - m under a prior) from the host; vectors B = [r | ymu], A = [r | r]
        const int nvec = prior ? 2 * n : n;
        const size_t vb = (size_t)nvec * N;
        IBO_TRY(ensure_pinned(g, vb + 2 * (size_t)n));
        IBO_TRY(g->cand.ensure(vb)); IBO_TRY(g->tmp.ensure(vb + (size_t)n * N)); IBO_TRY(g->outs.ensure(2 * (size_t)n + 1));
        double *hB = g->pin, *hout = g->pin + vb;
        pmu.resize(n);
        host.prepare(pts, n, hB, pmu.data());
        HIP_TRY(hipMemcpyAsync(g->cand.p, hB, sizeof(double) * vb, hipMemcpyHostToDevice, s));
        double *dB = g->cand.p, *dMb = g->tmp.p, *dout = g->outs.p + 1;
        // x2 = aMb(r, invR, r) for every point; x1 = aMb(r, invR, ymu) under a prior, else the dot of r with the cached inv(R) Y
        KERNEL_TRY(launch_legacy_aMb(g->W.p, dB, dB, dMb, dout + n, N, n, s));
        if (prior) KERNEL_TRY(launch_legacy_aMb(g->W.p, dB + (size_t)n * N, dB, dMb + (size_t)n * N, dout, N, n, s));
        else KERNEL_TRY(launch_legacy_dots(MbY, dB, dout, N, n, s));
        HIP_TRY(hipMemcpyAsync(hout, dout, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (int p = 0; p < n; p++) vals[p] = host.negated(pmu[p], hout[p], hout[n + p]);
        return 0;
    };
    ibo::DirectOptions o;
    o.maxiter = maxiter; o.maxtime = maxtime; o.maxsample = maxsample; o.compat = true; o.per_rectangle = false;
    ibo::DirectResult r = ibo::direct_minimize(ev, D, lb, ub, o);
    if (r.status) return r.status;
    *fmin = r.fmin;
    for (int i = 0; i < D; i++) xmin[i] = r.xmin[i];
    return IBO_OK;
}

// ------------------------------------------------------------------------ legacy libego symbols
extern "C" const double *acqmaxGP(int ndim, double *lb, double *ub, double *invR, double *X, double *Y, int nx,
                                  int acqfunc, int kerneltype, double *hyperparams, int npbases,
                                  double *pbasismeans, double *pbasisbeta, double pbasistheta,
                                  double *pbasislowerb, double *pbasiswidth, double parm, double noise,
                                  int maxiter, int maxtime, int maxsample)
{
    if (acqfunc < 0 || acqfunc > 2) {
        printf("[C++] unknown acquisition function\n");     // cpp/optimizeGP.cpp:342-345
        return NULL;
    }
    if (kerneltype < 0 || kerneltype > 3) {
        // the reference's switch has no such case and would evaluate uninitialised k* values (cpp/optimizeGP.cpp:67-113): refused
        fprintf(stderr, "[libibo_hip] acqmaxGP: unknown kernel type %d\n", kerneltype);
        return NULL;
    }
    ibo_gp *g = nullptr;
    int dev = 0;
    const char *e = getenv("IBO_DEVICE");
    if (e) dev = atoi(e);
    if (ibo_gp_create(dev, &g) != IBO_OK) { fprintf(stderr, "[libibo_hip] %s\n", g_err); return NULL; }
    // sf2: 1 for kernel types 0-2; magnitude^2 for Matern-5/2.  The reference reads
    // hyperparams[ndim] there (cpp/optimizeGP.cpp:313), which is the magnitude only for
    // ndim == 1 and out of bounds otherwise; the magnitude lives at hyperparams[1].
    double sf2 = 1.0;
    int nh = (kerneltype == IBO_K_SE_ARD) ? ndim : 1;
    if (kerneltype == IBO_K_MATERN5) sf2 = hyperparams[1] * hyperparams[1];
    double *res = nullptr;
    int rc;
    if (g_legacy_exact) {
        LegacySpec m;
        m.family = kerneltype; m.dim = ndim; m.rows = nx; m.obs = X; m.targets = Y; m.hyper = hyperparams;
        m.amp = kerneltype == IBO_K_MATERN5 ? exp(2.0 * log(hyperparams[1])) : 1.0;      // (cpp/optimizeGP.cpp:303-314)
        m.nbasis = npbases; m.centres = pbasismeans; m.weights = pbasisbeta; m.sharpness = pbasistheta; m.origin = pbasislowerb; m.extent = pbasiswidth;
        m.acq = acqfunc; m.parm = parm; m.noise = noise;
        std::vector<double> xo(ndim);
        double fmin = 0.0;
        rc = legacy_direct(g, m, invR, lb, ub, maxiter, maxtime, maxsample, &fmin, xo.data());
        if (rc == IBO_OK) {
            res = (double *)malloc(sizeof(double) * (ndim + 1));
            res[0] = fmin;
            for (int i = 0; i < ndim; i++) res[i + 1] = xo[i];
        }
        if (rc != IBO_OK) fprintf(stderr, "[libibo_hip] acqmaxGP failed: %s\n", g_err);
        ibo_gp_destroy(g);
        return res;
    }
    rc = fit_from_inverse(g, kerneltype, nx, ndim, X, Y, hyperparams, nh, sf2, noise, invR);
    if (rc == IBO_OK && npbases > 0)
        rc = ibo_gp_set_prior(g, npbases, pbasismeans, pbasisbeta, pbasistheta, pbasislowerb, pbasiswidth);
    if (rc == IBO_OK) {
        std::vector<double> xo(ndim);
        double opt = 0.0;
        rc = direct_on_gp(g, ndim, lb, ub, acqfunc, parm, IBO_ERF_LIBM, 1e-8, maxiter, maxtime, maxsample, 1,
                          &opt, xo.data(), nullptr);
        if (rc == IBO_OK) {
            res = (double *)malloc(sizeof(double) * (ndim + 1));
            res[0] = -opt;
            for (int i = 0; i < ndim; i++) res[i + 1] = xo[i];
        }
    }
    if (rc != IBO_OK) fprintf(stderr, "[libibo_hip] acqmaxGP failed: %s\n", g_err);
    ibo_gp_destroy(g);
    return res;
}

extern "C" const double *direct(objective_t objective, int ndim, double *lb, double *ub, int maxiter,
                                int maxtime, int maxsample)
{
    std::vector<double> x(ndim);
    ibo::batch_eval_t ev = [&](const double *pts, int n, double *vals) -> int {
        for (int p = 0; p < n; p++) {
            for (int i = 0; i < ndim; i++) x[i] = pts[(size_t)p * ndim + i];
            vals[p] = objective(ndim, x.data());
        }
        return 0;
    };
    ibo::DirectOptions o;
    o.maxiter = maxiter; o.maxtime = maxtime; o.maxsample = maxsample; o.compat = true; o.per_rectangle = true;
    ibo::DirectResult r = ibo::direct_minimize(ev, ndim, lb, ub, o);
    double *res = (double *)malloc(sizeof(double) * (ndim + 1));
    res[0] = r.fmin;
    for (int i = 0; i < ndim; i++) res[i + 1] = r.xmin[i];
    return res;
}

// libego's preference log-likelihood helper (cpp/helpers.cpp:30-56), host arithmetic: pairs are taken with
// stride 2 from the index array, a term is skipped when Phi(.)/sqrt 2 is exactly zero
extern "C" double logCDFs(int nprefinds, int *prefinds, double *x)
{
    const double Z = sqrt(2.0);
    double lcdf = 0.0;
    for (int i = 0; i + 1 < nprefinds; i += 2) {
        const double q = 0.5 * (1.0 + erf((x[prefinds[i]] - x[prefinds[i + 1]]) / Z));
        if (q / Z != 0.0) lcdf += log(q / Z);
    }
    return lcdf;
}

// host-callback DIRECT with the sample counter and the compat switch exposed
extern "C" int ibo_direct_host(objective_t objective, int ndim, const double *lb, const double *ub, int maxiter,
                               int maxtime, int maxsample, int compat, double *fmin, double *xmin, int64_t *nsamples)
{
    if (!objective || !lb || !ub) return fail(IBO_ERR_ARG, "NULL argument");
    std::vector<double> x(ndim);
    ibo::batch_eval_t ev = [&](const double *pts, int n, double *vals) -> int {
        for (int p = 0; p < n; p++) {
            for (int i = 0; i < ndim; i++) x[i] = pts[(size_t)p * ndim + i];
            vals[p] = objective(ndim, x.data());
        }
        return 0;
    };
    ibo::DirectOptions o;
    // bit 1 of `compat`: the objective is called on one batch per iteration (probes + guessed child centres), the schedule
    // the GPU objective runs under -- same (fmin, xmin, nsamples) as the per-rectangle call order (tested)
    o.maxiter = maxiter; o.maxtime = maxtime; o.maxsample = maxsample; o.compat = (compat & 1) != 0; o.per_rectangle = (compat & 2) == 0;
    ibo::DirectResult r = ibo::direct_minimize(ev, ndim, lb, ub, o);
    if (fmin) *fmin = r.fmin;
    if (xmin) for (int i = 0; i < ndim; i++) xmin[i] = r.xmin[i];
    if (nsamples) *nsamples = r.nsamples;
    return IBO_OK;
}
