// abi_core.hip -- the library-level part of the C ABI (include/ibo_abi.h): error channel, options, device memory with
// generations, the recycled buffers / streams / events, and the life cycle of a handle.
#include "abi_internal.h"

static thread_local char g_err[512] = "";

int ibo_fail(int code, const char *fmt, ...)
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

std::atomic<int> g_host_pipeline{1};  // ibo_set_option("host_pipeline", 0/1): chunked, overlapped host batches (0: one shot -- the test's comparator)
std::atomic<int> g_fused2_min_nb{104};  // ibo_set_option("fused2_min_nb"): block columns from which a single matrix takes the two-level order
std::atomic<int> g_gallery_prune{1};  // ibo_set_option("gallery_prune", 0/1/2): kept-state sweeps in levels of W's rows, the later ones only where a
                                 // tile's bound can still win (1); the same launches with every tile completed (2); the one-kernel sweep (0)
std::atomic<int> g_nlml_batch{0};     // ibo_set_option("nlml_batch", b): matrices per batched factorisation of ibo_nlml_grid (0: as many as 12 GB hold)
std::atomic<int> g_chol_left{1};      // ibo_set_option("chol_left", 0/1): ibo_nlml_grid factors in the left-looking outer order (update3.hip); 0: the
                                 // right-looking order of launch_cholesky_batched -- the same bits, the test's comparator
std::atomic<int> g_dot_override{-1};  // ibo_set_option("dot_form", -1/0/1): -1 auto, 0/1 force the difference / dot form of k* (tests)
std::atomic<int> g_legacy_exact{1};   // ibo_set_option("legacy_exact", 0/1): acqmaxGP evaluates libego's formulas in libego's operation order (legacy.hip)
std::atomic<int> g_force_path{0};     // ibo_set_option("sweep_path"): 0 auto, 1 gemv, 2 mfma, 3 panel-split (IBO_SWEEP_IMPL env / tests)
std::atomic<int> g_super_min_nb{kSuperFrom};   // ibo_set_option("super_min_nb", nb): single-matrix fits from nb block columns on run in super-panels (linalg.hip: launch_cholesky_super)
std::atomic<int> g_direct_resident{0};  // ibo_set_option("direct_resident", 0/1): ibo_direct_max evaluates its batches on a resident kernel (small2.hip) instead of launches -- off: measured slower, DESIGN 4.4
std::atomic<int> g_direct_idle_ms{20};  // ibo_set_option("direct_idle_ms", n): that kernel leaves when its mailbox stays silent this long
std::atomic<int> g_nlml_groups{2};    // IBO_NLML_GROUPS=1..4 (env): a batch of theta-points runs as that many sub-batches, each on its own stream(s); values do not depend on it
// The option switches above are process-wide configuration (atomics: setting one while another thread computes is a defined,
// if unspecified-moment, change); the per-device workspaces of ibo_nlml_grid / ibo_nlml_grad and ibo_trim are serialised by
// g_dev_mu (the exp table has its own lock, held only while it is created); handles are independent of each other (own stream,
// events, buffers) -- two threads may drive two handles on one device at once.  ONE handle is for one thread at a time.
std::mutex g_dev_mu[16];
std::atomic<size_t> g_pool_limit{(size_t)2 << 30};
static std::atomic<long long> g_arena_mb{1024};                  // ibo_set_option("arena_mb", n) / env IBO_ARENA_MB: MiB per slab of the buffer arena (0: none)
    // ibo_set_option("pool_limit_mb", n) / env IBO_POOL_LIMIT_MB

int use_device(int device)
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
        if (const char *am = getenv("IBO_ARENA_MB")) { if (atoll(am) >= 0) g_arena_mb = atoll(am); }
    });
    return IBO_OK;
}

// Device allocations are recycled: a Bayesian-optimisation loop builds a new model (a new handle, five N x N
// buffers) every round, and hipMalloc/hipFree of tens of megabytes cost more than the fit itself (~60 us per MB: the driver
// maps and clears what it hands out).  Two layers:
//   * the ARENA: slabs of g_arena_mb MiB (1 GiB unless configured; env IBO_ARENA_MB, ibo_set_option("arena_mb"); 0 = none), the first
//     one taken from the device when the library allocates its first buffer there, sub-allocated first-fit with coalescing.  A request
//     of up to half a slab is served from it in microseconds whatever sizes came before -- the hallucinated model of the first
//     fastUCBGallery call, the preference GP's matrices and the kept sweep state of the first round find warm memory, not only those
//     of the second (round 5: 9 of the first gallery call's 11 extra milliseconds were hipMalloc).  Further slabs are added while the
//     arena beyond its first slab stays within g_pool_limit;
//   * the free list of whole blocks for what is larger than half a slab (the likelihood grid's gigabytes): up to g_pool_limit bytes
//     (2 GiB unless configured), handed out again to requests of up to half their size less.
// ibo_trim() empties the free list and gives back every slab but the first (the library's standing reservation on that device).
struct PoolBlock { void *p; size_t bytes; };
static std::vector<PoolBlock> g_pool[16];
static size_t g_pool_bytes[16];
static std::mutex g_pool_mu;

struct ArenaSlab {
    char *base = nullptr;
    size_t bytes = 0, used = 0;
    std::vector<std::pair<size_t, size_t>> holes;        // (offset, length), ascending offsets, never adjacent
};
static std::vector<ArenaSlab> g_arena[16];
static bool g_arena_off[16];                              // the first slab could not be had: plain allocations from then on
static const size_t kArenaGrain = 512;

static void *slab_take(ArenaSlab &sl, size_t need)
{
    for (size_t i = 0; i < sl.holes.size(); i++)
        if (sl.holes[i].second >= need) {
            void *p = sl.base + sl.holes[i].first;
            if (sl.holes[i].second == need) sl.holes.erase(sl.holes.begin() + i);
            else { sl.holes[i].first += need; sl.holes[i].second -= need; }
            sl.used += need;
            return p;
        }
    return nullptr;
}
static void slab_give(ArenaSlab &sl, size_t off, size_t len)
{
    size_t i = 0;
    while (i < sl.holes.size() && sl.holes[i].first < off) i++;
    sl.holes.insert(sl.holes.begin() + i, std::make_pair(off, len));
    if (i + 1 < sl.holes.size() && sl.holes[i].first + sl.holes[i].second == sl.holes[i + 1].first) {
        sl.holes[i].second += sl.holes[i + 1].second;
        sl.holes.erase(sl.holes.begin() + i + 1);
    }
    if (i > 0 && sl.holes[i - 1].first + sl.holes[i - 1].second == sl.holes[i].first) {
        sl.holes[i - 1].second += sl.holes[i].second;
        sl.holes.erase(sl.holes.begin() + i);
    }
    sl.used -= len;
}
static void exec_sets_prewarm(int dev);
// one kernel of every translation unit (each defines its ibo_touch_*): their code objects are loaded now, not in the middle of the first
// call that needs them
void ibo_touch_small2(); void ibo_touch_sweep(); void ibo_touch_sweep2(); void ibo_touch_linalg(); void ibo_touch_assemble(); void ibo_touch_update3();
void ibo_touch_legacy(); void ibo_touch_comm();
void ibo_touch_s2fam_FAM_SE_0(); void ibo_touch_s2fam_FAM_SE_1(); void ibo_touch_s2fam_FAM_M3_0(); void ibo_touch_s2fam_FAM_M3_1();
void ibo_touch_s2fam_FAM_M5_0(); void ibo_touch_s2fam_FAM_M5_1();
static void load_code_objects()
{
    ibo_touch_small2(); ibo_touch_sweep(); ibo_touch_sweep2(); ibo_touch_linalg(); ibo_touch_assemble(); ibo_touch_update3();
    ibo_touch_legacy(); ibo_touch_comm();
    ibo_touch_s2fam_FAM_SE_0(); ibo_touch_s2fam_FAM_SE_1(); ibo_touch_s2fam_FAM_M3_0(); ibo_touch_s2fam_FAM_M3_1();
    ibo_touch_s2fam_FAM_M5_0(); ibo_touch_s2fam_FAM_M5_1();
    (void)hipGetLastError();
}
// (g_pool_mu held) one more slab; the first brings the start-up work with it
static bool arena_add_slab(int dev, size_t slab_bytes)
{
    std::vector<ArenaSlab> &A = g_arena[dev & 15];
    ArenaSlab sl;
    if (hipMalloc((void **)&sl.base, slab_bytes) != hipSuccess) {
        (void)hipGetLastError();
        if (A.empty()) g_arena_off[dev & 15] = true;
        return false;
    }
    sl.bytes = slab_bytes;
    sl.holes.push_back(std::make_pair((size_t)0, slab_bytes));
    const bool first = A.empty();
    A.push_back(sl);
    if (first) {
        exec_sets_prewarm(dev); load_code_objects();
        const double *tab = nullptr;
        (void)exp_table(dev, &tab);                  // (the sweeps' table: its upload is the process's first pageable copy, 9 ms of runtime set-up)
    }
    return true;
}
// (g_pool_mu held)
static void *arena_get(int dev, size_t bytes, size_t *got)
{
    const size_t slab_bytes = (size_t)g_arena_mb.load() << 20;
    if (slab_bytes == 0 || g_arena_off[dev & 15]) return nullptr;
    const size_t need = (bytes + kArenaGrain - 1) / kArenaGrain * kArenaGrain;
    std::vector<ArenaSlab> &A = g_arena[dev & 15];
    if (need > slab_bytes / 2 && (A.empty() || need > A[0].bytes / 2)) return nullptr;
    for (ArenaSlab &sl : A)
        if (void *p = slab_take(sl, need)) { *got = need; return p; }
    size_t beyond = 0;
    for (size_t i = 1; i < A.size(); i++) beyond += A[i].bytes;
    if (!A.empty() && beyond + slab_bytes > g_pool_limit) return nullptr;
    if (!arena_add_slab(dev, slab_bytes)) return nullptr;
    void *p = slab_take(A.back(), need);
    *got = need;
    return p;
}
// the library's start-up on a device, once: the arena's first slab, four stream / event / staging sets, every translation unit's code object.
// Called when the first handle is created (so that it, too, finds a ready-made set) and, failing that, by the first allocation.
void pool_warm(int dev)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    const size_t slab_bytes = (size_t)g_arena_mb.load() << 20;
    if (slab_bytes == 0 || g_arena_off[dev & 15] || !g_arena[dev & 15].empty()) return;
    (void)arena_add_slab(dev, slab_bytes);
}

void *pool_get(size_t bytes, size_t *got)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (void *p = arena_get(dev, bytes, got)) return p;
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

thread_local bool g_pool_quiet = false;       // the caller has synchronised the device already (ibo_gp_destroy: once for all its buffers)
void pool_put(void *p, size_t bytes)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!g_pool_quiet) (void)hipDeviceSynchronize(); // what hipFree would have waited for: nothing in flight uses p
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (ArenaSlab &sl : g_arena[dev & 15])
        if ((char *)p >= sl.base && (char *)p < sl.base + sl.bytes) { slab_give(sl, (size_t)((char *)p - sl.base), bytes); return; }
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

// four ready-made sets beside the arena's first slab (g_pool_mu held).  Four, because what a new stream costs is the hardware queue the
// runtime creates for it -- 9 ms each for the first four streams of a process (21 for the very first), next to nothing afterwards, when new
// streams share the queues that exist: with two sets, the model a gallery call makes while two other models were alive paid those 9 ms
static void exec_sets_prewarm(int dev)
{
    std::vector<ExecSet> &v = g_exec_pool[dev & 15];
    while (v.size() < 4) {
        ExecSet x;
        hipError_t e = hipStreamCreate(&x.stream);
        if (e == hipSuccess) e = hipEventCreate(&x.ev0);
        if (e == hipSuccess) e = hipEventCreate(&x.ev1);
        if (e == hipSuccess) e = hipEventCreate(&x.fit0);
        if (e == hipSuccess) e = hipEventCreate(&x.fit1);
        if (e == hipSuccess) e = hipHostMalloc((void **)&x.pin, ((size_t)1 << 16) * sizeof(double), hipHostMallocDefault);
        if (e == hipSuccess) { x.pin_cap = (size_t)1 << 16; e = hipHostMalloc((void **)&x.done_flag, 64, hipHostMallocDefault); }
        if (e != hipSuccess) { (void)hipGetLastError(); exec_set_free(x); return; }
        *x.done_flag = 0;
        v.push_back(x);
    }
}

void pool_trim(int dev)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (PoolBlock &b : g_pool[dev & 15]) (void)hipFree(b.p);
    g_pool[dev & 15].clear();
    g_pool_bytes[dev & 15] = 0;
    std::vector<ArenaSlab> &A = g_arena[dev & 15];
    for (size_t i = A.size(); i-- > 1;)                // every slab but the first, if nothing lives in it
        if (A[i].used == 0) { (void)hipFree(A[i].base); A.erase(A.begin() + i); }
    // (four stream / event sets with small staging stay, as the arena's first slab does: they hold no device memory worth the name, and a
    // model built right after a trim -- bench.py's preference GP -- paid 2.6 ms to make them again)
    std::vector<ExecSet> &v = g_exec_pool[dev & 15];
    size_t kept = 0;
    for (size_t i = 0; i < v.size(); i++) {
        if (kept < 4 && v[i].pin_cap <= ((size_t)1 << 16)) v[kept++] = v[i];
        else exec_set_free(v[i]);
    }
    v.resize(kept);
}

// ------------------------------------------------------------------------ library
// device-side time this process has measured with HIP events (fits, block extensions, candidate sweeps, likelihood grids and
// gradients -- DIRECT's small batches and the copies are not event-timed): what a reader relates a bench line to a busy-GPU sample with
static double g_gpu_ms[16];
static std::mutex g_gpu_ms_mu;
void gpu_time_add(int device, double ms)
{
    if (!(ms > 0.0)) return;
    std::lock_guard<std::mutex> lk(g_gpu_ms_mu);
    g_gpu_ms[device & 15] += ms;
}
extern "C" int ibo_gpu_time_ms(int device, double *ms)
{
    if (!ms) return fail(IBO_ERR_ARG, "ms is NULL");
    std::lock_guard<std::mutex> lk(g_gpu_ms_mu);
    *ms = g_gpu_ms[device & 15];
    return IBO_OK;
}

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
    if (!strcmp(key, "super_min_nb")) { if (value < 2 * kSuperPanel) return fail(IBO_ERR_ARG, "super_min_nb < %d", 2 * kSuperPanel); g_super_min_nb = value; return IBO_OK; }
    if (!strcmp(key, "direct_resident")) { g_direct_resident = value != 0; return IBO_OK; }
    if (!strcmp(key, "direct_idle_ms")) { if (value < 1 || value > 1000) return fail(IBO_ERR_ARG, "direct_idle_ms outside 1..1000"); g_direct_idle_ms = value; return IBO_OK; }
    if (!strcmp(key, "arena_mb")) { if (value < 0) return fail(IBO_ERR_ARG, "arena_mb < 0"); g_arena_mb = value; return IBO_OK; }
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
uint64_t alloc_generation(int device, const void *p, size_t bytes, size_t *offset)
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
    pool_warm(device);                               // (the first handle of the process on this device: the library's start-up, see pool_warm)
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
    g->done_count.release(); g->srv_ctl.release(); g->tall.release(); g->Pk2.release();
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

// host-in / host-out evaluation of M points: values of one acquisition (or the
// posterior) -- used by posterior_batch and by DIRECT's batches
int ensure_pinned(ibo_gp *g, size_t need)
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

