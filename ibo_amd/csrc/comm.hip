// comm.hip -- the one collective of the sharded candidate sweep: an arg-max
// exchange over RCCL (xGMI).  RCCL has no MAXLOC, so every rank writes
// (value, index, payload...) into its own slot of a world_size-slot buffer that
// is zero elsewhere and a single ncclAllReduce(sum) gathers all slots; the final
// (max value, lowest global index) reduction is then done identically on every
// rank.  16..(2+D)*8 bytes per rank: latency-bound, one collective per sweep.
//
// The sweep's result never visits the host on its way into the exchange (round 6): the arg-max kernel's (value, index) stay in
// HBM, slot_fill_kernel writes them and the winning candidate's coordinates into this rank's slot of the (zeroed) all-reduce
// buffer on the sweep's own stream, ncclAllReduce follows on that stream, and ONE copy brings world x (3 + D) doubles into
// pinned memory (ibo_comm_exchange_dev, behind ibo_acq_sweep_exchange).  The host-value entry points (ibo_comm_argmax: barriers,
// max-over-ranks timings; ibo_comm_allreduce_sum: the likelihood grid's gather) stage through the same pinned buffer.
//
// librccl is loaded lazily (dlopen) so single-GPU users and CPU-side symbol
// checks never pay for it.
#include "../../include/ibo_abi.h"
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <cmath>

// minimal RCCL surface (matches /opt/rocm/include/rccl/rccl.h)
typedef struct { char internal[128]; } rccl_unique_id_t;
typedef void *rccl_comm_t;
typedef int (*fn_get_unique_id)(rccl_unique_id_t *);
typedef int (*fn_comm_init_rank)(rccl_comm_t *, int, rccl_unique_id_t, int);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int /*dtype*/, int /*op*/, rccl_comm_t, hipStream_t);
typedef int (*fn_comm_destroy)(rccl_comm_t);
typedef const char *(*fn_get_error_string)(int);
typedef int (*fn_comm_count)(rccl_comm_t, int *);
enum { RCCL_FLOAT64 = 8, RCCL_SUM = 0 };

static struct {
    void *h = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_get_error_string err = nullptr;
    fn_comm_count comm_count = nullptr;
} R;

void ibo_internal_set_error(const char *msg);     // abi.hip: the one buffer behind ibo_last_error()
static int cfail(int code, const char *msg, int rc = 0)
{
    char c_err[256];
    snprintf(c_err, sizeof(c_err), "comm: %s (%s)", msg, (R.err && rc) ? R.err(rc) : "-");
    ibo_internal_set_error(c_err);
    fprintf(stderr, "[libibo_hip] %s\n", c_err);
    return code;
}

static int load_rccl()
{
    if (R.h) return IBO_OK;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *n : names) { R.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (R.h) break; }
    if (!R.h) return cfail(IBO_ERR_COMM, "cannot dlopen librccl.so");
    R.get_unique_id = (fn_get_unique_id)dlsym(R.h, "ncclGetUniqueId");
    R.comm_init_rank = (fn_comm_init_rank)dlsym(R.h, "ncclCommInitRank");
    R.all_reduce = (fn_all_reduce)dlsym(R.h, "ncclAllReduce");
    R.comm_destroy = (fn_comm_destroy)dlsym(R.h, "ncclCommDestroy");
    R.err = (fn_get_error_string)dlsym(R.h, "ncclGetErrorString");
    R.comm_count = (fn_comm_count)dlsym(R.h, "ncclCommCount");
    if (!R.get_unique_id || !R.comm_init_rank || !R.all_reduce || !R.comm_destroy)
        return cfail(IBO_ERR_COMM, "librccl.so lacks an expected symbol");
    return IBO_OK;
}

struct ibo_comm {
    int device, world, rank;
    rccl_comm_t comm;
    hipStream_t stream;
    double *dbuf;   // the all-reduce buffer (device)
    double *hpin;   // its pinned host mirror: every copy in either direction is truly asynchronous, nothing is allocated per call
    size_t cap;     // doubles, both
};

static int comm_reserve(ibo_comm *c, size_t n)
{
    if (n <= c->cap) return IBO_OK;
    size_t cap = n < 1024 ? 1024 : n;
    if (c->dbuf) (void)hipFree(c->dbuf);
    if (c->hpin) (void)hipHostFree(c->hpin);
    c->dbuf = nullptr; c->hpin = nullptr; c->cap = 0;
    if (hipMalloc((void **)&c->dbuf, cap * sizeof(double)) != hipSuccess) return cfail(IBO_ERR_HIP, "hipMalloc failed");
    if (hipHostMalloc((void **)&c->hpin, cap * sizeof(double), hipHostMallocDefault) != hipSuccess) return cfail(IBO_ERR_HIP, "hipHostMalloc failed");
    c->cap = cap;
    return IBO_OK;
}

extern "C" int ibo_comm_get_unique_id(unsigned char id[IBO_COMM_ID_BYTES])
{
    if (!id) return IBO_ERR_ARG;
    int rc = load_rccl();
    if (rc) return rc;
    rccl_unique_id_t u;
    int e = R.get_unique_id(&u);
    if (e) return cfail(IBO_ERR_COMM, "ncclGetUniqueId failed", e);
    memcpy(id, u.internal, IBO_COMM_ID_BYTES);
    return IBO_OK;
}

extern "C" int ibo_comm_init(int device, int world_size, int rank, const unsigned char id[IBO_COMM_ID_BYTES],
                             ibo_comm_t **out)
{
    if (!out || !id || world_size < 1 || rank < 0 || rank >= world_size) return IBO_ERR_ARG;
    int rc = load_rccl();
    if (rc) return rc;
    {
        int ndev = -1;
        hipError_t e1 = hipGetDeviceCount(&ndev);
        hipError_t e2 = hipSetDevice(device);
        if (e2 != hipSuccess) {
            fprintf(stderr, "[libibo_hip] comm: hipGetDeviceCount -> %s (%d devices); hipSetDevice(%d) -> %s\n",
                    hipGetErrorString(e1), ndev, device, hipGetErrorString(e2));
            return cfail(IBO_ERR_HIP, "hipSetDevice failed");
        }
    }
    ibo_comm *c = new ibo_comm();
    c->device = device; c->world = world_size; c->rank = rank; c->dbuf = nullptr; c->hpin = nullptr; c->cap = 0;
    rccl_unique_id_t u;
    memcpy(u.internal, id, IBO_COMM_ID_BYTES);
    int e = R.comm_init_rank(&c->comm, world_size, u, rank);
    if (e) { delete c; return cfail(IBO_ERR_COMM, "ncclCommInitRank failed", e); }
    if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return cfail(IBO_ERR_HIP, "hipStreamCreate failed"); }
    if (comm_reserve(c, (size_t)world_size * (3 + 64)) != IBO_OK) { (void)hipStreamDestroy(c->stream); delete c; return IBO_ERR_HIP; }
    *out = c;
    return IBO_OK;
}

extern "C" int ibo_comm_destroy(ibo_comm_t *c)
{
    if (!c) return IBO_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->dbuf) (void)hipFree(c->dbuf);
    if (c->hpin) (void)hipHostFree(c->hpin);
    R.comm_destroy(c->comm);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return IBO_OK;
}

// number of ranks RCCL itself reports for the communicator (ncclCommCount): bench.py prints it so a reader
// can see the collective really spanned N processes
extern "C" int ibo_comm_count(ibo_comm_t *c, int *nranks)
{
    if (!c || !nranks) return IBO_ERR_ARG;
    if (!R.comm_count) return cfail(IBO_ERR_COMM, "librccl.so lacks ncclCommCount");
    int e = R.comm_count(c->comm, nranks);
    if (e) return cfail(IBO_ERR_COMM, "ncclCommCount failed", e);
    return IBO_OK;
}

// the deterministic final reduction, shared with the host-side (gloo) tests
// through ibo_amd/multigpu.py which restates it in three lines
static void slot_argmax(const double *buf, int world, int slot, double *bv, int64_t *bi, int *br)
{
    double v = 0.0; int64_t i = -1; int r = -1;
    for (int k = 0; k < world; k++) {
        const double *s = buf + (size_t)k * slot;
        if (s[2] == 0.0) continue;                     // rank had no admissible candidate
        int64_t idx = (int64_t)s[1];
        if (r < 0 || s[0] > v || (s[0] == v && idx < i)) { v = s[0]; i = idx; r = k; }
    }
    *bv = v; *bi = i; *br = r;
}

extern "C" int ibo_comm_argmax(ibo_comm_t *c, double val, int64_t idx, const double *payload, int npayload,
                               double *best_val, int64_t *best_idx, double *best_payload, int *best_rank)
{
    if (!c || npayload < 0 || (npayload && !payload)) return IBO_ERR_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return cfail(IBO_ERR_HIP, "hipSetDevice failed");
    const int slot = 3 + npayload;          // value, index (exact in fp64 below 2^53), valid flag, payload
    const size_t n = (size_t)slot * c->world;
    if (int rc = comm_reserve(c, n)) return rc;
    double *h = c->hpin;
    memset(h, 0, n * sizeof(double));
    double *mine = h + (size_t)c->rank * slot;
    bool valid = idx >= 0 && val == val;
    mine[0] = valid ? val : 0.0; mine[1] = valid ? (double)idx : 0.0; mine[2] = valid ? 1.0 : 0.0;
    for (int k = 0; k < npayload; k++) mine[3 + k] = valid ? payload[k] : 0.0;
    if (hipMemcpyAsync(c->dbuf, h, n * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess)
        return cfail(IBO_ERR_HIP, "H2D failed");
    int e = R.all_reduce(c->dbuf, c->dbuf, n, RCCL_FLOAT64, RCCL_SUM, c->comm, c->stream);
    if (e) return cfail(IBO_ERR_COMM, "ncclAllReduce failed", e);
    if (hipMemcpyAsync(h, c->dbuf, n * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess)
        return cfail(IBO_ERR_HIP, "D2H failed");
    if (hipStreamSynchronize(c->stream) != hipSuccess) return cfail(IBO_ERR_HIP, "stream sync failed");
    double bv; int64_t bi; int br;
    slot_argmax(h, c->world, slot, &bv, &bi, &br);
    if (best_val) *best_val = bv;
    if (best_idx) *best_idx = bi;
    if (best_rank) *best_rank = br;
    if (best_payload && br >= 0)
        for (int k = 0; k < npayload; k++) best_payload[k] = h[(size_t)br * slot + 3 + k];
    return IBO_OK;
}

// this rank's slot of the exchange, filled where the sweep left its result: (value, global index) from the arg-max kernel's output
// words, the winner's D coordinates from the candidate array (row index - index_base); every other slot zero
__global__ void slot_fill_kernel(double *__restrict__ buf, int world, int slot, int rank, const double *__restrict__ res_v,
                                 const int64_t *__restrict__ res_i, const double *__restrict__ cand, int D, int64_t index_base)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= world * slot) return;
    const int r = e / slot, k = e - r * slot;
    double v = 0.0;
    if (r == rank) {
        const double val = res_v[0];
        const int64_t idx = res_i[0];
        const bool valid = idx >= 0 && val == val;
        if (valid) v = k == 0 ? val : (k == 1 ? (double)idx : (k == 2 ? 1.0 : cand[(size_t)(idx - index_base) * D + (k - 3)]));
    }
    buf[e] = v;
}

// The exchange behind ibo_acq_sweep_exchange (abi_sweep.hip): everything on the sweep's stream `s`, one synchronisation.
// local_*: this rank's own (value, index) as the sweep found them (index -1: no admissible candidate).
int ibo_comm_exchange_dev(ibo_comm *c, hipStream_t s, const double *res_v, const int64_t *res_i, const double *cand_dev, int D,
                          int64_t index_base, double *local_val, int64_t *local_idx, double *best_val, int64_t *best_idx,
                          double *best_x, int *best_rank)
{
    if (!c || !res_v || !res_i || !cand_dev || D < 1) return IBO_ERR_ARG;
    const int slot = 3 + D;
    const size_t n = (size_t)slot * c->world;
    if (int rc = comm_reserve(c, n)) return rc;
    hipLaunchKernelGGL(slot_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c->dbuf, c->world, slot, c->rank, res_v, res_i,
                       cand_dev, D, index_base);
    if (hipGetLastError() != hipSuccess) return cfail(IBO_ERR_HIP, "slot_fill_kernel launch failed");
    int e = R.all_reduce(c->dbuf, c->dbuf, n, RCCL_FLOAT64, RCCL_SUM, c->comm, s);
    if (e) return cfail(IBO_ERR_COMM, "ncclAllReduce failed", e);
    if (hipMemcpyAsync(c->hpin, c->dbuf, n * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return cfail(IBO_ERR_HIP, "D2H failed");
    if (hipStreamSynchronize(s) != hipSuccess) return cfail(IBO_ERR_HIP, "stream sync failed");
    const double *h = c->hpin, *mine = h + (size_t)c->rank * slot;
    if (local_val) *local_val = mine[2] != 0.0 ? mine[0] : -INFINITY;
    if (local_idx) *local_idx = mine[2] != 0.0 ? (int64_t)mine[1] : -1;
    double bv; int64_t bi; int br;
    slot_argmax(h, c->world, slot, &bv, &bi, &br);
    if (best_val) *best_val = bv;
    if (best_idx) *best_idx = bi;
    if (best_rank) *best_rank = br;
    if (best_x) for (int k = 0; k < D; k++) best_x[k] = br >= 0 ? h[(size_t)br * slot + 3 + k] : 0.0;
    return IBO_OK;
}

// in-place sum all-reduce of a host buffer (staged through HBM): the gather of the sharded
// marginal-likelihood grid (each rank fills its own theta slots, zeros elsewhere)
extern "C" int ibo_comm_allreduce_sum(ibo_comm_t *c, double *host_buf, int64_t n)
{
    if (!c || !host_buf || n < 1) return IBO_ERR_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return cfail(IBO_ERR_HIP, "hipSetDevice failed");
    if (int rc = comm_reserve(c, (size_t)n)) return rc;
    memcpy(c->hpin, host_buf, (size_t)n * sizeof(double));
    if (hipMemcpyAsync(c->dbuf, c->hpin, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess)
        return cfail(IBO_ERR_HIP, "H2D failed");
    int e = R.all_reduce(c->dbuf, c->dbuf, (size_t)n, RCCL_FLOAT64, RCCL_SUM, c->comm, c->stream);
    if (e) return cfail(IBO_ERR_COMM, "ncclAllReduce failed", e);
    if (hipMemcpyAsync(c->hpin, c->dbuf, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess)
        return cfail(IBO_ERR_HIP, "D2H failed");
    if (hipStreamSynchronize(c->stream) != hipSuccess) return cfail(IBO_ERR_HIP, "stream sync failed");
    memcpy(host_buf, c->hpin, (size_t)n * sizeof(double));
    return IBO_OK;
}

extern "C" int ibo_comm_barrier(ibo_comm_t *c)
{
    double v; int64_t i; int r;
    return ibo_comm_argmax(c, 0.0, 0, nullptr, 0, &v, &i, nullptr, &r);
}

void ibo_touch_comm() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, (const void *)slot_fill_kernel); }     // (see small2.hip: ibo_touch_small2)
