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

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <sched.h>
#include <thread>
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

// ---- host side ---------------------------------------------------------------------------------------------------------------
// What is shared with cpp/optimizeGP.cpp:67-236 is the ROUNDING SEQUENCE of every number, nothing else: each double below goes
// through the same IEEE operations in the same order as the reference's (the bit-for-bit tests against the compiled reference are
// the guard), under -ffp-contract=off, with the host's libm for exp / sqrt / erf.  Quantities that do not depend on the sample
// point (1 / h_d^2, sqrt 3, sqrt 5, 3 h^2) are formed once per call -- a rounded operation gives the same double whenever it runs.
namespace {

// one kernel family = how a coordinate difference enters the running sum, and what the sum becomes
template <int FAMILY> struct Radial;
template <> struct Radial<0> {                       // squared exponential, one length scale per dimension: sum w_d (d_d^2), w_d = 1 / h_d^2
    static double step(double sum, double diff, double w) { const double d2 = diff * diff; const double t = w * d2; return sum + t; }
    static double shape(double sum, double amp, const LegacyHost::Consts &) { return amp * exp(-.5 * sum); }
};
template <> struct Radial<1> {                       // squared exponential, one length scale: sum (d_d / h)^2
    static double step(double sum, double diff, double h) { const double q = diff / h; return sum + q * q; }
    static double shape(double sum, double amp, const LegacyHost::Consts &) { return amp * exp(-.5 * sum); }
};
template <> struct Radial<2> {                       // Matern-3/2 in s = sqrt 3 * sqrt(sum (d_d / h)^2)
    static double step(double sum, double diff, double h) { return Radial<1>::step(sum, diff, h); }
    static double shape(double sum, double amp, const LegacyHost::Consts &c)
    {
        const double s = c.root3 * sqrt(sum);
        return amp * (1.0 + s) * exp(-s);
    }
};
template <> struct Radial<3> {                       // Matern-5/2 in the unscaled distance: amp (1 + a + 5 r^2 / (3 h^2)) e^-a, a = sqrt 5 r / h
    static double step(double sum, double diff, double) { return sum + diff * diff; }
    static double shape(double sum, double amp, const LegacyHost::Consts &c)
    {
        const double dist = sqrt(sum);
        const double a = c.root5 * dist / c.h;
        const double quad = 5.0 * dist * dist / c.three_h2;
        return amp * (1.0 + a + quad) * exp(-a);
    }
};

template <int FAMILY>
void kstar_rows(const LegacyHost::Consts &c, double amp, int dim, const double *obs, int row0, int row1, const double *pt, double *out)
{
    for (int row = row0; row < row1; row++) {
        const double *o = obs + (size_t)row * dim;
        double sum = 0;
        for (int d = 0; d < dim; d++) sum = Radial<FAMILY>::step(sum, o[d] - pt[d], c.per_dim[FAMILY == 0 ? d : 0]);
        out[row] = Radial<FAMILY>::shape(sum, amp, c);
    }
}

// the cores this process may actually use: the affinity mask cut down to the cgroup's CPU quota (a GPU box shows 256 CPUs and grants 16)
int usable_cores()
{
    if (const char *e = getenv("IBO_HOST_THREADS")) { const int n = atoi(e); if (n >= 1) return n > 64 ? 64 : n; }
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        long long quota = 0, period = 0;
        if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) { const int q = (int)((quota + period - 1) / period); if (q < n) n = q; }
        fclose(f);
    }
    return n < 1 ? 1 : (n > 16 ? 16 : n);
}

}   // namespace

// A crew of host threads that lives as long as one acqmaxGP call: run(count, fn) hands the items 0 .. count-1 to whoever is free
// (the caller included) and returns when all are done.  Every item writes its own outputs, so the result does not depend on who ran what.
struct LegacyHost::Crew {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable wake, done;
    const std::function<void(int)> *job = nullptr;
    std::atomic<int> next{0};
    int count = 0, round = 0, working = 0;
    bool stop = false;

    explicit Crew(int helpers)
    {
        for (int t = 0; t < helpers; t++) threads.emplace_back([this] { loop(); });
    }
    ~Crew()
    {
        { std::lock_guard<std::mutex> l(mu); stop = true; }
        wake.notify_all();
        for (auto &t : threads) t.join();
    }
    void drain()
    {
        for (int i = next.fetch_add(1); i < count; i = next.fetch_add(1)) (*job)(i);
    }
    void loop()
    {
        int seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(mu);
                wake.wait(l, [&] { return stop || round != seen; });
                if (stop) return;
                seen = round;
            }
            drain();
            { std::lock_guard<std::mutex> l(mu); if (--working == 0) done.notify_one(); }
        }
    }
    void run(int n, const std::function<void(int)> &fn)
    {
        if (threads.empty() || n < 2) { for (int i = 0; i < n; i++) fn(i); return; }
        {
            std::lock_guard<std::mutex> l(mu);
            job = &fn; count = n; next = 0; working = (int)threads.size(); round++;
        }
        wake.notify_all();
        drain();
        std::unique_lock<std::mutex> l(mu);
        done.wait(l, [&] { return working == 0; });
    }
};

LegacyHost::LegacyHost(const LegacySpec &spec) : m(spec)
{
    k.per_dim.assign(m.dim, 0.0);
    if (m.family == 0) for (int d = 0; d < m.dim; d++) k.per_dim[d] = 1 / (m.hyper[d] * m.hyper[d]);      // 1 / pow(h, 2): powi(h, 2) = h * h
    else k.per_dim[0] = m.hyper[0];
    k.h = m.hyper[0];
    k.root3 = sqrt(3.0);
    k.root5 = sqrt(5.0);
    k.three_h2 = 3.0 * k.h * k.h;
    best = m.targets[0];
    for (int i = 0; i < m.rows; i++) if (m.targets[i] > best) best = m.targets[i];
    root2 = sqrt(2.);
    root2pi = sqrt(2. * M_PI);
    crew = new Crew(usable_cores() - 1);
}

LegacyHost::~LegacyHost() { delete crew; }

int LegacyHost::threads() const { return (int)crew->threads.size() + 1; }

// the RBF network's value at one point: sum_b beta_b exp(-theta |(pt - origin) / extent - centre_b|^2)
double LegacyHost::prior_at(const double *pt) const
{
    double total = 0.0;
    for (int b = 0; b < m.nbasis; b++) {
        const double *centre = m.centres + (size_t)b * m.dim;
        double dist2 = 0;
        for (int d = 0; d < m.dim; d++) { const double u = (pt[d] - m.origin[d]) / m.extent[d] - centre[d]; dist2 = dist2 + u * u; }
        total = total + m.weights[b] * exp(-m.sharpness * dist2);
    }
    return total;
}

// For n sample points: vecs[p] = k*(pt_p) (n vectors of `rows`), and under a mean prior prior_mu[p] and vecs[n + p] = targets - prior_mu[p].
// The work is cut into (point, 512-row slice) items for the crew.
void LegacyHost::prepare(const double *pts, int n, double *vecs, double *prior_mu) const
{
    const int slice = 512, per_pt = (m.rows + slice - 1) / slice;
    const bool prior = m.nbasis > 0;
    const std::function<void(int)> item = [&](int it) {
        const int p = it / per_pt, r0 = (it % per_pt) * slice, r1 = r0 + slice < m.rows ? r0 + slice : m.rows;
        const double *pt = pts + (size_t)p * m.dim;
        double *out = vecs + (size_t)p * m.rows;
        switch (m.family) {
        case 0: kstar_rows<0>(k, m.amp, m.dim, m.obs, r0, r1, pt, out); break;
        case 1: kstar_rows<1>(k, m.amp, m.dim, m.obs, r0, r1, pt, out); break;
        case 2: kstar_rows<2>(k, m.amp, m.dim, m.obs, r0, r1, pt, out); break;
        default: kstar_rows<3>(k, m.amp, m.dim, m.obs, r0, r1, pt, out); break;
        }
    };
    if (prior) {
        const std::function<void(int)> means = [&](int p) { prior_mu[p] = prior_at(pts + (size_t)p * m.dim); };
        crew->run(n, means);
        const std::function<void(int)> both = [&](int it) {
            item(it);
            const int p = it / per_pt, r0 = (it % per_pt) * slice, r1 = r0 + slice < m.rows ? r0 + slice : m.rows;
            double *resid = vecs + (size_t)(n + p) * m.rows;
            for (int i = r0; i < r1; i++) resid[i] = m.targets[i] - prior_mu[p];
        };
        crew->run(n * per_pt, both);
    } else {
        for (int p = 0; p < n; p++) prior_mu[p] = 0.0;       // (+0.0 added to the first contraction: exact)
        crew->run(n * per_pt, item);
    }
}

// the NEGATED acquisition from the two contractions: mean = prior_mu + c_mean, variance = clamp(1 + noise - c_var, 1e-8, 10)
double LegacyHost::negated(double prior_mu, double c_mean, double c_var) const
{
    const double mean = prior_mu + c_mean;
    double var = 1. + m.noise - c_var;
    var = var < 1e-8 ? 1e-8 : (var > 10. ? 10. : var);
    const double sd = sqrt(var);
    if (m.acq == 2) return -(mean + m.parm * sd);
    const double gain = mean - best - m.parm;
    const double u = gain / sd;
    const double Phi = 0.5 * (1. + erf(u / root2));
    if (m.acq == 1) return -Phi;
    const double phi = exp(-(u * u / 2.)) / root2pi;
    return -(gain * Phi + sd * phi);
}

void ibo_touch_legacy() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, (const void *)legacy_transpose_kernel); }     // (see small2.hip: ibo_touch_small2)
