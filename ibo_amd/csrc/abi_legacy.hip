// abi_legacy.hip -- libego's own symbols (acqmaxGP, direct, logCDFs) with the reference's exact signatures, and ibo_direct_host.
#include "abi_internal.h"

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
    if (ibo_gp_create(dev, &g) != IBO_OK) { fprintf(stderr, "[libibo_hip] %s\n", ibo_last_error()); return NULL; }
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
        if (rc != IBO_OK) fprintf(stderr, "[libibo_hip] acqmaxGP failed: %s\n", ibo_last_error());
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
    if (rc != IBO_OK) fprintf(stderr, "[libibo_hip] acqmaxGP failed: %s\n", ibo_last_error());
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

