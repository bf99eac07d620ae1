// abi_fit.hip -- the model side of the C ABI: fit (covariance, factorisation route, W = L^-1, alpha vectors), the block extension,
// the preference GP's device steps, accessors, ibo_cov_matrix and the ibo_spd_* helpers.
#include "abi_internal.h"

int make_kparams(int ktype, int D, const double *hyper, int nhyper, double sf2, KParams *kp)
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
    // from g_super_min_nb block columns on the single-level order runs in super-panels: the matrix and the ride-along's identity are the two
    // halves of ONE tall buffer (launch_cholesky_super), T and W are outputs only
    const bool super = super_order(Np);
    if (super) {
        IBO_TRY(g->tall.ensure(2 * (size_t)Np * Np)); IBO_TRY(g->Pk2.ensure(2 * (size_t)Np * Np));
    }
    double *work = super ? g->tall.p : g->T.p;                   // T is free until launch_trinv uses it as scratch
    double *eye = super ? g->tall.p + (size_t)Np * Np : g->W.p;
    // (with the working copy the same pass writes the identity the ride-along starts from and clears the info word)
    const bool one_pass = fused && !A_host;
    // (GP.R itself is not written here: 33 MB of stores at N = 2048 that only ibo_gp_get_R and ibo_pref_finish read -- ensure_R;
    // stage_data marked it stale)
    if (!A_host)
        KERNEL_TRY(launch_cov_fit(kp, N, g->Xp.p, g->DP, IBO_DIAG_UNIT_PLUS_NOISE, noise, work, Np, one_pass ? eye : nullptr, g->info.p, s));
    else {
        HIP_TRY(hipMemsetAsync(g->info.p, 0, sizeof(int), s));
        KERNEL_TRY(launch_pad_copy(g->A.p, N, N, work, Np, 1.0, s));
    }
    if (fused) {
        // the plain right-looking order: one launch per block column, out of place, with W = L^-1 riding along (E = I in W's buffer
        // turns into (L^-1)^T in Wp's, which is transposed into W and packed into T's buffer -- free by then -- in one pass; T and
        // Wp then trade places)
        if (!one_pass) KERNEL_TRY(launch_pad_copy(g->Xp.p, 0, 1, eye, Np, 1.0, s));       // identity
        if (super) KERNEL_TRY(launch_cholesky_super(g->tall.p, g->L.p, Np, g->diag64.p, g->info.p, s, g->Wp.p, g->Pk2.p, true));
        else KERNEL_TRY(launch_cholesky_fused(g->T.p, g->L.p, Np, g->diag64.p, g->info.p, s, g->W.p, g->Wp.p, true));
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
    gpu_time_add(g->device, g->fit_ms);
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
    gpu_time_add(g->device, g->fit_ms);
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
int fit_from_inverse(ibo_gp *g, int ktype, int N, int D, const double *X, const double *Y,
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
    gpu_time_add(g->device, g->fit_ms);
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
