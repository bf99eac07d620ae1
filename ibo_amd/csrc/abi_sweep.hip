// abi_sweep.hip -- the evaluation side of the C ABI: candidate sweeps (one-shot and with kept per-candidate state), host batches,
// and DIRECT on the GPU objective.
#include "abi_internal.h"

// 2^(j/2048), j < 2048: the table behind sweep2's exp (one per device, created on first use)
static std::atomic<double *> g_exp_tab[16];
static std::mutex g_exp_mu;                          // held only while a device's table is being created (never across a grid or a gradient)
int exp_table(int device, const double **out)
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
                     const double *cand_host = nullptr, bool device_result = false)
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
    const bool want_best = best_val || best_idx || device_result;      // device_result: (value, index) stay in res_v / res_i for the exchange
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
    if (!best_val && !best_idx) return IBO_OK;        // internal callers that only want the per-point outputs (or the result on the device)
    double hv; int64_t hi;
    HIP_TRY(hipMemcpyAsync(&hv, g->res_v.p, sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&hi, g->res_i.p, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipEventElapsedTime(&g->sweep_ms, g->ev0, g->ev1));
    gpu_time_add(g->device, g->sweep_ms);
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

// The sharded sweep's step in one call (the loop of ego/acquisition/gallery.py:93-134 cut over ranks): this rank's block is
// swept, the arg-max kernel's (value, global index) stay in HBM, a small kernel puts them and the winner's coordinates into the rank's
// slot of the all-reduce buffer, ncclAllReduce runs on the same stream and one copy brings every rank's slot to pinned host memory
// (csrc/comm.hip: ibo_comm_exchange_dev) -- one synchronisation per step, nothing staged through pageable memory.
extern "C" int ibo_acq_sweep_exchange(ibo_gp_t *g, ibo_comm_t *c, int incremental, int64_t M, const double *cand_dev, int acq, double parm,
                                      int erf_mode, double clamp_lo, double ymax, int n_excl, const double *excl_host, double excl_radius,
                                      int64_t index_base, double *local_val, int64_t *local_idx, double *best_val, int64_t *best_idx,
                                      double *best_x, int *best_rank)
{
    if (!g || !c) return fail(IBO_ERR_ARG, "NULL argument");
    IBO_TRY(use_device(g->device));
    IBO_TRY(run_sweep(g, M, cand_dev, acq, parm, erf_mode, clamp_lo, ymax, n_excl, excl_host, excl_radius, index_base, nullptr, nullptr, nullptr,
                      nullptr, nullptr, incremental != 0, true, false, nullptr, true));
    IBO_TRY(ibo_comm_exchange_dev(c, g->stream, g->res_v.p, g->res_i.p, cand_dev, g->D, index_base, local_val, local_idx, best_val, best_idx,
                                  best_x, best_rank));
    HIP_TRY(hipEventElapsedTime(&g->sweep_ms, g->ev0, g->ev1));      // (the exchange has synchronised the stream)
    gpu_time_add(g->device, g->sweep_ms);
    return IBO_OK;
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
// The resident evaluation server's host side (small2.hip: direct_server_kernel): started for the lifetime of one direct_on_gp call where
// the model and the acquisition admit it, fed through a mailbox in the handle's pinned staging, and -- whenever it is not there (not
// started, not resident in time, left on a deadline, a batch larger than its mailbox) -- replaced by the launches of eval_host_points:
// the same values bit for bit, so the search cannot tell.  Every host-side wait is bounded.
static double mono_us()
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}
struct DirectServer {
    ibo_gp *g = nullptr;
    bool active = false;
    int Mmax = 0, D = 0;
    unsigned long long seq = 0;
    volatile unsigned long long *bx = nullptr;      // the mailbox's words: [0] seq, [1] M, [2] done, [3] state
    double *cand = nullptr, *vals = nullptr;
    int batches = 0;
    const char *why = "";
    unsigned long long *stamps = nullptr;
    int stall_after = -1;
    double t_post[64], t_seen[64];                  // host clock (us) when batch b was posted / seen finished (diagnostics)

    int start(ibo_gp *gp, int acq, double parm, int erf_mode, double clamp_lo)
    {
        g = gp; D = g->D;
        if (!g_direct_resident.load()) { why = "switched off"; return IBO_OK; }
        if (g_force_path != 0 || !sweep2_fits(g->Npad)) { why = "model outside the small-batch kernels"; return IBO_OK; }
        SweepArgs a;
        memset(&a, 0, sizeof(a));
        a.kp = g->kp; a.N = g->N; a.Npad = g->Npad; a.DP = g->DP; a.M = 1;
        a.Xs = g->Xs.p; a.ak = g->ak.p; a.XA = g->XA.p; a.log_sf2 = log(g->kp.sf2);
        a.dot_form = (g_dot_override >= 0 && g->D <= IBO_DDOT) ? g_dot_override.load() : g->dot_form;
        a.Xp = g->Xp.p; a.W = g->W.p; a.Wp = g->Wp.p; a.alphaY = g->alphaY.p; a.alpha1 = g->alpha1.p;
        a.prior.nb = g->nb; a.prior.theta = g->ptheta; a.prior.means = g->pmeans.p; a.prior.beta = g->pbeta.p;
        a.prior.lowerb = g->plowerb.p; a.prior.width = g->pwidth.p;
        a.noise = g->noise; a.clamp_lo = clamp_lo; a.ymax = g->maxY; a.parm = parm; a.acq = acq; a.erf_mode = erf_mode;
        if (!direct_server_takes(a)) { why = "dimension or form outside the server's instantiations"; return IBO_OK; }
        Mmax = 2048;
        const size_t box_doubles = IBO_SRV_BOX_CAND + (size_t)Mmax * D + Mmax;
        IBO_TRY(ensure_pinned(g, box_doubles));
        IBO_TRY(exp_table(g->device, &a.exp_tab));
        IBO_TRY(g->small_ws.ensure(small_sweep_workspace(g->Npad, Mmax)));
        const bool stamping = getenv("IBO_SRV_STAMPS") != nullptr;
        if (const char *sa = getenv("IBO_SRV_STALL_AFTER")) stall_after = atoi(sa);
        IBO_TRY(g->srv_ctl.ensure(IBO_SRV_CTL_BYTES / sizeof(unsigned) + (stamping ? 64 * 2 * 8 * 2 : 0)));
        bx = (volatile unsigned long long *)g->pin;
        cand = g->pin + IBO_SRV_BOX_CAND; vals = cand + (size_t)Mmax * D;
        bx[0] = 0; bx[1] = 0; bx[2] = 0; bx[3] = 0;
        a.cand = cand; a.out_acq = vals;
        hipStream_t st = g->stream;
        HIP_TRY(hipMemsetAsync(g->srv_ctl.p, 0, IBO_SRV_CTL_BYTES + (stamping ? 64 * 2 * 8 * 8 : 0), st));
        stamps = stamping ? (unsigned long long *)(g->srv_ctl.p + IBO_SRV_CTL_BYTES / sizeof(unsigned)) : nullptr;
        int ncu = 0;
        HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, g->device));
        if (ncu < 8) { why = "device too small"; return IBO_OK; }
        KERNEL_TRY(launch_direct_server(a, g->srv_ctl.p, g->pin, g->small_ws.p, Mmax, ncu, g_direct_idle_ms.load(), st, stamps));
        // resident?  (bounded: ~2 ms on the device side, 20 ms here -- the kernel may sit behind other work in the queue)
        const double t0 = mono_us();
        while (bx[3] == 0 && mono_us() - t0 < 20000.0) {}
        if (bx[3] != IBO_SRV_READY) {
            why = bx[3] == IBO_SRV_NOT_RESIDENT ? "not every workgroup became resident" : "no answer from the device";
            __atomic_store_n((unsigned long long *)&bx[0], IBO_SRV_EXIT, __ATOMIC_RELEASE);
            HIP_TRY(hipStreamSynchronize(st));
            return IBO_OK;
        }
        active = true; seq = 0;
        return IBO_OK;
    }
    // 0: done by the server; 1: not taken (the caller runs the batch by launches); else an error code
    int eval(const double *pts, int n, double *out)
    {
        if (!active) return 1;
        if (n > Mmax) { IBO_TRY(stop()); why = "a batch beyond the mailbox"; return 1; }
        if (stall_after >= 0 && batches == stall_after) {
            // (diagnostics, env IBO_SRV_STALL_AFTER=k: the host stops feeding after k batches for longer than the kernel's idle deadline --
            // what a dead caller looks like from the device; tests/test_gpu_parity.py checks that the kernel has left and the call still ends right)
            struct timespec ts = {0, (long)(g_direct_idle_ms.load() + 15) * 1000000L};
            nanosleep(&ts, nullptr);
        }
        memcpy(cand, pts, sizeof(double) * (size_t)n * D);
        bx[1] = (unsigned long long)n;
        ++seq;
        __atomic_store_n((unsigned long long *)&bx[0], seq, __ATOMIC_RELEASE);
        const double t0 = mono_us();
        if (stamps && seq <= 64) t_post[seq - 1] = t0;
        for (int spin = 0;; spin++) {
            if (bx[2] == seq) break;
            if ((spin & 255) == 255 && (bx[3] != IBO_SRV_READY || mono_us() - t0 > 100000.0)) {
                // the kernel left (its own deadline, an abort) or does not answer: wait it out -- it is bounded -- and go on by launches
                active = false;
                why = bx[3] != IBO_SRV_READY ? "the server left on a deadline" : "no answer to a batch within 100 ms";
                __atomic_store_n((unsigned long long *)&bx[0], IBO_SRV_EXIT, __ATOMIC_RELEASE);
                HIP_TRY(hipStreamSynchronize(g->stream));
                if (bx[2] == seq) break;                 // (it did answer in the end)
                return 1;
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        if (stamps && seq <= 64) t_seen[seq - 1] = mono_us();
        memcpy(out, vals, sizeof(double) * n);
        batches++;
        return 0;
    }
    int stop()
    {
        if (!active) return IBO_OK;
        active = false;
        __atomic_store_n((unsigned long long *)&bx[0], IBO_SRV_EXIT, __ATOMIC_RELEASE);
        HIP_TRY(hipStreamSynchronize(g->stream));
        if (stamps) {
            std::vector<unsigned long long> h(64 * 2 * 8);
            HIP_TRY(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
            const int nb = batches < 64 ? batches : 64;
            double acc[2][6] = {}, hostrt = 0.0, relay = 0.0;
            for (int b = 1; b < nb; b++) {                // (batch 0 carries the start-up)
                for (int w = 0; w < 2; w++)
                    for (int k = 1; k < 6; k++) {
                        const unsigned long long t0 = h[(b * 2 + w) * 8], tk = h[(b * 2 + w) * 8 + k];
                        if (tk) acc[w][k] += (double)(tk - t0) * 0.01;
                    }
                hostrt += t_seen[b] - t_post[b];
                relay += (double)(h[(b * 2 + 0) * 8] - h[(b * 2 + 0) * 8 + 6]) * 0.01;
            }
            const int n = nb > 1 ? nb - 1 : 1;
            fprintf(stderr, "[libibo_hip] server stamps over %d batches (us after a workgroup saw the batch; wg 0 | last wg): phase-1 %.2f | %.2f  handover-1 %.2f | %.2f  "
                            "phase-2 %.2f | %.2f  handover-2 %.2f | %.2f  finish %.2f | %.2f;  mailbox seen -> relayed and seen by wg 0's own loop %.2f;  host post -> done seen %.2f\n", n,
                    acc[0][1] / n, acc[1][1] / n, acc[0][2] / n, acc[1][2] / n, acc[0][3] / n, acc[1][3] / n, acc[0][4] / n, acc[1][4] / n, acc[0][5] / n, acc[1][5] / n,
                    relay / n, hostrt / n);
        }
        return IBO_OK;
    }
};

int direct_on_gp(ibo_gp *g, int D, const double *lb, const double *ub, int acq, double parm, int erf_mode,
                        double clamp_lo, int maxiter, int maxtime, int maxsample, int compat,
                        double *opt, double *optx, int64_t *nsamples)
{
    if (D != g->D) return fail(IBO_ERR_ARG, "bounds have %d dimensions, model has %d", D, g->D);
    const bool dbg = getenv("IBO_DEBUG") != nullptr;
    double t_eval = 0.0; int n_batches = 0; int64_t n_pts = 0;
    DirectServer srv;
    IBO_TRY(srv.start(g, acq, parm, erf_mode, clamp_lo));
    ibo::batch_eval_t ev = [&](const double *pts, int n, double *vals) -> int {
        struct timespec a0, a1;
        if (dbg) clock_gettime(CLOCK_MONOTONIC, &a0);
        int rc = srv.eval(pts, n, vals);
        if (rc == 1) rc = eval_host_points(g, n, pts, acq, parm, erf_mode, clamp_lo, nullptr, nullptr, vals);
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
    const int src = srv.stop();
    g->srv_batches = srv.batches; g->srv_why = srv.why;
    if (dbg) fprintf(stderr, "[libibo_hip] DIRECT: %d iterations, %lld samples, %d batches (%lld points; %d by the resident server%s%s): %.2f ms total, %.2f ms in GPU evaluation\n",
                     r.iterations, (long long)r.nsamples, n_batches, (long long)n_pts, srv.batches, srv.why[0] ? "; " : "", srv.why,
                     (w1.tv_sec - w0.tv_sec) * 1e3 + (w1.tv_nsec - w0.tv_nsec) * 1e-6, t_eval);
    if (r.status) return r.status;
    if (src) return src;
    if (opt) *opt = -r.fmin;
    if (optx) for (int i = 0; i < D; i++) optx[i] = r.xmin[i];
    if (nsamples) *nsamples = r.nsamples;
    return IBO_OK;
}

// what the resident evaluation server did in the last ibo_direct_max on this handle: the batches it evaluated, and -- when some
// or all went through launches instead -- why ("" otherwise)
extern "C" int ibo_direct_server_info(ibo_gp_t *g, int *batches, const char **why)
{
    if (!g) return fail(IBO_ERR_ARG, "gp is NULL");
    if (batches) *batches = g->srv_batches;
    if (why) *why = g->srv_why;
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

