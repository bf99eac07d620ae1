// abi_nlml.hip -- the marginal-likelihood side of the C ABI: the batched theta-grid, one value + gradient, and ibo_trim (which owns
// their per-device workspaces).
#include "abi_internal.h"

// ------------------------------------------------------------------------ marginal-likelihood grid
struct NlmlWorkspace {
    DevBuf<double> dX, dY, dout, dL, d64, dP;       // dP: packed store of the trailing updates (update3.hip)
    DevBuf<KParams> dkp;                            // the theta-points' kernel parameters (one covariance launch per sub-batch)
    DevBuf<int> dinfo, dflags;                      // dflags: four hand-over words per matrix (chol_panel_fused_kernel)
    const double *padded = nullptr;                 // dL as it was when its matrices got their identity pad,
    int pad_Np = 0, pad_N = 0, pad_B = 0;           // and for which geometry
    hipStream_t streams[4] = {nullptr, nullptr, nullptr, nullptr};     // sub-batches of a grid run side by side (created on first use, kept)
    hipEvent_t t0[4] = {nullptr, nullptr, nullptr, nullptr}, t1[4] = {nullptr, nullptr, nullptr, nullptr};     // a sub-batch's span on its stream (ibo_gpu_time_ms)
};
static NlmlWorkspace g_nlml_ws[16];
static const int kSyrk3From = 1792;          // rows from which ibo_nlml_grad forms K^-1 = W^T W on the packed-operand kernel (launch_syrk3)
struct GradWorkspace {
    DevBuf<double> dX, dY, dL, dW, dT, dKi, d64, dal, da1, tmp, dpart, dout, dpiece, tall, Pk2;     // tall, Pk2: the super-panel order's (launch_cholesky_super)
    DevBuf<int> dinfo, dtasks, dsums;
    int plan_Np = 0, ntasks = 0, nsums = 0;         // launch_syrk3's lists on the device, for this Npad
    hipEvent_t t0 = nullptr, t1 = nullptr;          // the evaluation's span on the device (ibo_gpu_time_ms)
    // a learning loop calls with the same data and another theta dozens of times: what is on the device is kept (compared by content) and
    // the results come back through one pinned block behind one synchronisation
    std::vector<double> hostX, hostY;
    double *pin = nullptr;                          // IBO_GRAD_MAX + 4 doubles: gradient, two scalars, the info word
};
static GradWorkspace g_grad_ws[16];

extern "C" int ibo_trim(int device)
{
    IBO_TRY(use_device(device));
    std::lock_guard<std::mutex> lk(g_dev_mu[device & 15]);
    NlmlWorkspace &ws = g_nlml_ws[device & 15];
    ws.dX.release(); ws.dY.release(); ws.dout.release(); ws.dL.release(); ws.d64.release(); ws.dP.release(); ws.dinfo.release(); ws.dflags.release(); ws.dkp.release();
    ws.padded = nullptr;
    for (int g = 0; g < 4; g++) if (ws.streams[g]) {
        (void)hipStreamDestroy(ws.streams[g]); ws.streams[g] = nullptr;
        if (ws.t0[g]) { (void)hipEventDestroy(ws.t0[g]); (void)hipEventDestroy(ws.t1[g]); ws.t0[g] = ws.t1[g] = nullptr; }
    }
    GradWorkspace &gw = g_grad_ws[device & 15];
    gw.dX.release(); gw.dY.release(); gw.dL.release(); gw.dW.release(); gw.dT.release(); gw.dKi.release(); gw.d64.release();
    gw.dal.release(); gw.da1.release(); gw.tmp.release(); gw.dpart.release(); gw.dout.release(); gw.dinfo.release();
    gw.dpiece.release(); gw.dtasks.release(); gw.dsums.release(); gw.plan_Np = 0; gw.tall.release(); gw.Pk2.release();
    gw.hostX.clear(); gw.hostY.clear();              // (dX / dY went back to the pool: nothing of this data is on the device any more)
    if (gw.pin) { (void)hipHostFree(gw.pin); gw.pin = nullptr; }
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
    // the covariance pass forms the exponent on the MFMA unit (cov_grid_mfma_kernel) where every scaled point of every theta-point stays
    // within the dot form's accuracy guard: |x~|^2 <= sum_d w_d max_k x_kd^2 <= 1e5 (ibo_set_option("dot_form", 0) keeps the difference form)
    int dot_ok = g_dot_override.load() != 0 && D <= IBO_DDOT;
    if (dot_ok) {
        std::vector<double> xm(D, 0.0);
        for (int i = 0; i < N; i++)
            for (int d = 0; d < D; d++) { const double v = X[(size_t)i * D + d] * X[(size_t)i * D + d]; if (v > xm[d]) xm[d] = v; }
        for (int t = 0; t < n_theta && dot_ok; t++) {
            double b = 0.0;
            for (int d = 0; d < D; d++) b += kps[t].w[d] * xm[d];
            if (!(b <= 1e5)) dot_ok = 0;
        }
    }
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
                if (!ws.streams[g]) {
                    // a sub-batch's stream must run BESIDE the others': a new stream may share a hardware queue with one of them
                    // (assemble.hip: streams_run_side_by_side) -- then another is made, the rejected ones given back afterwards
                    hipStream_t rejected[6];
                    int nrej = 0;
                    for (;;) {
                        hipStream_t cand = nullptr;
                        HIP_TRY(hipStreamCreateWithFlags(&cand, hipStreamNonBlocking));
                        bool ok = true;
                        for (int g2 = 0; g2 < g && ok; g2++) KERNEL_TRY(streams_run_side_by_side(ws.streams[g2], cand, &ok));
                        if (ok || nrej == 6) { ws.streams[g] = cand; break; }
                        rejected[nrej++] = cand;
                    }
                    if (getenv("IBO_DEBUG")) fprintf(stderr, "[libibo_hip] sub-batch stream %d: %d candidate(s) shared a hardware queue with an earlier one%s\n", g, nrej, nrej == 6 ? " -- none found that does not" : "");
                    for (int r = 0; r < nrej; r++) (void)hipStreamDestroy(rejected[r]);
                }
                if (!ws.t0[g]) { HIP_TRY(hipEventCreate(&ws.t0[g])); HIP_TRY(hipEventCreate(&ws.t1[g])); }
                hipStream_t sg = ws.streams[g];
                HIP_TRY(hipEventRecord(ws.t0[g], sg));
                const int k0 = (int)((long long)nb * g / G), k1 = (int)((long long)nb * (g + 1) / G), ng = k1 - k0;
                grp[g] = CholGroup{dL.p + nn * k0, d64.p + (size_t)(Np / 64) * 4096 * k0, ws.dP.p + pws * k0, dinfo.p + t0 + k0, ng, sg,
                                   with_flags ? ws.dflags.p + 4 * k0 : nullptr};
                KERNEL_TRY(launch_cov_matrix_batched(ws.dkp.p + t0 + k0, ng, N, dX.p, D, IBO_DIAG_KERNEL_PLUS_NOISE, noise, dL.p + nn * k0, Np, nn, sg, dot_ok));
                KERNEL_TRY(launch_nlml_aug(dL.p + nn * k0, Np, N, dY.p, sg, ng, nn));
            }
            // (N a multiple of 64: the y row sits alone in the last block column, whose factor nobody reads -- it is left out)
            if (left) KERNEL_TRY(launch_cholesky_batched_left(grp, G, Np, nn, 4, pws, N + 1, N % 64 == 0 ? Np / 64 - 1 : Np / 64, N / 64));
            else KERNEL_TRY(launch_cholesky_batched(grp[0].L, Np, grp[0].diag64, grp[0].info, grp[0].batch, nn, 4, grp[0].stream, grp[0].Pk, pws));      // (G = 1)
            for (int g = 0; g < G; g++) {
                KERNEL_TRY(launch_nlml_reduce(grp[g].L, Np, N, dout.p + 2 * (size_t)(grp[g].info - dinfo.p), grp[g].stream, grp[g].batch, nn));
                HIP_TRY(hipEventRecord(ws.t1[g], grp[g].stream));
            }
            float span = 0.f;
            for (int g = 0; g < G; g++) {
                HIP_TRY(hipStreamSynchronize(ws.streams[g]));      // the next batch reuses the matrix slots
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, ws.t0[g], ws.t1[g]) == hipSuccess && ms > span) span = ms;
            }
            gpu_time_add(device, span);                            // (the sub-batches run side by side: the longest span, not the sum)
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
    if (ws.hostX.size() != (size_t)N * D || memcmp(ws.hostX.data(), X, sizeof(double) * N * D) != 0) {
        ws.hostX.clear();                            // (not valid while the copy is in flight or if it fails)
        HIP_TRY(hipMemcpy(dX.p, X, sizeof(double) * N * D, hipMemcpyHostToDevice));
        ws.hostX.assign(X, X + (size_t)N * D);
    }
    if (ws.hostY.size() != (size_t)Np || memcmp(ws.hostY.data(), Y, sizeof(double) * N) != 0) {
        std::vector<double> yp(Np, 0.0);
        for (int i = 0; i < N; i++) yp[i] = Y[i];
        ws.hostY.clear();
        HIP_TRY(hipMemcpy(dY.p, yp.data(), sizeof(double) * Np, hipMemcpyHostToDevice));
        ws.hostY.swap(yp);
    }
    if (!ws.pin) HIP_TRY(hipHostMalloc((void **)&ws.pin, sizeof(double) * (IBO_GRAD_MAX + 4), hipHostMallocDefault));
    // up to 2048 rows: the fit's route -- fused steps with W = L^-1 riding along (dT: the matrix being reduced, dKi: (L^-1)^T
    // until the transpose) -- instead of the three-kernel columns and the recursive-doubling inversion
    const bool fused = single_level_order(Np);
    if (!ws.t0) { HIP_TRY(hipEventCreate(&ws.t0)); HIP_TRY(hipEventCreate(&ws.t1)); }
    HIP_TRY(hipEventRecord(ws.t0, s));
    if (fused && super_order(Np)) {
        // (the fit's rule: from g_super_min_nb block columns on in super-panels -- the matrix and the ride-along's identity in one tall buffer; same bits)
        IBO_TRY(ws.tall.ensure(2 * nn)); IBO_TRY(ws.Pk2.ensure(2 * nn));
        KERNEL_TRY(launch_cov_fit(kp, N, dX.p, D, IBO_DIAG_KERNEL_PLUS_NOISE, noise, ws.tall.p, Np, ws.tall.p + nn, dinfo.p, s));
        KERNEL_TRY(launch_cholesky_super(ws.tall.p, dL.p, Np, d64.p, dinfo.p, s, dKi.p, ws.Pk2.p, true));
    } else if (fused) {
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
        // from 1792 rows the product runs on the packed-operand kernel (128 x 128 tiles, A fragments straight from L2), its long K ranges in pieces
        // of 256 columns below 2560 rows, 512 below 4096, 1024 from there (measured: 0.692 -> 0.668 ms per evaluation at N = 1792, 0.809 -> 0.766 at
        // 2048, 1.028 -> 1.002 at 2560, 1.436 -> 1.400 at 3072; below 1792 rows wtw_kernel's 64 x 64 tiles are as fast); the packed copy of W^T
        // goes where L was
        if (ws.plan_Np != Np) {
            std::vector<int> tasks, sums;
            int nslots = 0;
            syrk3_plan(Np, Np < 2560 ? 256 : (Np < 4096 ? 512 : 1024), tasks, sums, &nslots);
            IBO_TRY(ws.dtasks.ensure(tasks.size())); IBO_TRY(ws.dsums.ensure(sums.size() + 4)); IBO_TRY(ws.dpiece.ensure((size_t)(nslots + 1) * 16384));
            HIP_TRY(hipMemcpy(ws.dtasks.p, tasks.data(), sizeof(int) * tasks.size(), hipMemcpyHostToDevice));
            if (!sums.empty()) HIP_TRY(hipMemcpy(ws.dsums.p, sums.data(), sizeof(int) * sums.size(), hipMemcpyHostToDevice));
            ws.plan_Np = Np; ws.ntasks = (int)tasks.size() / 4; ws.nsums = (int)sums.size() / 4;
        }
        KERNEL_TRY(launch_syrk3(dKi.p, dL.p, dT.p, Np, ws.dtasks.p, ws.ntasks, ws.dsums.p, ws.nsums, ws.dpiece.p, s));
    } else if (fused) KERNEL_TRY(launch_wtw(dW.p, dKi.p, dT.p, Np, s, 1, 1));
    else KERNEL_TRY(launch_wtw(dW.p, dT.p, dKi.p, Np, s, 1));
    KERNEL_TRY(launch_nlml_grad(kp, gs, N, dX.p, D, Kinv, Np, dal.p, dpart.p, dout.p, s));
    HIP_TRY(hipEventRecord(ws.t1, s));
    HIP_TRY(hipMemcpyAsync(ws.pin, dout.p, sizeof(double) * (ngrad + 2), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(ws.pin + IBO_GRAD_MAX + 2, dinfo.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const double *res = ws.pin;
    int h = 0;
    memcpy(&h, ws.pin + IBO_GRAD_MAX + 2, sizeof(int));
    { float ms = 0.f; if (hipEventElapsedTime(&ms, ws.t0, ws.t1) == hipSuccess) gpu_time_add(device, ms); }
    if (h != 0) return fail(IBO_ERR_NOT_PD, "covariance matrix is not positive definite (pivot %d)", h);
    for (int i = 0; i < ngrad; i++) grad_host[i] = res[i];
    *nlml_host = 0.5 * res[ngrad] + res[ngrad + 1] + 0.5 * N * log(2.0 * M_PI);
    return IBO_OK;
}

