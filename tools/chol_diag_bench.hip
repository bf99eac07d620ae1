// Diagnostic: per-segment cycle stamps of chol_diag_kernel (built with -DIBO_STAMPS) on one 64x64 block.
//   hipcc --offload-arch=gfx950 -O3 -DIBO_STAMPS -I ibo_amd/csrc tools/chol_diag_bench.hip -o tools/chol_diag_bench
#include "../ibo_amd/csrc/linalg.hip"
#include "../ibo_amd/csrc/update3.hip"
#include <cstdio>
#include <vector>
int main()
{
    const int n = 64;
    std::vector<double> A(n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) A[i * n + j] = (i == j ? 1.1 : 0.0) + exp(-0.5 * (i - j) * (i - j) / 40.0);
    double *dA, *dD; int *info;
    hipMalloc(&dA, sizeof(double) * n * n); hipMalloc(&dD, sizeof(double) * n * n); hipMalloc(&info, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
        hipMemset(info, 0, 4);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), 0, 0, dA, n, 0, dD, info, (size_t)0, (size_t)0, (double *)nullptr);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long st[32];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_chol_stamps), sizeof(st));
        printf("rep %d: %.1f us by events; total stamps %llu ticks\n", rep, ms * 1e3, st[23] - st[31]);
        printf("  load %llu  sync %llu\n", st[0] - st[31], st[1] - st[0]);
        unsigned long long prev = st[1];
        for (int b = 0; b < 4; b++) {
            printf("  panel %d: chol %llu  sync %llu  (iii) %llu\n", b, st[2 + 5 * b] - prev,
                   st[4 + 5 * b] - st[2 + 5 * b], st[6 + 5 * b] - st[4 + 5 * b]);
            prev = st[6 + 5 * b];
        }
        printf("  inverses %llu  doubling %llu  store %llu\n", st[21] - prev, st[22] - st[21], st[23] - st[22]);
    }
    {   // fused step kernel stamps: block column 0 of a 1024 x 1024 matrix (120 workgroups)
        const int Np = 1024;
        std::vector<double> M((size_t)Np * Np);
        for (int i = 0; i < Np; i++)
            for (int j = 0; j < Np; j++) M[(size_t)i * Np + j] = (i == j ? 1.1 : 0.0) + exp(-0.5 * (i - j) * (double)(i - j) / 900.0);
        double *dW, *dO, *d64; int *dinfo;
        hipMalloc(&dW, sizeof(double) * Np * Np); hipMalloc(&dO, sizeof(double) * Np * Np); hipMalloc(&d64, sizeof(double) * 16 * 4096); hipMalloc(&dinfo, 4);
        for (int rep = 0; rep < 2; rep++) {
            hipMemcpy(dW, M.data(), sizeof(double) * Np * Np, hipMemcpyHostToDevice);
            hipMemset(dinfo, 0, 4);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(chol_step_kernel<false>, dim3(15 * 16 / 2), dim3(256), 0, 0, dW, dO, Np, 0, d64, dinfo, 15 * 16 / 2, 0,
                               (double *)nullptr, (double *)nullptr, 15 * 16 / 2);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long ss[2][16];
            hipMemcpyFromSymbol(ss, HIP_SYMBOL(g_step_stamps), sizeof(ss));
            for (int w = 0; w < 2; w++)
                printf("step kernel (%.1f us by events) wg %d: fetch-issue+load %llu  diag chain %llu  store+sync %llu  stash+sync %llu  2 trsm products %llu  sync %llu  X to LDS+sync %llu  update product %llu  store %llu   total %llu cycles\n",
                       ms * 1e3, w ? 7 : 0, ss[w][1] - ss[w][0], ss[w][2] - ss[w][1], ss[w][3] - ss[w][2], ss[w][4] - ss[w][3], ss[w][5] - ss[w][4],
                       ss[w][6] - ss[w][5], ss[w][7] - ss[w][6], ss[w][8] - ss[w][7], ss[w][9] - ss[w][8], ss[w][9] - ss[w][0]);
        }
    }
    {   // update kernel stamps: first K = 256 update of a batch of 16 matrices with 66 block rows
        const int Np = 66 * 64, B = 16;
        double *dL; hipMalloc(&dL, sizeof(double) * (size_t)Np * Np * B);
        hipMemset(dL, 0, sizeof(double) * (size_t)Np * Np * B);
        hipStream_t st = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0, 0);
            launch_update(dL, Np, 0, 4, 4, 66, B, (size_t)Np * Np, st);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double tiles = 62.0 * 63 / 2 * B;
            printf("update K=256, %d matrices of %d: %.3f ms  -> %.1f TFLOP/s\n", B, Np, ms, tiles * 2.0 * 64 * 64 * 256 / ms / 1e9);
        }
        unsigned long long us[4][32];
        hipMemcpyFromSymbol(us, HIP_SYMBOL(g_upd_stamps), sizeof(us));
        for (int w = 0; w < 4; w++) {
            printf("  wg %d:", w);
            for (int j = 0; j < 4; j++)
                printf("  [stash %llu bar %llu mma %llu bar %llu]", us[w][1 + 4 * j] - (j ? us[w][4 * j] : us[w][0]),
                       us[w][2 + 4 * j] - us[w][1 + 4 * j], us[w][3 + 4 * j] - us[w][2 + 4 * j], us[w][4 + 4 * j] - us[w][3 + 4 * j]);
            printf("  store %llu\n", us[w][20] - us[w][16]);
        }
        hipFree(dL);
    }
    std::vector<double> Lh(n * n);
    hipMemcpy(Lh.data(), dA, sizeof(double) * n * n, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) {
            double s = 0;
            for (int k = 0; k <= j; k++) s += Lh[i * n + k] * Lh[j * n + k];
            err = fmax(err, fabs(s - A[i * n + j]));
        }
    std::vector<double> Vh(n * n);
    hipMemcpy(Vh.data(), dD, sizeof(double) * n * n, hipMemcpyDeviceToHost);
    double err2 = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int k = 0; k < n; k++) s += Vh[i * n + k] * (k >= j ? Lh[k * n + j] : 0.0);
            err2 = fmax(err2, fabs(s - (i == j)));
        }
    printf("max |L L^T - A| = %.3e   max |V L - I| = %.3e\n", err, err2);
    return 0;
}
