// Ablation of the Cholesky trailing-update tile kernel (K = 256, 64x64 tiles, 4 waves): which part of a stage
// costs what.  V = 0 full; 1 no global fetch inside the loop; 2 no LDS stash (operands stay as first written);
// 3 fragments read once, MFMAs only; 4 no MFMAs (fetch + stash + fragment reads only); 5 / 6 / 7 the C tile not
// loaded / not stored / neither.
//   hipcc --offload-arch=gfx950 -O3 -w -I ibo_amd/csrc tools/update_ablation.hip -o tools/update_ablation
#include "../ibo_amd/csrc/linalg.hip"
#include <cstdio>
#include <vector>

template <int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void upd_variant(double *L, int Npad, int j0, int j1, int k0, int k1, int nsb, size_t lstride)
{
    __shared__ double As[64 * T64_LD];
    __shared__ double Bs[64 * T64_LD];
    TILE_IDS;
    L += blockIdx.z * lstride;
    const int nb = Npad / 64;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int sb = (q >> 6) * 8 + xcd, lt = q & 63;
    if (sb >= nsb) return;
    const int nsr = (nb - k0 + 7) / 8;
    int SK = 0, rem = sb;
    while (rem >= nsr - SK) { rem -= nsr - SK; SK++; }
    const int k = k0 + 8 * SK + (lt & 7), i = k0 + 8 * (SK + rem) + (lt >> 3);
    if (k >= k1 || i >= nb || i < k) return;
    double *C = L + (size_t)i * 64 * Npad + k * 64;
    const double *Ai = L + (size_t)i * 64 * Npad, *Ak = L + (size_t)k * 64 * Npad;
    if (V == 8) { if (Ai[0] == 12345.678) C[0] = 1.0; return; }        // dispatch + tile lookup only
    d2_t va[8], vb[8];
    tile64_fetch(Ai + j0 * 64, Npad, va);
    tile64_fetch(Ak + j0 * 64, Npad, vb);
    d4_t acc[2][2];
    for (int m = 0; m < 2; m++)
        for (int n = 0; n < 2; n++)
            for (int r = 0; r < 4; r++) acc[m][n][r] = (V == 5 || V == 7) ? 0.0 : C[(size_t)TILE_ROW(m, r) * Npad + TILE_COL(n)];
    double fa[2] = {1.0, 2.0}, fb[2] = {3.0, 4.0};
    if (V == 9) {                                  // first operand fetch + stash + barrier, nothing else
        tile64_stash<true>(As, va); tile64_stash(Bs, vb);
        __syncthreads();
        if (As[threadIdx.x] + Bs[threadIdx.x] == 12345.678) C[0] = 1.0;
        return;
    }
    for (int j = j0; j < j1; j++) {
        if (V != 2 || j == j0) { tile64_stash<true>(As, va); tile64_stash(Bs, vb); }
        __syncthreads();
        if (V != 1 && j + 1 < j1) { tile64_fetch(Ai + (j + 1) * 64, Npad, va); tile64_fetch(Ak + (j + 1) * 64, Npad, vb); }
        if (V == 3) {
            if (j == j0) { fa[0] = As[lane]; fa[1] = As[lane + 64]; fb[0] = Bs[lane]; fb[1] = Bs[lane + 64]; }
#pragma unroll
            for (int k4 = 0; k4 < 16; k4++)
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int n = 0; n < 2; n++) acc[m][n] = mfma_f64(fa[m], fb[n], acc[m][n]);
        } else if (V == 4) {
            const int wvv = threadIdx.x >> 6, wr2 = wvv >> 1, wc2 = wvv & 1;
#pragma unroll
            for (int k4 = 0; k4 < 16; k4++)
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    acc[m][0][0] += As[(wr2 * 32 + m * 16 + (lane & 15)) * T64_LD + k4 * 4 + (lane >> 4)];
                    acc[m][1][0] += Bs[(wc2 * 32 + m * 16 + (lane & 15)) * T64_LD + k4 * 4 + (lane >> 4)];
                }
        } else {
            tile64_mma_nt(As, Bs, acc);
        }
        if (j + 1 < j1) __syncthreads();
    }
    if (V == 6 || V == 7) {                        // no tile store: one value per wave keeps the work alive
        double s = 0.0;
        for (int m = 0; m < 2; m++)
            for (int n = 0; n < 2; n++)
                for (int r = 0; r < 4; r++) s += acc[m][n][r];
        if (s == 12345.678) C[0] = s;
        return;
    }
    for (int m = 0; m < 2; m++)
        for (int n = 0; n < 2; n++)
            for (int r = 0; r < 4; r++) C[(size_t)TILE_ROW(m, r) * Npad + TILE_COL(n)] = acc[m][n][r];
}

template <int V>
static void run(double *dL, int Np, int B, const char *what)
{
    const int nb = Np / 64, k0 = 4, nsr = (nb - k0 + 7) / 8;
    int nsb = nsr * (nsr + 1) / 2, groups = (nsb + 7) / 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(upd_variant<V>, dim3(groups * 512, 1, B), dim3(256), 0, 0, dL, Np, 0, 4, k0, nb, nsb, (size_t)Np * Np);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double tiles = (double)(nb - k0) * (nb - k0 + 1) / 2 * B;
    printf("%-58s %.3f ms   (%.1f TFLOP/s if it were the full kernel)\n", what, best, tiles * 2.0 * 64 * 64 * 256 / best / 1e9);
}

int main()
{
    const int Np = 66 * 64, B = 16;
    double *dL; hipMalloc(&dL, sizeof(double) * (size_t)Np * Np * B);
    std::vector<double> rnd((size_t)Np * Np);
    for (size_t e = 0; e < rnd.size(); e++) rnd[e] = (double)((e * 2654435761u) % 1000003) / 1000003.0 - 0.5;
    for (int b = 0; b < B; b++) hipMemcpy(dL + (size_t)b * Np * Np, rnd.data(), sizeof(double) * rnd.size(), hipMemcpyHostToDevice);
    run<0>(dL, Np, B, "full");
    run<1>(dL, Np, B, "no global fetch inside the K loop");
    run<2>(dL, Np, B, "no LDS stash after the first stage");
    run<3>(dL, Np, B, "MFMAs only (fragments read once)");
    run<4>(dL, Np, B, "no MFMAs (fetch + stash + fragment reads)");
    run<5>(dL, Np, B, "tile not loaded (accumulators start at 0)");
    run<6>(dL, Np, B, "tile not stored");
    run<7>(dL, Np, B, "tile neither loaded nor stored");
    run<8>(dL, Np, B, "dispatch + tile lookup only");
    run<9>(dL, Np, B, "first operand fetch + stash + barrier only");
    return 0;
}
