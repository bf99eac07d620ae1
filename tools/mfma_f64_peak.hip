// mfma_f64_peak.hip -- measured ceiling of v_mfma_f64_16x16x4_f64 on this device:
// back-to-back MFMAs on 16 independent accumulators per wave (the sweep kernel's
// register tile), random operands, 1 or 2 waves per SIMD.  Prints TFLOP/s and the
// in-kernel shader clock (s_memtime / s_memrealtime).  Diagnostic tool, not product.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const double *in, double *out, unsigned long long *clk, int iters)
{
    int l = threadIdx.x;
    double a[4], b[4];
    for (int i = 0; i < 4; i++) { a[i] = in[(l * 4 + i) & 4095]; b[i] = in[(l * 7 + i + 100) & 4095]; }
    d4 acc[4][4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) acc[i][j] = (d4){0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) for (int r = 0; r < 4; r++) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + l] = s;
    if (l == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int WAVES>
void run(const char *name, int blocks, int iters)
{
    double *in, *out; unsigned long long *clk;
    std::vector<double> h(4096);
    for (auto &v : h) v = (double)rand() / RAND_MAX * 2 - 1;
    hipMalloc(&in, 4096 * 8); hipMalloc(&out, (size_t)blocks * WAVES * 64 * 8); hipMalloc(&clk, blocks * 16);
    hipMemcpy(in, h.data(), 4096 * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<WAVES>, dim3(blocks), dim3(WAVES * 64), 0, 0, in, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(2 * blocks);
        hipMemcpy(c.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
        double flops = (double)blocks * WAVES * iters * 16 * 2048.0;
        double ghz = (double)c[0] / (double)c[1] * 0.1;
        double cyc_per_mfma = (double)c[0] / ((double)iters * 16);
        printf("%s blocks=%d iters=%d: %.3f ms  %.2f TFLOP/s  clock %.3f GHz  %.1f shader-cycles per MFMA per wave\n",
               name, blocks, iters, ms, flops / ms / 1e9, ghz, cyc_per_mfma);
    }
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s %s CUs=%d clock=%d MHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    run<4>("1 wave/SIMD ", 256, 20000);
    run<8>("2 waves/SIMD", 256, 20000);
    run<8>("2 waves/SIMD x4 blocks", 1024, 10000);
    return 0;
}
