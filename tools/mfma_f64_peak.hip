// mfma_f64_peak.hip -- measured ceiling of v_mfma_f64_16x16x4_f64 on this device:
// back-to-back MFMAs on 16 independent accumulators per wave (the sweep kernel's
// register tile), random operands, 1 or 2 waves per SIMD.  Prints TFLOP/s and the
// in-kernel shader clock (s_memtime / s_memrealtime).  Diagnostic tool, not product.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

// The same pipe with operands that CHANGE from one MFMA to the next, as they do in a real kernel (the loop above multiplies the same
// four a's and four b's for ever: little toggles, little power).  16 a's and 16 b's with full-entropy mantissas in registers; every
// MFMA of an unrolled group of 64 takes another pair.  What the device sustains here -- clock included -- is the ceiling a kernel
// fed with real data can reach; the nominal peak assumes the boost clock.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void kvar(const double *in, double *out, unsigned long long *clk, int iters)
{
    int l = threadIdx.x;
    double a[16], b[16];
    for (int i = 0; i < 16; i++) { a[i] = in[(l * 16 + i) & 4095]; b[i] = in[(l * 23 + i + 100) & 4095]; }
    d4 acc[4][4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) acc[i][j] = (d4){0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(4 * u + i) & 15], b[(4 * ((u + j) & 3) + ((i + j) & 3)) & 15], acc[i][j], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) for (int r = 0; r < 4; r++) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + l] = s;
    if (l == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int WAVES>
void runvar(const char *name, int blocks, int iters, bool zeros)
{
    double *in, *out; unsigned long long *clk;
    std::vector<double> h(4096);
    for (auto &v : h) {                                   // random sign, exponent around 1, 52 random mantissa bits
        unsigned long long m = ((unsigned long long)rand() << 31) ^ ((unsigned long long)rand() << 10) ^ (unsigned long long)rand();
        unsigned long long bits = ((unsigned long long)(rand() & 1) << 63) | ((1019ull + (rand() % 8)) << 52) | (m & 0xFFFFFFFFFFFFFull);
        double d; memcpy(&d, &bits, 8); v = zeros ? 0.0 : d;
    }
    hipMalloc(&in, 4096 * 8); hipMalloc(&out, (size_t)blocks * WAVES * 64 * 8); hipMalloc(&clk, blocks * 16);
    hipMemcpy(in, h.data(), 4096 * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kvar<WAVES>, dim3(blocks), dim3(WAVES * 64), 0, 0, in, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(2 * blocks);
        hipMemcpy(c.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
        double flops = (double)blocks * WAVES * iters * 16 * 2048.0;
        double ghz = 0; for (int b = 0; b < blocks; b++) ghz += (double)c[2 * b] / (double)c[2 * b + 1] * 0.1; ghz /= blocks;
        printf("%s blocks=%d iters=%d: %.3f ms  %.2f TFLOP/s  mean clock %.3f GHz\n", name, blocks, iters, ms, flops / ms / 1e9, ghz);
    }
    hipFree(in); hipFree(out); hipFree(clk);
}

// does fp64 VALU work share the MFMA pipe?  Each SIMD runs 2 MFMA waves (a full pipe) and
// VW extra waves doing independent v_fma_f64 chains.  If the units were separate the MFMA
// rate would stay at ~77 TFLOP/s.
template <int VW>
__global__ __launch_bounds__((8 + 4 * VW) * 64) void kmix(const double *in, double *out, int iters, int fma_per_iter)
{
    int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (w < 8) {
        double a[4], b[4];
        for (int i = 0; i < 4; i++) { a[i] = in[(l * 4 + i) & 4095]; b[i] = in[(l * 7 + i + 100) & 4095]; }
        d4 acc[4][4];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) acc[i][j] = (d4){0, 0, 0, 0};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        double s = 0;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) for (int r = 0; r < 4; r++) s += acc[i][j][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        double x[8], c = in[l], d = in[l + 64];
        for (int i = 0; i < 8; i++) x[i] = in[(l + i * 64) & 4095];
        int n = iters * fma_per_iter / 8;
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = __builtin_fma(x[i], c, d);
        }
        double s = 0;
        for (int i = 0; i < 8; i++) s += x[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

template <int VW>
void runmix(int fma_per_iter)
{
    const int blocks = 256, iters = 20000, threads = (8 + 4 * VW) * 64;
    double *in, *out;
    std::vector<double> h(4096);
    for (auto &v : h) v = (double)rand() / RAND_MAX * 0.5;
    hipMalloc(&in, 4096 * 8); hipMalloc(&out, (size_t)blocks * threads * 8);
    hipMemcpy(in, h.data(), 4096 * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kmix<VW>, dim3(blocks), dim3(threads), 0, 0, in, out, iters, fma_per_iter);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double mf = (double)blocks * 8 * iters * 16 * 2048.0, vf = (double)blocks * 4 * VW * 64 * (double)iters * fma_per_iter * 2.0;
        printf("mix: 2 MFMA waves/SIMD + %d VALU wave(s)/SIMD, %d v_fma_f64 per 16 MFMAs: %.3f ms  MFMA %.2f TFLOP/s + VALU %.2f TFLOP/s = %.2f\n",
               VW, fma_per_iter, ms, mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9);
    }
}

// cost of other instruction classes next to a saturated fp64 MFMA pipe: 2 MFMA waves per SIMD
// plus 1 wave per SIMD running OP (0 = v_fma_f64, 1 = v_fma_f32, 2 = v_add_u32/v_xor, 3 = v_ldexp_f64,
// 4 = v_rndne_f64, 5 = v_cvt_i32_f64) in 8 independent chains.  Extra time / #ops = pipe cycles per op.
template <int OP>
__global__ __launch_bounds__(12 * 64) void kops(const double *in, double *out, int iters, int ops_per_iter)
{
    int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (w < 8) {
        double a[4], b[4];
        for (int i = 0; i < 4; i++) { a[i] = in[(l * 4 + i) & 4095]; b[i] = in[(l * 7 + i + 100) & 4095]; }
        d4 acc[4][4];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) acc[i][j] = (d4){0, 0, 0, 0};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        double s = 0;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) for (int r = 0; r < 4; r++) s += acc[i][j][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        int n = iters * ops_per_iter / 8;
        double s = 0;
        if (OP == 0) {
            double x[8], c = in[l], d = in[l + 64];
            for (int i = 0; i < 8; i++) x[i] = in[(l + i * 64) & 4095];
            for (int it = 0; it < n; it++) {
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = __builtin_fma(x[i], c, d);
            }
            for (int i = 0; i < 8; i++) s += x[i];
        } else if (OP == 1) {
            float x[8], c = (float)in[l], d = (float)in[l + 64];
            for (int i = 0; i < 8; i++) x[i] = (float)in[(l + i * 64) & 4095];
            for (int it = 0; it < n; it++) {
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = __builtin_fmaf(x[i], c, d);
            }
            for (int i = 0; i < 8; i++) s += x[i];
        } else if (OP == 2) {
            unsigned x[8], c = (unsigned)(in[l] * 1e6);
            for (int i = 0; i < 8; i++) x[i] = (unsigned)(in[(l + i * 64) & 4095] * 1e6);
            for (int it = 0; it < n; it++) {
#pragma unroll
                for (int i = 0; i < 8; i++) { x[i] = (x[i] + c) ^ (unsigned)it; }
            }
            for (int i = 0; i < 8; i++) s += x[i];
        } else if (OP == 3) {
            double x[8]; int e = (int)(in[l] * 3);
            for (int i = 0; i < 8; i++) x[i] = in[(l + i * 64) & 4095];
            for (int it = 0; it < n; it++) {
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = __builtin_ldexp(x[i], e - (it & 1) * 2 * e);
            }
            for (int i = 0; i < 8; i++) s += x[i];
        } else if (OP == 4) {
            double x[8];
            for (int i = 0; i < 8; i++) x[i] = in[(l + i * 64) & 4095] * 1000;
            for (int it = 0; it < n; it++) {
#pragma unroll
                for (int i = 0; i < 8; i++) { double t = __builtin_rint(x[i]); asm volatile("" : "+v"(t)); x[i] = t; }
            }
            for (int i = 0; i < 8; i++) s += x[i];
        } else {
            double x[8]; int acc = 0;
            for (int i = 0; i < 8; i++) x[i] = in[(l + i * 64) & 4095] * 1000;
            for (int it = 0; it < n; it++) {
#pragma unroll
                for (int i = 0; i < 8; i++) { int t = (int)x[i]; asm volatile("" : "+v"(t)); acc += t; }
            }
            s = acc;
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

template <int OP>
void runops(const char *name, int ops_per_iter)
{
    const int blocks = 256, iters = 20000, threads = 12 * 64;
    double *in, *out;
    std::vector<double> h(4096);
    for (auto &v : h) v = (double)rand() / RAND_MAX * 0.5;
    hipMalloc(&in, 4096 * 8); hipMalloc(&out, (size_t)blocks * threads * 8);
    hipMemcpy(in, h.data(), 4096 * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kops<OP>, dim3(blocks), dim3(threads), 0, 0, in, out, iters, ops_per_iter);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    // baseline MFMA-only time for this shape: iters * 32 MFMAs * 64 cycles per SIMD
    double base_cycles = (double)iters * 32 * 64, ghz = 2.39;
    double extra = ms * 1e-3 * ghz * 1e9 - base_cycles;
    printf("ops next to MFMA: %-14s %4d per 32 MFMAs: %.3f ms  -> %.1f extra pipe cycles per op (at %.2f GHz)\n", name,
           ops_per_iter, ms, extra / ((double)iters * ops_per_iter), ghz);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s %s CUs=%d clock=%d MHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    run<4>("1 wave/SIMD ", 256, 20000);
    run<8>("2 waves/SIMD", 256, 20000);
    run<8>("2 waves/SIMD x4 blocks", 1024, 10000);
    runvar<8>("2 waves/SIMD, operands changing every MFMA, random mantissas, 17 ms", 256, 20000, false);
    runvar<8>("2 waves/SIMD, operands changing every MFMA, random mantissas, 140 ms", 1024, 40000, false);
    runvar<8>("2 waves/SIMD, the same loop on zeros, 140 ms", 1024, 40000, true);
    runmix<1>(0); runmix<1>(64); runmix<1>(256); runmix<2>(256);
    runops<0>("v_fma_f64", 128); runops<1>("v_fma_f32", 128); runops<2>("v_add+xor u32", 128);
    runops<3>("v_ldexp_f64", 128); runops<4>("v_rndne_f64", 128); runops<5>("v_cvt_i32_f64", 128);
    runops<1>("v_fma_f32", 512); runops<2>("v_add+xor u32", 512);
    return 0;
}
