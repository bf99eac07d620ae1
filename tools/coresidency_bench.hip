// coresidency_bench.hip -- can a chain of small launches (66 workgroups x 133 KB of LDS, ~15 us each, one stream) run beside a long kernel of
// FAT workgroups (one per CU: 140 KB of LDS) on another stream, when the long kernel leaves CUs free (190 workgroups on 256 CUs)?
//   hipcc -O3 --offload-arch=gfx950 tools/coresidency_bench.hip -o tools/coresidency_bench && tools/coresidency_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int LDSB>
__global__ __launch_bounds__(512) void spin_kernel(long long ticks, double *sink)      // s_memrealtime: 100 MHz
{
    __shared__ double buf[LDSB / 8];
    buf[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    double a = buf[(threadIdx.x * 7) & 511];
    while ((long long)__builtin_readcyclecounter() - t0 < ticks) a = a * 1.0000001 + 1e-9;
    if (a == 12345.678) sink[0] = a;
}

int main()
{
    double *sink;
    CHK(hipMalloc(&sink, 64));
    hipStream_t sa, sb;
    CHK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    CHK(hipFuncSetAttribute((const void *)spin_kernel<143360>, hipFuncAttributeMaxDynamicSharedMemorySize, 0));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    // the cycle counter's rate: calibrate one spin
    const long long chain_ticks = 36000, long_ticks = 4800000;           // s_memtime ticks (the base case below shows what a chain launch lasts)
    for (int fat_wgs : {0, 128, 160, 176, 184, 190, 256}) {
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipDeviceSynchronize());
            if (fat_wgs) hipLaunchKernelGGL(spin_kernel<143360>, dim3(fat_wgs), dim3(512), 0, sb, long_ticks, sink);
            CHK(hipEventRecord(e0, sa));
            for (int i = 0; i < 40; i++) hipLaunchKernelGGL(spin_kernel<136192>, dim3(66), dim3(512), 0, sa, chain_ticks, sink);
            CHK(hipEventRecord(e1, sa));
            CHK(hipDeviceSynchronize());
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("fat workgroups %4d (140 KB, %lld ticks): 40 chain launches of 66 x 133 KB take %.3f ms (%.1f us each)\n", fat_wgs, long_ticks, ms, ms * 1000 / 40);
        }
    }
    // the same with thin long workgroups (64 KB: two per CU)
    for (int thin_wgs : {380, 512, 4096}) {
        CHK(hipDeviceSynchronize());
        hipLaunchKernelGGL(spin_kernel<65536>, dim3(thin_wgs), dim3(512), 0, sb, long_ticks, sink);
        CHK(hipEventRecord(e0, sa));
        for (int i = 0; i < 40; i++) hipLaunchKernelGGL(spin_kernel<136192>, dim3(66), dim3(512), 0, sa, chain_ticks, sink);
        CHK(hipEventRecord(e1, sa));
        CHK(hipDeviceSynchronize());
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("thin workgroups %4d (64 KB): 40 chain launches take %.3f ms (%.1f us each)\n", thin_wgs, ms, ms * 1000 / 40);
    }
    return 0;
}
