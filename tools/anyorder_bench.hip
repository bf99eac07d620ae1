// anyorder_bench.hip -- does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) let a kernel start before its predecessor in the SAME stream has finished (gfx950)?
//   hipcc -O3 --offload-arch=gfx950 tools/anyorder_bench.hip -o tools/anyorder_bench && tools/anyorder_bench
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(256) void spin_kernel(long long ticks, double *sink)
{
    const long long t0 = wall_clock64();                    // 100 MHz
    double a = threadIdx.x;
    while (wall_clock64() - t0 < ticks) a = a * 1.0000001 + 1e-9;
    if (a == 12345.678) sink[0] = a;
}
int main()
{
    double *sink; CHK(hipMalloc(&sink, 64));
    hipStream_t s; CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    long long tA = 200000, tB = 100000;                     // 2 ms, 1 ms
    void *argsA[] = {&tA, &sink}, *argsB[] = {&tB, &sink};
    for (int flags = 0; flags < 2; flags++)
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0, s));
            CHK(hipExtLaunchKernel((const void *)spin_kernel, dim3(64), dim3(256), argsA, 0, s, nullptr, nullptr, 0));
            CHK(hipExtLaunchKernel((const void *)spin_kernel, dim3(64), dim3(256), argsB, 0, s, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0));
            CHK(hipEventRecord(e1, s));
            CHK(hipDeviceSynchronize());
            float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("second launch flags = %d: 2 ms kernel + 1 ms kernel in one stream take %.3f ms\n", flags, ms);
        }
    return 0;
}
