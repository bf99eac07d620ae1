// launch + completion latency of an empty kernel against the size of its by-value argument (gfx950, ROCm 7.2)
//   hipcc --offload-arch=gfx950 -O3 tools/kernarg_bench.hip -o tools/kernarg_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
template <int N> struct Blob { double v[N]; };
template <int N> __global__ void k(Blob<N> b, double *out) { if (threadIdx.x == 9999) out[0] = b.v[N - 1]; }
template <int N> void run(double *out)
{
    Blob<N> b; for (int i = 0; i < N; i++) b.v[i] = i;
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, b, out);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 200; i++) { hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, b, out); hipStreamSynchronize(0); }
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
    auto t1 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, b, out);
    hipStreamSynchronize(0);
    double us2 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count() / 2000;
    printf("argument %5zu bytes: launch + sync %.1f us, back-to-back %.2f us per launch\n", sizeof(b), us, us2);
}
int main()
{
    double *out; hipMalloc(&out, 8);
    run<8>(out); run<32>(out); run<64>(out); run<96>(out); run<128>(out); run<256>(out); run<384>(out); run<500>(out);
    return 0;
}
