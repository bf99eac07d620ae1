// launch_floor.hip -- what one dependent host <-> GPU round trip costs on this box, three ways:
//   (a) launch of an empty kernel whose only thread stores a sequence number into pinned host memory, host spins on it;
//   (b) the same with 4 KiB of kernel arguments (small2.hip's inline candidates);
//   (c) a RESIDENT kernel polling a pinned mailbox: host writes a sequence number, the kernel answers (no launch on the path);
//   (g3) the three dependent launches of (a3) as ONE instantiated hipGraph whose kernel nodes get new arguments before every launch
//        (hipGraphExecKernelNodeSetParams x 3 + hipGraphLaunch): what a DIRECT batch's three kernels would cost as a graph.
// Build: hipcc -O3 --offload-arch=gfx950 tools/launch_floor.hip -o tools/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <ctime>
#include <algorithm>
#include <vector>
struct Big { double v[480]; };
__global__ void flag_kernel(volatile unsigned long long *flag, unsigned long long seq) { *flag = seq; }
__global__ void flag_kernel_big(Big b, volatile unsigned long long *flag, unsigned long long seq) { *flag = seq; }
__global__ void resident_kernel(volatile unsigned long long *req, volatile unsigned long long *ack, unsigned long long last)
{
    unsigned long long seen = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (seen < last) {
        const unsigned long long r = __hip_atomic_load((unsigned long long *)req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (r != seen) { seen = r; __hip_atomic_store((unsigned long long *)ack, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) break;       // 2 s: never hang the box
    }
}
static double now_us() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }
static void report(const char *what, std::vector<double> &v)
{
    std::sort(v.begin(), v.end());
    printf("%-44s median %.2f us  p10 %.2f  p90 %.2f\n", what, v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10]);
}
int main()
{
    unsigned long long *flag; hipHostMalloc((void **)&flag, 128, hipHostMallocDefault); flag[0] = flag[8] = 0;
    hipStream_t s; hipStreamCreate(&s);
    const int R = 2000;
    std::vector<double> v; unsigned long long seq = 0;
    for (int pass = 0; pass < 2; pass++) {
        v.clear();
        for (int i = 0; i < R; i++) {
            const double t0 = now_us(); ++seq;
            hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, flag, seq);
            while (*(volatile unsigned long long *)flag != seq) {}
            v.push_back(now_us() - t0);
        }
    }
    report("(a) empty launch -> pinned flag", v);
    Big b = {};
    v.clear();
    for (int i = 0; i < R; i++) {
        const double t0 = now_us(); ++seq;
        hipLaunchKernelGGL(flag_kernel_big, dim3(1), dim3(64), 0, s, b, flag, seq);
        while (*(volatile unsigned long long *)flag != seq) {}
        v.push_back(now_us() - t0);
    }
    report("(b) the same with 3.8 KiB of arguments", v);
    v.clear();
    for (int i = 0; i < R; i++) {
        const double t0 = now_us(); ++seq;
        hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, flag + 16, seq);
        hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, flag + 16, seq);
        hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, flag, seq);
        while (*(volatile unsigned long long *)flag != seq) {}
        v.push_back(now_us() - t0);
    }
    report("(a3) three dependent empty launches", v);
    v.clear();
    for (int i = 0; i < R; i++) {
        const double t0 = now_us(); ++seq;
        hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, flag, seq);
        hipStreamSynchronize(s);
        v.push_back(now_us() - t0);
    }
    report("(d) empty launch + hipStreamSynchronize", v);
    // (g3) three kernel nodes in a chain, parameters refreshed per launch
    {
        hipGraph_t graph; hipGraphExec_t exec;
        hipGraphCreate(&graph, 0);
        hipGraphNode_t nodes[3];
        unsigned long long sq = seq;
        volatile unsigned long long *f0 = flag + 16, *f2 = flag;
        void *args[3][2] = {{(void *)&f0, (void *)&sq}, {(void *)&f0, (void *)&sq}, {(void *)&f2, (void *)&sq}};
        hipKernelNodeParams kp[3];
        for (int k = 0; k < 3; k++) {
            kp[k] = hipKernelNodeParams{};
            kp[k].func = (void *)flag_kernel; kp[k].gridDim = dim3(1); kp[k].blockDim = dim3(64); kp[k].sharedMemBytes = 0;
            kp[k].kernelParams = args[k]; kp[k].extra = nullptr;
            hipGraphAddKernelNode(&nodes[k], graph, k ? &nodes[k - 1] : nullptr, k ? 1 : 0, &kp[k]);
        }
        if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
            for (int pass = 0; pass < 2; pass++) {
                v.clear();
                for (int i = 0; i < R; i++) {
                    const double t0 = now_us(); sq = ++seq;
                    for (int k = 0; k < 3; k++) hipGraphExecKernelNodeSetParams(exec, nodes[k], &kp[k]);
                    hipGraphLaunch(exec, s);
                    while (*(volatile unsigned long long *)flag != seq) {}
                    v.push_back(now_us() - t0);
                }
            }
            report("(g3) the three launches as one hipGraph", v);
            v.clear();
            for (int i = 0; i < R; i++) {                       // without the parameter updates (a fixed graph): the launch alone
                const double t0 = now_us();
                hipGraphLaunch(exec, s);
                hipStreamSynchronize(s);
                v.push_back(now_us() - t0);
            }
            report("(g3s) the same graph, no updates, + sync", v);
            hipGraphExecDestroy(exec);
        } else printf("(g3) hipGraphInstantiate failed\n");
        hipGraphDestroy(graph);
    }
    // (c) resident kernel
    volatile unsigned long long *req = flag + 8, *ack = flag;
    *req = 0; *ack = 0;
    hipLaunchKernelGGL(resident_kernel, dim3(1), dim3(1), 0, s, req, ack, (unsigned long long)R);
    v.clear();
    for (unsigned long long i = 1; i <= (unsigned long long)R; i++) {
        const double t0 = now_us();
        *req = i;
        const double lim = t0 + 1e6;
        while (*ack != i && now_us() < lim) {}
        v.push_back(now_us() - t0);
    }
    hipStreamSynchronize(s);
    report("(c) resident kernel, pinned mailbox", v);
    return 0;
}
