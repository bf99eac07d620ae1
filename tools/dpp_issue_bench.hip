// Issue cost of fp64 VALU forms on one wave (gfx950): plain v_fma_f64, v_fmac_f64_dpp row_newbcast with and
// without the s_nop the DPP read hazard asks for, v_mov_b64_dpp, v_readlane pairs.
//   hipcc --offload-arch=gfx950 -O3 tools/dpp_issue_bench.hip -o tools/dpp_issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
__global__ void bench(double *p, unsigned long long *out)
{
    double a0 = p[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double x = a0 * 0.5, y = a0 * 0.25;
    unsigned long long t[11];
    auto T = [&](int i) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory"); t[i] = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    T(0);   // 128 independent plain fmac (8 accumulators round robin)
    REP16(asm volatile("v_fmac_f64 %0, %8, %9\n\tv_fmac_f64 %1, %8, %9\n\tv_fmac_f64 %2, %8, %9\n\tv_fmac_f64 %3, %8, %9\n\tv_fmac_f64 %4, %8, %9\n\tv_fmac_f64 %5, %8, %9\n\tv_fmac_f64 %6, %8, %9\n\tv_fmac_f64 %7, %8, %9"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
    T(1);   // 128 dependent plain fmac
    REP16(asm volatile("v_fmac_f64 %0, %1, %2\n\tv_fmac_f64 %0, %1, %2\n\tv_fmac_f64 %0, %1, %2\n\tv_fmac_f64 %0, %1, %2\n\tv_fmac_f64 %0, %1, %2\n\tv_fmac_f64 %0, %1, %2\n\tv_fmac_f64 %0, %1, %2\n\tv_fmac_f64 %0, %1, %2"
        : "+v"(a0) : "v"(x), "v"(y));)
    T(2);   // 128 independent fmac_dpp, no nop
#define D " row_newbcast:3 row_mask:0xf bank_mask:0xf"
    REP16(asm volatile("v_fmac_f64_dpp %0, %8, %9" D "\n\tv_fmac_f64_dpp %1, %8, %9" D "\n\tv_fmac_f64_dpp %2, %8, %9" D "\n\tv_fmac_f64_dpp %3, %8, %9" D "\n\tv_fmac_f64_dpp %4, %8, %9" D "\n\tv_fmac_f64_dpp %5, %8, %9" D "\n\tv_fmac_f64_dpp %6, %8, %9" D "\n\tv_fmac_f64_dpp %7, %8, %9" D
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
    T(3);   // 128 independent fmac_dpp, each behind s_nop 1
    REP16(asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %8, %9" D "\n\ts_nop 1\n\tv_fmac_f64_dpp %1, %8, %9" D "\n\ts_nop 1\n\tv_fmac_f64_dpp %2, %8, %9" D "\n\ts_nop 1\n\tv_fmac_f64_dpp %3, %8, %9" D "\n\ts_nop 1\n\tv_fmac_f64_dpp %4, %8, %9" D "\n\ts_nop 1\n\tv_fmac_f64_dpp %5, %8, %9" D "\n\ts_nop 1\n\tv_fmac_f64_dpp %6, %8, %9" D "\n\ts_nop 1\n\tv_fmac_f64_dpp %7, %8, %9" D
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
    T(4);   // 128 x (2 readlane + fma with SGPR pair)
    REP16(asm volatile(
        "v_readlane_b32 s20, %8, 3\n\tv_readlane_b32 s21, %9, 3\n\tv_fma_f64 %0, s[20:21], %10, %0\n\t"
        "v_readlane_b32 s22, %8, 4\n\tv_readlane_b32 s23, %9, 4\n\tv_fma_f64 %1, s[22:23], %10, %1\n\t"
        "v_readlane_b32 s20, %8, 5\n\tv_readlane_b32 s21, %9, 5\n\tv_fma_f64 %2, s[20:21], %10, %2\n\t"
        "v_readlane_b32 s22, %8, 6\n\tv_readlane_b32 s23, %9, 6\n\tv_fma_f64 %3, s[22:23], %10, %3\n\t"
        "v_readlane_b32 s20, %8, 7\n\tv_readlane_b32 s21, %9, 7\n\tv_fma_f64 %4, s[20:21], %10, %4\n\t"
        "v_readlane_b32 s22, %8, 8\n\tv_readlane_b32 s23, %9, 8\n\tv_fma_f64 %5, s[22:23], %10, %5\n\t"
        "v_readlane_b32 s20, %8, 9\n\tv_readlane_b32 s21, %9, 9\n\tv_fma_f64 %6, s[20:21], %10, %6\n\t"
        "v_readlane_b32 s22, %8, 10\n\tv_readlane_b32 s23, %9, 10\n\tv_fma_f64 %7, s[22:23], %10, %7"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
        : "v"(__double2loint(x)), "v"(__double2hiint(x)), "v"(y) : "s20", "s21", "s22", "s23");)
    T(5);   // 128 dependent fmac_dpp (acc is also the DPP source), s_nop 1 each
    REP16(asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1" D "\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %0, %1" D "\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %0, %1" D "\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %0, %1" D "\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %0, %1" D "\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %0, %1" D "\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %0, %1" D "\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %0, %1" D
        : "+v"(a0) : "v"(y));)
    T(6);   // 128 independent plain v_mul_f64 (VOP3)
    REP16(asm volatile("v_mul_f64 %0, %8, %9\n\tv_mul_f64 %1, %8, %9\n\tv_mul_f64 %2, %8, %9\n\tv_mul_f64 %3, %8, %9\n\tv_mul_f64 %4, %8, %9\n\tv_mul_f64 %5, %8, %9\n\tv_mul_f64 %6, %8, %9\n\tv_mul_f64 %7, %8, %9"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
    T(7);   // 128 x (s_nop 0 + dependent fmac)
    REP16(asm volatile("s_nop 0\n\tv_fmac_f64 %0, %1, %2\n\ts_nop 0\n\tv_fmac_f64 %0, %1, %2\n\ts_nop 0\n\tv_fmac_f64 %0, %1, %2\n\ts_nop 0\n\tv_fmac_f64 %0, %1, %2\n\ts_nop 0\n\tv_fmac_f64 %0, %1, %2\n\ts_nop 0\n\tv_fmac_f64 %0, %1, %2\n\ts_nop 0\n\tv_fmac_f64 %0, %1, %2\n\ts_nop 0\n\tv_fmac_f64 %0, %1, %2"
        : "+v"(a0) : "v"(x), "v"(y));)
    T(8);   // 128 x (s_setprio 0 + dependent fmac)
    REP16(asm volatile("s_setprio 0\n\tv_fmac_f64 %0, %1, %2\n\ts_setprio 0\n\tv_fmac_f64 %0, %1, %2\n\ts_setprio 0\n\tv_fmac_f64 %0, %1, %2\n\ts_setprio 0\n\tv_fmac_f64 %0, %1, %2\n\ts_setprio 0\n\tv_fmac_f64 %0, %1, %2\n\ts_setprio 0\n\tv_fmac_f64 %0, %1, %2\n\ts_setprio 0\n\tv_fmac_f64 %0, %1, %2\n\ts_setprio 0\n\tv_fmac_f64 %0, %1, %2"
        : "+v"(a0) : "v"(x), "v"(y));)
    T(9);   // 128 x (s_mov_b32 + dependent fmac)
    REP16(asm volatile("s_mov_b32 s20, 0\n\tv_fmac_f64 %0, %1, %2\n\ts_mov_b32 s20, 0\n\tv_fmac_f64 %0, %1, %2\n\ts_mov_b32 s20, 0\n\tv_fmac_f64 %0, %1, %2\n\ts_mov_b32 s20, 0\n\tv_fmac_f64 %0, %1, %2\n\ts_mov_b32 s20, 0\n\tv_fmac_f64 %0, %1, %2\n\ts_mov_b32 s20, 0\n\tv_fmac_f64 %0, %1, %2\n\ts_mov_b32 s20, 0\n\tv_fmac_f64 %0, %1, %2\n\ts_mov_b32 s20, 0\n\tv_fmac_f64 %0, %1, %2"
        : "+v"(a0) : "v"(x), "v"(y) : "s20");)
    T(10);
    p[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) for (int i = 0; i < 11; i++) out[i] = t[i];
}
int main()
{
    double *p; unsigned long long *o, h[11];
    hipMalloc(&p, 64 * 8); hipMalloc(&o, 88); hipMemset(p, 0, 64 * 8);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(bench, dim3(1), dim3(64), 0, 0, p, o);
        hipMemcpy(h, o, 88, hipMemcpyDeviceToHost);
    }
    const char *nm[10] = {"128 independent v_fmac_f64", "128 dependent v_fmac_f64", "128 independent v_fmac_f64_dpp", "128 independent s_nop 1 + v_fmac_f64_dpp",
                         "128 x (2 v_readlane + v_fma_f64 sgpr)", "128 dependent s_nop 1 + v_fmac_f64_dpp", "128 independent v_mul_f64",
                          "128 x (s_nop 0 + dependent v_fmac_f64)", "128 x (s_setprio 0 + dependent v_fmac_f64)", "128 x (s_mov_b32 + dependent v_fmac_f64)"};
    for (int i = 0; i < 10; i++) printf("%-45s %6llu ticks  = %.1f per group\n", nm[i], h[i + 1] - h[i], (h[i + 1] - h[i]) / 128.0);
    return 0;
}
