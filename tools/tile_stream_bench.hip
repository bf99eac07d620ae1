// tile_stream_bench.hip -- what moving the tiles of one right-looking step costs by itself, and whether the row-major layout is to blame:
// 256 workgroups of 512 threads walk the lower-triangle tiles (i, k) of an N x N matrix; per tile they read the 64 x 64 blocks X_i, X_k
// (one block column) and C(i, k) and write C(i, k) back -- the traffic of chol_pipe8_kernel's tile workgroups without their MFMAs --
//   (a) row-major, row stride N (a tile = 64 pieces of 512 bytes, N * 8 bytes apart);
//   (b) the same bytes with every 64 x 64 block contiguous (32 KiB).
// Build: hipcc -O3 --offload-arch=gfx950 tools/tile_stream_bench.hip -o tools/tile_stream_bench ;  run: tools/tile_stream_bench [N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2_t __attribute__((ext_vector_type(2)));
template <bool BLOCKED, bool WITH_C>
__global__ __launch_bounds__(512) void walk(double *__restrict__ A, const double *__restrict__ X, int N, int ntile, double *sink)
{
    const int nb = N / 64, t = threadIdx.x;
    d2_t s = {0.0, 0.0};
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        int k = 0, rem = tile;
        while (rem >= nb - k) { rem -= nb - k; k++; }
        const int i = k + rem;
        const size_t ld = BLOCKED ? 64 : (size_t)N;
        const double *Xi = BLOCKED ? X + (size_t)i * 4096 : X + (size_t)i * 64 * N;
        const double *Xk = BLOCKED ? X + (size_t)k * 4096 : X + (size_t)k * 64 * N;
        double *C = BLOCKED ? A + ((size_t)i * nb + k) * 4096 : A + (size_t)i * 64 * N + k * 64;
        d2_t va[4], vb[4], vc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            va[u] = *(const d2_t *)(Xi + (size_t)(16 * u + (t >> 5)) * ld + (t & 31) * 2);
            vb[u] = *(const d2_t *)(Xk + (size_t)(16 * u + (t >> 5)) * ld + (t & 31) * 2);
            if (WITH_C) vc[u] = *(const d2_t *)(C + (size_t)(16 * u + (t >> 5)) * ld + (t & 31) * 2);
            else vc[u] = d2_t{0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            vc[u] = vc[u] + va[u] * vb[u];
            if (WITH_C) *(d2_t *)(C + (size_t)(16 * u + (t >> 5)) * ld + (t & 31) * 2) = vc[u];
            else s += vc[u];
        }
    }
    if (!WITH_C && s.x == 1.2345e300) sink[0] = s.y;
}
template <bool BLOCKED, bool WITH_C>
static void run(const char *what, double *A, double *X, int N, double *sink)
{
    const int nb = N / 64, ntile = nb * (nb + 1) / 2;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((walk<BLOCKED, WITH_C>), dim3(256), dim3(512), 0, 0, A, X, N, ntile, sink);
    hipEventRecord(e0);
    const int R = 20;
    for (int r = 0; r < R; r++) hipLaunchKernelGGL((walk<BLOCKED, WITH_C>), dim3(256), dim3(512), 0, 0, A, X, N, ntile, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= R;
    const double bytes = (double)ntile * 32768.0 * (WITH_C ? 4 : 2);
    printf("%-44s %4d tiles  %7.1f us  %5.2f us per tile and CU  %6.2f TB/s (L2-side, %s)\n", what, ntile, ms * 1e3, ms * 1e3 * 256 / ntile,
           bytes / ms / 1e9, WITH_C ? "X_i + X_k + C in + C out" : "X_i + X_k only");
}
int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 4096;
    double *A, *X, *sink;
    hipMalloc(&A, (size_t)N * N * 8); hipMalloc(&X, (size_t)N * N * 8); hipMalloc(&sink, 64);
    hipMemset(A, 0, (size_t)N * N * 8); hipMemset(X, 0, (size_t)N * N * 8);
    printf("N = %d\n", N);
    run<false, true>("row-major, stride N", A, X, N, sink);
    run<true, true>("64 x 64 blocks contiguous", A, X, N, sink);
    run<false, false>("row-major, operands only", A, X, N, sink);
    run<true, false>("blocks contiguous, operands only", A, X, N, sink);
    return 0;
}
