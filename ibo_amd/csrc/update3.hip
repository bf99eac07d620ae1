// update3.hip -- LEFT-LOOKING trailing update of the batched two-level Cholesky (the marginal-likelihood grid,
// BASELINE config 5; linalg.hip: launch_cholesky_batched_left).
//
// update2.hip is right-looking: after every panel of 256 columns the whole trailing matrix takes one K = 256 update.
// Every 128 x 128 tile of C is then read and written once per panel -- 46 GB of the 108 GB a 64-theta grid at N = 4096
// moved through the fabric -- and each visit is a short K loop with a prologue (the tile from HBM) and an epilogue.
// Here the block columns [c0, c0 + width) of the NEXT panel are brought up to date in one go, just before they are
// factored:   C <- C - L[rows, 0 : c0] L[panel rows, 0 : c0]^T,   K = c0 (256 .. N - 256) columns deep.
// A tile of C is read once and written once per factorisation (8 GB per grid instead of 46), the K loop is 8 x longer
// on average, and both operands come from ONE packed copy of the finished block columns (chol_pack3_kernel) that
// grows by a panel per outer step:
//   Pk[((g NK8 + j) 64 + lane) 2 + h] = L[16 g + (lane & 15)][8 j + 4 h + (lane >> 4)],   NK8 = Npad / 8,
// a row-block's K range is contiguous (one aligned 16-byte load per lane and k8-step, scalar offsets only).
//
// Arithmetic: an element of C receives the terms -L_ik L_jk in ascending k, in the same groups of four (the k4-steps
// of v_mfma_f64_16x16x4) as in the right-looking order, where it received them panel by panel; storing and reloading
// the fp64 accumulator in between changes nothing.  There is no negated copy of the panel: the accumulators start as
// -C and the result is negated back -- every intermediate value is the exact negative of the right-looking one
// (round-to-nearest is symmetric).  So the factor is BIT-IDENTICAL to launch_cholesky_batched's (tested).
//
// Tile -> workgroup order: workgroups are dealt round-robin to the 8 XCDs (observed; speed only), so workgroup b gets
// entry (b & 7) C + (b >> 3) of the sequence "matrix 0's tiles, matrix 1's tiles, ..", C = its length / 8: one XCD works
// through whole matrices, whose two 128-row B strips (the panel's own rows, shared by ALL tiles of the matrix) then
// stream through that XCD's L2 once, and tiles (I, 0), (I, 1) -- same A strip -- are neighbours.
//
// Pad rows (>= nlive: identity rows whose factor entries left of the diagonal are exact zeros) are neither loaded
// nor multiplied nor stored: the y row the likelihood appends costs one 16-row block, not a 128-row tile.
#include "ibo_common.h"
#include <type_traits>
#include <vector>
#include <algorithm>

#define U3_NW 8
typedef double d2_t __attribute__((ext_vector_type(2)));
#define U3_KS 32                       // columns per LDS stage of B

typedef unsigned u3_v4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t u3_rsrc(const void *p, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)(bytes > 0x7fffffffu ? 0x7fffffffu : bytes), 0x00020000);
}
__device__ __forceinline__ double u3_lo(const u3_v4 &v) { return __hiloint2double((int)v.y, (int)v.x); }
__device__ __forceinline__ double u3_hi(const u3_v4 &v) { return __hiloint2double((int)v.w, (int)v.z); }

// rows [r0, Npad) x columns [c0, c0 + K) of L into the packed store (fragment order, see above)
__global__ __launch_bounds__(256) void chol_pack3_kernel(const double *__restrict__ L, int Npad, int r0, int c0, int K,
                                                         double *__restrict__ Pk, size_t lstride, size_t pstride)
{
    L += blockIdx.z * lstride; Pk += blockIdx.z * pstride;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)(Npad - r0) * K;
    if (e >= total) return;
    const int h = (int)(e & 1), lane = (int)((e >> 1) & 63), nk8 = K >> 3;
    const size_t gj = e >> 7;
    const int j = (int)(gj % nk8), g = (int)(gj / nk8);
    const int row = r0 + 16 * g + (lane & 15), col = c0 + 8 * j + 4 * h + (lane >> 4);
    const double v = L[(size_t)row * Npad + col];
    Pk[((((size_t)(row >> 4) * (Npad >> 3) + (col >> 3)) * 64 + lane) << 1) + h] = v;
}

// The ride-along's finished block columns [c0, c0 + K) of E^T-in-progress (Eout: row block i holds X_ij for j >= i and was never written left
// of its diagonal block) into the second half of a tall packed store: rows [0, rend), exact zeros where the block column lies left of the row's
// diagonal block -- a tall update's K range then needs no per-row start
__global__ __launch_bounds__(256) void chol_pack3e_kernel(const double *__restrict__ E, int Npad, int rend, int c0, int K, double *__restrict__ PkE)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)rend * K;
    if (e >= total) return;
    const int h = (int)(e & 1), lane = (int)((e >> 1) & 63), nk8 = K >> 3;
    const size_t gj = e >> 7;
    const int j = (int)(gj % nk8), g = (int)(gj / nk8);
    const int row = 16 * g + (lane & 15), col = c0 + 8 * j + 4 * h + (lane >> 4);
    const double v = (col >> 6) >= (row >> 6) ? E[(size_t)row * Npad + col] : 0.0;
    PkE[((((size_t)(row >> 4) * (Npad >> 3) + (col >> 3)) * 64 + lane) << 1) + h] = v;
}
int launch_chol_pack3e(const double *E, int Npad, int rend, int c0, int K, double *PkE, hipStream_t s)
{
    if (rend <= 0 || K <= 0) return 0;
    const size_t total = (size_t)rend * K;
    hipLaunchKernelGGL(chol_pack3e_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, E, Npad, rend, c0, K, PkE);
    return (int)hipGetLastError();
}

// The single live row-block below the last full 128-row tile (the likelihood's y row: 1 live row, 15 of pad) against ALL of
// the panel's column-blocks, one workgroup per matrix: wave w takes column-blocks 2 w and 2 w + 1 -- per k8-step one A and two
// B fragments straight from the packed store (no LDS, no barrier), four MFMAs -- with the loads eight steps ahead (a wave
// has 256 cycles of MFMA work per step, nothing like the latency of its loads).  As a 128-row tile the same block cost two
// workgroups half a tile time each: 2 of 5 tiles of the last panel.  Same MFMAs on the same operands per element: same bits.
__device__ __forceinline__ void u3_row_block(double *L, int Npad, int c0, int ncb, int g, const double *Pk, int wave, int lane, int kb8, int K)
{
    const int cbw = 2 * wave;
    if (cbw >= ncb) return;
    const int nk8s = Npad >> 3;
    const __amdgpu_buffer_rsrc_t rP = u3_rsrc(Pk, (size_t)Npad * Npad * sizeof(double));
    const unsigned lane16 = lane * 16;
    const bool two = cbw + 1 < ncb;
    double *Cw = L + (size_t)(16 * g + (lane >> 4)) * Npad + c0 + 16 * cbw + (lane & 15);
    d4_t acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; cb++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[cb][r] = (cb == 0 || two) ? -Cw[(size_t)(4 * r) * Npad + 16 * cb] : 0.0;
    const unsigned ba = (unsigned)(g * nk8s + kb8) * 1024u, bb0 = (unsigned)(((c0 >> 4) + cbw) * nk8s + kb8) * 1024u, bb1 = bb0 + (unsigned)nk8s * 1024u;
    u3_v4 R[8][3];
    auto fetch = [&](int j, u3_v4 (&F)[3]) {
        F[0] = __builtin_amdgcn_raw_buffer_load_b128(rP, lane16, ba + (unsigned)j * 1024u, 0);
        F[1] = __builtin_amdgcn_raw_buffer_load_b128(rP, lane16, bb0 + (unsigned)j * 1024u, 0);
        F[2] = __builtin_amdgcn_raw_buffer_load_b128(rP, lane16, bb1 + (unsigned)j * 1024u, 0);
    };
#pragma unroll
    for (int u = 0; u < 7; u++) fetch(u, R[u]);
    for (int j = 0; j < K / 8; j += 8) {                  // K is a multiple of 64
#pragma unroll
        for (int u = 0; u < 8; u++) {
            fetch(j + u + 7, R[(u + 7) & 7]);              // (past K: later columns' fragments or the bounds check's zeros, never used)
            const u3_v4 (&F)[3] = R[u];
            acc[0] = mfma_f64(u3_lo(F[0]), u3_lo(F[1]), acc[0]);
            acc[1] = mfma_f64(u3_lo(F[0]), u3_lo(F[2]), acc[1]);
            acc[0] = mfma_f64(u3_hi(F[0]), u3_hi(F[1]), acc[0]);
            acc[1] = mfma_f64(u3_hi(F[0]), u3_hi(F[2]), acc[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int cb = 0; cb < 2; cb++)
        if (cb == 0 || two) {
#pragma unroll
            for (int r = 0; r < 4; r++) Cw[(size_t)(4 * r) * Npad + 16 * cb] = -acc[cb][r];
        }
}

// C[rows >= c0][c0 .. c0 + 16 ncb) -= Pk-rows x Pk-rows^T over the columns k of the k8-steps [kb8, kb8 + K / 8).  The region
// starts ON the diagonal (first row = first column = c0): 128 x 128 tiles (I, J), J < ntc, J <= I; 8 waves = 4 (row pairs of 16-row
// blocks) x 2 (four column-blocks each).  Left-looking (launch_chol_update3): a panel's <= 256 columns, all finished columns
// (kb8 = 0, K = c0).  Right-looking (launch_chol_update2, the single-matrix fits beyond 2048 rows and the A/B path of the grid):
// the whole trailing matrix, the last panel's K = 256 columns.
__global__ __launch_bounds__(U3_NW * 64, 4) void chol_update3_kernel(double *L, int Npad, int c0, int ncb, int nlive_rb,
                                                                  const double *Pk, size_t lstride, size_t pstride,
                                                                  int tpm, int ntc, int batch, int chunk, int rowtile, int kb8, int K,
                                                                  const int4 *__restrict__ tasks, double *__restrict__ part)
{
    __shared__ __attribute__((aligned(16))) double lds_b[2][U3_KS / 8 * 8 * 128];      // [stage][k8-step][column-block][lane][2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // sequence entry of this workgroup (see the header): matrix m, tile t
    // (chunk = 0, fewer than 8 matrices: plain order -- neighbouring tiles on different XCDs find each other's strips in the
    // memory-side cache; whole-matrix chunks per XCD would leave most L2s without a matrix)
    // tasks (the SYRK use, launch_syrk3: C = P P^T of an upper triangular P): entry blockIdx.x is tile tasks[].x's piece of the K range
    // [kb8, kb8 + K / 8) k8-steps read from tasks[].y's fields; accumulators start from zero, the result is stored as it is -- into C, or
    // into slot `pslot` of `part` (128 x 128 doubles each) when the tile's K range is dealt in pieces
    int q = chunk ? (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    int pslot = 0;
    if (tasks) {
        const int4 tk = tasks[blockIdx.x];          // tile, first k8-step, columns, slot (0: C itself)
        q = tk.x; kb8 = tk.y; K = tk.z; pslot = tk.w;
    } else if ((chunk && (int)(blockIdx.x >> 3) >= chunk) || q >= batch * tpm) return;
    const int m = q / tpm, t = q - m * tpm;
    L += (size_t)m * lstride; Pk += (size_t)m * pstride;
    if (rowtile && t >= tpm - rowtile) {                 // the matrix's last entries: the lone live row-block below the full tiles,
        const int ri = t - (tpm - rowtile);              // 16 column-blocks (256 columns) per workgroup
        u3_row_block(L, Npad, c0 + 256 * ri, ncb - 16 * ri < 16 ? ncb - 16 * ri : 16, nlive_rb - 1, Pk, wave, lane, kb8, K);
        return;
    }
    // (0,0), (1,0), (1,1), (2,0), .. : row I holds min(I + 1, ntc) tiles
    int I, J;
    {
        const int tri = ntc * (ntc + 1) / 2;
        if (t < tri) { I = 0; int rem = t; while (rem > I) { rem -= I + 1; I++; } J = rem; }
        else { I = ntc + (t - tri) / ntc; J = (t - tri) % ntc; }
    }
    // A tile of the TALL matrix's second half (rows >= Npad: the ride-along's E, launch_cholesky_super) has nothing but exact zeros left of its
    // rows' diagonal block (chol_pack3e_kernel): its K range starts there -- a sixth of a super-panel's deep update is such products, and the
    // shortened tiles are the last ones dealt
    if (!tasks && c0 + 128 * I >= Npad) {
        const int ks = ((c0 + 128 * I - Npad) >> 6) << 6, d = ks - 8 * kb8;
        if (d > 0 && d < K) { kb8 += d >> 3; K -= d; }
    }
    const int wr = wave >> 1, wc = wave & 1;
    const int gA = ((c0 + 128 * I) >> 4) + 2 * wr;               // first of this wave's two row-blocks (numbered from row 0)
    const int gB = (c0 + 128 * J) >> 4;                          // first of the tile's eight column-blocks, as row-blocks of the panel
    const int cb0 = 8 * J + 4 * wc;                              // this wave's first column-block inside the panel
    const int nk8s = Npad >> 3;                                  // k8-steps per row-block of the packed store
    // (the packed store has Npad rows -- or, under the fit's ride-along, the 16 nlive_rb > Npad rows of the TALL matrix [A ; E], whose second
    // half packs E's finished columns: launch_cholesky_super)
    const size_t pbytes = (size_t)(16 * nlive_rb > Npad ? 16 * nlive_rb : Npad) * Npad * sizeof(double);
    const __amdgpu_buffer_rsrc_t rP = u3_rsrc(Pk, pbytes);
    const unsigned lane16 = lane * 16;
    // which of this wave's blocks exist: live rows only, the panel's columns only.  A wave with nothing to do runs the same
    // loop (it stages B; its own fragments are pad rows -- exact zeros -- or the bounds check's zeros) and stores nothing.
    const bool rowin[2] = {gA < nlive_rb, gA + 1 < nlive_rb};
    // (tile-local row-blocks 2 wr, 2 wr + 1 against column-blocks 4 wc ..: wave (wr < 2, wc = 1) of a diagonal tile is all upper triangle)
    const bool upper = I == J && wc == 1 && wr < 2;
    const int ni = (cb0 < ncb && !upper) ? (rowin[1] ? 2 : (rowin[0] ? 1 : 0)) : 0;      // live row-blocks of this wave

    // B staging: 32 fragments (k8-step f>>3, column-block f&7) of 1 KiB per stage, four per wave, in two halves
    auto fetch_b = [&](int st, int half, u3_v4 (&v)[2]) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int f = wave + 8 * (2 * half + u);
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rP, lane16, (unsigned)(((gB + (f & 7)) * nk8s + kb8 + st * (U3_KS / 8) + (f >> 3)) * 1024), 0);
        }
    };
    auto stash_b = [&](int b, int half, const u3_v4 (&v)[2]) {
#pragma unroll
        for (int u = 0; u < 2; u++) *(u3_v4 *)&lds_b[b][((wave + 8 * (2 * half + u)) * 64 + lane) * 2] = v[u];
    };
    const int nstage = K / U3_KS;                         // even: K is a multiple of 64
    u3_v4 vb[2];
    fetch_b(0, 0, vb); stash_b(0, 0, vb);
    fetch_b(0, 1, vb); stash_b(0, 1, vb);
    // A fragments: straight from the packed store, a ring of four register pairs, THREE k8-steps ahead -- the operand
    // strips stream from HBM (a row-block's K range is read once per tile pair), and a step of one wave lasts about
    // 4 x 16 MFMAs of pipe time, so three steps cover the memory latency with room to spare.
    // The loop body below is branch-free straight-line code (two stages = eight steps per trip, every LDS offset an
    // immediate, the prefetch of the stage after the last one reading harmless addresses): with a conditional load in the
    // body the compiler's s_waitcnt placement falls back to vmcnt(0) at the first use after it -- the MFMAs then wait
    // for the prefetch they have just issued (measured: MFMA pipe 71 % busy, and the same 52 TFLOP/s as update2.hip,
    // whose loop has that flaw, although the traffic had fallen from 108 to 42 GB).
    // accumulators <- MINUS the tile.  Element r of block (i, cb): row 16 (gA + i) + (lane>>4) + 4 r, column c0 + 16 (cb0 + cb) + (lane&15)
    d4_t acc[2][4];
    double *Cw = L + (size_t)(16 * gA + (lane >> 4)) * Npad + c0 + 16 * cb0 + (lane & 15);
    size_t ldc = Npad;
    if (pslot) {            // a piece's own 128 x 128 slot
        Cw = part + (size_t)(pslot - 1) * 16384 + (size_t)(32 * wr + (lane >> 4)) * 128 + 64 * wc + (lane & 15);
        ldc = 128;
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int cb = 0; cb < 4; cb++) {
            const bool in = i < ni && cb0 + cb < ncb && !tasks;
#pragma unroll
            for (int r = 0; r < 4; r++) acc[i][cb][r] = in ? -Cw[(size_t)(16 * i + 4 * r) * ldc + 16 * cb] : 0.0;
        }

    u3_v4 A[4][2];
    const unsigned abase0 = (unsigned)(gA * nk8s + kb8) * 1024u, abase1 = (unsigned)((gA + 1) * nk8s + kb8) * 1024u;
    // NI = live row-blocks of this wave (2; 1: the y row's block, whose neighbour is pad; 0: nothing -- pad rows, columns
    // beyond the panel, or the blocks of a diagonal tile that lie wholly above the diagonal): the loop exists in three
    // straight-line versions chosen once per wave, so the idle waves of a thin tile leave the MFMA pipe to their neighbours
    auto run = [&](auto ni_tag) {
        constexpr int NI = decltype(ni_tag)::value;
        auto fetch_a = [&](int j, u3_v4 (&Aj)[2]) {       // k8-step j (past K: later columns' fragments or the bounds check's zeros, never used)
            if (NI >= 1) Aj[0] = __builtin_amdgcn_raw_buffer_load_b128(rP, lane16, abase0 + (unsigned)j * 1024u, 0);
            if (NI >= 2) Aj[1] = __builtin_amdgcn_raw_buffer_load_b128(rP, lane16, abase1 + (unsigned)j * 1024u, 0);
        };
        fetch_a(0, A[0]);
        fetch_a(1, A[1]);
        fetch_a(2, A[2]);
        __syncthreads();
        // one k8-step: 8 NI MFMAs on the fragments in CUR and the four B fragment pairs of this wave's column-blocks
        auto mma = [&](const double *kb, int j8, const u3_v4 (&CUR)[2]) {
            if (NI == 0) return;
            d2_t b[4];
#pragma unroll
            for (int cb = 0; cb < 4; cb++) b[cb] = *(const d2_t *)&kb[((j8 * 8 + cb) * 64) * 2];
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    const double av = h ? u3_hi(CUR[i]) : u3_lo(CUR[i]);
#pragma unroll
                    for (int cb = 0; cb < 4; cb++) acc[i][cb] = mfma_f64(av, h ? b[cb].y : b[cb].x, acc[i][cb]);
                }
        };
        // a stage whose B fragments sit in LDS buffer BUF (a compile-time constant); it stages the next one into the other
        auto stage = [&](int st, auto buf_tag) {
            constexpr int BUF = decltype(buf_tag)::value;
            const double *kb = &lds_b[BUF][(4 * wc * 64 + lane) * 2];
            const int j = st * (U3_KS / 8);
            fetch_a(j + 3, A[3]); fetch_b(st + 1, 0, vb);
            mma(kb, 0, A[0]);
            __builtin_amdgcn_sched_barrier(0);
            fetch_a(j + 4, A[0]);
            mma(kb, 1, A[1]);
            stash_b(BUF ^ 1, 0, vb);
            __builtin_amdgcn_sched_barrier(0);
            fetch_a(j + 5, A[1]); fetch_b(st + 1, 1, vb);
            mma(kb, 2, A[2]);
            __builtin_amdgcn_sched_barrier(0);
            fetch_a(j + 6, A[2]);
            mma(kb, 3, A[3]);
            stash_b(BUF ^ 1, 1, vb);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        };
        for (int st = 0; st < nstage; st += 2) {
            stage(st, std::integral_constant<int, 0>{});
            stage(st + 1, std::integral_constant<int, 1>{});
        }
    };
    if (ni == 0) { run(std::integral_constant<int, 0>{}); return; }      // this wave only helps staging B
    run(std::integral_constant<int, 2>{});                   // (ni = 1, the y row's block: its neighbour is pad -- zeros -- and is not stored;
                                                             //  a third, one-block version of the loop made hipcc spill 340 B/lane)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int cb = 0; cb < 4; cb++) {
            if (i < ni && cb0 + cb < ncb) {
#pragma unroll
                for (int r = 0; r < 4; r++) Cw[(size_t)(16 * i + 4 * r) * ldc + 16 * cb] = tasks ? acc[i][cb][r] : -acc[i][cb][r];
            }
        }
}

// Pk: Npad * Npad doubles per matrix, `pstride` apart.  nlive: rows >= nlive are identity pad (never touched).
int launch_chol_pack3(const double *L, int Npad, int r0, int c0, int K, int batch, size_t lstride, double *Pk, size_t pstride,
                      hipStream_t s)
{
    if (r0 >= Npad || K <= 0) return 0;
    const size_t total = (size_t)(Npad - r0) * K;
    hipLaunchKernelGGL(chol_pack3_kernel, dim3((unsigned)((total + 255) / 256), 1, batch), dim3(256), 0, s, L, Npad, r0, c0, K, Pk,
                       lstride, pstride);
    return (int)hipGetLastError();
}

// region: rows >= c0 (live ones), columns [c0, c0 + width), updated with the packed columns [kbeg, kend) (multiples of 64)
int launch_chol_update3_range(double *L, int Npad, int c0, int width, int kbeg, int kend, int nlive, int batch, size_t lstride,
                                 const double *Pk, size_t pstride, hipStream_t s)
{
    if (width <= 0 || kend <= kbeg) return 0;
    const int nlive_rb = (nlive + 15) / 16;
    const int rows = 16 * nlive_rb - c0;                     // live rows at and below the region's first row
    if (rows <= 0) return 0;
    const int ncb = width / 16;
    // 16 live rows below the last full tile (the likelihood's y row when N is a multiple of 128): one row-block workgroup
    // per matrix instead of a 128-row tile per tile column
    const int rowtile = (rows % 128 == 16 && rows > 128) ? (ncb + 15) / 16 : 0;
    const int nrt = rowtile ? rows / 128 : (rows + 127) / 128;
    int ntc = (width + 127) / 128;
    if (ntc > nrt) ntc = nrt;                                // tile columns beyond the last tile row lie above the diagonal
    const int tpm = ntc * (ntc + 1) / 2 + (nrt - ntc) * ntc + rowtile;
    const long long total = (long long)tpm * batch;
    const int chunk = batch >= 8 ? (int)((total + 7) / 8) : 0;
    hipLaunchKernelGGL(chol_update3_kernel, dim3((unsigned)(chunk ? 8 * chunk : total)), dim3(U3_NW * 64), 0, s, L, Npad, c0, ncb, nlive_rb, Pk,
                       lstride, pstride, tpm, ntc, batch, chunk, rowtile, kbeg / 8, kend - kbeg, (const int4 *)nullptr, (double *)nullptr);
    return (int)hipGetLastError();
}

int launch_chol_update3(double *L, int Npad, int c0, int width, int nlive, int batch, size_t lstride, const double *Pk,
                        size_t pstride, hipStream_t s)
{
    if (c0 <= 0) return 0;
    return launch_chol_update3_range(L, Npad, c0, width, 0, c0, nlive, batch, lstride, Pk, pstride, s);
}

// RIGHT-LOOKING use of the same kernel: after the block columns [p0, pend) are finished, the whole trailing matrix takes their
// K = 64 (pend - p0) update.  ws: Npad * Npad doubles per matrix (`wstride` apart) -- the packed store, of which only the
// panel's columns, rows >= 64 pend, are written and read.  Lpanel: where the finished columns live (out-of-place factorisations).
int launch_chol_update2(double *L, int Npad, int p0, int pend, int batch, size_t lstride, double *ws, size_t wstride,
                        hipStream_t s, const double *Lpanel)
{
    if (!Lpanel) Lpanel = L;
    const int r0 = 64 * pend;
    if (r0 >= Npad) return 0;
    int rc = launch_chol_pack3(Lpanel, Npad, r0, 64 * p0, 64 * (pend - p0), batch, lstride, ws, wstride, s);
    if (rc) return rc;
    return launch_chol_update3_range(L, Npad, r0, Npad - r0, 64 * p0, 64 * pend, Npad, batch, lstride, ws, wstride, s);
}


// ---- C = P P^T for an UPPER triangular P (K^-1 = W^T W with P = W^T as the ride-along leaves it: ibo_nlml_grad), lower 128 x 128 tiles of C.
// Tile row I starts its K range at its own first row (everything left of it is zero), so tile (0, 0) is N deep and the last tiles 64: dealt whole,
// the first row's tiles ARE the launch (561 tiles at N = 4096 are one round of the chip).  Here K ranges beyond `piece` columns go out in pieces
// (pieces of a tile sum in a fixed order afterwards: syrk3_sum_kernel), longest pieces first.
// P's fragments: as chol_pack3_kernel, with exact zeros where the block column lies left of the row's diagonal block (the ride-along never wrote
// there); only columns from the row's 128-row tile on are ever read.
__global__ __launch_bounds__(256) void syrk3_pack_kernel(const double *__restrict__ P, int Npad, double *__restrict__ Pk)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int nk8 = Npad >> 3;
    if (e >= (size_t)Npad * Npad) return;
    const int h = (int)(e & 1), lane = (int)((e >> 1) & 63);
    const size_t gj = e >> 7;
    const int j = (int)(gj % nk8), g = (int)(gj / nk8);
    const int row = 16 * g + (lane & 15), col = 8 * j + 4 * h + (lane >> 4);
    if (col < (row & ~127)) return;
    Pk[e] = (col >> 6) >= (row >> 6) ? P[(size_t)row * Npad + col] : 0.0;
}
// C tile += its later pieces, in piece order (sums[]: tile index, first slot, number of slots)
__global__ __launch_bounds__(256) void syrk3_sum_kernel(double *__restrict__ C, int Npad, const double *__restrict__ part, const int4 *__restrict__ sums)
{
    const int4 sm = sums[blockIdx.x];            // x: tile row I, y: tile column J, z: first slot, w: slots
    const int I = sm.x, J = sm.y;
    for (int e = blockIdx.y * 2048 + threadIdx.x; e < (int)(blockIdx.y + 1) * 2048; e += 256) {       // (eight workgroups per tile)
        const int r = e >> 7, c = e & 127;
        const int row = 128 * I + r, col = 128 * J + c;
        if (row >= Npad || col >= Npad || (I == J && (c >> 6) > (r >> 6))) continue;      // (a diagonal tile's upper 64-block is never formed)
        double v = C[(size_t)row * Npad + col];
        for (int k = 0; k < sm.w; k++) v += part[(size_t)(sm.z + k) * 16384 + e];
        C[(size_t)row * Npad + col] = v;
    }
}

// host side of the task list for one Npad (kept by the caller's workspace): tasks (int4 per workgroup), sums (int4 per tile dealt in pieces)
void syrk3_plan(int Npad, int piece, std::vector<int> &tasks, std::vector<int> &sums, int *nslots)
{
    const int nt = (Npad + 127) / 128;
    struct T { int len, t, kb8, pslot; };
    std::vector<T> all;
    int slot = 0, t = 0;
    for (int I = 0; I < nt; I++)
        for (int J = 0; J <= I; J++, t++) {
            const int k0 = 128 * I, len = Npad - k0;                 // (a multiple of 64)
            const int np = len > piece + piece / 4 ? (len + piece - 1) / piece : 1;
            const int per = ((len / np + 63) / 64) * 64;
            int first = -1, cnt = 0;
            for (int p = 0, k = k0; k < Npad; p++, k += per) {
                const int l = k + per <= Npad ? per : Npad - k;
                int ps = 0;
                if (p > 0) { ps = ++slot; if (first < 0) first = slot - 1; cnt++; }
                all.push_back(T{l, t, k / 8, ps});
            }
            if (cnt) { sums.push_back(I); sums.push_back(J); sums.push_back(first); sums.push_back(cnt); }
        }
    std::stable_sort(all.begin(), all.end(), [](const T &a, const T &b) { return a.len > b.len; });
    for (const T &a : all) { tasks.push_back(a.t); tasks.push_back(a.kb8); tasks.push_back(a.len); tasks.push_back(a.pslot); }
    *nslots = slot;
}

// P: Npad x Npad row-major upper triangular (blocks left of the diagonal blocks never read); Pk: Npad^2 doubles of scratch; C: the product's lower tiles;
// tasks_dev / sums_dev / part: syrk3_plan's lists on the device and nslots x 16384 doubles
int launch_syrk3(const double *P, double *Pk, double *C, int Npad, const int *tasks_dev, int ntasks, const int *sums_dev, int nsums, double *part,
                 hipStream_t s)
{
    const size_t total = (size_t)Npad * Npad;
    hipLaunchKernelGGL(syrk3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, P, Npad, Pk);
    const int nt = (Npad + 127) / 128;
    hipLaunchKernelGGL(chol_update3_kernel, dim3((unsigned)ntasks), dim3(U3_NW * 64), 0, s, C, Npad, 0, Npad / 16, Npad / 16, (const double *)Pk,
                       (size_t)0, (size_t)0, nt * (nt + 1) / 2, nt, 1, 0, 0, 0, 0, (const int4 *)tasks_dev, part);
    if (nsums) hipLaunchKernelGGL(syrk3_sum_kernel, dim3((unsigned)nsums, 8), dim3(256), 0, s, C, Npad, (const double *)part, (const int4 *)sums_dev);
    return (int)hipGetLastError();
}

void ibo_touch_update3() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, (const void *)chol_pack3_kernel); }     // (see small2.hip: ibo_touch_small2)
