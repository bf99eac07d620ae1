// update2.hip -- the trailing update of the two-level blocked Cholesky (linalg.hip: launch_cholesky_batched,
// panel = P block columns), second design:   C <- C - A B^T,   A = L[i rows][panel columns], B = L[k rows][panel columns],
// for every block of the trailing matrix at or below the diagonal, K = 64 P (256) columns deep.  It carries
// nearly all the flops of a large factorisation (N = 4096: the marginal-likelihood grid, BASELINE config 5).
//
// The first kernel (chol_update_kernel) works on 64 x 64 tiles with both operand strips copied through LDS by the
// workgroup's own VALU: one LDS fragment read per MFMA, 2.6 address / copy / negation instructions per MFMA on the
// pipe the fp64 MFMAs need (every VALU instruction beside them costs ~12 pipe cycles), 38 TFLOP/s.  Here:
//   * the panel is packed ONCE per outer step into MFMA fragment order (pack_panel_kernel) -- twice, a negated
//     copy for the A role, a plain one for the B role -- so an operand fragment is one aligned 16-byte load per
//     lane, with no index arithmetic, straight from L2;
//   * a workgroup of 8 waves owns a 128 x 128 tile, two workgroups per CU (while one loads or stores its tile --
//     the matrices stream from HBM -- the other computes): wave (wr, wc) keeps 2 row-blocks x 4 column-blocks of
//     16 x 16 accumulators (64 VGPRs); its A fragments come directly from the packed panel by buffer_load_dwordx4
//     with a scalar offset, two 8-column steps ahead (a ring of three: the panel is streamed, not cache-resident),
//     the B fragments of the tile's 8 column-blocks are staged in LDS once per 32-column stage and read back with
//     one ds_read_b128 per 4 MFMAs -- 0.25 LDS reads, 0.125 global loads and no VALU instruction per MFMA in the loop;
//   * accumulators start as the tile itself and the k4-steps run in ascending order, exactly the order of the
//     first kernel: results are bit-identical to it (tested).
// Blocks of a diagonal tile that lie strictly above the diagonal are computed and stored like the rest: the strict
// upper triangle of the matrix being factored is scratch for every caller (abi.hip: L_upper_dirty).
#include "ibo_common.h"

#define U2_NW 8
#define U2_TM 128
#define U2_TN 128
#define U2_KS 32                       // columns per LDS stage of B

typedef unsigned u2_v4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t u2_rsrc(const void *p, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)(bytes > 0x7fffffffu ? 0x7fffffffu : bytes), 0x00020000);
}
__device__ __forceinline__ double u2_lo(const u2_v4 &v) { return __hiloint2double((int)v.y, (int)v.x); }
__device__ __forceinline__ double u2_hi(const u2_v4 &v) { return __hiloint2double((int)v.w, (int)v.z); }

// Fragment order of the panel rows below the panel (rows r0 .. Npad-1, columns c0 .. c0+K-1 of L):
//   P[((g nk8 + j) 64 + lane) 2 + h] = L[r0 + 16 g + (lane&15)][c0 + 8 j + 4 h + (lane>>4)],  nk8 = K / 8;
// PA receives the negated values.
__global__ __launch_bounds__(256) void pack_panel_kernel(const double *__restrict__ L, int Npad, int r0, int c0, int K,
                                                         double *__restrict__ PA, double *__restrict__ PB, size_t lstride,
                                                         size_t pstride)
{
    L += blockIdx.z * lstride; PA += blockIdx.z * pstride; PB += blockIdx.z * pstride;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)(Npad - r0) * K;
    if (e >= total) return;
    const int h = (int)(e & 1), lane = (int)((e >> 1) & 63), nk8 = K >> 3;
    const size_t gj = e >> 7;
    const int j = (int)(gj % nk8), g = (int)(gj / nk8);
    const double v = L[(size_t)(r0 + 16 * g + (lane & 15)) * Npad + c0 + 8 * j + 4 * h + (lane >> 4)];
    PA[e] = -v;
    PB[e] = v;
}

__global__ __launch_bounds__(U2_NW * 64, 4) void chol_update2_kernel(double *L, int Npad, int r0, int K, const double *PA,
                                                                  const double *PB, size_t lstride, size_t pstride)
{
    __shared__ __attribute__((aligned(16))) double lds_b[2][U2_KS / 8 * 8 * 128];      // [stage][k8-step][column-block][lane][2]
    L += blockIdx.z * lstride; PA += blockIdx.z * pstride; PB += blockIdx.z * pstride;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = Npad - r0;                                     // trailing size
    // tile (I, J), J <= I: rows r0 + 128 I .., columns r0 + 128 J ..
    int I = 0, rem = blockIdx.x;
    while (rem > I) { rem -= I + 1; I++; }
    const int J = rem;
    const int wr = wave >> 1, wc = wave & 1;
    const int gA = (U2_TM * I + 32 * wr) >> 4;                   // first of this wave's two row-blocks (trailing numbering)
    const int gB = (U2_TN * J) >> 4;                             // first of the tile's eight column-blocks
    const int nrb = T >> 4, nk8 = K >> 3;
    const size_t pbytes = (size_t)T * K * sizeof(double);
    const __amdgpu_buffer_rsrc_t rA = u2_rsrc(PA, pbytes), rB = u2_rsrc(PB, pbytes);
    const unsigned lane16 = lane * 16;

    // accumulators <- the tile.  Element r of block (i, cb): row 16 (gA + i) + (lane>>4) + 4 r, column 16 (gB + 4 wc + cb) + (lane&15)
    d4_t acc[2][4];
    double *Cw = L + (size_t)(r0 + 16 * gA + (lane >> 4)) * Npad + r0 + 16 * (gB + 4 * wc) + (lane & 15);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int cb = 0; cb < 4; cb++) {
            const bool in = gA + i < nrb && gB + 4 * wc + cb < nrb;
#pragma unroll
            for (int r = 0; r < 4; r++) acc[i][cb][r] = in ? Cw[(size_t)(16 * i + 4 * r) * Npad + 16 * cb] : 0.0;
        }

    // B staging: 32 fragments (k8-step f>>3, column-block f&7) of 1 KiB per stage, four per wave, fetched and
    // written to the idle LDS buffer in two halves (8 staging registers instead of 16)
    auto fetch_b = [&](int st, int half, u2_v4 (&v)[2]) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int f = wave + 8 * (2 * half + u);
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rB, lane16, (unsigned)(((gB + (f & 7)) * nk8 + st * (U2_KS / 8) + (f >> 3)) * 1024), 0);
        }
    };
    auto stash_b = [&](int b, int half, const u2_v4 (&v)[2]) {
#pragma unroll
        for (int u = 0; u < 2; u++) *(u2_v4 *)&lds_b[b][((wave + 8 * (2 * half + u)) * 64 + lane) * 2] = v[u];
    };
    const int nstage = K / U2_KS;
    u2_v4 vb[2];
    fetch_b(0, 0, vb); stash_b(0, 0, vb);
    fetch_b(0, 1, vb); stash_b(0, 1, vb);
    u2_v4 A0[2], A1[2], A2[2];
    const unsigned abase0 = (unsigned)(gA * nk8) * 1024u, abase1 = (unsigned)((gA + 1) * nk8) * 1024u;
    auto fetch_a = [&](int j, u2_v4 (&A)[2]) {            // k8-step j of the panel (past its end: the next row-block's data or the bounds check's zeros, never used)
        A[0] = __builtin_amdgcn_raw_buffer_load_b128(rA, lane16, abase0 + (unsigned)j * 1024u, 0);
        A[1] = __builtin_amdgcn_raw_buffer_load_b128(rA, lane16, abase1 + (unsigned)j * 1024u, 0);
    };
    fetch_a(0, A0);
    fetch_a(1, A1);
    __syncthreads();
    auto step = [&](int st, int j8, const u2_v4 (&CUR)[2], u2_v4 (&NXT2)[2]) {
        fetch_a(st * (U2_KS / 8) + j8 + 2, NXT2);
        const double *kb = &lds_b[st & 1][(4 * wc * 64 + lane) * 2];                    // this wave's four column-blocks
        const bool more = st + 1 < nstage;
        if (more && (j8 == 0 || j8 == 2)) fetch_b(st + 1, j8 >> 1, vb);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            double bf[4];
#pragma unroll
            for (int cb = 0; cb < 4; cb++) bf[cb] = kb[((j8 * 8 + cb) * 64) * 2 + h];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const double av = h ? u2_hi(CUR[i]) : u2_lo(CUR[i]);
#pragma unroll
                for (int cb = 0; cb < 4; cb++) acc[i][cb] = mfma_f64(av, bf[cb], acc[i][cb]);
            }
        }
        if (more && (j8 == 1 || j8 == 3)) stash_b((st + 1) & 1, j8 >> 1, vb);
        // keep every step's loads where they are written: left to itself the scheduler sinks the A fetches down to
        // their first use two steps later, which turns the ring of three into no prefetch at all
        __builtin_amdgcn_sched_barrier(0);
    };
    // twelve k8-steps = three stages per trip, so that the ring of three A registers returns to its start
    for (int st = 0; st < nstage; st += 3) {
        step(st, 0, A0, A2); step(st, 1, A1, A0); step(st, 2, A2, A1); step(st, 3, A0, A2);
        __syncthreads();
        if (st + 1 >= nstage) break;
        step(st + 1, 0, A1, A0); step(st + 1, 1, A2, A1); step(st + 1, 2, A0, A2); step(st + 1, 3, A1, A0);
        __syncthreads();
        if (st + 2 >= nstage) break;
        step(st + 2, 0, A2, A1); step(st + 2, 1, A0, A2); step(st + 2, 2, A1, A0); step(st + 2, 3, A2, A1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int cb = 0; cb < 4; cb++) {
            if (gA + i < nrb && gB + 4 * wc + cb < nrb) {
#pragma unroll
                for (int r = 0; r < 4; r++) Cw[(size_t)(16 * i + 4 * r) * Npad + 16 * cb] = acc[i][cb][r];
            }
        }
}

// ws: 2 T K doubles per matrix (T = Npad - 64 pend), `wstride` doubles apart
int launch_chol_update2(double *L, int Npad, int p0, int pend, int batch, size_t lstride, double *ws, size_t wstride,
                        hipStream_t s, const double *Lpanel)
{
    if (!Lpanel) Lpanel = L;                 // out-of-place factorisations keep the finished block columns in another matrix
    const int r0 = 64 * pend, c0 = 64 * p0, K = 64 * (pend - p0), T = Npad - r0;
    if (T <= 0) return 0;
    double *PA = ws, *PB = ws + (size_t)T * K;
    const size_t total = (size_t)T * K;
    hipLaunchKernelGGL(pack_panel_kernel, dim3((unsigned)((total + 255) / 256), 1, batch), dim3(256), 0, s, Lpanel, Npad, r0, c0, K, PA, PB,
                       lstride, wstride);
    const int nI = (T + U2_TM - 1) / U2_TM;
    const int tiles = nI * (nI + 1) / 2;
    hipLaunchKernelGGL(chol_update2_kernel, dim3(tiles, 1, batch), dim3(U2_NW * 64), 0, s, L, Npad, r0, K, PA, PB, lstride, wstride);
    return (int)hipGetLastError();
}
