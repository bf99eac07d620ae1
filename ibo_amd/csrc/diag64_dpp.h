// diag64_dpp.h -- factor the 64x64 diagonal block of the blocked Cholesky and invert the factor: the sequential
// chain of every factorisation in this library (N/64 times per fit), second design.
//
// The chain is bound by the instruction issue of ONE wave (6 cycles per fp64 VALU instruction, dependent or not:
// tools/dpp_issue_bench), so what counts is the number of instructions between one pivot and the next.  The first
// design held one matrix row per lane and fetched the pivot row's values with v_readlane (two per double) -- three
// instructions, 17 cycles, per (pivot, column) pair.  gfx950's DP-ALU DPP mode `row_newbcast:n` (lane n of every
// 16-lane row broadcast to that row) is the one cross-lane operand an fp64 VOP2 instruction accepts, and
// v_fmac_f64 has a VOP2 form:
//       v_fmac_f64_dpp  acc, x row_newbcast:k, -y        acc -= x[lane k of my row] * y
// is a whole rank-1 update step in one 6-cycle instruction.  For it the 16 rows of the panel's diagonal block are
// REPLICATED in all four 16-lane rows of the wave (lane l holds block row l & 15), so every lane finds L[k][j] in
// lane k of its own DPP row; the rows below the diagonal block ride in a second register set, one row per lane,
// and take their multipliers from the same replicated registers: 2 instructions per pair for 16 + 48 rows.
// The second register set has lanes to spare (at most 48 rows lie below), and the spare lanes 48..63 carry the
// rows of a 16x16 IDENTITY: what "rows below" turn into is (row) L16^-T, so those lanes end the panel holding the
// inverse of the diagonal factor, transposed -- for no instruction at all.  (Round 1 inverted the four factors in
// a separate 6.5 k-cycle phase after the last panel.)
// The assembly of the 64x64 inverse from the 16x16 inverses (recursive doubling, fp64 MFMA tiles out of LDS) runs on
// the waves that idle while wave 0 factors the next panel, as do the trailing-update tiles the next panel does not
// need; after the last panel three 16x16x16 products remain.
#pragma once
#define DPP_TAIL "row_mask:0xf bank_mask:0xf"
// Hazards.  Two wait states must lie between a VALU write of a VGPR and a DPP read of it, one between a
// transcendental result and its first use, and the hazard recogniser does not look inside inline asm.  s_nop costs
// more than an fp64 instruction here (fmac_dpp 6.1 cycles, s_nop 1 + fmac_dpp 15.6), so every statement of the
// pivot sequence is `asm volatile` -- volatile statements keep their program order -- and the order is written so
// that useful instructions fill the wait states; s_nop appears only where the sequence leaves nothing to put there.
template <int NOPS>
__device__ __forceinline__ void dpp_wait()                      // NOPS wait states, in program order with the DPP statements
{
    if constexpr (NOPS == 1) asm volatile("s_nop 0");
    else if constexpr (NOPS >= 2) asm volatile("s_nop 1");
}
template <int N>                                                // a -= x[lane N of this 16-lane row] * ya,  w -= the same * yw
__device__ __forceinline__ void dpp_fnma2(double &a, double &w, double x, double ya, double yw)
{
    asm volatile("v_fmac_f64_dpp %0, %2, -%3 row_newbcast:%5 " DPP_TAIL "\n\tv_fmac_f64_dpp %1, %2, -%4 row_newbcast:%5 " DPP_TAIL
                 : "+v"(a), "+v"(w) : "v"(x), "v"(ya), "v"(yw), "i"(N));
}

// ---- panel factorisation (wave 0).  a[k]: row o + (lane & 15) of the diagonal block, column o + k (replicated in
// the four DPP rows); w[k]: row o + 16 + lane, column o + k, or row lane - 48 of the identity.
// Per pivot J:   s = 1/sqrt(a[J]) in the lanes of row J (v_rsq_f64 + one third-order correction; formed in every lane
// from its own a[J], broadcast from where it is right);  a[J] *= s, w[J] *= s;  a[K] -= L[K][J] a[J], w[K] -= L[K][J] w[J].
// Order: pair (J-1, J) -> v_rsq -> pair (J-1, J+1) -> correction chain of pivot J -> the other pairs of pivot J-1
// (they fill the wait states) -> scale J -> pair (J, J+1) ...   A pivot <= 0 (or NaN) turns s and with it the rest of
// the block into NaN: nothing traps, nothing sits on the dependent chain, and the first NaN column is found afterwards.
__device__ __forceinline__ double panel_rsq(double d)
{
    double y0;
    asm volatile("v_rsq_f64 %0, %1" : "=v"(y0) : "v"(d));
    return y0;
}
__device__ __forceinline__ double panel_chain(double d, double y0)
{
    // (round 6, measured and removed: one Newton step s = y0 (1 + e / 2) instead of the third-order correction below -- four dependent instructions
    // for five, the same |L - L_numpy| (7.5e-15): 0.272 against 0.272 ms at N = 1024, 0.572 / 0.570 at 2048.  The correction is not on the
    // pivot's critical path: the previous pivot's pair updates issue beside it, and it is their 2 x (15 - J) DPP instructions that a pivot costs.)
    double s, e, p;
    asm volatile("v_mul_f64 %1, %3, -%4\n\t"                    // -d y0
                 "v_fma_f64 %1, %1, %3, 1.0\n\t"                // e = 1 - d y0^2
                 "v_fma_f64 %2, %1, %5, 0.5\n\t"                // p = 1/2 + 3e/8
                 "v_mul_f64 %1, %3, %1\n\t"
                 "v_fma_f64 %0, %1, %2, %3"                      // s = y0 (1 + e p)
                 : "=&v"(s), "=&v"(e), "=&v"(p) : "v"(y0), "v"(d), "s"(0.375));
    return s;
}
template <int J, int NOPS>                                      // NOPS: wait states still missing between s and its DPP read
__device__ __forceinline__ void panel_scale(double (&a)[16], double (&w)[16], double s)
{
    double sb;
    dpp_wait<NOPS>();
    // the trailing wait state (with the second multiplication) is for the DPP read of a[J] by the pairs that follow
    asm volatile("v_mov_b64_dpp %0, %3 row_newbcast:%4 " DPP_TAIL "\n\tv_mul_f64 %1, %1, %0\n\tv_mul_f64 %2, %2, %0\n\ts_nop 0"
                 : "=&v"(sb), "+v"(a[J]), "+v"(w[J]) : "v"(s), "i"(J));
}
template <int J, int K, int KEND>
__device__ __forceinline__ void panel_pairs(double (&a)[16], double (&w)[16])
{
    if constexpr (K < KEND) {
        dpp_fnma2<K>(a[K], w[K], a[J], a[J], w[J]);
        panel_pairs<J, K + 1, KEND>(a, w);
    }
}
template <int J>
__device__ __forceinline__ void panel_from(double (&a)[16], double (&w)[16])
{
    // a[J], w[J] are scaled; pivots J + 1 .. remain
    if constexpr (J < 15) {
        panel_pairs<J, J + 1, J + 2>(a, w);
        const double y0 = panel_rsq(a[J + 1]);
        if constexpr (J + 2 < 16) panel_pairs<J, J + 2, J + 3>(a, w);
        else dpp_wait<1>();
        const double s = panel_chain(a[J + 1], y0);
        panel_pairs<J, J + 3, 16>(a, w);
        constexpr int filled = (J + 3 < 16 ? 16 - (J + 3) : 0) * 2;     // instructions between the chain and the broadcast
        panel_scale<J + 1, (filled >= 2 ? 0 : 2 - filled)>(a, w, s);
        panel_from<J + 1>(a, w);
    }
}
// Factor the panel of 16 columns at o: S[o.., o..o+15] becomes the factor's columns (the diagonal block's strict
// upper triangle is left with by-products of the elimination: scratch), V[o..o+15][o..o+15] the inverse of the
// diagonal block.  `Id`: a 16x16 identity in LDS (row stride TD).  One wave.
template <int TD>
__device__ __forceinline__ void diag64_panel(double *S, double *V, const double *Id, int o, int pivot0, int *info)
{
    const int lane = threadIdx.x & 63, dr = lane & 15;
    const int rb = min(o + 16 + lane, 63);
    const double *wrow = (lane >= 48) ? Id + (lane - 48) * TD : S + rb * SD + o;
    double a[16], w[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        a[k] = S[(o + dr) * SD + o + k];
        w[k] = wrow[k];
    }
    {
        const double y0 = panel_rsq(a[0]);
        dpp_wait<1>();
        panel_scale<0, 2>(a, w, panel_chain(a[0], y0));
    }
    panel_from<0>(a, w);
    // the last row of the block has an entry in every column: its first NaN is the first failed pivot
    if (info) {
        const double last = a[15];
        if (__builtin_amdgcn_readlane((int)(last != last), 15)) {
            int bad = 15;
#pragma unroll
            for (int k = 14; k >= 0; k--)
                if (__builtin_amdgcn_readlane((int)(a[k] != a[k]), 15)) bad = k;
            if (lane == 0) atomicCAS(info, 0, pivot0 + o + bad + 1);
        }
    }
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 16; k++) S[(o + dr) * SD + o + k] = a[k];
    }
    if (o + 16 + lane < 64) {
#pragma unroll
        for (int k = 0; k < 16; k++) S[rb * SD + o + k] = w[k];
    }
    if (lane >= 48) {       // identity row i = lane - 48 became column i of the inverse (its entries k < i are still exact zeros)
#pragma unroll
        for (int k = 0; k < 16; k++) V[(o + k) * SD + o + lane - 48] = w[k];
    }
}

// one wave: 16x16 tiles out of LDS
template <int K, int LDA = SD, int LDB = SD>
__device__ __forceinline__ d4_t diag64_mm(const double *Am, const double *Bm) { return lds_mm16<false, K, LDA, LDB>(Am, Bm); }
template <int LD = SD>
__device__ __forceinline__ void diag64_put(double *Dst, int r0, int c0, d4_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < 4; r++) Dst[(r0 + MM16_ROW(r)) * LD + c0 + MM16_COL] = v[r];
}
// trailing-update tile (it, kt) of the panel at o:  S[o+16+16 it.., o+16+16 kt..] -= X_it X_kt^T
__device__ __forceinline__ void diag64_update_tile(double *S, int o, int it, int kt)
{
    const int lane = threadIdx.x & 63;
    d4_t acc = lds_mm16<true, 16>(S + (o + 16 + 16 * it) * SD + o, S + (o + 16 + 16 * kt) * SD + o);
    double *C = S + (o + 16 + 16 * it) * SD + o + 16 + 16 * kt;
#pragma unroll
    for (int r = 0; r < 4; r++) C[MM16_ROW(r) * SD + MM16_COL] -= acc[r];
}

// S: the 64x64 block (row stride SD), V: zeros, T: a 16x16 identity in its first 16 rows (diag64_load leaves them
// so); T has row stride TD >= 49 (the chain touches its columns 0..47 only: a kernel that uses T for nothing else gives it 64 x 49 doubles and
// stays under the 96 KB of LDS that fit beside a 64 KB workgroup of another kernel on the same CU).  On return S holds the factor (lower triangle; the strict upper
// part of the off-diagonal 16-blocks is scratch), V its inverse.  T is scratch.  Called by all threads of the workgroup (waves 0..3 work); ends
// with a barrier.  With 16-blocks L_ij of the factor and V_i = L_ii^-1:
//   [L11 0; L21 L22]^-1 = [V11 0; -V22 L21 V11, V22]   at the 32- and at the 64-level.
// `side`: work for the waves beyond the fourth of an eight-wave workgroup (chol_pipe8_kernel), called once per panel b = 0..3 between
// the barriers: it must be short enough (~3 k cycles) not to hold the chain's barriers up, and it must stay off wave 4, which shares
// its SIMD -- hence the fp64 pipe -- with wave 0.
struct Diag64NoSide { __device__ __forceinline__ void operator()(int) const {} };
template <int TD = SD, typename Side = Diag64NoSide>
__device__ __forceinline__ void diag64_factor_invert(double *S, double *V, double *T, int pivot0, int *info, Side side = Side())
{
    const int t = threadIdx.x, wv = t >> 6;
    for (int b = 0; b < 4; b++) {
        const int o = 16 * b;
        if (wv == 0) {
            diag64_panel<TD>(S, V, T, o, pivot0, info);
        } else if (wv > 3) {
            // (an eight-wave workgroup: waves 4..7 keep the barriers' count -- chol_step8_kernel -- or do the caller's side work)
            side(b);
        } else if (b == 1) {
            // the tiles of panel 0's update that panel 1 does not read: (1,1), (2,1), (2,2)
            diag64_update_tile(S, 0, wv == 1 ? 1 : 2, wv == 3 ? 2 : 1);
        } else if (b == 2) {
            if (wv == 3) diag64_update_tile(S, 16, 1, 1);
            if (wv == 1) {                          // upper-left 32x32 node: V[16..31][0..15] = -V1 (L10 V0)
                diag64_put<TD>(T, 16, 0, diag64_mm<16>(S + 16 * SD, V));
                diag64_put(V, 16, 0, -diag64_mm<16, SD, TD>(V + 16 * SD + 16, T + 16 * TD));
            }
        } else if (b == 3 && wv < 3) {
            // everything of the 64-level that does not need V3, for the column half c = 0 / 16 of this wave:
            //   T' = L[32..63][0..31] V[0..31][c..]              (rows 32..63 of T)
            //   V[32..47][c..] = -V2 T'[32..47]
            //   T32 = L32 V2  (both waves form it; same values)   (T[48..63][32..47])
            //   Q = T'[48..63] - T32 T'[32..47]                   (T[16..31][c..]);  what remains is -V3 Q
            const int c = 16 * (wv - 1), lane = t & 63;
            diag64_put<TD>(T, 32, c, diag64_mm<32>(S + 32 * SD, V + c));
            diag64_put<TD>(T, 48, c, diag64_mm<32>(S + 48 * SD, V + c));
            diag64_put<TD>(T, 48, 32, diag64_mm<16>(S + 48 * SD + 32, V + 32 * SD + 32));
            diag64_put(V, 32, c, -diag64_mm<16, SD, TD>(V + 32 * SD + 32, T + 32 * TD + c));
            const d4_t q = diag64_mm<16, TD, TD>(T + 48 * TD + 32, T + 32 * TD + c);
#pragma unroll
            for (int r = 0; r < 4; r++)
                T[(16 + MM16_ROW(r)) * TD + c + MM16_COL] = T[(48 + MM16_ROW(r)) * TD + c + MM16_COL] - q[r];
        }
        __syncthreads();
        // the part of the trailing update the next panel reads: tiles (it, 0)
        if (wv < 3 - b) diag64_update_tile(S, o, wv, 0);
        if (b < 3) __syncthreads();
    }
    // V3 is in place: V[48..63][32..47] = -V3 T32,  V[48..63][c..] = -V3 Q
    if (wv == 0) diag64_put(V, 48, 32, -diag64_mm<16, SD, TD>(V + 48 * SD + 48, T + 48 * TD + 32));
    else if (wv < 3) diag64_put(V, 48, 16 * (wv - 1), -diag64_mm<16, SD, TD>(V + 48 * SD + 48, T + 16 * TD + 16 * (wv - 1)));
    __syncthreads();
}
