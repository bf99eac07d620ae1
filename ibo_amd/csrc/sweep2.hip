// sweep2.hip -- the candidate sweep's main kernel, second design (large batches, dot-form distances).
//
// Same mathematics as sweep.hip (GP_Maximizer::posterior + negei/negpi/negucb,
// cpp/optimizeGP.cpp:57-236; GaussianProcess.posterior, ego/gaussianprocess/__init__.py:169-228):
//   k*_i = k(x_i, c);  mu = m(c) + aY.k* - m(c) a1.k*;  q = |W k*|^2;  s2 = clamp(1+noise-q);  acq; arg-max.
//
// What bounds this kernel is the fp64 MFMA pipe, which executes on the fp64 FMA units: every VALU
// instruction of any type issued beside it costs ~10 pipe cycles against 64 for an MFMA
// (tools/mfma_f64_peak).  sweep.hip spends 1.3 of its 1.47 VALU instructions per MFMA on generating k*:
// D+1 FMAs and a 16-instruction exp per (row, candidate), and every k* row is regenerated for each
// 512-row panel of W (1.5x at N = 1024, 2.5x at N = 2048).  This kernel removes most of that:
//
//   * 32 candidates per workgroup instead of 64: with the same 64 accumulator VGPRs a wave now owns
//     4 row-blocks x 2 candidate-blocks, so one panel is 16 waves x 4 x 16 = 1024 rows -- no
//     regeneration up to N = 1024, 1.5x at N = 2048;
//   * the exponent y = a_k + b_c + x~_k . c~ is itself a small GEMM, [x~ | a_k | 1] (N x (D+2)) times
//     [c~ | 1 | b_c]^T, so it is done by the MFMA unit: ceil((D+2)/4) MFMAs give a 16-row x
//     16-candidate tile of y (2 for D <= 6, 5 for D = 16) instead of 4 (D+1) VALU FMAs;
//   * the MFMA output layout (lane l, element r = row (l>>4)+4r, column l&15) IS the B-fragment layout
//     of the following V += W K* MFMA for k4-step r, so the four k* values of a lane go to LDS with four
//     linear 512-byte wave stores and no address arithmetic; per stage every wave produces exactly one
//     16 x 16 tile (16 waves = 128 rows x 32 candidates per stage);
//   * W's A-fragments come straight from L2 by buffer_load_dwordx4 with a scalar row/column offset and
//     ONE lane-offset VGPR (no 64-bit per-lane pointers), one 8-column step ahead, and only for
//     row-blocks that still have non-zeros in that step (the triangle is skipped for loads and MFMAs).
//
// Used for batches > 8192 candidates when the dot form is admissible (ibo_gp.dot_form; otherwise, and
// for the small-batch SPLIT / GEMV paths, sweep.hip's kernels run).
#include "ibo_common.h"
#include <atomic>
#include <type_traits>
#include <cfloat>

#define S2_NW 16
#define S2_KCH 128                     // k* rows per LDS stage
#define S2_PANEL 1024                  // rows of W per pass: 16 waves x 4 row-blocks x 16
// rows of the alpha vectors held in LDS at a time: 64 KiB of the 160, or 48 KiB where the candidate tile is wide
// (17..32 dimensions: 32 x 37 doubles instead of 32 x 21)
__host__ __device__ constexpr int s2_awin(int ka4) { return ka4 <= 5 ? 4096 : 3072; }

#include "sweep2_dev.h"

// KA4 = ceil((D + 2) / 4): k4-steps of the exponent GEMM
// What a tile's bound must reach to be refreshed / completed, given the threshold word: the threshold less a slack.  The bound
// argument is exact mathematics (fewer rows of W: larger variance; EI and UCB grow with it), but the reference's EI formula is not
// monotone in floating point where it is tiny: Phi(z) = (1 + erf(z / sqrt 2)) / 2 carries ~1e-16 of absolute rounding noise, so an EI
// below ~1e-14 (times the data's scale) is noise, and two candidates' order there is whatever the noise says.  With the slack --
// 1e-9 of the threshold plus 1e-13 (1 + |ymax| + |parm|) -- a tile is never dropped on the strength of such digits; when the best value
// itself is down there, every admissible tile is completed, as a full sweep would.  (tools/fuzz_gallery.py found the three cases
// in 420 that taught this.)
__device__ __forceinline__ double s2_part_limit(unsigned long long th, double slack_abs);

// order-preserving encoding of a double as an unsigned integer (atomicMax over values); 0 decodes to a NaN: "no value yet"
__device__ __forceinline__ unsigned long long s2_enc(double x)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double s2_dec(unsigned long long e)
{
    return __longlong_as_double((long long)((e >> 63) ? (e & 0x7fffffffffffffffull) : ~e));
}

__device__ __forceinline__ double s2_part_limit(unsigned long long th, double slack_abs)
{
    if (th == 0ull) return -DBL_MAX;
    const double t = s2_dec(th);
    return t - (1e-9 * fabs(t) + slack_abs);
}

template <int FAM, int KA4, bool BIGN, bool PART = false>   // BIGN: more than s2_awin(KA4) rows -- the alpha vectors' window moves
                                                            // PART: rows [a.part_lo, a.part_hi) of W only, no means (kept-state sweeps)
__global__ __launch_bounds__(S2_NW * 64) void sweep2_kernel(SweepArgs a)
{
    // (PART, a.part_all == 2: the launch's workgroups are the entries of a compact LIST of tiles -- a.tile_sel holds tile numbers, not flags)
    const unsigned tb = (PART && a.part_all == 2) ? (unsigned)a.tile_sel[blockIdx.x] : blockIdx.x;
    if (PART && a.part_lo > 0) {
        // a later level: only tiles that stand at the level before it and whose bound reaches the threshold (a bound below it cannot
        // win: the threshold is a value some complete candidate attains, or the cut that picks the first tiles to complete)
        if (a.tile_done[tb] != a.part_level - 1) return;
        if (a.part_all == 2) { /* listed: selected by part_mark_kernel */ }
        else if (a.tile_sel) { if (!a.tile_sel[tb]) return; }
        else if (!a.part_all) {
            // (no threshold yet: every tile with an admissible candidate; a tile whose candidates are all excluded never needs its variance)
            if (!(a.tile_ub[tb] >= s2_part_limit(*a.part_thresh, a.part_slack))) return;
        }
    }
    constexpr int TCAND = IBO_S2_TCAND, CBW = TCAND / 16, RBW = 4, KA = 4 * KA4, S2_AWIN = s2_awin(KA4);
    static_assert(CBW == 2 && S2_NW * RBW * 16 == S2_PANEL && (S2_KCH / 16) * CBW == S2_NW, "tile geometry");
    __shared__ double lds_k[2][S2_KCH * TCAND];    // K* stages in B-fragment order: [k4-step][cand-block][lane]
    __shared__ double lds_c[TCAND * (KA + 1)];            // augmented, scaled candidates [cand][KA]
    __shared__ double lds_q[S2_NW][TCAND];
    __shared__ double lds_m[2][S2_NW][16];
    __shared__ double lds_tab[2048];                // 2^(j/2048)
    // alphaY[AW], alpha1[AW]: a window of AW = min(rows padded to 128, s2_awin) rows of both vectors -- all of them up to
    // 4096 observations; beyond, the last panel (the one that forms the mean) moves the window as its stages advance
    extern __shared__ __attribute__((aligned(16))) double lds_alpha[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t tile0 = (int64_t)tb * TCAND;
    const int D = a.kp.D;
#ifdef IBO_STAMPS   // diagnostic build (tools/stamp_sweep2.py): a tile's entry / prologue done / panels done / exit, and where it ran
    unsigned long long st2[4];
    st2[0] = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- candidates of this tile: c~ = c sqrt(w), then the two extra columns 1 and b_c
    const int NA128 = (a.Npad + 127) & ~127;
    lds_tab[tid] = a.exp_tab[tid];
    lds_tab[tid + 1024] = a.exp_tab[tid + 1024];
    const int AW = BIGN ? S2_AWIN : NA128;
    if (!PART || a.part_means) {
        for (int e = tid; e < AW; e += S2_NW * 64) {                // both vectors are zero beyond N (abi.hip pads them)
            lds_alpha[e] = a.alphaY[e];
            lds_alpha[AW + e] = a.alpha1[e];
        }
    }
    for (int e = tid; e < TCAND * KA; e += S2_NW * 64) {
        const int c = e / KA, col = e - c * KA;
        int64_t gi = tile0 + c;
        if (gi > a.M - 1) gi = a.M - 1;
        lds_c[c * (KA + 1) + col] = (col < D) ? a.cand[gi * D + col] * a.kp.sw[col] : (col == D ? 1.0 : 0.0);
    }
    __syncthreads();
    if (tid < TCAND) {
        double n2 = 0.0;
        for (int d = 0; d < D; d++) { const double v = lds_c[tid * (KA + 1) + d]; n2 = fma(v, v, n2); }
        // A candidate more than 775 length scales from the origin (hence > 450 from every observation: |x~| <= 316
        // where the dot form is in use) has k* = 0 exactly; it is pulled in to that radius, where k* is still 0, so
        // that the exponent stays within what s2_exp's integer arithmetic covers (|y| < 7e5).
        if (n2 > 6e5) {
            const double sc = sqrt(6e5 / n2);
            for (int d = 0; d < D; d++) lds_c[tid * (KA + 1) + d] *= sc;
            n2 = 6e5;
        }
        lds_c[tid * (KA + 1) + D + 1] = fma(-0.5, n2, FAM == FAM_SE ? a.log_sf2 : 0.0);
    }
    __syncthreads();
    // this wave generates the 16 x 16 tile (row-tile rt, candidate block gcb) of every stage
    const int rt = wave >> 1, gcb = wave & 1;
    // its B-fragments of the exponent GEMM, c~aug[candidate 16 gcb + (lane&15)][4 s + (lane>>4)], are re-read from
    // LDS at every generation (KA4 reads, no VALU) rather than held in 2 KA4 registers
    const double *cfrag = &lds_c[(16 * gcb + (lane & 15)) * (KA + 1) + (lane >> 4)];

    const int Npad = a.Npad;
    const int nk8 = Npad >> 3;
    const int npanel = (Npad + S2_PANEL - 1) / S2_PANEL;
    const int rem = Npad % S2_PANEL;
    const __amdgpu_buffer_rsrc_t rW = s2_rsrc(a.Wp, (size_t)Npad * Npad * sizeof(double));
    const unsigned lane16 = lane * 16;
    const __amdgpu_buffer_rsrc_t rXA = s2_rsrc(a.XA, (size_t)(NA128 / 16) * KA4 * 64 * sizeof(double));
    const unsigned lane8 = lane * 8;
    const double *aY_quad = lds_alpha + (lane >> 4), *a1_quad = lds_alpha + AW + (lane >> 4);
    // SIMD balance: a stage's eight partly active row-blocks (2, 4, .., 16 active steps) belong to eight
    // consecutive waves; waves w and w+4 share a SIMD, so blocks k and 7-k of each group of eight go to waves
    // that do -- every SIMD then carries the same MFMA count in every stage
    const int pw = (wave & 8) | ((wave & 4) ? 11 - (wave & 7) : (wave & 7));

    if (lane < 16) { lds_q[wave][lane] = 0.0; lds_q[wave][16 + lane] = 0.0; }
#ifdef IBO_STAMPS
    st2[1] = __builtin_amdgcn_s_memrealtime();
#endif

    // the exponent GEMM's A-fragments of the tile this wave generates in stage t (rows 128 t + 16 rt ..)
    auto load_xa = [&](int k0, double (&xa)[KA4]) {
        const int tile = (k0 >> 4) + rt;
#pragma unroll
        for (int s = 0; s < KA4; s++) xa[s] = s2_ld_f64(rXA, lane8, (unsigned)((tile * KA4 + s) * 512));
    };
    // k* rows [k0 + 16 rt, +16) x candidates of block gcb -> stage buffer b
    auto gen = [&](int k0, int b, const double (&xa)[KA4], double &muY, double &mu1, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int tile = (k0 >> 4) + rt;
        d4_t y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KA4; s++) y = mfma_f64(xa[s], cfrag[4 * s], y);
        double *dst = &lds_k[b][(4 * rt * CBW + gcb) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double ay = 0.0, a1v = 0.0;
            if (LAST) { const int kw = tile * 16 + 4 * r - (BIGN ? (k0 / S2_AWIN) * S2_AWIN : 0); ay = aY_quad[kw]; a1v = a1_quad[kw]; }
            double kv;
            if (FAM == FAM_SE) kv = s2_exp(y[r], lds_tab);
            else {                                                  // z = |x~ - c~|^2 = -2y
                const double z = fmax(-2.0 * y[r], 0.0);
                const double rr = sqrt_fast((FAM == FAM_M3 ? 3.0 : 5.0) * z);
                const double poly = FAM == FAM_M3 ? 1.0 + rr : fma(rr, fma(rr, 1.0 / 3.0, 1.0), 1.0);
                kv = a.kp.sf2 * poly * s2_exp(-rr, lds_tab);
            }
            if (LAST) { muY = fma(ay, kv, muY); mu1 = fma(a1v, kv, mu1); }
            dst[r * CBW * 64] = kv;
            __builtin_amdgcn_sched_barrier(0);                     // one element at a time: keeps the temporaries few
        }
    };

    auto run_panel = [&](int p, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        // (PART: p counts 1024-row panels from a.part_lo)
        const int row0 = PART ? a.part_lo + p * S2_PANEL : (rem == 0) ? p * S2_PANEL : (p == 0 ? 0 : rem + (p - 1) * S2_PANEL);
        const int row1 = PART ? min(row0 + S2_PANEL, a.part_hi) : (rem == 0) ? row0 + S2_PANEL : (p == 0 ? rem : row0 + S2_PANEL);
        const int nstage = (row1 + S2_KCH - 1) / S2_KCH;
        // Row-blocks of this wave, ascending: g = row0/16 + pw + 16 e, e < ne (a short panel has fewer than RBW);
        // row-block g has non-zeros in 8-column steps j < 2g + 2.  They sit in the LAST ne slots, so that the slots
        // still active at any step are always a suffix (RBW-NA .. RBW-1) -- what the four fixed-shape loops below need.
        const int nrb = (row1 - row0) >> 4;
        const int ne = nrb > pw ? min(RBW, (nrb - pw + S2_NW - 1) / S2_NW) : 0;
        int last8[RBW];
        unsigned wbase[RBW];
#pragma unroll
        for (int i = 0; i < RBW; i++) {
            const int e = i - (RBW - ne);
            const int g = (row0 >> 4) + pw + S2_NW * (e < 0 ? 0 : e);
            last8[i] = e >= 0 ? 2 * g + 2 : 0;
            wbase[i] = (unsigned)g * (unsigned)nk8 * 1024u;        // bytes: fragment (g, j) sits at (g nk8 + j) * 1024
        }
        d4_t acc[RBW][CBW];
#pragma unroll
        for (int i = 0; i < RBW; i++)
#pragma unroll
            for (int cb = 0; cb < CBW; cb++) acc[i][cb] = (d4_t){0.0, 0.0, 0.0, 0.0};
        double muY = 0.0, mu1 = 0.0;      // partial means: candidate (gcb, lane&15), rows of this wave's tiles (LAST only)

        double xa[KA4];
        load_xa(0, xa);
        gen(0, 0, xa, muY, mu1, last_tag);
        // fragments of the first stage's first step
        v4u_t A0[RBW], A1[RBW];
#pragma unroll
        for (int i = 0; i < RBW; i++) A0[i] = __builtin_amdgcn_raw_buffer_load_b128(rW, lane16, wbase[i], 0);
        __syncthreads();
        for (int t = 0; t < nstage; t++) {
            const int j0 = t * (S2_KCH / 8);
            // steps of this stage in which row-block i is active: jj < n[i]; n[] ascends with i, all even
            int n[RBW];
#pragma unroll
            for (int i = 0; i < RBW; i++) {
                int v = last8[i] - j0;
                v = v < 0 ? 0 : v;
                n[i] = v > S2_KCH / 8 ? S2_KCH / 8 : v;
            }
            const bool more = t + 1 < nstage;
            constexpr bool XA_EARLY = KA4 <= 4;      // D = 15, 16: five fragment registers more would spill; fetch late
            // the next stage's fragments of X go out first; its k* is generated after the first (all row-blocks
            // active) range of steps, when they have long arrived
            if (XA_EARLY && more) load_xa((t + 1) * S2_KCH, xa);
            const double *kb = &lds_k[t & 1][lane];
            // Two 8-column steps (jj, jj+1; jj a compile-time constant, so every LDS offset is an immediate) with
            // the NA largest row-blocks of the wave active (slots RBW-NA .. RBW-1), straight line: fragments of
            // step jj+1 are fetched while step jj's MFMAs issue, those of step jj+2 -- the next pair's, or the next
            // stage's first -- during step jj+1.  A fetch past a row-block's last step is harmless (zeros of the
            // upper triangle, or zeros from the buffer's bounds check) and happens once per range.
            auto pair = [&](auto jj_tag, auto na_tag) {
                constexpr int JJ = decltype(jj_tag)::value, NA = decltype(na_tag)::value;
                const unsigned so = (unsigned)(j0 + JJ) * 1024u;
#pragma unroll
                for (int i = RBW - NA; i < RBW; i++) A1[i] = __builtin_amdgcn_raw_buffer_load_b128(rW, lane16, wbase[i] + so + 1024u, 0);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const double b0 = kb[((JJ * 2 + h) * CBW + 0) * 64], b1 = kb[((JJ * 2 + h) * CBW + 1) * 64];
#pragma unroll
                    for (int i = RBW - NA; i < RBW; i++) {
                        const double av = h ? s2_hi(A0[i]) : s2_lo(A0[i]);
                        acc[i][0] = mfma_f64(av, b0, acc[i][0]);
                        acc[i][1] = mfma_f64(av, b1, acc[i][1]);
                    }
                }
#pragma unroll
                for (int i = RBW - NA; i < RBW; i++) A0[i] = __builtin_amdgcn_raw_buffer_load_b128(rW, lane16, wbase[i] + so + 2048u, 0);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const double b0 = kb[((JJ * 2 + 2 + h) * CBW + 0) * 64], b1 = kb[((JJ * 2 + 2 + h) * CBW + 1) * 64];
#pragma unroll
                    for (int i = RBW - NA; i < RBW; i++) {
                        const double av = h ? s2_hi(A1[i]) : s2_lo(A1[i]);
                        acc[i][0] = mfma_f64(av, b0, acc[i][0]);
                        acc[i][1] = mfma_f64(av, b1, acc[i][1]);
                    }
                }
            };
            // steps [lo, hi) with NA row-blocks active: eight guarded copies of the pair, entered and left by
            // scalar branches -- no vector instruction is spent on loop control or addresses
            auto range = [&](int lo, int hi, auto na_tag) {
                if (lo >= hi) return;
                if (0 >= lo && 0 < hi) pair(std::integral_constant<int, 0>{}, na_tag);
                if (2 >= lo && 2 < hi) pair(std::integral_constant<int, 2>{}, na_tag);
                if (4 >= lo && 4 < hi) pair(std::integral_constant<int, 4>{}, na_tag);
                if (6 >= lo && 6 < hi) pair(std::integral_constant<int, 6>{}, na_tag);
                if (8 >= lo && 8 < hi) pair(std::integral_constant<int, 8>{}, na_tag);
                if (10 >= lo && 10 < hi) pair(std::integral_constant<int, 10>{}, na_tag);
                if (12 >= lo && 12 < hi) pair(std::integral_constant<int, 12>{}, na_tag);
                if (14 >= lo && 14 < hi) pair(std::integral_constant<int, 14>{}, na_tag);
            };
            range(0, n[0], std::integral_constant<int, 4>{});
            if (!XA_EARLY && more) load_xa((t + 1) * S2_KCH, xa);
            if (BIGN && LAST && more && ((t + 1) * S2_KCH) % S2_AWIN == 0) {
                // the next stage's rows start a new window of the alpha vectors (nobody reads the old one any more: the
                // stage that used it was generated before the last barrier)
                const int wb = (t + 1) * S2_KCH;
                for (int e = tid; e < S2_AWIN; e += S2_NW * 64) {
                    lds_alpha[e] = wb + e < NA128 ? a.alphaY[wb + e] : 0.0;
                    lds_alpha[AW + e] = wb + e < NA128 ? a.alpha1[wb + e] : 0.0;
                }
                __syncthreads();
            }
            if (more) gen((t + 1) * S2_KCH, (t + 1) & 1, xa, muY, mu1, last_tag);
            range(n[0], n[1], std::integral_constant<int, 3>{});
            range(n[1], n[2], std::integral_constant<int, 2>{});
            range(n[2], n[3], std::integral_constant<int, 1>{});
            __syncthreads();
        }
        if (PART && LAST) {
            // the first part of a kept state also forms the means: its last panel has generated the k* rows below a.part_hi and added
            // their terms; the rows from there on are generated here for the two dot products alone (no W, no barrier: the stage
            // buffers they land in are not read again) -- half of what the separate means pass regenerated
            for (int t = nstage; t < NA128 / S2_KCH; t++) {
                load_xa(t * S2_KCH, xa);
                gen(t * S2_KCH, t & 1, xa, muY, mu1, last_tag);
            }
        }
        // |V|^2 down the rows of this panel: acc[i][cb][r] is row 16 g_i + (lane>>4) + 4r, candidate 16 cb + (lane&15)
#pragma unroll
        for (int cb = 0; cb < CBW; cb++) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < RBW; i++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    // PART: rows appended to the model after the state was formed are zsum's, whenever this tile is completed
                    if (PART && 16 * ((row0 >> 4) + pw + S2_NW * (i - (RBW - ne))) + (lane >> 4) + 4 * r >= a.part_rows) continue;
                    s = fma(acc[i][cb][r], acc[i][cb][r], s);
                }
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (lane < 16) lds_q[wave][cb * 16 + lane] += s;
        }
        if (LAST) {
            muY += __shfl_xor(muY, 16); muY += __shfl_xor(muY, 32);
            mu1 += __shfl_xor(mu1, 16); mu1 += __shfl_xor(mu1, 32);
            if (lane < 16) { lds_m[0][wave][lane] = muY; lds_m[1][wave][lane] = mu1; }
        }
    };

    // the last panel sees every k: it also forms the mean.  A short panel (N not a multiple of 1024) comes first.
    if (PART) {
        const int np = (a.part_hi - a.part_lo + S2_PANEL - 1) / S2_PANEL;
        if (a.part_means) {
            for (int p = 0; p + 1 < np; p++) run_panel(p, std::false_type{});
            run_panel(np - 1, std::true_type{});
        } else {
            for (int p = 0; p < np; p++) run_panel(p, std::false_type{});
        }
        __syncthreads();
        if (tid < TCAND) {
            double q = 0.0;
#pragma unroll
            for (int w = 0; w < S2_NW; w++) q += lds_q[w][tid];
            const int64_t li = tile0 + tid;
            // q_a: the first level; q_b: the later levels, added in level order (a tile always takes them in order, so its bits do not
            // depend on WHEN it was taken further)
            if (li < a.M) {
                if (a.part_lo == 0) a.qpart[li] = q;
                else a.qpart[4 * a.M + li] = a.part_level == 1 ? q : a.qpart[4 * a.M + li] + q;
            }
            if (a.part_means) {
                const int c = tid;
                double my = 0.0, m1 = 0.0;
#pragma unroll
                for (int w = 0; w < S2_NW / 2; w++) { my += lds_m[0][2 * w + (c >> 4)][c & 15]; m1 += lds_m[1][2 * w + (c >> 4)][c & 15]; }
                if (li < a.M) { a.qpart[a.M + li] = my; a.qpart[2 * a.M + li] = m1; }
            }
        }
        if (a.part_lo > 0 && tid == 0) a.tile_done[tb] = a.part_level;
        return;
    }
    for (int p = 0; p + 1 < npanel; p++) run_panel(p, std::false_type{});
    run_panel(npanel - 1, std::true_type{});
#ifdef IBO_STAMPS
    st2[2] = __builtin_amdgcn_s_memrealtime();
#endif
    __syncthreads();
    // hand (q, aY.k*, a1.k*) of every candidate to acq_finish_kernel: the acquisition's erf/exp/sqrt chain on one
    // wave would keep the other fifteen (and the MFMA pipe) waiting at the end of every tile
    if (tid < TCAND) {
        const int c = tid;
        double q = 0.0, my = 0.0, m1 = 0.0;
#pragma unroll
        for (int w = 0; w < S2_NW; w++) q += lds_q[w][c];
#pragma unroll
        for (int w = 0; w < S2_NW / 2; w++) { my += lds_m[0][2 * w + (c >> 4)][c & 15]; m1 += lds_m[1][2 * w + (c >> 4)][c & 15]; }
        const int64_t li = tile0 + c;
        if (li < a.M) { a.qpart[li] = q; a.qpart[a.M + li] = my; a.qpart[2 * a.M + li] = m1; }
    }
#ifdef IBO_STAMPS
    st2[3] = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && a.mupart) {
        unsigned long long *d = (unsigned long long *)a.mupart + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 4; i++) d[i] = st2[i];
        d[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_ID
        d[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));     // XCC_ID
    }
#endif
}

// second half of the sweep: mean prior, variance clamp, EI / PI / UCB, optional per-candidate outputs, exclusion
// balls, and a (max value, lowest index) partial per 256 candidates -- all from sweep2_kernel's three numbers per
// candidate.  Summation order and formulas are those of s2_finish (shared with sweep.hip's epilogue).
__global__ __launch_bounds__(256) void acq_finish_kernel(SweepArgs a)
{
    __shared__ double sv[4];
    __shared__ int64_t si[4];
    const int64_t li = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = li < a.M;
    const int64_t gi = valid ? li : a.M - 1;
    bool excl;
    const double q = a.state5 ? (a.qpart[gi] + a.qpart[4 * a.M + gi]) + a.qpart[3 * a.M + gi] : a.qpart[gi];
    double val = s2_finish(a, a.cand + gi * a.kp.D, q, a.qpart[a.M + gi], a.qpart[2 * a.M + gi], li, valid, excl);
    int64_t idx = a.index_base + li;
    if (!valid || excl || !(val == val)) { val = -INFINITY; idx = INT64_MAX; }
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(val, o);
        const int64_t oi = __shfl_xor(idx, o);
        if (ov > val || (ov == val && oi < idx)) { val = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = val; si[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++)
            if (sv[w] > val || (sv[w] == val && si[w] < idx)) { val = sv[w]; idx = si[w]; }
        a.part_val[blockIdx.x] = val; a.part_idx[blockIdx.x] = idx;
    }
}

// The kept state's values per TILE of 32 candidates (same arithmetic as acq_finish_kernel): tile_ub = the largest value of the
// tile -- exact where the tile is complete, an upper bound where q_b is missing (EI and UCB grow with the variance) --, and the
// maximum over the complete tiles into *part_best.  No arg-max: this only decides which tiles must be completed (the caller
// passes no per-candidate output arrays).
__global__ __launch_bounds__(256) void acq_bound_kernel(SweepArgs a)
{
    const int64_t li = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = li < a.M;
    const int64_t gi = valid ? li : a.M - 1;
    bool excl;
    const double q = (a.qpart[gi] + a.qpart[4 * a.M + gi]) + a.qpart[3 * a.M + gi];
    // A tile that has not folded in every appended row (lazy refresh) carries means formed before those rows existed.  Row i moves a
    // candidate's mean by nu_i (W y)_i, nu = W k*, and |nu_i| <= sqrt(q) <= a.nu_max = sf2_k / sqrt(sf2_fit) -- the PRIOR standard
    // deviation in units of the fitted matrix (1 for the squared exponentials; the magnitude for SV / Matern kernels; run_sweep
    // derives it and switches the lazy mode off where no bound exists).  So the tile's value is bounded by the acquisition at
    // mean + nu_max sum |(W y)_i| -- nothing when the appended observations sit on the posterior mean (the gallery's
    // hallucinations), anything when they do not: then no tile is skipped.  (A mean prior multiplies a second vector, (W 1)_i, that is
    // never small: the caller refreshes every tile then.)
    const int64_t tile = gi >> 5;
    const bool fresh = !a.tile_rows || a.part_rows + a.tile_rows[tile] >= a.rank_hi;
    double margin = 0.0;
    if (!fresh) {
        for (int i = a.part_rows; i < a.rank_hi; i++) margin += fabs(a.wy[i]);
        margin = margin > 0.0 ? margin * a.nu_max : 0.0;        // (rows with (W y)_i = 0 exactly move nothing, whatever nu is)
    }
    double val = s2_finish(a, a.cand + gi * a.kp.D, q, a.qpart[a.M + gi] + margin, a.qpart[2 * a.M + gi], li, valid, excl);
    if (!valid || excl || !(val == val)) val = -INFINITY;
    for (int o = 16; o > 0; o >>= 1) val = fmax(val, __shfl_xor(val, o));          // the 32 candidates of a tile: half a wave
    if ((threadIdx.x & 31) == 0 && valid) {
        a.tile_ub[tile] = val;
        if (a.tile_done[tile] == a.part_nlev - 1 && fresh && val > -INFINITY) atomicMax(a.part_best, s2_enc(val));
    }
}

// The cut that picks the FIRST tiles to complete, when no tile is complete yet: the bound of the tile ranked ~3 % from the top (or ~240th)
// among 1024 evenly spaced tiles (one workgroup, bitonic sort in LDS).  The tiles at or above it very likely hold the maximum,
// and whatever value they reach is the threshold for everyone else.
// (round 4: at most ~256 tiles -- one per CU: the completion of the first tiles is a launch whose length is whole rounds of the chip, and
// 633 tiles cost three rounds, 2.6 of a 5.2 ms sweep at BASELINE config 3's shape, where 256 cost one)
__global__ __launch_bounds__(1024) void part_select_kernel(const double *__restrict__ tile_ub, int64_t ntiles, unsigned long long *thresh)
{
    __shared__ double v[1024];
    const int t = threadIdx.x;
    const int64_t n = ntiles < 1024 ? ntiles : 1024;
    v[t] = t < n ? tile_ub[(int64_t)t * ntiles / n] : -INFINITY;
    __syncthreads();
    for (int k = 2; k <= 1024; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int o = t ^ j;
            if (o > t) {
                const bool desc = (t & k) == 0;
                const double x = v[t], y = v[o];
                if (desc ? x < y : x > y) { v[t] = y; v[o] = x; }
            }
            __syncthreads();
        }
    // 3 % of the ranking, but not more than 256 tiles in all: the sample only estimates how many tiles lie above a cut, so the
    // tiles at or above it are counted and the cut is raised while they are too many (at most four passes over the bounds)
    __shared__ int cnt;
    int64_t k = n * 3 / 100;
    const int64_t k1 = n * 240 / ntiles;
    if (k1 < k) k = k1;
    for (int pass = 0; pass < 4; pass++) {
        const double cut = v[k];
        if (t == 0) cnt = 0;
        __syncthreads();
        int c = 0;
        for (int64_t i = t; i < ntiles; i += 1024) c += tile_ub[i] >= cut;
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if ((t & 63) == 0 && c) atomicAdd(&cnt, c);
        __syncthreads();
        const int total = cnt;
        __syncthreads();
        if (total <= 256 || k == 0) break;
        int64_t kn = k * 230 / total;
        k = kn < k ? kn : k - 1;
    }
    if (t == 0) {
        const double cut = v[k];
        *thresh = cut > -INFINITY ? s2_enc(cut) : 0ull;
    }
}

// Refresh of a kept sweep state after the model has grown by one row (ibo_gp_extend): with W' = [[W, 0], [w^T, 1/d]]
// the new candidate-side quantity is q' = q + (w'_last . k*')^2, and the means are re-formed from the current alpha
// vectors -- three dot products against the regenerated k* per candidate, O(N) instead of the O(N^2) of W K*.
// Same tile geometry and k* generation as sweep2_kernel (each wave produces the 16 x 16 tile (wave>>1, wave&1) of
// every 128-row stage) without the MFMA phase, the LDS stages and the barriers.  a.qpart is the state [3][M].
template <int FAM, int KA4>
__global__ __launch_bounds__(S2_NW * 64) void sweep2_rank1_kernel(SweepArgs a)
{
    constexpr int TCAND = IBO_S2_TCAND, KA = 4 * KA4;
    __shared__ double lds_c[TCAND * (KA + 1)];
    __shared__ double lds_m[3][S2_NW][16];
    __shared__ double lds_tab[2048];
    extern __shared__ __attribute__((aligned(16))) double lds_vec[];     // alphaY, alpha1, new row of W: NA128 each
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned tb = (a.tile_rows && a.part_all == 2) ? (unsigned)a.tile_sel[blockIdx.x] : blockIdx.x;      // (a compact list of tiles, as in sweep2_kernel)
    const int64_t tile0 = (int64_t)tb * TCAND;
    const int D = a.kp.D, Npad = a.Npad;
    const int NA128 = (Npad + 127) & ~127;
    // one row (a.rank1_row; < 0: the means only), or -- a kept state whose tiles are refreshed lazily (a.tile_rows) -- every appended
    // row this tile has not folded in yet, in order: the squares enter zsum in row order whenever the tile catches up, so its bits
    // do not depend on when that is; the last row's pass leaves the means formed from the current alpha vectors
    int row0 = a.rank1_row, row1 = a.rank1_row;
    if (a.tile_rows) {
        if (a.part_all != 2 && a.tile_sel && !a.tile_sel[tb]) return;
        row0 = a.part_rows + a.tile_rows[tb]; row1 = a.rank_hi - 1;
        if (row0 > row1) return;
    }
    lds_tab[tid] = a.exp_tab[tid];
    lds_tab[tid + 1024] = a.exp_tab[tid + 1024];
    for (int e = tid; e < NA128; e += S2_NW * 64) {
        lds_vec[e] = a.alphaY[e];
        lds_vec[NA128 + e] = a.alpha1[e];
    }
    for (int e = tid; e < TCAND * KA; e += S2_NW * 64) {
        const int c = e / KA, col = e - c * KA;
        int64_t gi = tile0 + c;
        if (gi > a.M - 1) gi = a.M - 1;
        lds_c[c * (KA + 1) + col] = (col < D) ? a.cand[gi * D + col] * a.kp.sw[col] : (col == D ? 1.0 : 0.0);
    }
    __syncthreads();
    if (tid < TCAND) {                               // as sweep2_kernel: radius guard, then b_c
        double n2 = 0.0;
        for (int d = 0; d < D; d++) { const double v = lds_c[tid * (KA + 1) + d]; n2 = fma(v, v, n2); }
        if (n2 > 6e5) {
            const double sc = sqrt(6e5 / n2);
            for (int d = 0; d < D; d++) lds_c[tid * (KA + 1) + d] *= sc;
            n2 = 6e5;
        }
        lds_c[tid * (KA + 1) + D + 1] = fma(-0.5, n2, FAM == FAM_SE ? a.log_sf2 : 0.0);
    }
    __syncthreads();
    const int rt = wave >> 1, gcb = wave & 1;
    const double *cfrag = &lds_c[(16 * gcb + (lane & 15)) * (KA + 1) + (lane >> 4)];
    const __amdgpu_buffer_rsrc_t rXA = s2_rsrc(a.XA, (size_t)(NA128 / 16) * KA4 * 64 * sizeof(double));
    const unsigned lane8 = lane * 8;
    const double *vq = lds_vec + (lane >> 4);
    for (int row = row0; row <= row1; row++) {
    for (int e = tid; e < NA128; e += S2_NW * 64) lds_vec[2 * NA128 + e] = (row >= 0 && e <= row) ? a.W[(size_t)row * Npad + e] : 0.0;
    __syncthreads();
    const int nstage = row >= 0 ? (row + 1 + S2_KCH - 1) / S2_KCH : NA128 / S2_KCH;      // (row < 0: only the means, over every row)
    double muY = 0.0, mu1 = 0.0, nu = 0.0;
    double xa[KA4], xn[KA4];
#pragma unroll
    for (int s = 0; s < KA4; s++) xa[s] = s2_ld_f64(rXA, lane8, (unsigned)((rt * KA4 + s) * 512));
    for (int t = 0; t < nstage; t++) {
        const int tile = t * (S2_KCH / 16) + rt;
        if (t + 1 < nstage) {
#pragma unroll
            for (int s = 0; s < KA4; s++) xn[s] = s2_ld_f64(rXA, lane8, (unsigned)(((tile + S2_KCH / 16) * KA4 + s) * 512));
        }
        d4_t y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KA4; s++) y = mfma_f64(xa[s], cfrag[4 * s], y);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int k = tile * 16 + 4 * r;
            const double ay = vq[k], a1v = vq[NA128 + k], om = vq[2 * NA128 + k];
            double kv;
            if (FAM == FAM_SE) kv = s2_exp(y[r], lds_tab);
            else {
                const double z = fmax(-2.0 * y[r], 0.0);
                const double rr = sqrt_fast((FAM == FAM_M3 ? 3.0 : 5.0) * z);
                const double poly = FAM == FAM_M3 ? 1.0 + rr : fma(rr, fma(rr, 1.0 / 3.0, 1.0), 1.0);
                kv = a.kp.sf2 * poly * s2_exp(-rr, lds_tab);
            }
            muY = fma(ay, kv, muY); mu1 = fma(a1v, kv, mu1); nu = fma(om, kv, nu);
        }
#pragma unroll
        for (int s = 0; s < KA4; s++) xa[s] = xn[s];
    }
    muY += __shfl_xor(muY, 16); muY += __shfl_xor(muY, 32);
    mu1 += __shfl_xor(mu1, 16); mu1 += __shfl_xor(mu1, 32);
    nu += __shfl_xor(nu, 16); nu += __shfl_xor(nu, 32);
    if (lane < 16) { lds_m[0][wave][lane] = muY; lds_m[1][wave][lane] = mu1; lds_m[2][wave][lane] = nu; }
    __syncthreads();
    if (tid < TCAND) {
        const int c = tid;
        double my = 0.0, m1 = 0.0, v = 0.0;
#pragma unroll
        for (int w = 0; w < S2_NW / 2; w++) {
            my += lds_m[0][2 * w + (c >> 4)][c & 15]; m1 += lds_m[1][2 * w + (c >> 4)][c & 15]; v += lds_m[2][2 * w + (c >> 4)][c & 15];
        }
        const int64_t li = tile0 + c;
        if (li < a.M) {
            if (row >= 0) a.qpart[3 * a.M + li] = fma(v, v, a.qpart[3 * a.M + li]);
            if (row == row1) { a.qpart[a.M + li] = my; a.qpart[2 * a.M + li] = m1; }
        }
    }
    __syncthreads();                                 // the row's vector and the partial sums are about to be rewritten
    }
    if (tid == 0 && a.tile_rows) a.tile_rows[tb] = row1 + 1 - a.part_rows;
}

// which tiles the next refresh / completion launches take: those whose bound reaches the threshold and that are not yet exact
__global__ void part_mark_kernel(const double *__restrict__ tile_ub, const int *__restrict__ tile_done, const int *__restrict__ tile_rows, int rows_all,
                                 const unsigned long long *__restrict__ thresh, int64_t ntiles, int *__restrict__ tile_sel, int all, double slack_abs, int last_level,
                                 int *__restrict__ list, int *__restrict__ count)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool sel = false;
    if (t < ntiles) {
        const unsigned long long th = *thresh;
        const bool complete = tile_done[t] == last_level;
        const bool exact = complete && tile_rows[t] >= rows_all;
        // all = 2: the complete tiles that lag behind (refreshing them is cheap and their values make the threshold)
        sel = !exact && (all == 2 ? complete : (all || tile_ub[t] >= s2_part_limit(th, slack_abs)));
        tile_sel[t] = sel;
    }
    // the selected tiles as a compact list (any order: tiles are independent), its length in *count -- the host reads the length and
    // launches exactly that many workgroups instead of one early-exit workgroup per tile of the array
    if (list) {
        const unsigned long long bal = __ballot(sel);
        const int lane = threadIdx.x & 63, n = __popcll(bal);
        int base = 0;
        if (lane == 0 && n) base = atomicAdd(count, n);
        base = __shfl(base, 0);
        if (sel) list[base + __popcll(bal & ((1ull << lane) - 1))] = (int)t;
    }
}

// XA: the observations as A-fragments of the exponent GEMM.  Row k of the augmented matrix is
// [x~_k (D), a_k, 1, 0...] (KA = 4 KA4 columns); fragment (tile, s) holds rows 16 tile + (lane&15), column
// 4 s + (lane>>4) at XA[(tile KA4 + s) 64 + lane].  Rows >= N (padding up to a multiple of 128) are zero.
__global__ void pack_xa_kernel(const double *__restrict__ Xs, const double *__restrict__ ak, int N, int DP, int D,
                               int KA4, int ntile, double *__restrict__ XA)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= ntile * KA4 * 64) return;
    const int lane = e & 63, s = (e >> 6) % KA4, tile = (e >> 6) / KA4;
    const int k = 16 * tile + (lane & 15), col = 4 * s + (lane >> 4);
    double v = 0.0;
    if (k < N) v = col < D ? Xs[(size_t)k * DP + col] : (col == D ? ak[k] : (col == D + 1 ? 1.0 : 0.0));
    XA[e] = v;
}

int launch_pack_xa(const double *Xs, const double *ak, int N, int Npad, int DP, int D, double *XA, hipStream_t s)
{
    const int KA4 = (D + 2 + 3) / 4;
    const int ntile = (Npad + 127) / 128 * 8;
    const int total = ntile * KA4 * 64;
    hipLaunchKernelGGL(pack_xa_kernel, dim3((total + 255) / 256), dim3(256), 0, s, Xs, ak, N, DP, D, KA4, ntile, XA);
    return (int)hipGetLastError();
}

// dynamic LDS: a window of the two alpha vectors, at most s2_awin rows (64 or 48 KiB); static: 64 KiB of k* stages,
// 16 KiB exp table, candidates (32 x 21 doubles up to 16 dimensions, 32 x 37 up to 32), q and mean partials (4 + 4 KiB):
// 93.3 / 97.3 KiB.  Any N whose packed W the 2 GiB buffer descriptor covers (16384 rows) fits; the refresh kernel keeps
// three whole vectors (24 B/row + 27.3 / 31.3 KiB).
bool sweep2_fits(int Npad) { return Npad <= 16384; }
bool sweep2_rank1_fits(int Npad, int D) { return (size_t)((Npad + 127) & ~127) * 24 + (D <= 18 ? 28 : 32) * 1024 <= 160 * 1024; }

template <int FAM, int KA4, bool BIGN>
static int launch_s2_var(const SweepArgs &a, int64_t ntiles, hipStream_t s)
{
    const int na128 = (a.Npad + 127) & ~127;
    const int dyn = (BIGN ? s2_awin(KA4) : na128) * 16;
    static std::atomic<int> granted[16];                          // per instantiation AND device (the attribute is per device): largest dynamic size already allowed
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dyn > granted[dev & 15]) {
        hipError_t e = hipFuncSetAttribute((const void *)sweep2_kernel<FAM, KA4, BIGN>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
        if (e != hipSuccess) return (int)e;
        granted[dev & 15] = dyn;
    }
    hipLaunchKernelGGL((sweep2_kernel<FAM, KA4, BIGN>), dim3((unsigned)ntiles), dim3(S2_NW * 64), dyn, s, a);
    return (int)hipGetLastError();
}
template <int FAM, int KA4>
static int launch_s2_one(const SweepArgs &a, int64_t ntiles, hipStream_t s)
{
    return ((a.Npad + 127) & ~127) > s2_awin(KA4) ? launch_s2_var<FAM, KA4, true>(a, ntiles, s) : launch_s2_var<FAM, KA4, false>(a, ntiles, s);
}

template <int FAM, int KA4>
static int launch_s2_rank1_one(const SweepArgs &a, int64_t ntiles, hipStream_t s)
{
    const int dyn = ((a.Npad + 127) & ~127) * 24;
    static std::atomic<int> granted[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dyn > granted[dev & 15]) {
        hipError_t e = hipFuncSetAttribute((const void *)sweep2_rank1_kernel<FAM, KA4>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
        if (e != hipSuccess) return (int)e;
        granted[dev & 15] = dyn;
    }
    hipLaunchKernelGGL((sweep2_rank1_kernel<FAM, KA4>), dim3((unsigned)ntiles), dim3(S2_NW * 64), dyn, s, a);
    return (int)hipGetLastError();
}

template <int FAM>
static int launch_s2_rank1_fam(const SweepArgs &a, int64_t ntiles, hipStream_t s)
{
    switch ((a.kp.D + 2 + 3) / 4) {
    case 1: return launch_s2_rank1_one<FAM, 1>(a, ntiles, s);
    case 2: return launch_s2_rank1_one<FAM, 2>(a, ntiles, s);
    case 3: return launch_s2_rank1_one<FAM, 3>(a, ntiles, s);
    case 4: return launch_s2_rank1_one<FAM, 4>(a, ntiles, s);
    case 5: return launch_s2_rank1_one<FAM, 5>(a, ntiles, s);
    case 6: return launch_s2_rank1_one<FAM, 6>(a, ntiles, s);
    case 7: return launch_s2_rank1_one<FAM, 7>(a, ntiles, s);
    case 8: return launch_s2_rank1_one<FAM, 8>(a, ntiles, s);
    default: return launch_s2_rank1_one<FAM, 9>(a, ntiles, s);
    }
}

template <int FAM>
static int launch_s2_fam(const SweepArgs &a, int64_t ntiles, hipStream_t s)
{
    switch ((a.kp.D + 2 + 3) / 4) {
    case 1: return launch_s2_one<FAM, 1>(a, ntiles, s);
    case 2: return launch_s2_one<FAM, 2>(a, ntiles, s);
    case 3: return launch_s2_one<FAM, 3>(a, ntiles, s);
    case 4: return launch_s2_one<FAM, 4>(a, ntiles, s);
    case 5: return launch_s2_one<FAM, 5>(a, ntiles, s);
    case 6: return launch_s2_one<FAM, 6>(a, ntiles, s);
    case 7: return launch_s2_one<FAM, 7>(a, ntiles, s);
    case 8: return launch_s2_one<FAM, 8>(a, ntiles, s);
    default: return launch_s2_one<FAM, 9>(a, ntiles, s);
    }
}

// ---- kept-state sweeps that run the second part of W only where it can matter ---------------------------------------------
template <int FAM, int KA4>
static int launch_s2_part_one(const SweepArgs &a, int64_t ntiles, hipStream_t s)
{
    const int dyn = a.part_means ? ((a.Npad + 127) & ~127) * 16 : 0;      // the alpha vectors, when this launch also forms the means
    static std::atomic<int> granted[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dyn > granted[dev & 15]) {
        hipError_t e = hipFuncSetAttribute((const void *)sweep2_kernel<FAM, KA4, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
        if (e != hipSuccess) return (int)e;
        granted[dev & 15] = dyn;
    }
    hipLaunchKernelGGL((sweep2_kernel<FAM, KA4, false, true>), dim3((unsigned)ntiles), dim3(S2_NW * 64), dyn, s, a);
    return (int)hipGetLastError();
}
template <int FAM>
static int launch_s2_part_fam(const SweepArgs &a, int64_t ntiles, hipStream_t s)
{
    switch ((a.kp.D + 2 + 3) / 4) {
    case 1: return launch_s2_part_one<FAM, 1>(a, ntiles, s);
    case 2: return launch_s2_part_one<FAM, 2>(a, ntiles, s);
    case 3: return launch_s2_part_one<FAM, 3>(a, ntiles, s);
    case 4: return launch_s2_part_one<FAM, 4>(a, ntiles, s);
    case 5: return launch_s2_part_one<FAM, 5>(a, ntiles, s);
    case 6: return launch_s2_part_one<FAM, 6>(a, ntiles, s);
    case 7: return launch_s2_part_one<FAM, 7>(a, ntiles, s);
    case 8: return launch_s2_part_one<FAM, 8>(a, ntiles, s);
    default: return launch_s2_part_one<FAM, 9>(a, ntiles, s);
    }
}
static int launch_s2_part(const SweepArgs &a, int lo, int hi, unsigned long long *thresh, hipStream_t s, int means = 0, int level = 0, int64_t grid = 0)
{
    const int64_t ntiles = grid > 0 ? grid : (a.M + IBO_S2_TCAND - 1) / IBO_S2_TCAND;      // (grid: the length of the compact list, a.part_all == 2)
    SweepArgs b = a;
    b.part_lo = lo; b.part_hi = hi; b.part_thresh = thresh; b.part_means = means; b.part_level = level;
    if (a.kp.family == FAM_SE) return launch_s2_part_fam<FAM_SE>(b, ntiles, s);
    if (a.kp.family == FAM_M3) return launch_s2_part_fam<FAM_M3>(b, ntiles, s);
    return launch_s2_part_fam<FAM_M5>(b, ntiles, s);
}
static int launch_s2_bound(const SweepArgs &a0, hipStream_t s)
{
    SweepArgs a = a0;
    a.out_mu = a.out_s2 = a.out_acq = nullptr;
    hipError_t e = hipMemsetAsync(a.part_best, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(acq_bound_kernel, dim3((unsigned)((a.M + 255) / 256)), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

// the part kernel has no moving alpha window and the means come from the refresh kernel: both must fit
bool sweep2_part_fits(int Npad, int D) { return Npad >= 512 && ((Npad + 127) & ~127) <= s2_awin((D + 2 + 3) / 4) && sweep2_rank1_fits(Npad, D); }
// The levels of a kept state (round 4; rounds 2-3 had two: rows [0, h) and [h, N) with h = N/2).  q = |W k*|^2 is a sum over W's rows
// with every term >= 0, so ANY prefix of the rows bounds the variance from above, and W is triangular: rows [0, N/8) are 1/64 of the
// MFMA work.  Level 0 = rows [0, h0) for every candidate (plus the means, which need every k* row); level l = rows [h_{l-1}, h_l) for the
// tiles whose bound still reaches the best exact value.  Splits at ~N/8, N/4, N/2 (multiples of 128: the k* stage; distinct; at least
// 256 rows; the last level ends at the model's rows): tools/argmax_bound_probe3.py -- on BASELINE config 3's shape no tile outside the top 3 % of the
// level-0 ranking survives even N/8 rows, 4.4 % of a full sweep's MFMA work against 27 % with the single split at N/2.
int sweep2_part_levels(int Npad, int *h)
{
    const int top = (Npad + 127) & ~127;
    int n = 0;
    for (int f = 8; f >= 2; f >>= 1) {
        // (at least 256 rows: the bound from 128 rows is too loose to be worth its 1/64 -- at N = 1024 half the tiles survived it,
        // 4.6 ms for the first sweep where splits at 256 and 512 take 4.1)
        const int v = ((Npad / f + 64) / 128) * 128;
        if (v >= 256 && v < top && (n == 0 || v > h[n - 1])) h[n++] = v;
    }
    if (n == 0) h[n++] = ((Npad / 2 + 64) / 128) * 128;
    return n + 1;
}
static std::atomic<int> g_part_maxlev{4};              // ibo_set_option("part_levels", 2..4): at most this many levels (2: the round-3 split at N/2)
void set_part_levels(int v) { g_part_maxlev = v < 2 ? 2 : (v > 4 ? 4 : v); }
// the splits of a state with `nlev` levels: the finest ones are dropped first (2 levels: the round-3 split at N/2)
static void part_splits(int Npad, int nlev, int *h)
{
    int all[3];
    const int n = sweep2_part_levels(Npad, all) - 1;
    const int keep = nlev - 1 < n ? nlev - 1 : n;
    for (int i = 0; i < keep; i++) h[i] = all[n - keep + i];
}
int sweep2_part_nlev(int Npad)                         // levels a NEW state gets (the switch is read when a state is formed, never afterwards)
{
    int all[3];
    const int n = sweep2_part_levels(Npad, all);
    return n < g_part_maxlev ? n : g_part_maxlev.load();
}

// levels [from, nlev) for the tiles the launch's filter lets through (a tile runs level l only if it stands at l - 1, so a tile taken
// further by one launch is picked up by the next)
static int launch_s2_levels(const SweepArgs &a, int from, unsigned long long *thresh, hipStream_t s)
{
    int h[3];
    const int nlev = a.part_nlev, hi = (a.part_rows + 15) & ~15;
    part_splits(a.Npad, nlev, h);
    for (int l = from < 1 ? 1 : from; l < nlev; l++) {
        const int rc = launch_s2_part(a, h[l - 1], l + 1 < nlev ? h[l] : hi, thresh, s, 0, l);
        if (rc) return rc;
    }
    return 0;
}

// First sweep of a kept state (a.qpart = [q_a, aY.k*, a1.k*, zsum, q_b][M], zsum and q_b and a.tile_done zeroed by the caller;
// a.part_best / a.part_thresh: two device words): level 0 for everyone with the means, the bounds, then every later level for the tiles at
// the top of the bound ranking, then -- against the best value THOSE reached -- level by level for whoever's bound still reaches it, the
// bounds re-formed after each level.  A tile left incomplete has a bound below a value that a complete candidate attains: it cannot hold the
// maximum, and the arg-max over (exact where complete, bound elsewhere) is the arg-max of the full sweep.  prune = false completes every tile.
int launch_sweep2_pruned(const SweepArgs &a_in, bool prune, hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    SweepArgs a = a_in;
    a.tile_sel = nullptr;                            // (the first sweep selects by threshold; a_in.tile_rows is all zeros: everyone is fresh)
    int h[3];
    const int nlev = a.part_nlev;                    // (set by the caller from sweep2_part_nlev when the state is formed)
    part_splits(a.Npad, nlev, h);
    const int64_t ntiles = (a.M + IBO_S2_TCAND - 1) / IBO_S2_TCAND;
    if (e0) (void)hipEventRecord(e0, s);
    int rc = launch_s2_part(a, 0, h[0], a.part_thresh, s, 1);        // (the means ride along with the first level)
    if (rc) return rc;
    if (prune) {
        if ((rc = launch_s2_bound(a, s))) return rc;                               // bounds; nothing complete yet
        hipLaunchKernelGGL(part_select_kernel, dim3(1), dim3(1024), 0, s, (const double *)a.tile_ub, ntiles, a.part_thresh);
        if ((rc = launch_s2_levels(a, 1, a.part_thresh, s))) return rc;            // the top of the ranking, all the way
        // everyone who can still reach what they reached: one level at a time, over a compact list of the tiles concerned -- usually
        // nobody (BASELINE config 3's shape: no tile outside the first ones survives even the first level), and then that is all
        int *sel = a_in.tile_sel, *list = sel + ntiles, *counters = sel + 2 * ntiles;
        if ((rc = (int)hipMemsetAsync(counters, 0, 16 * sizeof(int), s))) return rc;
        SweepArgs al = a;
        al.part_all = 2; al.tile_sel = list;
        for (int l = 1; l < nlev; l++) {
            if ((rc = launch_s2_bound(a, s))) return rc;                           // (best exact value so far; every tile's bound at its own level)
            int n = 0;
            hipLaunchKernelGGL(part_mark_kernel, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, s, (const double *)a.tile_ub, (const int *)a.tile_done,
                               (const int *)a.tile_rows, 0, (const unsigned long long *)a.part_best, ntiles, sel, 0, a.part_slack, nlev - 1, list, counters + l);
            hipError_t e = hipMemcpyAsync(&n, counters + l, sizeof(int), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) return (int)e;
            if (n == 0) break;
            if ((rc = launch_s2_part(al, h[l - 1], l + 1 < nlev ? h[l] : ((a.part_rows + 15) & ~15), a.part_best, s, 0, l, n))) return rc;
        }
    } else {
        SweepArgs b = a;
        b.part_all = 1;
        if ((rc = launch_s2_levels(b, 1, a.part_thresh, s))) return rc;
    }
    if (e1) (void)hipEventRecord(e1, s);
    const int64_t nfin = (a.M + 255) / 256;
    hipLaunchKernelGGL(acq_finish_kernel, dim3((unsigned)nfin), dim3(256), 0, s, a);
    rc = (int)hipGetLastError();
    if (rc) return rc;
    return launch_argmax_final(a, nfin, s);
}

// after the rank-1 launches of a refresh on such a state (zsum and the means are current): the tiles whose bound now reaches the
// best complete value get their remaining levels; to be followed by acq_finish_kernel (launch_sweep2_refresh does both)
int launch_sweep2_complete(const SweepArgs &a, hipStream_t s)
{
    int rc = launch_s2_bound(a, s);
    if (rc) return rc;
    return launch_s2_levels(a, 1, a.part_best, s);
}

// every incomplete tile of such a state gets its remaining levels (a caller that needs each candidate's full variance)
int launch_sweep2_pruned_finish_all(const SweepArgs &a, hipStream_t s)
{
    SweepArgs b = a;
    b.part_all = 1; b.tile_sel = nullptr;
    return launch_s2_levels(b, 1, a.part_thresh, s);
}

// rows [row_first, row_last] were appended to the model since a.qpart (the kept state [3][M]) was last brought up to
// date: one refresh launch per row, then the acquisition as after a full sweep
// The same for a state with incomplete tiles (a.tile_done, a.tile_rows, a.tile_sel set): nothing is refreshed that cannot matter.
// The complete tiles -- few, and the best of earlier rounds -- fold in the rows they are missing; the best value they now reach is
// the threshold: whoever else's bound (from its stale state: means widened by the drift margin, variance only too large) reaches it
// folds its rows in and gets its remaining levels; the arg-max then runs over everyone -- a tile left stale has a bound below a value
// that an exact candidate attains.
// lazy = false: every tile is refreshed and completed by the same launches (the reference the lazy run is held to, and the route
// when a mean prior is set).
static int launch_sweep2_refresh_lazy(const SweepArgs &a0, bool lazy, hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    const int64_t ntiles = (a0.M + IBO_S2_TCAND - 1) / IBO_S2_TCAND;
    const int rows_all = a0.rank_hi - a0.part_rows;
    const unsigned nmark = (unsigned)((ntiles + 255) / 256);
    // a0.tile_sel: [flags | list | 16 counters] (abi.hip sizes it so)
    int *list = a0.tile_sel + ntiles, *counters = a0.tile_sel + 2 * ntiles;
    if (e0) (void)hipEventRecord(e0, s);
    int rc;
    // mark the tiles at or above *thresh that are not exact (all = 1: every one; 2: the complete ones that lag behind); returns how many
    // (the host waits for the number -- ~15 us -- and what follows runs on exactly that many workgroups, or not at all)
    int slot = 0;
    auto mark = [&](unsigned long long *thresh, int all, int *n) -> int {
        int *cnt = counters + (slot++ & 15);
        hipLaunchKernelGGL(part_mark_kernel, dim3(nmark), dim3(256), 0, s, (const double *)a0.tile_ub, (const int *)a0.tile_done, (const int *)a0.tile_rows,
                           rows_all, (const unsigned long long *)thresh, ntiles, a0.tile_sel, all, a0.part_slack, a0.part_nlev - 1, list, cnt);
        hipError_t e = hipMemcpyAsync(n, cnt, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        return (int)e;
    };
    auto rank1 = [&](const SweepArgs &a, int64_t grid) -> int {          // fold the appended rows into the listed (or flagged) tiles
        if (a.kp.family == FAM_SE) return launch_s2_rank1_fam<FAM_SE>(a, grid, s);
        if (a.kp.family == FAM_M3) return launch_s2_rank1_fam<FAM_M3>(a, grid, s);
        return launch_s2_rank1_fam<FAM_M5>(a, grid, s);
    };
    if ((rc = (int)hipMemsetAsync(counters, 0, 16 * sizeof(int), s))) return rc;
    SweepArgs al = a0;                                // the same launch arguments over the compact list
    al.rank1_row = 0; al.part_all = 2; al.tile_sel = list;
    if (!lazy) {
        SweepArgs a = a0;
        a.rank1_row = 0;
        hipLaunchKernelGGL(part_mark_kernel, dim3(nmark), dim3(256), 0, s, (const double *)a0.tile_ub, (const int *)a0.tile_done, (const int *)a0.tile_rows,
                           rows_all, (const unsigned long long *)a0.part_thresh, ntiles, a0.tile_sel, 1, a0.part_slack, a0.part_nlev - 1, (int *)nullptr, (int *)nullptr);
        if ((rc = rank1(a, ntiles))) return rc;
        if ((rc = launch_s2_levels(a0, 1, a0.part_thresh, s))) return rc;          // (the flags select every tile that is not exact: every level for them)
    } else {
        // the complete tiles catch up (cheap, and their values make the threshold); then, level by level, whoever's bound -- re-formed
        // after each level, from fresh means once a tile has been refreshed -- still reaches the best exact value goes one level further
        int n = 0;
        if ((rc = mark(a0.part_thresh, 2, &n))) return rc;
        if (n > 0 && (rc = rank1(al, n))) return rc;
        if ((rc = launch_s2_bound(a0, s))) return rc;
        int h[3];
        part_splits(a0.Npad, a0.part_nlev, h);
        for (int l = 1; l < a0.part_nlev; l++) {
            if ((rc = mark(a0.part_best, 0, &n))) return rc;
            if (n == 0) break;                         // nobody's bound reaches the best exact value: the round is decided
            if ((rc = rank1(al, n))) return rc;
            if ((rc = launch_s2_part(al, h[l - 1], l + 1 < a0.part_nlev ? h[l] : ((a0.part_rows + 15) & ~15), a0.part_best, s, 0, l, n))) return rc;
            if (l + 1 < a0.part_nlev && (rc = launch_s2_bound(a0, s))) return rc;
        }
    }
    if (e1) (void)hipEventRecord(e1, s);
    const int64_t nfin = (a0.M + 255) / 256;
    hipLaunchKernelGGL(acq_finish_kernel, dim3((unsigned)nfin), dim3(256), 0, s, a0);
    rc = (int)hipGetLastError();
    if (rc) return rc;
    return launch_argmax_final(a0, nfin, s);
}

int launch_sweep2_refresh(const SweepArgs &a0, int row_first, int row_last, hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    if (a0.tile_done && a0.tile_rows) return launch_sweep2_refresh_lazy(a0, a0.part_lazy != 0, s, e0, e1);
    const int64_t ntiles = (a0.M + IBO_S2_TCAND - 1) / IBO_S2_TCAND;
    if (e0) (void)hipEventRecord(e0, s);
    for (int r = row_first; r <= row_last; r++) {
        SweepArgs a = a0;
        a.rank1_row = r;
        int rc;
        if (a.kp.family == FAM_SE) rc = launch_s2_rank1_fam<FAM_SE>(a, ntiles, s);
        else if (a.kp.family == FAM_M3) rc = launch_s2_rank1_fam<FAM_M3>(a, ntiles, s);
        else rc = launch_s2_rank1_fam<FAM_M5>(a, ntiles, s);
        if (rc) return rc;
    }
    if (a0.tile_done) {                              // a state with incomplete tiles: whoever's bound has caught up is completed first
        int rc = launch_sweep2_complete(a0, s);
        if (rc) return rc;
    }
    if (e1) (void)hipEventRecord(e1, s);
    const int64_t nfin = (a0.M + 255) / 256;
    hipLaunchKernelGGL(acq_finish_kernel, dim3((unsigned)nfin), dim3(256), 0, s, a0);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    return launch_argmax_final(a0, nfin, s);
}

int launch_sweep2(const SweepArgs &a, hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    const int64_t ntiles = (a.M + IBO_S2_TCAND - 1) / IBO_S2_TCAND;
    int rc;
    if (e0) (void)hipEventRecord(e0, s);
    if (a.kp.family == FAM_SE) rc = launch_s2_fam<FAM_SE>(a, ntiles, s);
    else if (a.kp.family == FAM_M3) rc = launch_s2_fam<FAM_M3>(a, ntiles, s);
    else rc = launch_s2_fam<FAM_M5>(a, ntiles, s);
    if (e1) (void)hipEventRecord(e1, s);
    if (rc) return rc;
    const int64_t nfin = (a.M + 255) / 256;
    hipLaunchKernelGGL(acq_finish_kernel, dim3((unsigned)nfin), dim3(256), 0, s, a);
    rc = (int)hipGetLastError();
    if (rc) return rc;
    return launch_argmax_final(a, nfin, s);
}
