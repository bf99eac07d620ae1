// sweep2_kernels.h -- the templated kernels of sweep2.hip and their launchers.  The 3 families x 9 exponent-GEMM depths x 4 kernels
// (sweep2_kernel plain / moving alpha window / PART, sweep2_rank1_kernel) are 108 heavy instantiations: sweep2_fam.hip compiles them in six
// pieces (family x {full sweeps, kept-state kernels}) that build side by side; sweep2.hip holds the host logic and the small kernels.
#pragma once
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
int launch_s2_rank1_fam(const SweepArgs &a, int64_t ntiles, hipStream_t s)
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
int launch_s2_fam(const SweepArgs &a, int64_t ntiles, hipStream_t s)
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
int launch_s2_part_fam(const SweepArgs &a, int64_t ntiles, hipStream_t s)
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
