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

#include "sweep2_kernels.h"

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


// the templated kernels' launchers are instantiated in sweep2_fam.hip's six objects, not here
extern template int launch_s2_fam<FAM_SE>(const SweepArgs &, int64_t, hipStream_t);
extern template int launch_s2_rank1_fam<FAM_SE>(const SweepArgs &, int64_t, hipStream_t);
extern template int launch_s2_part_fam<FAM_SE>(const SweepArgs &, int64_t, hipStream_t);
extern template int launch_s2_fam<FAM_M3>(const SweepArgs &, int64_t, hipStream_t);
extern template int launch_s2_rank1_fam<FAM_M3>(const SweepArgs &, int64_t, hipStream_t);
extern template int launch_s2_part_fam<FAM_M3>(const SweepArgs &, int64_t, hipStream_t);
extern template int launch_s2_fam<FAM_M5>(const SweepArgs &, int64_t, hipStream_t);
extern template int launch_s2_rank1_fam<FAM_M5>(const SweepArgs &, int64_t, hipStream_t);
extern template int launch_s2_part_fam<FAM_M5>(const SweepArgs &, int64_t, hipStream_t);


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

void ibo_touch_sweep2() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, (const void *)acq_finish_kernel); }     // (see small2.hip: ibo_touch_small2)
