// direct_host.cpp -- DIRECT with batched objective evaluation.
//
// The search is the one the reference's native optimiser performs
// (cpp/direct.cpp:329-581: driver and potentially-optimal test; :111-141 sample
// mapping and incumbent; :146-235 division), restructured so that the objective
// is called on arrays of points: per iteration all split-point probes of all
// potentially-optimal rectangles form one batch and all child centres a second
// one (GPU objective), or per rectangle (host callbacks, identical call order
// to the reference).  The incumbent is then replayed in the reference's
// sequential order, so (fmin, xmin, nsamples) are those of the sequential run.
//
// Quirks kept on purpose: probes at lb+len/3 and lb+2len/3 order
// the dimensions, child centres are sampled separately; strict '<' incumbent;
// reverse-order division; I1 slope seeded with DBL_MIN; epsilon = 1e-9;
// with compat the longest-side search is seeded from dimension 0 without
// looking at fixed[0], which stalls the search when dimension 0 is fixed.
#include "direct_host.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <emmintrin.h>

namespace ibo {
namespace {

// Rectangles, structure-of-arrays, in insertion order.  A divided rectangle is only marked dead: indices stay valid, the
// relative order of the living -- which is what the reference's erase-from-the-vector leaves, and what the order of the
// potentially-optimal set depends on -- is unchanged, and nothing is copied (the compaction pass cost 20-40 us per
// iteration with a few thousand rectangles in 8 dimensions).
// Every rectangle also carries the number of its size class: half-diagonals d come from a small set (one per
// division depth pattern), compared with exact equality as the reference compares them, and the potentially-optimal test
// only needs, per class, the smallest y.  Classes are numbered as they appear; `by_size` lists them in ascending d.
struct Pool {
    int D = 0;
    std::vector<double> lb, ub, ctr;   // size() * D
    std::vector<double> y, d;
    std::vector<int> cls;              // size class of each rectangle
    std::vector<char> alive;
    std::vector<double> cls_d;         // per class: its d
    std::vector<int> by_size;          // class numbers, ascending d
    // per class, kept up to date by push / erase_many: the living members, the smallest y among them and the members that
    // attain it -- all the potentially-optimal test looks at (it used to walk every rectangle twice per iteration)
    std::vector<std::vector<int>> members, argmin;
    std::vector<double> cls_best;
    std::vector<int> pos;              // where rectangle j sits in members[cls[j]]
    size_t size() const { return y.size(); }
    // a pool is kept from search to search (Workspace below): emptied, its storage -- the per-class vectors' too -- stays
    void reset(int dims)
    {
        D = dims;
        lb.clear(); ub.clear(); ctr.clear(); y.clear(); d.clear(); cls.clear(); alive.clear(); pos.clear();
        cls_d.clear(); by_size.clear(); cls_best.clear();
        last_dd = -1.0; last_cls = -1;
    }
    double last_dd = -1.0; int last_cls = -1;      // (the two children of a cut have the same d: every second look-up is the one before)
    int class_of(double dd)
    {
        if (dd == last_dd) return last_cls;
        const int c = class_find(dd);
        last_dd = dd; last_cls = c;
        return c;
    }
    int class_find(double dd)
    {
        size_t lo = 0, hi = by_size.size();
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (cls_d[by_size[mid]] < dd) lo = mid + 1; else hi = mid;
        }
        if (lo < by_size.size() && cls_d[by_size[lo]] == dd) return by_size[lo];
        const int c = (int)cls_d.size();
        cls_d.push_back(dd);
        by_size.insert(by_size.begin() + lo, c);
        if ((size_t)c == members.size()) { members.emplace_back(); argmin.emplace_back(); }
        else { members[c].clear(); argmin[c].clear(); }      // (an earlier search's vectors: their storage is reused)
        cls_best.push_back(0.0);
        return c;
    }
    void push(const double *l, const double *u, const double *c, double yy, double dd)
    {
        lb.insert(lb.end(), l, l + D); ub.insert(ub.end(), u, u + D); ctr.insert(ctr.end(), c, c + D);
        const int j = (int)y.size();
        y.push_back(yy); d.push_back(dd);
        const int k = class_of(dd);
        cls.push_back(k); alive.push_back(1);
        if (members[k].empty() || yy < cls_best[k]) { cls_best[k] = yy; argmin[k].assign(1, j); }
        else if (yy == cls_best[k]) argmin[k].push_back(j);
        pos.push_back((int)members[k].size());
        members[k].push_back(j);
    }
    void erase_many(const std::vector<size_t> &dead)
    {
        for (size_t j : dead) {
            alive[j] = 0;
            const int c = cls[j];
            std::vector<int> &m = members[c];
            m[pos[j]] = m.back(); pos[m.back()] = pos[j]; m.pop_back();
            std::vector<int> &a = argmin[c];
            for (size_t k = 0; k < a.size(); k++) if (a[k] == (int)j) { a.erase(a.begin() + k); break; }
            if (a.empty() && !m.empty()) {                  // the class lost its best member(s): look again
                double b = y[m[0]];
                for (int k : m) if (y[k] < b) b = y[k];
                cls_best[c] = b;
                for (int k : m) if (y[k] == b) a.push_back(k);
                std::sort(a.begin(), a.end());
            }
        }
    }
};

struct Division {             // the two-phase division of one rectangle
    size_t src;
    std::vector<double> lb, ub, ctr;   // copy of the rectangle
    double y;
    std::vector<int> dims;             // divided dimensions, in probe order
    std::vector<int> dims_probe;       // the same (plan_guess's slots are numbered by it)
    std::vector<double> probes;        // 2*dims x D unit-cube points
    std::vector<double> probe_vals;
    std::vector<double> kid_lb, kid_ub, kid_ctr, kid_d;   // 2*dims children
    std::vector<double> kid_vals;
    double mid_d;
    // child centres guessed before the probe values are known (one GPU batch per iteration instead of two, see plan_guess)
    std::vector<double> guess;         // 2*dims x D: (dimension dims[q], side 0 / 1) at 2 q + side
    std::vector<double> guess_vals;
};

struct Search {
    int D;
    std::vector<double> lo, hi;
    std::vector<char> fixed;
    double fmin = DBL_MAX;
    std::vector<double> xmin;
    int64_t nsamples = 0;
    bool compat;
    const batch_eval_t *eval;
    std::vector<double> mapped;
    mutable std::vector<double> sc_best, sc_l, sc_u, sc_cl, sc_cu, sc_cc;   // plan_children's scratch (no allocation per division)
    mutable std::vector<int> sc_order;

    // unit cube -> caller's box (cpp/direct.cpp:113-120)
    int evaluate(const double *unit, int n, double *vals)
    {
        mapped.resize((size_t)n * D);
        for (int p = 0; p < n; p++)
            for (int i = 0; i < D; i++)
                mapped[(size_t)p * D + i] = fixed[i] ? lo[i] : unit[(size_t)p * D + i] * (hi[i] - lo[i]) + lo[i];
        return (*eval)(mapped.data(), n, vals);
    }
    // incumbent update in sequential order (cpp/direct.cpp:122-139)
    void account(const double *unit, double v)
    {
        nsamples++;
        if (v < fmin) {
            fmin = v;
            for (int i = 0; i < D; i++) xmin[i] = lo[i] + (hi[i] - lo[i]) * unit[i];
        }
    }

    void plan_probes(Division &dv) const
    {
        const double *l = dv.lb.data(), *u = dv.ub.data();
        double maxlen;
        if (compat) {
            maxlen = u[0] - l[0];
            for (int i = 1; i < D; i++) if (!fixed[i] && u[i] - l[i] > maxlen) maxlen = u[i] - l[i];
        } else {
            maxlen = -1.0;
            for (int i = 0; i < D; i++) if (!fixed[i] && u[i] - l[i] > maxlen) maxlen = u[i] - l[i];
        }
        dv.dims.clear(); dv.probes.clear();
        for (int i = 0; i < D; i++) {
            if (!fixed[i] && u[i] - l[i] == maxlen) {
                dv.dims.push_back(i);
                size_t o = dv.probes.size();
                dv.probes.insert(dv.probes.end(), dv.ctr.begin(), dv.ctr.end());
                dv.probes.insert(dv.probes.end(), dv.ctr.begin(), dv.ctr.end());
                dv.probes[o + i] = l[i] + maxlen / 3.;
                dv.probes[o + D + i] = l[i] + 2. * maxlen / 3.;
            }
        }
        dv.probe_vals.assign(dv.dims.size() * 2, 0.0);
        dv.dims_probe = dv.dims;
    }

    // The child centres do not depend on the ORDER in which the longest sides are cut, although the children's boxes do:
    // a child of side dd has the rectangle's centre in every other coordinate -- a side cut earlier has shrunk to its
    // middle third [s1, s2] by then, whose centre s1 + (s2 - s1) / 2 is the old l + (u - l) / 2 -- and the centre of an
    // outer third in dd.  In floating point the two expressions agree bit for bit except once in ~10^4 intervals, so the
    // children are sampled in the SAME batch as the probes, at the centres computed here, and plan_children's real
    // centres are compared with them afterwards: only a centre that differs in some bit is evaluated again, so every
    // value is the objective at exactly the point the sequential search samples.
    void plan_guess(Division &dv) const
    {
        const size_t m = dv.dims.size();
        dv.guess.resize(2 * m * D);
        for (size_t q = 0; q < m; q++) {
            const int dd = dv.dims[q];
            const double l = dv.lb[dd], u = dv.ub[dd], w = u - l;
            const double s1 = l + w / 3., s2 = l + 2. * w / 3.;
            double *g0 = &dv.guess[(2 * q) * D], *g1 = g0 + D;
            for (int i = 0; i < D; i++) g0[i] = g1[i] = dv.ctr[i];
            g0[dd] = l + (s1 - l) / 2.;
            g1[dd] = s2 + (u - s2) / 2.;
        }
        dv.guess_vals.assign(2 * m, 0.0);
    }
    // after plan_children: children whose centre is the guessed one take its value; the others are appended to `redo`
    // (as indices into dv.kid_ctr) for a second evaluation
    void take_guess(Division &dv, std::vector<size_t> &redo) const
    {
        const size_t m = dv.dims.size();
        redo.clear();
        for (size_t q = 0; q < m; q++) {
            const int dd = sc_order[q];
            size_t pos = 0;
            while (dv.dims_probe[pos] != dd) pos++;
            for (int side = 0; side < 2; side++) {
                const double *kc = &dv.kid_ctr[(2 * q + side) * D], *gc = &dv.guess[(2 * pos + side) * D];
                if (std::memcmp(kc, gc, sizeof(double) * D) == 0) dv.kid_vals[2 * q + side] = dv.guess_vals[2 * pos + side];
                else redo.push_back(2 * q + side);
            }
        }
    }

    void plan_children(Division &dv) const
    {
        const size_t m = dv.dims.size();
        std::vector<double> &best = sc_best;
        best.resize(m);
        for (size_t q = 0; q < m; q++) {
            double f1 = dv.probe_vals[2 * q], f2 = dv.probe_vals[2 * q + 1];
            best[q] = (f1 < f2) ? f1 : f2;
        }
        // ascending by value as cpp/direct.cpp:194 orders them: std::sort with a value-only comparison.  libstdc++ sorts
        // <= 16 elements by (stable) insertion; beyond that -- more than 16 longest sides, i.e. 17..32 dimensions -- the
        // order among EQUAL values is whatever its introsort leaves, so the library routine itself is called
        std::vector<int> &order = sc_order;
        order.assign(dv.dims.begin(), dv.dims.end());
        if (m > 16) {
            std::vector<std::pair<unsigned, double>> iv(m);
            for (size_t a = 0; a < m; a++) iv[a] = std::make_pair((unsigned)order[a], best[a]);
            std::sort(iv.begin(), iv.end(), [](const std::pair<unsigned, double> &a, const std::pair<unsigned, double> &b) { return a.second < b.second; });
            for (size_t a = 0; a < m; a++) { order[a] = (int)iv[a].first; best[a] = iv[a].second; }
        } else {
            for (size_t a = 1; a < m; a++) {
                int da = order[a]; double va = best[a];
                size_t b = a;
                while (b > 0 && va < best[b - 1]) { order[b] = order[b - 1]; best[b] = best[b - 1]; b--; }
                order[b] = da; best[b] = va;
            }
        }
        std::vector<double> &l = sc_l, &u = sc_u, &cl = sc_cl, &cu = sc_cu, &cc = sc_cc;
        l.assign(dv.lb.begin(), dv.lb.end()); u.assign(dv.ub.begin(), dv.ub.end()); cl.resize(D); cu.resize(D); cc.resize(D);
        // a child is the rectangle as cut so far with ONE side replaced: written straight into the division's arrays (the sums below run over
        // the dimensions in the same order with the same operands as a copy-then-modify would give them)
        dv.kid_lb.resize(2 * m * D); dv.kid_ub.resize(2 * m * D); dv.kid_ctr.resize(2 * m * D); dv.kid_d.resize(2 * m);
        auto add_kid = [&](size_t k, int side_dim, double lo_v, double hi_v) {
            double *kl = &dv.kid_lb[k * D], *ku = &dv.kid_ub[k * D], *kc = &dv.kid_ctr[k * D];
            double dd = 0.0;
            for (int i = 0; i < D; i++) {
                const double a = i == side_dim ? lo_v : l[i], b = i == side_dim ? hi_v : u[i];
                kl[i] = a; ku[i] = b;
                kc[i] = a + (b - a) / 2.;
                dd += (a - kc[i]) * (a - kc[i]);
            }
            dv.kid_d[k] = std::sqrt(dd);
        };
        (void)cl; (void)cu; (void)cc;
        for (size_t q = 0; q < m; q++) {
            int dd = order[q];
            double w = u[dd] - l[dd];
            double s1 = l[dd] + w / 3.;
            double s2 = l[dd] + 2. * w / 3.;
            add_kid(2 * q, dd, l[dd], s1);
            add_kid(2 * q + 1, dd, s2, u[dd]);
            l[dd] = s1; u[dd] = s2;
        }
        double md = 0.0;
        for (int i = 0; i < D; i++) md += (l[i] - dv.ctr[i]) * (l[i] - dv.ctr[i]);
        dv.mid_d = std::sqrt(md);
        dv.lb = l; dv.ub = u;           // the middle rectangle keeps centre and value
        dv.kid_vals.assign(2 * m, 0.0);
    }

    // replay in reference order and materialise the new rectangles
    void apply(const Division &dv, Pool &pool)
    {
        const size_t m = dv.dims.size();
        for (size_t q = 0; q < 2 * m; q++) account(&dv.probes[q * D], dv.probe_vals[q]);
        for (size_t q = 0; q < 2 * m; q++) account(&dv.kid_ctr[q * D], dv.kid_vals[q]);
        for (size_t q = 0; q < 2 * m; q++)
            pool.push(&dv.kid_lb[q * D], &dv.kid_ub[q * D], &dv.kid_ctr[q * D], dv.kid_vals[q], dv.kid_d[q]);
        pool.push(dv.lb.data(), dv.ub.data(), dv.ctr.data(), dv.y, dv.mid_d);
    }
};

void make_division(const Pool &pool, size_t j, Division &dv)      // dv is reused from iteration to iteration: its vectors keep their storage
{
    const int D = pool.D;
    dv.src = j;
    dv.lb.assign(pool.lb.begin() + j * D, pool.lb.begin() + (j + 1) * D);
    dv.ub.assign(pool.ub.begin() + j * D, pool.ub.begin() + (j + 1) * D);
    dv.ctr.assign(pool.ctr.begin() + j * D, pool.ctr.begin() + (j + 1) * D);
    dv.y = pool.y[j];
    dv.mid_d = pool.d[j];
}

// potentially-optimal rectangles (cpp/direct.cpp:378-471).  The reference tests every rectangle j against every
// other one: rejected if a rectangle of the same size d has a smaller y, or one with a larger d has y <= y_j
// (slope <= 0), or min slope to the larger ones < max slope from the smaller ones (maxI1 seeded with DBL_MIN,
// so non-positive slopes from below never count); kept otherwise subject to the epsilon test.  All of that
// only involves, per distinct size, the smallest y of that size (a slope to a fixed size is monotone in y, in
// floating point too): same decisions, same floating-point values, in two passes over the living rectangles --
// O(R + candidates x sizes) instead of O(R^2), with no sort (the size classes are kept by the pool).
// FINITE objective values are assumed for "same decisions": the reference's all-pairs scan selects a rectangle whose
// value is NaN (every comparison with it is false) and lets equal +/-inf values through a NaN slope, while here a NaN
// never becomes a class minimum and equal infinities are rejected by the suffix-minimum test.  The GP acquisitions
// this search is run on are finite by construction (clamped variance, finite means); a host callback that returns
// NaN or inf gets a well-defined search, not the reference's trajectory.
// Two things keep the test off most classes.  (1) The decision is a function of (y, d) alone: it is taken once per CLASS -- for the class's best
// value -- and holds for every member that attains it.  (2) A class that is not a vertex of the lower convex hull of the points (d, best y) has
// two hull vertices around it, h1 smaller and h2 larger, and in exact arithmetic the slope from h1 exceeds the slope to h2: the full test would
// find minI2 < maxI1.  In floating point that is not a proof -- but the full test's maxI1 is a maximum over the very quotient s1 formed here and its
// minI2 a minimum over the very quotient s2, so "s2 <= 0" or "s1 > DBL_MIN and s2 < s1" IS one (minI2 <= s2 < s1 <= maxI1): such a class is
// rejected by two divisions instead of one per class; everything else -- hull vertices, near-ties, NaNs -- takes the full test.  The hull itself
// is only a way of finding good witnesses: no decision rests on how it was rounded.  The same set in the same (pool) order as the full test on
// every candidate (potentially_optimal_ref, compared call by call under IBO_DIRECT_SELFCHECK: tools/direct_host_check.cpp).
void potentially_optimal(const Pool &pool, double fmin, std::vector<size_t> &out)
{
    const double eps = 10e-10;
    const double *Y = pool.y.data(), *Dd = pool.d.data();
    out.clear();
    // (scratch kept between calls: this runs once per iteration of every search)
    static thread_local std::vector<double> gd, gy, sufmin;    // the classes that have members, ascending d; their smallest y; min of gy beyond
    static thread_local std::vector<int> gc, hull, before;     // their class numbers; the hull's vertices; the last hull vertex at or before g
    gd.clear(); gy.clear(); gc.clear();
    for (int c : pool.by_size)
        if (!pool.members[c].empty()) { gc.push_back(c); gd.push_back(pool.cls_d[c]); gy.push_back(pool.cls_best[c]); }
    const size_t G = gd.size();
    // A rectangle is rejected as soon as some LARGER class holds a value <= its own (the slope to it is <= 0: minI2 <= 0).
    // With the minimum of gy over the larger classes at hand that is one comparison, and it removes all but the few classes on
    // the descending staircase before any slope is formed -- the same decisions as forming them all (a quotient with a positive
    // denominator is <= 0 exactly when its numerator is).  NaN values fall through to the full test.
    sufmin.assign(G + 1, DBL_MAX);
    for (size_t h = G; h-- > 0;) sufmin[h] = (gy[h] < sufmin[h + 1] || gy[h] != gy[h]) ? gy[h] : sufmin[h + 1];
    // lower hull by the monotone chain (a vertex that lies on or above the segment around it is dropped), over the classes that passed the
    // comparison above only: they are the hull's vertices from the lowest value on (beyond it a vertex lies below everything larger), and no
    // other class asks for witnesses
    hull.clear();
    before.assign(G, 0);
    for (size_t g = 0; g < G; g++) {
        if (g + 1 < G && sufmin[g + 1] <= gy[g]) continue;
        while (hull.size() >= 2) {
            const int a = hull[hull.size() - 2], b = hull.back();
            if ((gy[b] - gy[a]) * (gd[g] - gd[a]) >= (gy[g] - gy[a]) * (gd[b] - gd[a])) hull.pop_back();
            else break;
        }
        hull.push_back((int)g);
    }
    for (size_t k = 0; k < hull.size(); k++) {
        const size_t g1 = k + 1 < hull.size() ? (size_t)hull[k + 1] : G;
        for (size_t g = (size_t)hull[k]; g < g1; g++) before[g] = (int)k;
    }
    for (size_t g = 0; g < G; g++) {
        if (g + 1 < G && sufmin[g + 1] <= gy[g]) continue;      // every member of the class fails the minI2 test
        const std::vector<int> &am = pool.argmin[gc[g]];
        if (am.empty()) continue;
        const size_t j = (size_t)am[0];
        const double yj = Y[j], dj = Dd[j];
        const size_t k = (size_t)before[g];
        if ((size_t)hull[k] != g && k + 1 < hull.size()) {      // not a vertex: the two vertices around it as witnesses
            const size_t h1 = (size_t)hull[k], h2 = (size_t)hull[k + 1];
            const double s2 = (gy[h2] - yj) / (gd[h2] - dj);
            if (s2 <= 0.) continue;
            const double s1 = (yj - gy[h1]) / (dj - gd[h1]);
            if (s1 > DBL_MIN && s2 < s1) continue;
        }
        // (the slopes two at a time: SSE2 is the x86-64 baseline, a division per class and candidate was 5 us of every iteration; the
        // quotients are the scalar ones and a maximum / minimum of finite values does not depend on the order it is taken in)
        double maxI1 = DBL_MIN, minI2 = DBL_MAX;
        {
            const __m128d vy = _mm_set1_pd(yj), vd = _mm_set1_pd(dj);
            __m128d mn = _mm_set1_pd(DBL_MAX);
            size_t h = g + 1;
            for (; h + 2 <= G; h += 2)
                mn = _mm_min_pd(_mm_div_pd(_mm_sub_pd(_mm_loadu_pd(&gy[h]), vy), _mm_sub_pd(_mm_loadu_pd(&gd[h]), vd)), mn);
            for (; h < G; h++) { const double v = (gy[h] - yj) / (gd[h] - dj); if (v < minI2) minI2 = v; }
            double t[2];
            _mm_storeu_pd(t, mn); if (t[0] < minI2) minI2 = t[0]; if (t[1] < minI2) minI2 = t[1];
            // (the slopes from the smaller classes only matter through "is one of them above minI2": not formed for a rectangle the test
            // below rejects anyway, and the scan stops at the first that is)
            if (minI2 > 0. && minI2 != DBL_MAX) {
                bool above = false;                     // the rejection below already holds (the maximum only grows: it keeps holding)
                for (h = 0; h + 2 <= g && !above; h += 2) {
                    _mm_storeu_pd(t, _mm_div_pd(_mm_sub_pd(vy, _mm_loadu_pd(&gy[h])), _mm_sub_pd(vd, _mm_loadu_pd(&gd[h]))));
                    if (t[0] > maxI1) maxI1 = t[0];
                    if (t[1] > maxI1) maxI1 = t[1];
                    above = maxI1 != DBL_MIN && minI2 < maxI1;
                }
                for (; h < g && !above; h++) { const double v = (yj - gy[h]) / (dj - gd[h]); if (v > maxI1) maxI1 = v; above = maxI1 != DBL_MIN && minI2 < maxI1; }
            }
        }
        if (minI2 <= 0.) continue;
        if (maxI1 != DBL_MIN && minI2 != DBL_MAX && minI2 < maxI1) continue;
        bool take;
        if (minI2 == DBL_MAX) take = true;
        else if (fmin == 0.0) take = (yj <= dj * minI2);
        else take = (eps <= (fmin - yj) / std::fabs(fmin) + (dj / std::fabs(fmin)) * minI2);
        if (take) for (int m : am) out.push_back((size_t)m);
    }
    std::sort(out.begin(), out.end());             // pool order: the order the reference collects them in
}

#ifdef IBO_DIRECT_SELFCHECK
// the full test on every candidate rectangle, as it stood before the per-class decision and the hull witnesses: the comparator
void potentially_optimal_ref(const Pool &pool, double fmin, std::vector<size_t> &out)
{
    const double eps = 10e-10;
    const size_t n = pool.size();
    const double *Y = pool.y.data(), *Dd = pool.d.data();
    const int *cls = pool.cls.data();
    const char *alive = pool.alive.data();
    out.clear();
    const size_t C = pool.cls_d.size();
    // (scratch kept between calls: this runs once per iteration of every search)
    static thread_local std::vector<double> gd, gy, sufmin;    // the classes that have members, ascending d; their smallest y; min of gy beyond
    static thread_local std::vector<int> rank;
    static thread_local std::vector<size_t> cand;
    gd.clear(); gy.clear(); cand.clear();
    rank.assign(C, -1);
    for (int c : pool.by_size)
        if (!pool.members[c].empty()) { rank[c] = (int)gd.size(); gd.push_back(pool.cls_d[c]); gy.push_back(pool.cls_best[c]); }
    const size_t G = gd.size();
    // A rectangle is rejected as soon as some LARGER class holds a value <= its own (the slope to it is <= 0: minI2 <= 0).
    // With the minimum of gy over the larger classes at hand that is one comparison, and it removes all but the few classes on
    // the descending staircase before any slope is formed -- the same decisions as forming them all (a quotient with a positive
    // denominator is <= 0 exactly when its numerator is).  NaN values fall through to the full test.
    sufmin.assign(G + 1, DBL_MAX);
    for (size_t h = G; h-- > 0;) sufmin[h] = (gy[h] < sufmin[h + 1] || gy[h] != gy[h]) ? gy[h] : sufmin[h + 1];
    for (int c : pool.by_size) {
        if (pool.members[c].empty()) continue;
        const size_t g = (size_t)rank[c];
        if (g + 1 < G && sufmin[g + 1] <= gy[g]) continue;      // every member of the class fails the minI2 test
        for (int j : pool.argmin[c]) cand.push_back((size_t)j);
    }
    std::sort(cand.begin(), cand.end());           // pool order: the order the reference collects them in
    (void)n; (void)alive;
    for (size_t j : cand) {
        const size_t g = (size_t)rank[cls[j]];
        const double yj = Y[j], dj = Dd[j];
        // (the slopes two at a time: SSE2 is the x86-64 baseline, a division per class and candidate was 5 us of every iteration; the
        // quotients are the scalar ones and a maximum / minimum of finite values does not depend on the order it is taken in)
        double maxI1 = DBL_MIN, minI2 = DBL_MAX;
        {
            const __m128d vy = _mm_set1_pd(yj), vd = _mm_set1_pd(dj);
            __m128d mn = _mm_set1_pd(DBL_MAX);
            size_t h = g + 1;
            for (; h + 2 <= G; h += 2)
                mn = _mm_min_pd(_mm_div_pd(_mm_sub_pd(_mm_loadu_pd(&gy[h]), vy), _mm_sub_pd(_mm_loadu_pd(&gd[h]), vd)), mn);
            for (; h < G; h++) { const double v = (gy[h] - yj) / (gd[h] - dj); if (v < minI2) minI2 = v; }
            double t[2];
            _mm_storeu_pd(t, mn); if (t[0] < minI2) minI2 = t[0]; if (t[1] < minI2) minI2 = t[1];
            // (the slopes from the smaller classes only matter through "is one of them above minI2": not formed for a rectangle the test
            // below rejects anyway, and the scan stops at the first that is)
            if (minI2 > 0. && minI2 != DBL_MAX) {
                bool above = false;                     // the rejection below already holds (the maximum only grows: it keeps holding)
                for (h = 0; h + 2 <= g && !above; h += 2) {
                    _mm_storeu_pd(t, _mm_div_pd(_mm_sub_pd(vy, _mm_loadu_pd(&gy[h])), _mm_sub_pd(vd, _mm_loadu_pd(&gd[h]))));
                    if (t[0] > maxI1) maxI1 = t[0];
                    if (t[1] > maxI1) maxI1 = t[1];
                    above = maxI1 != DBL_MIN && minI2 < maxI1;
                }
                for (; h < g && !above; h++) { const double v = (yj - gy[h]) / (dj - gd[h]); if (v > maxI1) maxI1 = v; above = maxI1 != DBL_MIN && minI2 < maxI1; }
            }
        }
        if (minI2 <= 0.) continue;
        if (maxI1 != DBL_MIN && minI2 != DBL_MAX && minI2 < maxI1) continue;
        bool take;
        if (minI2 == DBL_MAX) take = true;
        else if (fmin == 0.0) take = (yj <= dj * minI2);
        else take = (eps <= (fmin - yj) / std::fabs(fmin) + (dj / std::fabs(fmin)) * minI2);
        if (take) out.push_back(j);
    }
}

#endif

// What a search allocates -- the pool's arrays (megabytes at the default budget), a vector per size class, the divisions' vectors, the batch --
// is kept per thread from one search to the next: a Bayesian-optimisation loop runs one search per iteration, a gallery seven per call, and
// growing ~110 per-class vectors and a dozen arrays from nothing was a tenth of a search's tree logic.  A search that starts while this
// thread's workspace is in use (an objective that itself calls direct) works on one of its own; a workspace beyond 64 MB is given back.
struct Workspace {
    Pool pool;
    std::vector<size_t> pot, dead, redo, fix_div, fix_kid;
    std::vector<Division> divs;
    std::vector<double> pts, vals;
    bool busy = false;
};
struct WorkspaceLease {
    Workspace *w;
    bool own;
    WorkspaceLease()
    {
        static thread_local Workspace kept;
        own = kept.busy;
        w = own ? new Workspace : &kept;
        w->busy = true;
    }
    ~WorkspaceLease()
    {
        if (own) { delete w; return; }
        w->busy = false;
        if (w->pool.lb.capacity() * sizeof(double) * 3 > ((size_t)64 << 20)) { Workspace fresh; std::swap(*w, fresh); }
    }
};

}  // namespace

DirectResult direct_minimize(const batch_eval_t &eval, int D, const double *lb, const double *ub,
                             const DirectOptions &opt)
{
    DirectResult res;
    const time_t start = time(nullptr);
    WorkspaceLease lease;
    Workspace &ws = *lease.w;
    Search S;
    S.D = D; S.compat = opt.compat; S.eval = &eval;
    S.lo.assign(lb, lb + D); S.hi.assign(ub, ub + D);
    S.fixed.resize(D); S.xmin.assign(D, 0.0);
    for (int i = 0; i < D; i++) S.fixed[i] = (lb[i] == ub[i]);

    Pool &pool = ws.pool;
    pool.reset(D);
    {   // room for every rectangle the sample budget allows (a division appends two children per cut and the shrunken rectangle): no
        // reallocation while the tree grows (the vectors' growth copies were a tenth of a run's tree logic)
        const size_t cap = (size_t)(opt.maxsample > 0 && opt.maxsample < 200000 ? opt.maxsample : 200000) * 3 / 2 + 4096;
        pool.lb.reserve(cap * D); pool.ub.reserve(cap * D); pool.ctr.reserve(cap * D);
        pool.y.reserve(cap); pool.d.reserve(cap); pool.cls.reserve(cap); pool.alive.reserve(cap); pool.pos.reserve(cap);
    }
    {   // the unit cube, its centre, and its first division (cpp/direct.cpp:352-357)
        Division dv;
        dv.src = 0;
        dv.lb.assign(D, 0.0); dv.ub.assign(D, 1.0); dv.ctr.resize(D);
        double dd = 0.0;
        for (int i = 0; i < D; i++) { dv.ctr[i] = 0.0 + (1.0 - 0.0) / 2.; dd += (0.0 - dv.ctr[i]) * (0.0 - dv.ctr[i]); }
        double y0;
        if (opt.per_rectangle) {
            if ((res.status = S.evaluate(dv.ctr.data(), 1, &y0))) return res;
            S.account(dv.ctr.data(), y0);
            dv.y = y0; dv.mid_d = std::sqrt(dd);
            S.plan_probes(dv);
            if (!dv.dims.empty() && (res.status = S.evaluate(dv.probes.data(), (int)dv.dims.size() * 2, dv.probe_vals.data()))) return res;
            S.plan_children(dv);
            if (!dv.dims.empty() && (res.status = S.evaluate(dv.kid_ctr.data(), (int)dv.dims.size() * 2, dv.kid_vals.data()))) return res;
        } else {
            // centre, probes and guessed child centres in one batch (none of the points depends on a value)
            S.plan_probes(dv);
            S.plan_guess(dv);
            const size_t c = dv.dims.size() * 2;
            std::vector<double> p0(dv.ctr), v0(1 + 2 * c, 0.0);
            p0.insert(p0.end(), dv.probes.begin(), dv.probes.end());
            p0.insert(p0.end(), dv.guess.begin(), dv.guess.end());
            if ((res.status = S.evaluate(p0.data(), (int)(1 + 2 * c), v0.data()))) return res;
            y0 = v0[0];
            S.account(dv.ctr.data(), y0);
            dv.y = y0; dv.mid_d = std::sqrt(dd);
            for (size_t q = 0; q < c; q++) { dv.probe_vals[q] = v0[1 + q]; dv.guess_vals[q] = v0[1 + c + q]; }
            S.plan_children(dv);
            std::vector<size_t> redo0;
            S.take_guess(dv, redo0);
            for (size_t k : redo0)
                if ((res.status = S.evaluate(&dv.kid_ctr[k * D], 1, &dv.kid_vals[k]))) return res;
        }
        S.apply(dv, pool);
    }

    std::vector<size_t> &pot = ws.pot, &dead = ws.dead, &redo = ws.redo, &fix_div = ws.fix_div, &fix_kid = ws.fix_kid;
    std::vector<Division> &divs = ws.divs;
    std::vector<double> &pts = ws.pts, &vals = ws.vals;
    bool done = false;
    int iteration = 0;
    while (iteration < opt.maxiter && !done) {
        iteration++;
        potentially_optimal(pool, S.fmin, pot);
#ifdef IBO_DIRECT_SELFCHECK
        {
            std::vector<size_t> chk;
            potentially_optimal_ref(pool, S.fmin, chk);
            if (chk != pot) { fprintf(stderr, "direct_host.cpp: potentially_optimal disagrees with its comparator (%zu against %zu rectangles)\n", pot.size(), chk.size()); abort(); }
        }
#endif
        if (pot.empty()) {
            printf("[cdirect] could not divide any more\n");
            break;
        }
        dead.clear();
        if (divs.size() < pot.size()) divs.resize(pot.size());
        const size_t ndiv = pot.size();
        for (size_t q = 0; q < ndiv; q++) make_division(pool, pot[ndiv - 1 - q], divs[q]);     // reverse order
        if (opt.per_rectangle) {
            for (size_t qd = 0; qd < ndiv; qd++) {
                Division &dv = divs[qd];
                S.plan_probes(dv);
                int np = (int)dv.dims.size() * 2;
                if (np && (res.status = S.evaluate(dv.probes.data(), np, dv.probe_vals.data()))) return res;
                S.plan_children(dv);
                if (np && (res.status = S.evaluate(dv.kid_ctr.data(), np, dv.kid_vals.data()))) return res;
                S.apply(dv, pool);
                dead.push_back(dv.src);
                if (S.nsamples > (int64_t)(unsigned)opt.maxsample) { done = true; break; }
                if (time(nullptr) - start > opt.maxtime) { done = true; break; }
            }
        } else {
            // ONE batch per iteration: every rectangle's probes and, behind them, its guessed child centres (plan_guess)
            pts.clear();
            for (size_t qd = 0; qd < ndiv; qd++) {
                Division &dv = divs[qd];
                S.plan_probes(dv);
                S.plan_guess(dv);
                pts.insert(pts.end(), dv.probes.begin(), dv.probes.end());
                pts.insert(pts.end(), dv.guess.begin(), dv.guess.end());
            }
            int np = (int)(pts.size() / D);
            vals.assign(np, 0.0);
            if (np && (res.status = S.evaluate(pts.data(), np, vals.data()))) return res;
            size_t o = 0;
            pts.clear(); fix_div.clear(); fix_kid.clear();
            for (size_t qd = 0; qd < ndiv; qd++) {
                Division &dv = divs[qd];
                const size_t c = dv.dims.size() * 2;
                for (size_t q = 0; q < c; q++) { dv.probe_vals[q] = vals[o + q]; dv.guess_vals[q] = vals[o + c + q]; }
                o += 2 * c;
                S.plan_children(dv);
                S.take_guess(dv, redo);
                for (size_t k : redo) {                  // a centre that differs from its guess in some bit: sampled again
                    fix_div.push_back(qd); fix_kid.push_back(k);
                    pts.insert(pts.end(), dv.kid_ctr.begin() + k * D, dv.kid_ctr.begin() + (k + 1) * D);
                }
            }
            if (!fix_div.empty()) {
                vals.assign(fix_div.size(), 0.0);
                if ((res.status = S.evaluate(pts.data(), (int)fix_div.size(), vals.data()))) return res;
                for (size_t f = 0; f < fix_div.size(); f++) divs[fix_div[f]].kid_vals[fix_kid[f]] = vals[f];
            }
            for (size_t qd = 0; qd < ndiv; qd++) {
                Division &dv = divs[qd];
                S.apply(dv, pool);
                dead.push_back(dv.src);
                if (S.nsamples > (int64_t)(unsigned)opt.maxsample) { done = true; break; }
                if (time(nullptr) - start > opt.maxtime) { done = true; break; }
            }
        }
        // children were appended behind the old rectangles, so removing the divided ones afterwards, in one
        // pass, leaves the pool in the order one-by-one removal would
        pool.erase_many(dead);
        if (time(nullptr) - start > opt.maxtime) break;
        if (S.nsamples > (int64_t)(unsigned)opt.maxsample) break;
    }
    res.fmin = S.fmin; res.xmin = S.xmin; res.nsamples = S.nsamples; res.iterations = iteration;
    return res;
}

}  // namespace ibo
