// direct_host_check.cpp -- the host side of DIRECT (ibo_amd/csrc/direct_host.cpp) on its own, for two uses:
//   bash tools/sanitize_host.sh          builds it with -fsanitize=address,undefined and runs every case below (CPU only; the GPU build
//                                        cannot run under a sanitizer on this pool)
//   ./direct_host_check time             the tree logic's milliseconds per run beside a cheap objective (what tools/time_direct.py cannot
//                                        separate from the GPU's batches)
// Built with -DIBO_DIRECT_SELFCHECK (the script does) every call of the potentially-optimal test is also compared with its comparator -- the full
// slope test on every candidate rectangle -- and a difference aborts.
// The cases: 1 .. 12 dimensions, both evaluation schedules (per rectangle / one batch per iteration) which must agree in (fmin, xmin, samples),
// the dimension-0 quirk on and off, degenerate boxes (lb == ub in a dimension), tiny and exhausted sample budgets, an objective with ties
// everywhere (a constant), one with non-finite values, an evaluator that aborts, and a search nested inside an objective.
#include "direct_host.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>

static int g_fail = 0;
#define CHECK(c, ...) do { if (!(c)) { g_fail++; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

struct Obj {
    int D, kind;
    long calls = 0;
    double operator()(const double *x) {
        calls++;
        double s = 0;
        switch (kind) {
        case 0: for (int d = 0; d < D; d++) s += (x[d] - 0.3 - 0.05 * d) * (x[d] - 0.3 - 0.05 * d); return s;                 // bowl
        case 1: for (int d = 0; d < D; d++) s += std::sin(5 * x[d] + d) * std::exp(-x[d]); return -s * s;                       // wavy
        case 2: return 1.0;                                                                                                    // ties everywhere
        case 3: for (int d = 0; d < D; d++) s += x[d]; return s > 0.9 * D ? std::numeric_limits<double>::infinity() : -s;       // +inf in a corner
        case 4: for (int d = 0; d < D; d++) s += std::fabs(x[d] - 0.5); return s;                                              // kink at the first centre
        case 5: for (int d = 0; d < D; d++) s += (x[d] - 0.7) * (x[d] - 0.7); return -std::fmax(0.0, 0.05 * D - s);              // an EI-like surface: exact zeros (ties) over most of the box
        case 6: return 0.0;                                                                                                    // fmin = 0: the test's other branch
        default: for (int d = 0; d < D; d++) s += x[d]; return std::floor(4.0 * s) / 4.0;                                       // terraces: ties inside every size class
        }
    }
};

static ibo::DirectResult run(Obj &o, const std::vector<double> &lb, const std::vector<double> &ub, int maxiter, int maxsample, bool compat, bool per_rect)
{
    ibo::batch_eval_t ev = [&](const double *p, int n, double *v) -> int {
        for (int i = 0; i < n; i++) v[i] = o(p + (size_t)i * o.D);
        return 0;
    };
    ibo::DirectOptions opt; opt.maxiter = maxiter; opt.maxtime = 30; opt.maxsample = maxsample; opt.compat = compat; opt.per_rectangle = per_rect;
    return ibo::direct_minimize(ev, o.D, lb.data(), ub.data(), opt);
}

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "time")) {
        for (int D : {2, 4, 8}) {
            std::vector<double> lb(D, 0.0), ub(D, 1.0);
            double tobj = 0;
            Obj o{D, 1};
            ibo::batch_eval_t ev = [&](const double *p, int n, double *v) -> int {
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < n; i++) v[i] = o(p + (size_t)i * D);
                tobj += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                return 0;
            };
            ibo::DirectOptions opt; opt.maxiter = 50; opt.maxsample = 10000; opt.per_rectangle = false;
            double best = 1e9;
            long ns = 0;
            for (int rep = 0; rep < 200; rep++) {
                tobj = 0;
                auto t0 = std::chrono::steady_clock::now();
                ibo::DirectResult r = ibo::direct_minimize(ev, D, lb.data(), ub.data(), opt);
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() - tobj;
                if (ms < best) best = ms;
                ns = (long)r.nsamples;
            }
            printf("D=%d: tree logic %.3f ms per run (%ld samples, 50 iterations)\n", D, best, ns);
        }
        return 0;
    }
    int ncase = 0;
    for (int D = 1; D <= 12; D++)
        for (int kind = 0; kind < 8; kind++)
            for (int compat = 0; compat < 2; compat++)
                for (int budget : {1, 7, 200, 10000}) {
                    std::vector<double> lb(D), ub(D);
                    for (int d = 0; d < D; d++) { lb[d] = -0.25 * d; ub[d] = 1.0 + 0.5 * d; }
                    if (D >= 3 && kind == 0) ub[1] = lb[1];                       // a fixed dimension
                    if (D >= 2 && kind == 4 && compat) ub[0] = lb[0];             // dimension 0 fixed: the quirk's own case
                    Obj a{D, kind}, b{D, kind};
                    const int iters = D <= 6 ? 30 : 12;
                    ibo::DirectResult ra = run(a, lb, ub, iters, budget, compat != 0, true);
                    ibo::DirectResult rb = run(b, lb, ub, iters, budget, compat != 0, false);
                    ncase++;
                    CHECK(ra.status == 0 && rb.status == 0, "status %d %d (D=%d kind=%d)", ra.status, rb.status, D, kind);
                    CHECK(ra.nsamples == rb.nsamples, "samples %ld vs %ld (D=%d kind=%d compat=%d budget=%d)", (long)ra.nsamples, (long)rb.nsamples, D, kind, compat, budget);
                    CHECK(ra.fmin == rb.fmin || (std::isnan(ra.fmin) && std::isnan(rb.fmin)), "fmin %.17g vs %.17g (D=%d kind=%d compat=%d budget=%d)", ra.fmin, rb.fmin, D, kind, compat, budget);
                    CHECK((int)ra.xmin.size() == D && (int)rb.xmin.size() == D, "xmin sizes");
                    for (int d = 0; d < D && d < (int)ra.xmin.size() && d < (int)rb.xmin.size(); d++) {
                        CHECK(ra.xmin[d] == rb.xmin[d], "xmin[%d] %.17g vs %.17g (D=%d kind=%d)", d, ra.xmin[d], rb.xmin[d], D, kind);
                        CHECK(ra.xmin[d] >= lb[d] && ra.xmin[d] <= ub[d], "xmin[%d] outside the box (D=%d kind=%d)", d, D, kind);
                    }
                    CHECK(ra.nsamples >= 1, "no sample taken");
                }
    // an evaluator that gives up: its code comes back, nothing else is touched afterwards
    {
        int calls = 0;
        ibo::batch_eval_t ev = [&](const double *, int n, double *v) -> int { for (int i = 0; i < n; i++) v[i] = 0.5; return ++calls >= 3 ? 42 : 0; };
        std::vector<double> lb(4, 0.0), ub(4, 1.0);
        ibo::DirectOptions opt; opt.per_rectangle = false;
        ibo::DirectResult r = ibo::direct_minimize(ev, 4, lb.data(), ub.data(), opt);
        ncase++;
        CHECK(r.status == 42, "abort code %d", r.status);
    }
    // a search inside an objective (the per-thread workspace is in use: the inner search takes one of its own), against the same inner search run
    // on its own and added by hand
    {
        std::vector<double> lb2(2, 0.0), ub2(2, 1.0), lb3(3, -1.0), ub3(3, 2.0);
        auto inner_at = [&](double shift, int64_t *ns) {
            ibo::batch_eval_t in = [&](const double *p, int n, double *v) -> int {
                for (int i = 0; i < n; i++) v[i] = (p[2 * i] - shift) * (p[2 * i] - shift) + (p[2 * i + 1] - 0.25) * (p[2 * i + 1] - 0.25);
                return 0; };
            ibo::DirectOptions io; io.maxiter = 8; io.per_rectangle = false;
            ibo::DirectResult r = ibo::direct_minimize(in, 2, lb2.data(), ub2.data(), io);
            if (ns) *ns = r.nsamples;
            return r.fmin;
        };
        std::vector<double> seen_x, seen_v;
        ibo::batch_eval_t outer = [&](const double *p, int n, double *v) -> int {
            for (int i = 0; i < n; i++) {
                const double sh = 0.5 + 0.1 * p[3 * i];
                v[i] = inner_at(sh, nullptr) + std::fabs(p[3 * i + 1] - 0.4) + p[3 * i + 2] * p[3 * i + 2];
                seen_x.push_back(sh); seen_v.push_back(v[i] - std::fabs(p[3 * i + 1] - 0.4) - p[3 * i + 2] * p[3 * i + 2]);
            }
            return 0; };
        ibo::DirectOptions oo; oo.maxiter = 6; oo.per_rectangle = false;
        ibo::DirectResult r = ibo::direct_minimize(outer, 3, lb3.data(), ub3.data(), oo);
        ncase++;
        CHECK(r.status == 0 && r.nsamples > 10, "nested search: status %d, %ld samples", r.status, (long)r.nsamples);
        for (size_t i = 0; i < seen_x.size(); i += 7) {
            const double alone = inner_at(seen_x[i], nullptr);
            CHECK(alone == seen_v[i] || std::fabs(alone - seen_v[i]) <= 1e-15, "nested search: inner result %.17g inside against %.17g alone", seen_v[i], alone);
        }
    }
    printf("%d cases, %d failure(s)\n", ncase, g_fail);
    return g_fail ? 1 : 0;
}
