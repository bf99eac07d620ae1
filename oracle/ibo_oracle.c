/*
 * ibo_oracle.c -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain-C CPU restatement of the native half of the reference's GP-posterior +
 * acquisition hot path (the part the reference ships as cpp/libego), written
 * from the algorithm, not from the source text.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load this; the product
 * (ibo_amd/) never does.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here against (a) golden vectors produced by the real reference in the build
 * container (tests/golden/make_golden.py imports the lib2to3-converted Python
 * half and the g++-built cpp/ half) and (b) oracle/_ref/libego.so, the
 * reference's own C++ compiled from /root/reference/cpp (oracle/Makefile),
 * when that file is present.
 *
 * Every function cites the reference lines it restates
 * (paths relative to /root/reference).
 */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* kernel type codes: ego/acquisition/__init__.py:323-333 */
enum { ORC_K_SE_ARD = 0, ORC_K_SE_ISO = 1, ORC_K_MATERN3 = 2, ORC_K_MATERN5 = 3 };
/* acquisition codes: ego/acquisition/__init__.py:309-321 */
enum { ORC_ACQ_EI = 0, ORC_ACQ_PI = 1, ORC_ACQ_UCB = 2 };
/* erf flavour: libm (cpp/optimizeGP.cpp:200-204) or the Numerical-Recipes
 * Chebyshev fit with truncated constants (ego/gaussianprocess/__init__.py:55-77) */
enum { ORC_ERF_LIBM = 0, ORC_ERF_NR = 1 };

/* ------------------------------------------------------------------------ */
/* covariance k(a,b)                                                         */
/*   cpp/optimizeGP.cpp:67-113 (k* loop, 4 kernel types)                     */
/*   ego/gaussianprocess/kernel.py:87-89,147-149,207-210,246-249            */
/* hyper: ARD -> D length scales, ISO -> [theta], Matern -> [theta, mag].    */
/* sf2 is passed explicitly because the two halves of the reference disagree */
/* (C++ forces 1 for types 0-2, cpp/optimizeGP.cpp:303-310; Python applies   */
/* magnitude^2 for Matern and SV kernels, kernel.py:66-68,203,243).          */
/* ------------------------------------------------------------------------ */
double orc_cov(int ktype, int D, const double *a, const double *b,
               const double *hyper, double sf2)
{
    double z = 0.0;
    int j;
    switch (ktype) {
    case ORC_K_SE_ARD:
        for (j = 0; j < D; j++) {
            double t = a[j] - b[j];
            z += (1.0 / (hyper[j] * hyper[j])) * (t * t);
        }
        return sf2 * exp(-0.5 * z);
    case ORC_K_SE_ISO:
        for (j = 0; j < D; j++) {
            double t = (a[j] - b[j]) / hyper[0];
            z += t * t;
        }
        return sf2 * exp(-0.5 * z);
    case ORC_K_MATERN3:
        for (j = 0; j < D; j++) {
            double t = (a[j] - b[j]) / hyper[0];
            z += t * t;
        }
        z = sqrt(3.0) * sqrt(z);
        return sf2 * (1.0 + z) * exp(-z);
    case ORC_K_MATERN5:
        /* intended formula, SURVEY 7.3-5: sf2 (1 + s5 r/t + 5 r^2 / 3 t^2) exp(-s5 r/t) */
        for (j = 0; j < D; j++) {
            double t = a[j] - b[j];
            z += t * t;
        }
        z = sqrt(z);
        return sf2 * (1.0 + sqrt(5.0) * z / hyper[0] + 5.0 * z * z / (3.0 * hyper[0] * hyper[0]))
               * exp(-(sqrt(5.0) * z / hyper[0]));
    }
    return NAN;
}

/* ------------------------------------------------------------------------ */
/* R = K(X,X) with the diagonal forced to 1+noise                            */
/*   ego/gaussianprocess/__init__.py:134-149 (_computeCorrelations)          */
/* ------------------------------------------------------------------------ */
void orc_build_R(int ktype, int N, int D, const double *X, const double *hyper,
                 double sf2, double noise, double *R)
{
    int i, j;
    for (i = 0; i < N; i++) {
        R[(size_t)i * N + i] = 1.0 + noise;
        for (j = 0; j < i; j++) {
            double v = orc_cov(ktype, D, X + (size_t)i * D, X + (size_t)j * D, hyper, sf2);
            R[(size_t)i * N + j] = v;
            R[(size_t)j * N + i] = v;
        }
    }
}

/* K = covMatrix(X) (diagonal from cov itself): ego/gaussianprocess/kernel.py:46-53 */
void orc_cov_matrix(int ktype, int N, int D, const double *X, const double *hyper,
                    double sf2, double *K)
{
    int i, j;
    for (i = 0; i < N; i++)
        for (j = 0; j <= i; j++) {
            double v = orc_cov(ktype, D, X + (size_t)i * D, X + (size_t)j * D, hyper, sf2);
            K[(size_t)i * N + j] = v;
            K[(size_t)j * N + i] = v;
        }
}

/* ------------------------------------------------------------------------ */
/* lower Cholesky, A = L L^T (what numpy.linalg.cholesky computes;           */
/* ego/gaussianprocess/__init__.py:299).  Returns 0, or k+1 if pivot k <= 0. */
/* ------------------------------------------------------------------------ */
int orc_cholesky(int N, const double *A, double *L)
{
    int i, j, k;
    memset(L, 0, sizeof(double) * (size_t)N * N);
    for (j = 0; j < N; j++) {
        double s = A[(size_t)j * N + j];
        for (k = 0; k < j; k++) s -= L[(size_t)j * N + k] * L[(size_t)j * N + k];
        if (!(s > 0.0)) return j + 1;
        L[(size_t)j * N + j] = sqrt(s);
        for (i = j + 1; i < N; i++) {
            double t = A[(size_t)i * N + j];
            for (k = 0; k < j; k++) t -= L[(size_t)i * N + k] * L[(size_t)j * N + k];
            L[(size_t)i * N + j] = t / L[(size_t)j * N + j];
        }
    }
    return 0;
}

/* forward substitution L x = b */
static void fwd_solve(int N, const double *L, const double *b, double *x)
{
    int i, k;
    for (i = 0; i < N; i++) {
        double s = b[i];
        for (k = 0; k < i; k++) s -= L[(size_t)i * N + k] * x[k];
        x[i] = s / L[(size_t)i * N + i];
    }
}

/* ------------------------------------------------------------------------ */
/* RBF-network prior mean                                                    */
/*   cpp/optimizeGP.cpp:116-133 ; ego/gaussianprocess/prior.py:60-66         */
/* ------------------------------------------------------------------------ */
double orc_prior_mu(int D, const double *x, int nb, const double *means,
                    const double *beta, double theta, const double *lowerb,
                    const double *width)
{
    double mu = 0.0;
    int i, j;
    for (i = 0; i < nb; i++) {
        double d = 0.0;
        for (j = 0; j < D; j++) {
            double t = (x[j] - lowerb[j]) / width[j] - means[(size_t)i * D + j];
            d += t * t;
        }
        mu += beta[i] * exp(-theta * d);
    }
    return mu;
}

/* a^T M b as matvec then dot: cpp/optimizeGP.cpp:174-191 */
static double aMb(int N, const double *a, const double *M, const double *b, double *tmp)
{
    int i, j;
    double x = 0.0;
    for (i = 0; i < N; i++) {
        double s = 0.0;
        const double *row = M + (size_t)i * N;
        for (j = 0; j < N; j++) s += row[j] * b[j];
        tmp[i] = s;
    }
    for (i = 0; i < N; i++) x += tmp[i] * a[i];
    return x;
}

/* ------------------------------------------------------------------------ */
/* native-path posterior at one point: cpp/optimizeGP.cpp:57-170             */
/*   mu = m + k*^T invR (Y-m);  sigma = sqrt(clamp(1+noise - k*^T invR k*,   */
/*   1e-8, 10)).  Two full N^2 matvecs per point, invR (Y-m) recomputed      */
/*   every time -- this is the cost shape of the reference.                  */
/* ------------------------------------------------------------------------ */
typedef struct {
    int D, N, ktype, acq, erf_mode, nb;
    const double *invR, *X, *Y, *hyper;
    const double *pmeans, *pbeta, *plowerb, *pwidth;
    double ptheta, sf2, noise, maxY, parm, clamp_lo;
    double *r, *tmp, *ymu;
    long nevals;
} orc_gp_t;

static void posterior_native(orc_gp_t *g, const double *x, double *mu, double *sigma)
{
    int i;
    double ypred, sig2;
    for (i = 0; i < g->N; i++)
        g->r[i] = orc_cov(g->ktype, g->D, g->X + (size_t)i * g->D, x, g->hyper, g->sf2);
    if (g->nb > 0) {
        double m = orc_prior_mu(g->D, x, g->nb, g->pmeans, g->pbeta, g->ptheta, g->plowerb, g->pwidth);
        for (i = 0; i < g->N; i++) g->ymu[i] = g->Y[i] - m;
        ypred = m + aMb(g->N, g->r, g->invR, g->ymu, g->tmp);
    } else {
        ypred = aMb(g->N, g->r, g->invR, g->Y, g->tmp);
    }
    sig2 = 1.0 + g->noise - aMb(g->N, g->r, g->invR, g->r, g->tmp);
    if (sig2 < g->clamp_lo) sig2 = g->clamp_lo;
    else if (sig2 > 10.0) sig2 = 10.0;
    *sigma = sqrt(sig2);
    *mu = ypred;
    g->nevals++;
}

/* NR erf: ego/gaussianprocess/__init__.py:55-70 */
double orc_erf_nr(double z)
{
    double t = 1.0 / (1.0 + 0.5 * fabs(z));
    double p = 0.17087277;
    double ans;
    p = -0.82215223 + t * p;
    p = 1.48851587 + t * p;
    p = -1.13520398 + t * p;
    p = 0.27886807 + t * p;
    p = -0.18628806 + t * p;
    p = 0.09678418 + t * p;
    p = 0.37409196 + t * p;
    p = 1.00002368 + t * p;
    ans = 1.0 - t * exp(-z * z - 1.26551223 + t * p);
    return z >= 0.0 ? ans : -ans;
}

/* ------------------------------------------------------------------------ */
/* acquisition value (POSITIVE: the thing being maximised) from (mu, sigma)  */
/*   libm flavour: cpp/optimizeGP.cpp:194-236 (negei/negpi/negucb)           */
/*   NR flavour:   ego/acquisition/__init__.py:150-164,107-110,68-71 with    */
/*                 CDF/PDF of ego/gaussianprocess/__init__.py:72-76          */
/* For UCB 'parm' is the multiplier of sigma in both flavours.               */
/* ------------------------------------------------------------------------ */
double orc_acq_value(int acq, int erf_mode, double mu, double sigma, double maxY, double parm)
{
    double ydiff, Z, cdf, pdf;
    if (acq == ORC_ACQ_UCB) return mu + parm * sigma;
    ydiff = mu - maxY - parm;
    Z = ydiff / sigma;
    if (erf_mode == ORC_ERF_LIBM) {
        cdf = 0.5 * (1.0 + erf(Z / sqrt(2.0)));
        pdf = exp(-(Z * Z / 2.0)) / sqrt(2.0 * M_PI);
    } else {
        cdf = 0.5 * (1.0 + orc_erf_nr(Z * 0.707106));
        pdf = exp(-(Z * Z / 2.0)) * 0.398942;
    }
    if (acq == ORC_ACQ_PI) return cdf;
    return ydiff * cdf + sigma * pdf;
}

static double neg_acq(orc_gp_t *g, const double *x)
{
    double mu, sigma;
    posterior_native(g, x, &mu, &sigma);
    return -orc_acq_value(g->acq, g->erf_mode, mu, sigma, g->maxY, g->parm);
}

static int gp_init(orc_gp_t *g, int D, const double *invR, const double *X, const double *Y,
                   int N, int acq, int ktype, const double *hyper, double sf2, int nb,
                   const double *pmeans, const double *pbeta, double ptheta,
                   const double *plowerb, const double *pwidth, double parm, double noise,
                   int erf_mode, double clamp_lo)
{
    int i;
    memset(g, 0, sizeof(*g));
    g->D = D; g->N = N; g->ktype = ktype; g->acq = acq; g->erf_mode = erf_mode; g->nb = nb;
    g->invR = invR; g->X = X; g->Y = Y; g->hyper = hyper;
    g->pmeans = pmeans; g->pbeta = pbeta; g->plowerb = plowerb; g->pwidth = pwidth;
    g->ptheta = ptheta; g->sf2 = sf2; g->noise = noise; g->parm = parm; g->clamp_lo = clamp_lo;
    /* maxY: cpp/optimizeGP.cpp:316-321 */
    g->maxY = Y[0];
    for (i = 0; i < N; i++) if (Y[i] > g->maxY) g->maxY = Y[i];
    g->r = (double *)malloc(sizeof(double) * N);
    g->tmp = (double *)malloc(sizeof(double) * N);
    g->ymu = (double *)malloc(sizeof(double) * N);
    return (g->r && g->tmp && g->ymu) ? 0 : -1;
}
static void gp_free(orc_gp_t *g) { free(g->r); free(g->tmp); free(g->ymu); }

/* ------------------------------------------------------------------------ */
/* candidate sweep in the reference's cost shape: M independent calls of the */
/* native posterior + acquisition.  Outputs are optional (NULL to skip).     */
/* best_idx = first index attaining the maximum (strict >), the convention   */
/* of numpy.argmax and of DIRECT's strict '<' on the negated value           */
/* (cpp/direct.cpp:124).                                                     */
/* ------------------------------------------------------------------------ */
int orc_sweep_native(int D, const double *invR, const double *X, const double *Y, int N,
                     int acq, int ktype, const double *hyper, double sf2, int nb,
                     const double *pmeans, const double *pbeta, double ptheta,
                     const double *plowerb, const double *pwidth, double parm, double noise,
                     int erf_mode, double clamp_lo,
                     long M, const double *cand,
                     double *out_mu, double *out_s2, double *out_acq,
                     double *best_val, long *best_idx)
{
    orc_gp_t g;
    long c, bi = -1;
    double bv = -DBL_MAX;
    if (gp_init(&g, D, invR, X, Y, N, acq, ktype, hyper, sf2, nb, pmeans, pbeta, ptheta,
                plowerb, pwidth, parm, noise, erf_mode, clamp_lo)) return -1;
    for (c = 0; c < M; c++) {
        double mu, sigma, v;
        posterior_native(&g, cand + (size_t)c * D, &mu, &sigma);
        v = orc_acq_value(acq, erf_mode, mu, sigma, g.maxY, parm);
        if (out_mu) out_mu[c] = mu;
        if (out_s2) out_s2[c] = sigma * sigma;
        if (out_acq) out_acq[c] = v;
        if (bi < 0 || v > bv) { bv = v; bi = c; }
    }
    if (best_val) *best_val = bv;
    if (best_idx) *best_idx = bi;
    gp_free(&g);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Python-path posterior: ego/gaussianprocess/__init__.py:169-228            */
/*   Lr = L \ k*;  mu = m + Lr . (L \ (Y-m));  s2 = clip(1+noise-|Lr|^2,     */
/*   1e-7, 10).  m is the scalar prior mean at the query point.             */
/* ------------------------------------------------------------------------ */
int orc_posterior_chol(int ktype, int D, int N, const double *X, const double *Y,
                       const double *L, const double *hyper, double sf2, double noise,
                       int nb, const double *pmeans, const double *pbeta, double ptheta,
                       const double *plowerb, const double *pwidth,
                       long M, const double *q, double *out_mu, double *out_s2)
{
    double *r = (double *)malloc(sizeof(double) * N);
    double *Lr = (double *)malloc(sizeof(double) * N);
    double *d = (double *)malloc(sizeof(double) * N);
    double *Ld = (double *)malloc(sizeof(double) * N);
    long c; int i;
    if (!r || !Lr || !d || !Ld) return -1;
    for (c = 0; c < M; c++) {
        const double *x = q + (size_t)c * D;
        double m = 0.0, mu = 0.0, ss = 0.0, s2;
        if (nb > 0) m = orc_prior_mu(D, x, nb, pmeans, pbeta, ptheta, plowerb, pwidth);
        for (i = 0; i < N; i++) {
            r[i] = orc_cov(ktype, D, X + (size_t)i * D, x, hyper, sf2);
            d[i] = Y[i] - m;
        }
        fwd_solve(N, L, r, Lr);
        fwd_solve(N, L, d, Ld);
        for (i = 0; i < N; i++) { mu += Lr[i] * Ld[i]; ss += Lr[i] * Lr[i]; }
        s2 = (1.0 + noise) - ss;
        if (s2 < 10e-8) s2 = 10e-8;
        if (s2 > 10.0) s2 = 10.0;
        out_mu[c] = m + mu;
        if (out_s2) out_s2[c] = s2;
    }
    free(r); free(Lr); free(d); free(Ld);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* best-effort CPU sweep: the same posterior with alpha =    */
/* invR (Y - m) cached, the variance through the lower-triangular W = L^-1     */
/* (N^2/2 multiply-adds instead of 2 N^2) and OpenMP over the candidates.      */
/* Not the reference's cost shape -- reported next to it as "what all the host */
/* cores can do with the obvious algebra".  No prior mean (bench workload).    */
/* ------------------------------------------------------------------------ */
int orc_sweep_fast(int D, const double *W, const double *alpha, const double *X, int N,
                   int acq, int ktype, const double *hyper, double sf2, double parm, double noise,
                   int erf_mode, double clamp_lo, double maxY, long M, const double *cand,
                   double *out_acq, double *best_val, long *best_idx, int *threads_used)
{
    long c;
    int nthreads = 1;
#ifdef _OPENMP
    /* *threads_used on entry: threads to use (<= 0: the runtime's default).  A container's CPU quota is usually far below
       the CPUs it can see; the caller knows it (oracle.py: effective_cores) and the OpenMP runtime does not. */
    if (threads_used && *threads_used > 0) omp_set_num_threads(*threads_used);
#pragma omp parallel
    {
#pragma omp single
        nthreads = omp_get_num_threads();
    }
#endif
    /* ORC_CB candidates share every pass over W: a row of the 4 MB triangle (N = 1024) is loaded once for eight dot products
       instead of once per candidate -- the unblocked loop re-streamed W from memory for every evaluation and ran 6x slower
       per core than the reference-shaped leg it is meant to beat.  Each candidate's sums keep their order (k ascending):
       same values as the unblocked loop. */
#define ORC_CB 8
#pragma omp parallel
    {
        double *r = (double *)malloc(sizeof(double) * (size_t)N * ORC_CB);      /* r[k * ORC_CB + b] */
        long c0;
#pragma omp for schedule(dynamic, 8)
        for (c0 = 0; c0 < M; c0 += ORC_CB) {
            const int nb = (int)(M - c0 < ORC_CB ? M - c0 : ORC_CB);
            double mu[ORC_CB], q[ORC_CB];
            int i, k, b;
            for (b = 0; b < ORC_CB; b++) { mu[b] = 0.0; q[b] = 0.0; }
            for (i = 0; i < N; i++)
                for (b = 0; b < ORC_CB; b++) {
                    const double v = b < nb ? orc_cov(ktype, D, X + (size_t)i * D, cand + (size_t)(c0 + b) * D, hyper, sf2) : 0.0;
                    r[(size_t)i * ORC_CB + b] = v;
                    mu[b] += alpha[i] * v;
                }
            for (i = 0; i < N; i++) {
                const double *w = W + (size_t)i * N;
                double v[ORC_CB];
                for (b = 0; b < ORC_CB; b++) v[b] = 0.0;
                for (k = 0; k <= i; k++) {
                    const double wk = w[k];
                    const double *rk = r + (size_t)k * ORC_CB;
                    for (b = 0; b < ORC_CB; b++) v[b] += wk * rk[b];
                }
                for (b = 0; b < ORC_CB; b++) q[b] += v[b] * v[b];
            }
            for (b = 0; b < nb; b++) {
                double s2 = 1.0 + noise - q[b];
                if (s2 < clamp_lo) s2 = clamp_lo; else if (s2 > 10.0) s2 = 10.0;
                out_acq[c0 + b] = orc_acq_value(acq, erf_mode, mu[b], sqrt(s2), maxY, parm);
            }
        }
        free(r);
    }
    {
        long bi = -1; double bv = -DBL_MAX;
        for (c = 0; c < M; c++) if (bi < 0 || out_acq[c] > bv) { bv = out_acq[c]; bi = c; }
        if (best_val) *best_val = bv;
        if (best_idx) *best_idx = bi;
    }
    if (threads_used) *threads_used = nthreads;
    return 0;
}

/* ======================================================================== */
/* DIRECT (dividing rectangles) exactly as the reference's native optimiser  */
/* runs it: cpp/direct.cpp:329-581 (driver), :111-141 (samplef),             */
/* :146-235 (divrec), :49-74 (Rectangle), cpp/direct.h:43-74.                */
/* Quirks kept on purpose: split-point samples order the      */
/* dimensions and the child centres are sampled again; FMIN uses strict '<'; */
/* potentially-optimal rectangles are divided in reverse index order;        */
/* maxI1 starts at DBL_MIN (smallest positive normal, not -inf); dimension 0 */
/* seeds maxlength without the fixed[] test; whole-second maxtime.           */
/* ======================================================================== */
typedef double (*orc_objective_t)(int, double *);
typedef double (*orc_objective_ctx_t)(void *, int, const double *);

typedef struct {
    double *lb, *ub, *center;   /* n each, one allocation */
    double y, d;
} rect_t;

typedef struct {
    int n;
    double *lowerb, *upperb;
    int *fixed;
    double FMIN;
    double *XMIN;
    unsigned nsamples;
    orc_objective_t fn;
    orc_objective_ctx_t fnc;
    void *ctx;
    double *scratch;
} dstate_t;

static void rect_alloc(rect_t *r, int n)
{
    r->lb = (double *)malloc(sizeof(double) * 3 * n);
    r->ub = r->lb + n;
    r->center = r->ub + n;
}
static void rect_free(rect_t *r) { free(r->lb); }
static void rect_copy(rect_t *dst, const rect_t *src, int n)
{
    rect_alloc(dst, n);
    memcpy(dst->lb, src->lb, sizeof(double) * 3 * n);
    dst->y = src->y; dst->d = src->d;
}

/* cpp/direct.cpp:111-141 */
static double d_sample(dstate_t *S, const double *x)
{
    int i; double y;
    for (i = 0; i < S->n; i++)
        S->scratch[i] = S->fixed[i] ? S->lowerb[i]
                                    : x[i] * (S->upperb[i] - S->lowerb[i]) + S->lowerb[i];
    y = S->fn ? S->fn(S->n, S->scratch) : S->fnc(S->ctx, S->n, S->scratch);
    S->nsamples += 1;
    if (y < S->FMIN) {
        S->FMIN = y;
        for (i = 0; i < S->n; i++)
            S->XMIN[i] = S->lowerb[i] + (S->upperb[i] - S->lowerb[i]) * x[i];
    }
    return y;
}

/* cpp/direct.cpp:49-66: centre, half-diagonal, sample the centre */
static void rect_make(rect_t *r, dstate_t *S, const double *lb, const double *ub)
{
    int i; double d = 0.0;
    rect_alloc(r, S->n);
    for (i = 0; i < S->n; i++) {
        r->lb[i] = lb[i]; r->ub[i] = ub[i];
        r->center[i] = lb[i] + (ub[i] - lb[i]) / 2.;
        d += (lb[i] - r->center[i]) * (lb[i] - r->center[i]);
    }
    r->d = sqrt(d);
    r->y = d_sample(S, r->center);
}

typedef struct { rect_t *v; size_t len, cap; } rvec_t;
static void rvec_push(rvec_t *a, rect_t r)
{
    if (a->len == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 64;
        a->v = (rect_t *)realloc(a->v, a->cap * sizeof(rect_t));
    }
    a->v[a->len++] = r;
}

/* cpp/direct.cpp:194 sorts its (dimension, value) pairs with std::sort(.., sortByVal), and WHICH of several equal
 * values comes first decides the order the rectangle is cut in, so the sort is restated as the reference's standard
 * library performs it.  Dependency outside /root/reference: libstdc++ (GCC; bits/stl_algo.h, unchanged since 4.x):
 *   std::sort = __introsort_loop(first, last, 2 floor(log2 n)) + __final_insertion_sort, threshold 16:
 *   ranges longer than 16 are cut by __unguarded_partition_pivot (median of first+1 / middle / last-1 moved to the
 *   front, then Hoare's unguarded partition around it), right part first by recursion, left part by iteration;
 *   what is left (pieces of <= 16) is finished by one insertion sort over the first 16 elements and an unguarded
 *   insertion sort over the rest.  Up to 16 elements this is a plain stable insertion sort.
 * The depth-limit fallback (heap sort after 2 log2 n bad partitions) is not restated: it aborts loudly. */
typedef struct { int d; double v; } iv_t;
static void iv_swap(iv_t *a, iv_t *b) { iv_t t = *a; *a = *b; *b = t; }
static void iv_linear_insert(iv_t *last)                      /* __unguarded_linear_insert */
{
    iv_t val = *last, *next = last - 1;
    while (val.v < next->v) { *last = *next; last = next; next--; }
    *last = val;
}
static void iv_insertion(iv_t *first, iv_t *last)             /* __insertion_sort */
{
    iv_t *i;
    if (first == last) return;
    for (i = first + 1; i != last; i++) {
        if (i->v < first->v) {
            iv_t val = *i;
            memmove(first + 1, first, (size_t)(i - first) * sizeof(iv_t));
            *first = val;
        } else iv_linear_insert(i);
    }
}
static iv_t *iv_partition_pivot(iv_t *first, iv_t *last)      /* __unguarded_partition_pivot */
{
    iv_t *mid = first + (last - first) / 2, *a = first + 1, *b = mid, *c = last - 1, *lo, *hi;
    if (a->v < b->v) {                                        /* __move_median_to_first(first, a, b, c) */
        if (b->v < c->v) iv_swap(first, b);
        else if (a->v < c->v) iv_swap(first, c);
        else iv_swap(first, a);
    } else if (a->v < c->v) iv_swap(first, a);
    else if (b->v < c->v) iv_swap(first, c);
    else iv_swap(first, b);
    lo = first + 1; hi = last;                                /* __unguarded_partition(first + 1, last, first) */
    for (;;) {
        while (lo->v < first->v) lo++;
        hi--;
        while (first->v < hi->v) hi--;
        if (!(lo < hi)) return lo;
        iv_swap(lo, hi);
        lo++;
    }
}
static void iv_introsort_loop(iv_t *first, iv_t *last, int depth)
{
    while (last - first > 16) {
        iv_t *cut;
        if (depth == 0) { fprintf(stderr, "oracle: std::sort's heap-sort fallback is not restated\n"); abort(); }
        depth--;
        cut = iv_partition_pivot(first, last);
        iv_introsort_loop(cut, last, depth);
        last = cut;
    }
}
static void iv_sort(int *dim, double *val, int m)
{
    iv_t *a;
    int i, lg = 0;
    if (m < 2) return;
    a = (iv_t *)malloc(sizeof(iv_t) * (size_t)m);
    for (i = 0; i < m; i++) { a[i].d = dim[i]; a[i].v = val[i]; }
    while ((m >> (lg + 1)) > 0) lg++;
    iv_introsort_loop(a, a + m, 2 * lg);
    if (m > 16) {                                             /* __final_insertion_sort */
        iv_insertion(a, a + 16);
        for (i = 16; i < m; i++) iv_linear_insert(a + i);
    } else iv_insertion(a, a + m);
    for (i = 0; i < m; i++) { dim[i] = a[i].d; val[i] = a[i].v; }
    free(a);
}

/* cpp/direct.cpp:146-235; appends the new rectangles to 'out' */
static void d_divide(dstate_t *S, const rect_t *rec, rvec_t *out)
{
    int n = S->n, i, k, m = 0;
    int *dim = (int *)malloc(sizeof(int) * n);
    double *val = (double *)malloc(sizeof(double) * n);
    double *s = (double *)malloc(sizeof(double) * n);
    double *lbt = (double *)malloc(sizeof(double) * 2 * n), *ubt = lbt + n;
    double maxlength = rec->ub[0] - rec->lb[0];
    rect_t old;
    for (i = 1; i < n; i++)
        if (!S->fixed[i] && rec->ub[i] - rec->lb[i] > maxlength) maxlength = rec->ub[i] - rec->lb[i];
    for (i = 0; i < n; i++) {
        if (!S->fixed[i] && rec->ub[i] - rec->lb[i] == maxlength) {
            double f1, f2;
            memcpy(s, rec->center, sizeof(double) * n);
            s[i] = rec->lb[i] + maxlength / 3.;
            f1 = d_sample(S, s);
            s[i] = rec->lb[i] + 2. * maxlength / 3.;
            f2 = d_sample(S, s);
            dim[m] = i; val[m] = (f1 < f2) ? f1 : f2; m++;
        }
    }
    /* ascending by value (cpp/direct.cpp:194, std::sort with sortByVal): see iv_sort */
    iv_sort(dim, val, m);
    rect_copy(&old, rec, n);
    for (k = 0; k < m; k++) {
        int dd = dim[k];
        double w = old.ub[dd] - old.lb[dd];
        double split1 = old.lb[dd] + w / 3.;
        double split2 = old.lb[dd] + 2. * w / 3.;
        rect_t child;
        memcpy(lbt, old.lb, sizeof(double) * n);
        memcpy(ubt, old.ub, sizeof(double) * n);
        ubt[dd] = split1;
        rect_make(&child, S, lbt, ubt);
        rvec_push(out, child);
        ubt[dd] = old.ub[dd];
        lbt[dd] = split2;
        old.lb[dd] = split1;
        old.ub[dd] = split2;
        rect_make(&child, S, lbt, ubt);
        rvec_push(out, child);
    }
    {
        double d = 0.0;
        for (i = 0; i < n; i++) d += (old.lb[i] - old.center[i]) * (old.lb[i] - old.center[i]);
        old.d = sqrt(d);
    }
    rvec_push(out, old);
    free(dim); free(val); free(s); free(lbt);
}

static double *direct_run(dstate_t *S, int n, const double *lb, const double *ub,
                          int maxiter, int maxtime, int maxsample, long *nsamples_out)
{
    time_t start = time(NULL);
    rvec_t recs = {0, 0, 0};
    rect_t first;
    double *zero = (double *)calloc(n, sizeof(double));
    double *one = (double *)malloc(sizeof(double) * n);
    double *res;
    const double epsilon = 10e-10;
    int iteration = 0, done = 0, i;
    size_t *potopts = NULL; size_t npot, potcap = 0;

    S->n = n;
    S->lowerb = (double *)malloc(sizeof(double) * n);
    S->upperb = (double *)malloc(sizeof(double) * n);
    S->fixed = (int *)malloc(sizeof(int) * n);
    S->XMIN = (double *)calloc(n, sizeof(double));
    S->scratch = (double *)malloc(sizeof(double) * n);
    S->FMIN = DBL_MAX; S->nsamples = 0;
    for (i = 0; i < n; i++) {
        S->lowerb[i] = lb[i]; S->upperb[i] = ub[i];
        S->fixed[i] = (lb[i] == ub[i]);
        one[i] = 1.0;
    }
    rect_make(&first, S, zero, one);
    d_divide(S, &first, &recs);
    rect_free(&first);

    while (iteration < maxiter && !done) {
        size_t j, nrec = recs.len;
        long ind;
        iteration++;
        npot = 0;
        /* potentially-optimal scan, cpp/direct.cpp:378-471 */
        for (j = 0; j < nrec; j++) {
            double maxI1 = DBL_MIN, minI2 = DBL_MAX;
            int stop = 0; size_t q;
            const rect_t *Rj = &recs.v[j];
            for (q = 0; q < nrec; q++) {
                const rect_t *Ri = &recs.v[q];
                if (q == j) continue;
                if (Ri->d < Rj->d) {
                    double v = (Rj->y - Ri->y) / (Rj->d - Ri->d);
                    if (v > maxI1) maxI1 = v;
                } else if (Ri->d > Rj->d) {
                    double v = (Ri->y - Rj->y) / (Ri->d - Rj->d);
                    if (v < minI2) {
                        minI2 = v;
                        if (minI2 <= 0.) { stop = 1; break; }
                    }
                } else if (Rj->y > Ri->y) { stop = 1; break; }
                if (maxI1 != DBL_MIN && minI2 != DBL_MAX && minI2 < maxI1) { stop = 1; break; }
            }
            if (!stop) {
                int take = 0;
                if (minI2 == DBL_MAX) take = 1;
                else if (S->FMIN == 0.0) take = (Rj->y <= Rj->d * minI2);
                else take = (epsilon <= (S->FMIN - Rj->y) / fabs(S->FMIN) + (Rj->d / fabs(S->FMIN)) * minI2);
                if (take) {
                    if (npot == potcap) {
                        potcap = potcap ? potcap * 2 : 64;
                        potopts = (size_t *)realloc(potopts, potcap * sizeof(size_t));
                    }
                    potopts[npot++] = j;
                }
            }
        }
        if (npot == 0) {
            printf("[cdirect] could not divide any more\n");
            break;
        }
        for (ind = (long)npot - 1; ind >= 0; ind--) {
            size_t jj = potopts[ind];
            rect_t victim = recs.v[jj];
            d_divide(S, &victim, &recs);
            /* erase element jj (cpp/direct.cpp:484) */
            memmove(&recs.v[jj], &recs.v[jj + 1], (recs.len - jj - 1) * sizeof(rect_t));
            recs.len--;
            rect_free(&victim);
            if (S->nsamples > (unsigned)maxsample) { done = 1; break; }
            if (time(NULL) - start > maxtime) { done = 1; break; }
        }
        if (time(NULL) - start > maxtime) break;
        if (S->nsamples > (unsigned)maxsample) break;
    }

    res = (double *)malloc(sizeof(double) * (n + 1));
    res[0] = S->FMIN;
    for (i = 0; i < n; i++) res[i + 1] = S->XMIN[i];
    if (nsamples_out) *nsamples_out = (long)S->nsamples;
    for (i = 0; (size_t)i < recs.len; i++) rect_free(&recs.v[i]);
    free(recs.v); free(potopts); free(zero); free(one);
    free(S->lowerb); free(S->upperb); free(S->fixed); free(S->XMIN); free(S->scratch);
    return res;
}

/* same contract as the reference's `direct` (cpp/direct.h:76): returns a
 * malloc'd [fmin, xmin...]; adds an optional sample counter */
double *orc_direct(orc_objective_t fn, int n, double *lb, double *ub,
                   int maxiter, int maxtime, int maxsample, long *nsamples_out)
{
    dstate_t S; memset(&S, 0, sizeof(S));
    S.fn = fn;
    return direct_run(&S, n, lb, ub, maxiter, maxtime, maxsample, nsamples_out);
}

static double gp_ctx_obj(void *ctx, int n, const double *x) { (void)n; return neg_acq((orc_gp_t *)ctx, x); }

/* restatement of acqmaxGP (cpp/optimizeGP.cpp:262-349) with explicit sf2,
 * erf flavour and clamp.  result[0] = min of the NEGATED acquisition. */
double *orc_acqmax_gp(int D, double *lb, double *ub, const double *invR, const double *X,
                      const double *Y, int N, int acq, int ktype, const double *hyper, double sf2,
                      int nb, const double *pmeans, const double *pbeta, double ptheta,
                      const double *plowerb, const double *pwidth, double parm, double noise,
                      int erf_mode, double clamp_lo,
                      int maxiter, int maxtime, int maxsample, long *nsamples_out)
{
    orc_gp_t g; dstate_t S; double *res;
    if (acq < 0 || acq > 2) return NULL;
    if (gp_init(&g, D, invR, X, Y, N, acq, ktype, hyper, sf2, nb, pmeans, pbeta, ptheta,
                plowerb, pwidth, parm, noise, erf_mode, clamp_lo)) return NULL;
    memset(&S, 0, sizeof(S));
    S.fnc = gp_ctx_obj; S.ctx = &g;
    res = direct_run(&S, D, lb, ub, maxiter, maxtime, maxsample, nsamples_out);
    gp_free(&g);
    return res;
}

/* ------------------------------------------------------------------------ */
/* negative log marginal likelihood, Cholesky branch, no gradient:           */
/*   ego/gaussianprocess/trainhyper.py:47-75                                 */
/*   K = covMatrix(X) + noise I; nlml = Y.a/2 + sum log diag L + N/2 log 2pi */
/* Returns NAN when K is not positive definite.                              */
/* ------------------------------------------------------------------------ */
double orc_nlml(int ktype, int N, int D, const double *X, const double *Y,
                const double *hyper, double sf2, double noise)
{
    double *K = (double *)malloc(sizeof(double) * (size_t)N * N);
    double *L = (double *)malloc(sizeof(double) * (size_t)N * N);
    double *z = (double *)malloc(sizeof(double) * N);
    double v = NAN; int i;
    orc_cov_matrix(ktype, N, D, X, hyper, sf2, K);
    for (i = 0; i < N; i++) K[(size_t)i * N + i] += noise;
    if (orc_cholesky(N, K, L) == 0) {
        double quad = 0.0, logdet = 0.0;
        fwd_solve(N, L, Y, z);           /* Y.alpha = |L^-1 Y|^2 */
        for (i = 0; i < N; i++) { quad += z[i] * z[i]; logdet += log(L[(size_t)i * N + i]); }
        v = 0.5 * quad + logdet + 0.5 * N * log(2.0 * M_PI);
    }
    free(K); free(L); free(z);
    return v;
}

void orc_free(void *p) { free(p); }
