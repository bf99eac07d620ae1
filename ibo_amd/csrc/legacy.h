// legacy.h -- libego's arithmetic in libego's order, for the legacy symbol acqmaxGP (legacy.hip)
#pragma once
#include <hip/hip_runtime.h>

int launch_legacy_transpose(const double *M, double *MT, int N, hipStream_t s);
int launch_legacy_aMb(const double *MT, const double *B, const double *A, double *Mb, double *out, int N, int nvec, hipStream_t s);
int launch_legacy_dots(const double *Mb, const double *A, double *out, int N, int nvec, hipStream_t s);
void legacy_kstar(int kerneltype, int NA, int NX, const double *X, const double *hyperparams, double sf2, const double *x, double *r);
double legacy_prior_mean(int NA, const double *x, int npbases, const double *pbasismeans, const double *pbasisbeta, double pbasistheta,
                         const double *pbasislowerb, const double *pbasiswidth);
double legacy_neg_acq(int acqfunc, double prior_mu, double x1, double x2, double noise, double maxY, double parm);
