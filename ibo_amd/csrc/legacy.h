// legacy.h -- libego's arithmetic in libego's order, for the legacy symbol acqmaxGP (legacy.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <vector>

int launch_legacy_transpose(const double *M, double *MT, int N, hipStream_t s);
int launch_legacy_aMb(const double *MT, const double *B, const double *A, double *Mb, double *out, int N, int nvec, hipStream_t s);
int launch_legacy_dots(const double *Mb, const double *A, double *out, int N, int nvec, hipStream_t s);

// what acqmaxGP was called with (borrowed pointers, valid for the call)
struct LegacySpec {
    int family = 0;             // IBO_K_*
    int dim = 0, rows = 0;
    const double *obs = nullptr, *targets = nullptr, *hyper = nullptr;
    double amp = 1.0;           // libego's sf2: 1 for kernel types 0-2, magnitude^2 for Matern-5/2
    int nbasis = 0;             // RBF-network mean prior (0: none)
    const double *centres = nullptr, *weights = nullptr, *origin = nullptr, *extent = nullptr;
    double sharpness = 0.0;
    int acq = 0;                // IBO_ACQ_EI / PI / UCB
    double parm = 0.0, noise = 0.0;
};

// The host half of libego's objective: k*, the mean prior and the acquisition formula with the host's libm, the k* vectors of a
// batch of sample points spread over a crew of host threads (every number is formed by one thread alone, in libego's order).
class LegacyHost {
public:
    struct Consts { std::vector<double> per_dim; double h, root3, root5, three_h2; };
    explicit LegacyHost(const LegacySpec &spec);
    ~LegacyHost();
    LegacyHost(const LegacyHost &) = delete;
    LegacyHost &operator=(const LegacyHost &) = delete;
    void prepare(const double *pts, int n, double *vecs, double *prior_mu) const;
    double negated(double prior_mu, double c_mean, double c_var) const;
    double prior_at(const double *pt) const;
    double incumbent() const { return best; }
    int threads() const;
private:
    struct Crew;
    LegacySpec m;
    Consts k;
    double best, root2, root2pi;
    Crew *crew;
};
