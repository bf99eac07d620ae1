// direct_host.h -- DIRECT (dividing rectangles) tree logic on the host with the
// objective evaluated in batches.  Same search as the reference's native
// optimiser (cpp/direct.cpp:329-581); see direct_host.cpp.
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

namespace ibo {

// evaluate n points (n x D, row-major, already mapped to the caller's box);
// write n objective values (to be MINIMISED).  Return non-zero to abort.
typedef std::function<int(const double *pts, int n, double *vals)> batch_eval_t;

struct DirectOptions {
    int maxiter = 50;
    int maxtime = 30;          // whole seconds, as the reference (time(NULL))
    int maxsample = 10000;
    bool compat = true;        // reproduce the dimension-0 maxlength quirk (cpp/direct.cpp:156-164)
    bool per_rectangle = false;  // true: evaluate rectangle by rectangle (host callbacks, exact call order)
                                 // false: one batch of split points + one batch of centres per iteration
};

struct DirectResult {
    double fmin = 0.0;
    std::vector<double> xmin;
    int64_t nsamples = 0;
    int iterations = 0;
    int status = 0;            // 0 ok, otherwise the evaluator's error code
};

DirectResult direct_minimize(const batch_eval_t &eval, int D, const double *lb, const double *ub,
                             const DirectOptions &opt);

}  // namespace ibo
