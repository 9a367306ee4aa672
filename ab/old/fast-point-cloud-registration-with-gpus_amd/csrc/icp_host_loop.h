// icp_host_loop.h -- the host half of the ICP driver loop, free of any device call: error series, stop
// rule, minimisation, transform composition.  The device loop (icp_api.cpp) and the host-only C ABI
// (icp_host_loop_*, used by the multi-rank CPU tests) run this one implementation.
#pragma once
#include <vector>

#include "../../include/icp_mi355x.h"

namespace icp {

struct HostLoop {
    icp_params prm{};
    int applied = 0;      // transforms applied so far (= index k of the error being produced next)
    int iterations = 0;   // the reference's loop counter at exit
    bool done = false;
    bool have_rt = false; // R, t of the last advance() wait to be applied
    double n_total = 0.0; // moving points that contributed (summed over ranks)
    double R[9], t[3], T[16];
    std::vector<double> err;

    int begin(const icp_params& p);
    // the caller has applied (R, t) to the moving cloud: compose T (with the values rounded to the storage
    // precision, i.e. exactly what the kernel multiplied by), count the pass
    void note_applied();
    // true when the pass that is about to be enqueued is the last one whatever its error turns out to be
    bool next_is_final() const { return applied + (have_rt ? 1 : 0) >= prm.max_iter; }
    // feed the (rank-reduced) moment vector of the enqueue that followed note_applied(): E[k], the stop
    // rule of src/ICP_CPU.c:267-269, and -- unless the loop ended -- the next R, t.
    int advance(const double* mom);
};

}  // namespace icp
