// icp_host_loop.cpp -- see icp_host_loop.h.  Reference statements: src/ICP_CPU.c:239-248,257-270,
// src/ICP_point_to_point.cu:412-423, src/ICP_point_to_plane.cu:585-593,619-628.
#include "icp_host_loop.h"

#include <cmath>
#include <cstring>
#include <new>

#include "icp_host_math.h"

namespace icp {

int HostLoop::begin(const icp_params& p)
{
    if (p.max_iter < 1) return ICP_ERR_INVALID;
    if (p.metric != ICP_POINT_TO_POINT && p.metric != ICP_POINT_TO_PLANE) return ICP_ERR_INVALID;
    if (p.precision != ICP_F32 && p.precision != ICP_F64) return ICP_ERR_INVALID;
    *this = HostLoop();
    prm = p;
    err.assign((size_t)p.max_iter + 1, 0.0);
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
    return ICP_OK;
}

void HostLoop::note_applied()
{
    double Tk[16] = {0};
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) Tk[a * 4 + b] = prm.precision == ICP_F64 ? R[a * 3 + b] : (double)(float)R[a * 3 + b];
        Tk[a * 4 + 3] = prm.precision == ICP_F64 ? t[a] : (double)(float)t[a];
    }
    Tk[15] = 1.0;
    double Tn[16];
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += Tk[a * 4 + k] * T[k * 4 + b];
            Tn[a * 4 + b] = s;
        }
    std::memcpy(T, Tn, sizeof Tn);
    applied += 1;
    have_rt = false;
}

int HostLoop::advance(const double* mom)
{
    if (done) return ICP_ERR_STATE;
    if (mom[ICP_MOM_CNT] > 0) n_total = mom[ICP_MOM_CNT];
    const int k = applied;
    if (k >= 1) {
        if (!(n_total > 0)) return ICP_ERR_INVALID;
        // E[k] = || q[idx_{k-1}] - p_k ||_2 / sqrt(N)   (src/ICP_CPU.c:266)
        err[k] = std::sqrt(mom[ICP_MOM_ERR]) / std::sqrt(n_total);
        const bool stop = !prm.fixed_iterations && ((err[k] < prm.tol) || (std::fabs(err[k] - err[k - 1]) < prm.tol));
        if (stop) {
            iterations = k - 1;  // break before the counter is incremented (src/ICP_CPU.c:267)
            done = true;
        } else {
            iterations = k;
            if (k > prm.max_iter - 1) done = true;  // src/ICP_CPU.c:268-269
        }
    }
    if (!done) {
        const int rc = prm.metric == ICP_POINT_TO_PLANE ? solve_point_to_plane(mom, R, t, nullptr)
                                                        : solve_point_to_point(mom, R, t);
        if (rc != ICP_OK) {
            done = true;
            return rc;
        }
        have_rt = true;
    }
    return ICP_OK;
}

}  // namespace icp

// ---- host-only C ABI: the same state machine without a device (multi-rank CPU tests, custom drivers) ----
struct icp_host_loop {
    icp::HostLoop H;
};

extern "C" {

int icp_host_loop_create(const icp_params* prm, icp_host_loop** out)
{
    if (!prm || !out) return ICP_ERR_INVALID;
    *out = nullptr;
    icp_host_loop* h = new (std::nothrow) icp_host_loop();
    if (!h) return ICP_ERR_NOMEM;
    const int rc = h->H.begin(*prm);
    if (rc != ICP_OK) {
        delete h;
        return rc;
    }
    *out = h;
    return ICP_OK;
}

void icp_host_loop_destroy(icp_host_loop* h) { delete h; }

int icp_host_loop_advance(icp_host_loop* h, const double* mom, int* done, double* R9, double* t3)
{
    if (!h || !mom) return ICP_ERR_INVALID;
    const int rc = h->H.advance(mom);
    if (done) *done = h->H.done ? 1 : 0;
    if (rc == ICP_OK && h->H.have_rt) {
        if (R9) std::memcpy(R9, h->H.R, sizeof h->H.R);
        if (t3) std::memcpy(t3, h->H.t, sizeof h->H.t);
    }
    return rc;
}

int icp_host_loop_note_applied(icp_host_loop* h)
{
    if (!h || !h->H.have_rt) return ICP_ERR_STATE;
    h->H.note_applied();
    return ICP_OK;
}

int icp_host_loop_state(icp_host_loop* h, int* iterations, int* passes, double* err, int err_cap, double* T16)
{
    if (!h) return ICP_ERR_INVALID;
    if (iterations) *iterations = h->H.iterations;
    if (passes) *passes = h->H.applied;
    if (err) {
        const int cnt = (int)h->H.err.size() < err_cap ? (int)h->H.err.size() : err_cap;
        for (int i = 0; i < cnt; ++i) err[i] = h->H.err[i];
    }
    if (T16) std::memcpy(T16, h->H.T, sizeof h->H.T);
    return ICP_OK;
}

}  // extern "C"
