// icp_host_math.cpp -- the tiny dense solves that north_star keeps on the host:
//   3x3 SVD -> R = U*Vt        replaces cusolverDnSgesvd + cublasSgemm (src/ICP_point_to_point.cu:369-381)
//                              and LAPACKE_dgesvd + dgemm (src/ICP_CPU.c:240-246)
//   6x6 Cholesky (upper)       replaces cusolverDnSpotrf/Spotrs (src/ICP_point_to_plane.cu:576-581)
//   symmetric 3x3 eigen-solve  replaces LAPACKE_ssyev (src/ICP_point_to_plane.cu:429-438)
// Everything is fp64.  No device code here; the CPU test-suite calls these through the C ABI.
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/icp_mi355x.h"
#include "icp_host_math.h"

namespace icp {

// One-sided Jacobi run on the ROWS of a working copy W = A: row rotations (accumulated in Ut) drive
// the rows mutually orthogonal, W = Ut^T... i.e. J * A = diag(s) * Vt with J orthogonal.  Then
// A = J^T * diag(s) * Vt and the orthogonal polar factor is R = J^T * Vt_normalised = U * Vt.
// Row orientation keeps the rotation accumulation on the U side.
void svd3_rows(const double A[9], double U[9], double S[3], double Vt[9])
{
    double W[3][3], J[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            W[i][j] = A[i * 3 + j];
            J[i][j] = i == j ? 1.0 : 0.0;
        }
    const double eps = 2.220446049250313e-16;
    for (int sweep = 0; sweep < 64; ++sweep) {
        bool changed = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double app = 0, aqq = 0, apq = 0;
                for (int k = 0; k < 3; ++k) {
                    app += W[p][k] * W[p][k];
                    aqq += W[q][k] * W[q][k];
                    apq += W[p][k] * W[q][k];
                }
                if (apq == 0.0 || std::fabs(apq) <= eps * std::sqrt(app * aqq)) continue;
                changed = true;
                const double tau = (aqq - app) / (2.0 * apq);
                const double t = std::copysign(1.0, tau) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int k = 0; k < 3; ++k) {
                    const double wp = W[p][k], wq = W[q][k];
                    W[p][k] = c * wp - s * wq;
                    W[q][k] = s * wp + c * wq;
                    const double jp = J[p][k], jq = J[q][k];
                    J[p][k] = c * jp - s * jq;
                    J[q][k] = s * jp + c * jq;
                }
            }
        if (!changed) break;
    }
    double nrm[3];
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 3; ++i) nrm[i] = std::sqrt(W[i][0] * W[i][0] + W[i][1] * W[i][1] + W[i][2] * W[i][2]);
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (nrm[ord[j]] > nrm[ord[i]]) std::swap(ord[i], ord[j]);
    double v[3][3];
    bool have[3] = {false, false, false};
    const double tiny = 1e-14 * (nrm[ord[0]] > 0 ? nrm[ord[0]] : 1.0);
    for (int k = 0; k < 3; ++k) {
        const int r = ord[k];
        S[k] = nrm[r];
        if (nrm[r] > tiny) {
            for (int c = 0; c < 3; ++c) v[k][c] = W[r][c] / nrm[r];
            have[k] = true;
        }
    }
    auto cross = [](const double* a, const double* b, double* c) {
        c[0] = a[1] * b[2] - a[2] * b[1];
        c[1] = a[2] * b[0] - a[0] * b[2];
        c[2] = a[0] * b[1] - a[1] * b[0];
    };
    if (!have[0]) { v[0][0] = 1; v[0][1] = 0; v[0][2] = 0; }
    if (!have[1]) {
        int mi = 0;
        for (int c = 1; c < 3; ++c) if (std::fabs(v[0][c]) < std::fabs(v[0][mi])) mi = c;
        double e[3] = {0, 0, 0};
        e[mi] = 1;
        cross(v[0], e, v[1]);
        const double nn = std::sqrt(v[1][0] * v[1][0] + v[1][1] * v[1][1] + v[1][2] * v[1][2]);
        for (int c = 0; c < 3; ++c) v[1][c] /= nn;
    }
    if (!have[2]) cross(v[0], v[1], v[2]);
    for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 3; ++c) {
            Vt[k * 3 + c] = v[k][c];
            U[c * 3 + k] = J[ord[k]][c];  // U = J^T with columns permuted like the singular values
        }
}

int solve_point_to_point(const double* mom, double* R, double* t)
{
    const double n = mom[ICP_MOM_CNT];
    if (!(n > 0)) return ICP_ERR_INVALID;
    double pb[3], qb[3];
    for (int a = 0; a < 3; ++a) {
        pb[a] = mom[ICP_MOM_SP + a] / n;
        qb[a] = mom[ICP_MOM_SQ + a] / n;
    }
    double N[9];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) N[a * 3 + b] = mom[ICP_MOM_SQP + a * 3 + b] - n * qb[a] * pb[b];
    double U[9], S[3], Vt[9];
    svd3_rows(N, U, S, Vt);
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += U[a * 3 + k] * Vt[k * 3 + b];
            R[a * 3 + b] = s;  // no det(R) check: src/ICP_CPU.c:246 has none
        }
    for (int a = 0; a < 3; ++a) t[a] = qb[a] - (R[a * 3 + 0] * pb[0] + R[a * 3 + 1] * pb[1] + R[a * 3 + 2] * pb[2]);
    return ICP_OK;
}

// Upper Cholesky C = Ut*U of the 6x6 SPD matrix given by its upper triangle (row-major packed, 21).
int solve_point_to_plane(const double* mom, double* R, double* t, double* x6)
{
    double Uc[6][6];
    {
        int o = ICP_MOM_C;
        for (int a = 0; a < 6; ++a)
            for (int c = a; c < 6; ++c) Uc[a][c] = mom[o++];
    }
    for (int k = 0; k < 6; ++k) {
        double d = Uc[k][k];
        for (int r = 0; r < k; ++r) d -= Uc[r][k] * Uc[r][k];
        if (!(d > 0.0)) return ICP_ERR_SINGULAR;  // potrf's info > 0
        d = std::sqrt(d);
        Uc[k][k] = d;
        for (int c = k + 1; c < 6; ++c) {
            double s = Uc[k][c];
            for (int r = 0; r < k; ++r) s -= Uc[r][k] * Uc[r][c];
            Uc[k][c] = s / d;
        }
    }
    double y[6], x[6];
    for (int i = 0; i < 6; ++i) {  // Ut y = b
        double s = mom[ICP_MOM_B + i];
        for (int r = 0; r < i; ++r) s -= Uc[r][i] * y[r];
        y[i] = s / Uc[i][i];
    }
    for (int i = 5; i >= 0; --i) {  // U x = y
        double s = y[i];
        for (int c = i + 1; c < 6; ++c) s -= Uc[i][c] * x[c];
        x[i] = s / Uc[i][i];
    }
    if (x6) std::memcpy(x6, x, sizeof x);
    const double cx = std::cos(x[0]), cy = std::cos(x[1]), cz = std::cos(x[2]);
    const double sx = std::sin(x[0]), sy = std::sin(x[1]), sz = std::sin(x[2]);
    // R = Rz(gamma) Ry(beta) Rx(alpha), row-major (src/CUDA/CPU_ICP_point_to-plane.cpp:381-383)
    R[0] = cy * cz; R[1] = cz * sx * sy - cx * sz; R[2] = cx * cz * sy + sx * sz;
    R[3] = cy * sz; R[4] = cx * cz + sx * sy * sz; R[5] = cx * sy * sz - cz * sx;
    R[6] = -sy;     R[7] = cy * sx;                R[8] = cx * cy;
    t[0] = x[3]; t[1] = x[4]; t[2] = x[5];
    return ICP_OK;
}

// cyclic Jacobi on the full symmetric matrix built from the upper triangle
void eigh3(const double A[9], double w[3], double Z[9])
{
    double a[3][3], v[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            a[i][j] = j >= i ? A[i * 3 + j] : A[j * 3 + i];
            v[i][j] = i == j ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 64; ++sweep) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        const double dia = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (off <= 1e-34 * dia || off == 0.0) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double apq = a[p][q];
                if (apq == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                const double t = std::copysign(1.0, theta) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                const double app = a[p][p], aqq = a[q][q];
                a[p][p] = app - t * apq;
                a[q][q] = aqq + t * apq;
                a[p][q] = a[q][p] = 0.0;
                const int r = 3 - p - q;  // the remaining index
                const double arp = a[r][p], arq = a[r][q];
                a[r][p] = a[p][r] = c * arp - s * arq;
                a[r][q] = a[q][r] = s * arp + c * arq;
                for (int k = 0; k < 3; ++k) {
                    const double vp = v[k][p], vq = v[k][q];
                    v[k][p] = c * vp - s * vq;
                    v[k][q] = s * vp + c * vq;
                }
            }
    }
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (a[ord[j]][ord[j]] < a[ord[i]][ord[i]]) std::swap(ord[i], ord[j]);
    for (int k = 0; k < 3; ++k) {
        w[k] = a[ord[k]][ord[k]];
        for (int i = 0; i < 3; ++i) Z[i * 3 + k] = v[i][ord[k]];
    }
}

int shard_range(int64_t n, int rank, int world, int64_t* begin, int64_t* count)
{
    if (n < 0 || world <= 0 || rank < 0 || rank >= world) return ICP_ERR_INVALID;
    const int64_t base = n / world, rem = n % world;
    *begin = rank * base + (rank < rem ? rank : rem);
    *count = base + (rank < rem ? 1 : 0);
    return ICP_OK;
}

}  // namespace icp

extern "C" {
int icp_solve_point_to_point(const double* mom, double* R9, double* t3)
{
    if (!mom || !R9 || !t3) return ICP_ERR_INVALID;
    return icp::solve_point_to_point(mom, R9, t3);
}
int icp_solve_point_to_plane(const double* mom, double* R9, double* t3, double* x6)
{
    if (!mom || !R9 || !t3) return ICP_ERR_INVALID;
    return icp::solve_point_to_plane(mom, R9, t3, x6);
}
int icp_shard_range(int64_t n, int rank, int world, int64_t* begin, int64_t* count)
{
    if (!begin || !count) return ICP_ERR_INVALID;
    return icp::shard_range(n, rank, world, begin, count);
}
int icp_eigh3(const double* A9, double* w3, double* Z9)
{
    if (!A9 || !w3 || !Z9) return ICP_ERR_INVALID;
    icp::eigh3(A9, w3, Z9);
    return ICP_OK;
}
}
