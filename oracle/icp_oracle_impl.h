/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Precision-generic body of the CPU restatement,
 * included twice by icp_oracle.c (REAL = double, SUF = f64; REAL = float, SUF = f32).
 *
 * Every function restates, in plain scalar C, one statement group of the reference's
 * CPU programs (citations are relative to /root/reference/):
 *   REAL=double : src/ICP_CPU.c                         (fp64 point-to-point, MKL d- and vd-routines)
 *   REAL=float  : src/CUDA/CPU_ICP_point_to_point.cpp   (fp32 twin, MKL s- and vs-routines)
 *
 * Layout is the reference CPU layout: SoA, "3 x N row-major"  (x[0..n) y[0..n) z[0..n)).
 * The file must be compiled with -ffp-contract=off: the reference's VML calls round every
 * sub / square / add separately (ICP_CPU.c:227-231).
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* ---- matching: ICP_CPU.c:220-234 / CPU_ICP_point_to_point.cpp:188-203 -------------------
 * for every moving point j: p1 = q - repmat(pt_j); p1 = p1^2; dist = (p1x + p1y) + p1z;
 * q_idx[j] = cblas_i?amin(dist)  -> FIRST index of the minimum |dist| (dist >= 0).      */
void FN(orc_nn)(const REAL* pt, int n, const REAL* q, int m, int* q_idx)
{
    const REAL* qx = q;
    const REAL* qy = q + (size_t)m;
    const REAL* qz = q + 2 * (size_t)m;
    /* (the moving points are independent: orc_set_threads(k > 1) spreads them over k threads for the all-cores CPU
     * baseline of bench.py; the default, 1, is the reference's scalar loop) */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(orc_threads_)
#endif
    for (int j = 0; j < n; j++) {
        const REAL px = pt[j], py = pt[j + (size_t)n], pz = pt[j + 2 * (size_t)n];
        REAL best = 0;
        int besti = 0;
        for (int c = 0; c < m; c++) {
            REAL dx = qx[c] - px; /* vdSub(q, p1)          :227 */
            REAL dy = qy[c] - py;
            REAL dz = qz[c] - pz;
            dx = dx * dx;         /* vdSqr                 :228 */
            dy = dy * dy;
            dz = dz * dz;
            REAL d = dx + dy;     /* vdAdd(p1x, p1y)       :230 */
            d = d + dz;           /* vdAdd(dist, p1z)      :231 */
            if (c == 0 || d < best) { /* idamin: first minimum  :232 */
                best = d;
                besti = c;
            }
        }
        q_idx[j] = besti;
    }
}

/* ---- centroid + deviation: ICP_CPU.c:342-366 ----------------------------------------------
 * sequential sums of the gathered cloud, bar = sum * (1/n), mark = cloud[index] - bar.
 * index == NULL means the identity (p_idx[k] = k, ICP_CPU.c:213).                          */
void FN(orc_centroid_deviation)(const REAL* cloud, int cloud_size, const int* index, REAL* bar, REAL* mark)
{
    REAL x = 0, y = 0, z = 0;
    const size_t cs = (size_t)cloud_size;
    for (int i = 0; i < cloud_size; i++) {
        const int k = index ? index[i] : i;
        x += cloud[k];
        y += cloud[k + cs];
        z += cloud[k + 2 * cs];
    }
    if (sizeof(REAL) == 8) {
        const REAL inv = (REAL)1.0 / (REAL)cloud_size;
        bar[0] = x * inv;
        bar[1] = y * inv;
        bar[2] = z * inv;
    } else { /* the fp32 twin divides: CPU_ICP_point_to_point.cpp:355-357 */
        bar[0] = x / (REAL)cloud_size;
        bar[1] = y / (REAL)cloud_size;
        bar[2] = z / (REAL)cloud_size;
    }
    for (int i = 0; i < cloud_size; i++) {
        const int k = index ? index[i] : i;
        for (int j = 0; j < 3; j++) mark[i + cs * j] = cloud[k + cs * j] - bar[j];
    }
}

/* ---- minimisation: ICP_CPU.c:237-248 ------------------------------------------------------
 * N = q_mark * p_mark^T (3x3 row-major, K = n), N = U S Vt, R = U * Vt (no det check),
 * G = R * p_bar, t = q_bar - G.  The SVD itself is done in double for both precisions
 * (LAPACKE_?gesvd is third-party; R = U*Vt is the orthogonal polar factor of N and does
 * not depend on the SVD's sign/ordering conventions when N is non-singular).               */
int FN(orc_p2p_minimize)(const REAL* pt, int n, const REAL* q, int m, const int* q_idx,
                         REAL* R /*9 row-major*/, REAL* t /*3*/, REAL* N_out /*9 or NULL*/)
{
    const size_t ns = (size_t)n;
    REAL* q_mark = (REAL*)malloc(3 * ns * sizeof(REAL));
    REAL* p_mark = (REAL*)malloc(3 * ns * sizeof(REAL));
    if (!q_mark || !p_mark) { free(q_mark); free(p_mark); return -1; }
    REAL q_bar[3], p_bar[3];
    /* the reference passes q_size as cloud_size and iterates it; here n == m in every
     * reference program, so the gathered model cloud has n entries (ICP_CPU.c:237). */
    {
        /* gather-aware centroid of q over q_idx, with q's own stride m */
        REAL x = 0, y = 0, z = 0;
        const size_t ms = (size_t)m;
        for (int i = 0; i < n; i++) {
            const int k = q_idx[i];
            x += q[k];
            y += q[k + ms];
            z += q[k + 2 * ms];
        }
        /* ICP_CPU.c:355 multiplies by 1.0/n; the fp32 twin divides (CPU_ICP_point_to_point.cpp:355-363) */
        if (sizeof(REAL) == 8) {
            const REAL inv = (REAL)1.0 / (REAL)n;
            q_bar[0] = x * inv; q_bar[1] = y * inv; q_bar[2] = z * inv;
        } else {
            q_bar[0] = x / (REAL)n; q_bar[1] = y / (REAL)n; q_bar[2] = z / (REAL)n;
        }
        for (int i = 0; i < n; i++) {
            const int k = q_idx[i];
            for (int j = 0; j < 3; j++) q_mark[i + ns * j] = q[k + ms * j] - q_bar[j];
        }
    }
    FN(orc_centroid_deviation)(pt, n, NULL, p_bar, p_mark);

    REAL Nm[9];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            REAL s = 0;
            for (int i = 0; i < n; i++) s += q_mark[i + ns * a] * p_mark[i + ns * b];
            Nm[a * 3 + b] = s;
        }
    free(q_mark);
    free(p_mark);
    if (N_out) for (int i = 0; i < 9; i++) N_out[i] = Nm[i];

    double Nd[9], U[9], S[3], Vt[9];
    for (int i = 0; i < 9; i++) Nd[i] = (double)Nm[i];
    orc_svd3(Nd, U, S, Vt);
    REAL Rr[9];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += U[a * 3 + k] * Vt[k * 3 + b]; /* :246 */
            Rr[a * 3 + b] = (REAL)s;
        }
    for (int a = 0; a < 3; a++) {
        REAL g = 0;
        for (int k = 0; k < 3; k++) g += Rr[a * 3 + k] * p_bar[k]; /* G = R*p_bar :247 */
        t[a] = q_bar[a] - g;                                      /* :248 */
    }
    for (int i = 0; i < 9; i++) R[i] = Rr[i];
    return 0;
}

/* ---- transformation: ICP_CPU.c:251-253 ---- C = R*pt (gemm), pt = C + repmat(t) -------- */
void FN(orc_transform)(REAL* pt, int n, const REAL* R, const REAL* t)
{
    const size_t ns = (size_t)n;
    for (int i = 0; i < n; i++) {
        const REAL x = pt[i], y = pt[i + ns], z = pt[i + 2 * ns];
        for (int a = 0; a < 3; a++) {
            REAL c = R[a * 3 + 0] * x;
            c = c + R[a * 3 + 1] * y;
            c = c + R[a * 3 + 2] * z;
            pt[i + ns * a] = c + t[a];
        }
    }
}

/* ---- error: ICP_CPU.c:257-266 ---- E = || q[q_idx] - pt ||_2 / sqrt(n) ------------------ */
REAL FN(orc_rms_error)(const REAL* pt, int n, const REAL* q, int m, const int* q_idx)
{
    const size_t ns = (size_t)n, ms = (size_t)m;
    REAL s = 0;
    for (int k = 0; k < 3; k++)
        for (int i = 0; i < n; i++) {
            const REAL c = q[q_idx[i] + ms * k] - pt[i + ns * k];
            s += c * c;
        }
    return (REAL)(sqrt((double)s) / pow((double)n, 0.5));
}

/* ---- the driver loop: ICP_CPU.c:217-271 ---------------------------------------------------
 * E has max_iter+1 entries, E[0] = 0.  Stop rule (:267-269): after computing E[i+1] break if
 * E[i+1] < tol or |E[i+1]-E[i]| < tol (i NOT incremented); else i++ and break when
 * i > max_iter-1.  Returns num_iterations (= i at exit) and the number of matching passes
 * executed in *passes.  T_total (row-major 4x4, double) is the left-composed product of
 * every [R_k | t_k] that was applied.  idx_last receives the last pass's correspondences.
 * fixed != 0 disables the tolerance test (ICP_standard.cu:369 runs a fixed 40 passes).     */
int FN(orc_icp_p2p)(const REAL* D, const REAL* M, int n, int m, int max_iter, double tol, int fixed,
                    REAL* E, double* T_total, int* idx_last, REAL* pt_out, int* passes)
{
    const size_t ns = (size_t)n;
    REAL* pt = (REAL*)malloc(3 * ns * sizeof(REAL));
    int* q_idx = (int*)malloc(ns * sizeof(int));
    if (!pt || !q_idx) { free(pt); free(q_idx); return -1; }
    memcpy(pt, D, 3 * ns * sizeof(REAL));
    for (int k = 0; k <= max_iter; k++) E[k] = 0;
    double T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    int i = 0, npass = 0;
    while (1) {
        FN(orc_nn)(pt, n, M, m, q_idx);
        npass++;
        REAL R[9], t[3];
        if (FN(orc_p2p_minimize)(pt, n, M, m, q_idx, R, t, NULL)) { free(pt); free(q_idx); return -1; }
        FN(orc_transform)(pt, n, R, t);
        {
            double Tk[16] = {R[0], R[1], R[2], t[0], R[3], R[4], R[5], t[1], R[6], R[7], R[8], t[2], 0, 0, 0, 1};
            double Tn[16];
            for (int a = 0; a < 4; a++)
                for (int b = 0; b < 4; b++) {
                    double s = 0;
                    for (int k = 0; k < 4; k++) s += Tk[a * 4 + k] * T[k * 4 + b];
                    Tn[a * 4 + b] = s;
                }
            memcpy(T, Tn, sizeof T);
        }
        E[i + 1] = FN(orc_rms_error)(pt, n, M, m, q_idx);
        if (!fixed && ((E[i + 1] < tol) || (fabs((double)E[i + 1] - (double)E[i]) < tol))) break;
        i++;
        if (i > max_iter - 1) break;
    }
    if (T_total) memcpy(T_total, T, sizeof T);
    if (idx_last) memcpy(idx_last, q_idx, ns * sizeof(int));
    if (pt_out) memcpy(pt_out, pt, 3 * ns * sizeof(REAL));
    if (passes) *passes = npass;
    free(pt);
    free(q_idx);
    return i;
}

#undef FN
#undef CAT
#undef CAT_
