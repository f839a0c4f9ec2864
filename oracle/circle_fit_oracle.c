/*
 * circle_fit_oracle.c -- CPU ORACLE for the landmark-extraction front end.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as nuslam_oracle.h).  Plain-C restatement of
 * nuslam/src/circle_fit_library.cpp: circleFit (:15-134), clusterPoints (:136-206), classifyCluster (:208-250).
 *
 * PARITY PINNING: PINNED by the reference's own Catch2 known answers, nuslam/tests/circle_tests.cpp:38-40 and :67-69
 * (centre and radius of two point sets, Approx = 1.2e-5 relative) -- restated in tests/test_circle_fit.py.
 *
 * Third-party arithmetic: the reference calls Armadillo's svd(), eig_sym() and solve() (circle_fit_library.cpp:65,
 * 88,103) on a n x 4 and two 4 x 4 matrices.  Armadillo (unpinned, absent here) forwards them to LAPACK
 * dgesdd / dsyev / dgesv; the results are defined up to rounding and up to the sign of singular / eigen vectors, and
 * the fitted circle is invariant under both sign choices (Y = V diag(s) V^T does not depend on the signs of V's
 * columns; a, b, R^2 are ratios that do not change under A -> -A).  Restated here with the textbook algorithms:
 * one-sided (Hestenes) Jacobi for the SVD of Z, cyclic Jacobi for the symmetric 4x4 eigenproblem, Gaussian
 * elimination with partial pivoting for the 4x4 solve.
 */
#include <limits.h>
#include <math.h>
#include <string.h>

enum { CF_OK = 0, CF_TOO_FEW = 1 /* marker.id = -1, :73-77 */, CF_DEGENERATE = 2 };

/* columns of a 4x4 matrix are stored column-major: M[i + 4*j] */

/* One-sided Jacobi SVD of Z (n x 4, column-major with leading dimension n, destroyed): on return the columns of Z are
 * U diag(s); V (4x4) holds the right singular vectors; s descending like LAPACK's. */
static void svd_n_by_4(double* Z, int n, double s[4], double V[16])
{
    for (int i = 0; i < 16; ++i) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                const double* zp = Z + (size_t)p * n;
                const double* zq = Z + (size_t)q * n;
                for (int i = 0; i < n; ++i) {
                    alpha += zp[i] * zp[i];
                    beta += zq[i] * zq[i];
                    gamma += zp[i] * zq[i];
                }
                if (gamma == 0.0) continue;
                const double rel = fabs(gamma) / sqrt(alpha * beta);
                if (rel > off) off = rel;
                if (rel < 1e-16) continue;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                double* wp = Z + (size_t)p * n;
                double* wq = Z + (size_t)q * n;
                for (int i = 0; i < n; ++i) {
                    const double a = wp[i], b = wq[i];
                    wp[i] = c * a - sn * b;
                    wq[i] = sn * a + c * b;
                }
                for (int i = 0; i < 4; ++i) {
                    const double a = V[i + 4 * p], b = V[i + 4 * q];
                    V[i + 4 * p] = c * a - sn * b;
                    V[i + 4 * q] = sn * a + c * b;
                }
            }
        if (off < 1e-15) break;
    }
    for (int k = 0; k < 4; ++k) {
        double a = 0.0;
        const double* z = Z + (size_t)k * n;
        for (int i = 0; i < n; ++i) a += z[i] * z[i];
        s[k] = sqrt(a);
    }
    for (int a = 0; a < 3; ++a)           /* sort descending, V columns along */
        for (int b = a + 1; b < 4; ++b)
            if (s[b] > s[a]) {
                double t = s[a]; s[a] = s[b]; s[b] = t;
                for (int i = 0; i < 4; ++i) { t = V[i + 4 * a]; V[i + 4 * a] = V[i + 4 * b]; V[i + 4 * b] = t; }
            }
}

/* cyclic Jacobi for a symmetric 4x4 (destroyed); eigenvalues in w, eigenvectors in the columns of E */
static void eig_sym4(double* A, double w[4], double E[16])
{
    for (int i = 0; i < 16; ++i) E[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int p = 0; p < 4; ++p) {
            diag += A[p + 4 * p] * A[p + 4 * p];
            for (int q = p + 1; q < 4; ++q) off += A[p + 4 * q] * A[p + 4 * q];
        }
        if (off <= 1e-32 * diag || off == 0.0) break;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                const double apq = A[p + 4 * q];
                if (apq == 0.0) continue;
                const double theta = (A[q + 4 * q] - A[p + 4 * p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(1.0 + theta * theta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < 4; ++k) {          /* A <- A J */
                    const double akp = A[k + 4 * p], akq = A[k + 4 * q];
                    A[k + 4 * p] = c * akp - sn * akq;
                    A[k + 4 * q] = sn * akp + c * akq;
                }
                for (int k = 0; k < 4; ++k) {          /* A <- J^T A */
                    const double apk = A[p + 4 * k], aqk = A[q + 4 * k];
                    A[p + 4 * k] = c * apk - sn * aqk;
                    A[q + 4 * k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < 4; ++k) {
                    const double ekp = E[k + 4 * p], ekq = E[k + 4 * q];
                    E[k + 4 * p] = c * ekp - sn * ekq;
                    E[k + 4 * q] = sn * ekp + c * ekq;
                }
            }
    }
    for (int k = 0; k < 4; ++k) w[k] = A[k + 4 * k];
}

/* x = M^-1 b, Gaussian elimination with partial pivoting (what dgesv does); returns 0 or 1 (singular) */
static int solve4(const double* M, const double b[4], double x[4])
{
    double a[4][5];
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) a[i][j] = M[i + 4 * j];
        a[i][4] = b[i];
    }
    for (int k = 0; k < 4; ++k) {
        int piv = k;
        for (int i = k + 1; i < 4; ++i)
            if (fabs(a[i][k]) > fabs(a[piv][k])) piv = i;
        if (a[piv][k] == 0.0) return 1;
        if (piv != k)
            for (int j = 0; j < 5; ++j) { const double t = a[k][j]; a[k][j] = a[piv][j]; a[piv][j] = t; }
        for (int i = k + 1; i < 4; ++i) {
            const double l = a[i][k] / a[k][k];
            for (int j = k; j < 5; ++j) a[i][j] -= l * a[k][j];
        }
    }
    for (int i = 3; i >= 0; --i) {
        double acc = a[i][4];
        for (int j = i + 1; j < 4; ++j) acc -= a[i][j] * x[j];
        x[i] = acc / a[i][i];
    }
    return 0;
}

/* circleFit, circle_fit_library.cpp:15-134.  xs, ys: n points; work: 4n doubles of scratch.
 * out = {centre x, centre y, radius}.  Returns CF_OK, CF_TOO_FEW (n < 4: the reference returns marker.id = -1). */
int orc_circle_fit(const double* xs, const double* ys, int n, double* work, double out[3])
{
    if (n < 4) return CF_TOO_FEW;                                   /* :73-77, s.size() < 4 */
    double x_hat = 0, y_hat = 0;
    for (int i = 0; i < n; ++i) { x_hat += xs[i] / n; y_hat += ys[i] / n; }   /* :21-25 */
    double z_bar = 0;
    double* Z = work;                                               /* n x 4 column-major: z, x, y, 1   :36-47 */
    for (int j = 0; j < n; ++j) {
        const double px = xs[j] - x_hat, py = ys[j] - y_hat;        /* :28-32 */
        const double z = (px * px) + (py * py);
        z_bar += z / n;
        Z[j] = z; Z[j + (size_t)n] = px; Z[j + (size_t)2 * n] = py; Z[j + (size_t)3 * n] = 1.0;
    }
    double Hinv[16];                                                /* :57-61 */
    for (int i = 0; i < 16; ++i) Hinv[i] = (i % 5 == 0) ? 1.0 : 0.0;
    Hinv[0 + 4 * 0] = 0.0; Hinv[0 + 4 * 3] = 0.5; Hinv[3 + 4 * 0] = 0.5; Hinv[3 + 4 * 3] = -2 * z_bar;
    double s[4], V[16], A[4];
    svd_n_by_4(Z, n, s, V);                                         /* :64-67 */
    if (s[3] < 1e-12) {                                             /* :79-81 */
        for (int i = 0; i < 4; ++i) A[i] = V[i + 4 * 3];
    } else {
        double Y[16], T[16], Qm[16], Ycopy[16];
        for (int i = 0; i < 4; ++i)                                 /* Y = V diag(s) V^T   :83 */
            for (int j = 0; j < 4; ++j) {
                double acc = 0.0;
                for (int k = 0; k < 4; ++k) acc += V[i + 4 * k] * s[k] * V[j + 4 * k];
                Y[i + 4 * j] = acc;
            }
        for (int i = 0; i < 4; ++i)                                 /* Q = Y Hinv Y         :84 */
            for (int j = 0; j < 4; ++j) {
                double acc = 0.0;
                for (int k = 0; k < 4; ++k) acc += Y[i + 4 * k] * Hinv[k + 4 * j];
                T[i + 4 * j] = acc;
            }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double acc = 0.0;
                for (int k = 0; k < 4; ++k) acc += T[i + 4 * k] * Y[k + 4 * j];
                Qm[i + 4 * j] = acc;
            }
        for (int i = 0; i < 4; ++i)                                 /* eig_sym reads one triangle: symmetrise */
            for (int j = i + 1; j < 4; ++j) { const double m = 0.5 * (Qm[i + 4 * j] + Qm[j + 4 * i]); Qm[i + 4 * j] = m; Qm[j + 4 * i] = m; }
        double w[4], E[16];
        eig_sym4(Qm, w, E);                                         /* :88 */
        int eig_index = 0;                                          /* :91-101 smallest positive eigenvalue */
        double eig_max = INT_MAX;
        for (int i = 0; i < 4; ++i)
            if (w[i] > 0 && w[i] < eig_max) { eig_index = i; eig_max = w[i]; }
        double Astar[4];
        for (int i = 0; i < 4; ++i) Astar[i] = E[i + 4 * eig_index];
        memcpy(Ycopy, Y, sizeof(Y));
        if (solve4(Ycopy, Astar, A)) return CF_DEGENERATE;          /* :103 */
    }
    const double a = -A[1] / (2 * A[0]);                            /* :107-110 */
    const double b = -A[2] / (2 * A[0]);
    const double R2 = ((A[1] * A[1]) + (A[2] * A[2]) - 4 * A[0] * A[3]) / (4 * (A[0] * A[0]));
    out[0] = a + x_hat;                                             /* :120-121 */
    out[1] = b + y_hat;
    out[2] = sqrt(R2);                                              /* tube_radius; the marker's scale is 2 * this (:124) */
    return CF_OK;
}

/* classifyCluster, circle_fit_library.cpp:208-250: standard deviation (degrees) of the inscribed angles < 10 */
int orc_classify_cluster(const double* xs, const double* ys, int n, double* std_dev_out)
{
    const double PI = 3.14159265358979323846;
    if (n < 3) {
        /* no inscribed angle: angles.size() == 0 (unsigned), the loops of :217,:231,:237 do not run and
         * std_dev = sqrt(0.0 / 0) = NaN, NaN < 10 is false (:243-249).  1- and 2-point clusters do reach this
         * function: the erase loop of clusterPoints exempts the cluster that follows a discarded one. */
        if (std_dev_out) *std_dev_out = NAN;
        return 0;
    }
    const double p2x = xs[0], p2y = ys[0], p3x = xs[n - 1], p3y = ys[n - 1];
    const int cnt = n - 2;
    double mean = 0.0;
    for (int i = 1; i < n - 1; ++i) {
        const double num = p2y * (xs[i] - p3x) + ys[i] * (p3x - p2x) + p3y * (p2x - xs[i]);
        const double den = (p2x - xs[i]) * (xs[i] - p3x) + (p2y - ys[i]) * (ys[i] - p3y);
        mean += (((double)180 / PI) * atan2(num, den)) / cnt;
    }
    double sd = 0.0;
    for (int i = 1; i < n - 1; ++i) {
        const double num = p2y * (xs[i] - p3x) + ys[i] * (p3x - p2x) + p3y * (p2x - xs[i]);
        const double den = (p2x - xs[i]) * (xs[i] - p3x) + (p2y - ys[i]) * (ys[i] - p3y);
        const double ang = ((double)180 / PI) * atan2(num, den);
        sd += (ang - mean) * (ang - mean);
    }
    sd = sqrt(sd / cnt);
    if (std_dev_out) *std_dev_out = sd;
    return sd < 10 ? 1 : 0;
}

/* clusterPoints, circle_fit_library.cpp:136-206, for a 360-ray scan.  Output: cluster id per emitted point
 * (point_cluster[k]), its coordinates, and the number of clusters AFTER the reference's erase loop (:197-204, which
 * skips the element following each erased one).  Returns the number of emitted points; *n_clusters_out clusters,
 * cluster_of[] compacted to the surviving ones, -1 for points of discarded clusters. */
int orc_cluster_points(const float* ranges, double minRange, double maxRange, double* px, double* py, int* cluster_of,
                       int* n_clusters_out)
{
    const double PI = 3.14159265358979323846;
    const double threshold = 0.04;
    int curr = 0, npts = 0, nclusters = 0, cur_cluster_open = 0;
    int sizes[361];
    memset(sizes, 0, sizeof(sizes));
    /* points of the open (not yet pushed) cluster carry id nclusters; wrap-around points go to cluster 0 */
    while (curr < 360) {
        if ((ranges[curr] > maxRange) || (ranges[curr] < minRange)) { curr += 1; continue; }
        const int next = (curr + 1) % 360;
        const double cd = ranges[curr], nd = ranges[next];
        const double x = ranges[curr] * cos((PI / (double)180) * curr);
        const double y = ranges[curr] * sin((PI / (double)180) * curr);
        if (fabs(cd - nd) < threshold) {
            if (next < curr) {
                if (nclusters == 0) { /* clusters[0] does not exist: undefined behaviour in the reference; drop */ }
                else { px[npts] = x; py[npts] = y; cluster_of[npts] = 0; sizes[0]++; npts++; }
            } else {
                px[npts] = x; py[npts] = y; cluster_of[npts] = nclusters; sizes[nclusters]++; npts++;
                cur_cluster_open = 1;
                curr += 1;
            }
        } else {
            px[npts] = x; py[npts] = y; cluster_of[npts] = nclusters; sizes[nclusters]++; npts++;
            nclusters++;
            cur_cluster_open = 0;
            curr += 1;
        }
        if (next < curr) break;
    }
    if (cur_cluster_open) {           /* the open cluster is never pushed (:146-195): its points are lost */
        for (int k = 0; k < npts; ++k) if (cluster_of[k] == nclusters) cluster_of[k] = -1;
    }
    /* erase loop :197-204 -- erasing element i shifts the rest down and the loop still increments i */
    int alive[361], map[361];
    int cnt = nclusters;
    for (int i = 0; i < cnt; ++i) alive[i] = i;
    for (int i = 0; i < cnt; ++i)
        if (sizes[alive[i]] < 3) {
            for (int j = i; j < cnt - 1; ++j) alive[j] = alive[j + 1];
            cnt--;
        }
    for (int i = 0; i < 361; ++i) map[i] = -1;
    for (int i = 0; i < cnt; ++i) map[alive[i]] = i;
    for (int k = 0; k < npts; ++k) if (cluster_of[k] >= 0) cluster_of[k] = map[cluster_of[k]];
    *n_clusters_out = cnt;
    return npts;
}
