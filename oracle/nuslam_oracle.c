/*
 * nuslam_oracle.c -- CPU ORACLE (test infrastructure only; see nuslam_oracle.h for the rules,
 * the parity-pinning statement and the two evaluation modes).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared   (oracle/Makefile)
 * -ffp-contract=off: every multiply and add below is rounded separately, in the order written;
 * the HIP kernels are compiled the same way so the two sides do the same arithmetic.
 *
 * All matrices are column-major (Armadillo's layout): X(i,j) = X[i + j*ld].
 */
#include "nuslam_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

struct orc_ekf {
    int n;        /* landmarks                    slam_library.hpp:31 */
    int len;      /* 3 + 2n                       slam_library.hpp:30 */
    int seen;     /* seen_landmarks               slam_library.hpp:32 */
    int mode;
    double* state; /* len */
    double* P;     /* len x len */
    double Q[9];   /* 3x3 col-major */
    double R[4];   /* 2x2 col-major */
};

static int g_threads = 1; /* scalar by default; the cpu_baseline leg and large-N checks raise it explicitly */

void orc_set_threads(int nthreads) { g_threads = nthreads > 0 ? nthreads : 1; }
int orc_get_threads(void)
{
#ifdef _OPENMP
    return g_threads;
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ rigid2d */

/* rigid2d/src/rigid2d.cpp:9-13 -- atan2(sin, cos), not fmod. */
double orc_normalize_angle(double rad) { return atan2(sin(rad), cos(rad)); }

/* Transform2D is {cos, sin, x, y} (rigid2d.hpp:171-175). */
static void tf_translation(double x, double y, double T[4]) /* rigid2d.cpp:154-160 */
{ T[0] = 1; T[1] = 0; T[2] = x; T[3] = y; }
static void tf_rotation(double rad, double T[4])            /* rigid2d.cpp:162-168 */
{ T[0] = cos(rad); T[1] = sin(rad); T[2] = 0; T[3] = 0; }
static void tf_inv(const double T[4], double out[4])        /* rigid2d.cpp:187-196 */
{
    const double c = T[0], s = T[1], x = T[2], y = T[3];
    out[0] = c;
    out[1] = -s;
    out[2] = (-x * c) + (-y * s);
    out[3] = (x * s) + (-y * c);
}
static void tf_mul(const double L[4], const double Rr[4], double out[4]) /* rigid2d.cpp:198-209 (lhs *= rhs) */
{
    const double c = L[0], s = L[1], x = L[2], y = L[3];
    const double m00 = (c * Rr[0]) - (s * Rr[1]);
    const double m10 = (s * Rr[0]) + (c * Rr[1]);
    const double m02 = (c * Rr[2]) - (s * Rr[3]) + x;
    const double m12 = (s * Rr[2]) + (c * Rr[3]) + y;
    out[0] = m00; out[1] = m10; out[2] = m02; out[3] = m12;
}

void orc_tf_make(double x, double y, double rad, double T[4])  /* Transform2D(trans, radians), rigid2d.cpp:170-176 */
{ T[0] = cos(rad); T[1] = sin(rad); T[2] = x; T[3] = y; }
void orc_tf_inv(const double T[4], double out[4]) { tf_inv(T, out); }
void orc_tf_mul(const double L[4], const double Rr[4], double out[4]) { tf_mul(L, Rr, out); }
void orc_tf_point(const double T[4], double x, double y, double out[2]) /* rigid2d.cpp:178-185 */
{
    out[0] = (x * T[0]) + (y * (-T[1])) + T[2];
    out[1] = (x * T[1]) + (y * T[0]) + T[3];
}

/* broadcast_map2odom_tf, nuslam/src/slam.cpp:175-210 */
void orc_map_to_odom(const double odom[3], const double state[3], double out[3])
{
    double T_ob[4], T_mb[4], T_bo[4], T_mo[4];
    orc_tf_make(odom[0], odom[1], odom[2], T_ob);          /* :179-182 */
    orc_tf_make(state[1], state[2], state[0], T_mb);       /* :186-188 */
    tf_inv(T_ob, T_bo);
    tf_mul(T_mb, T_bo, T_mo);                              /* :191 */
    out[0] = T_mo[2];
    out[1] = T_mo[3];
    out[2] = orc_normalize_angle(asin(T_mo[1]));           /* :194 */
}

/* Transform2D::operator()(Twist2D) -- the adjoint, rigid2d.cpp:254-261.  tw = {dth, dx, dy}. */
void orc_transform_twist(const double T[4], const double tw[3], double out[3])
{
    const double c = T[0], s = T[1], x = T[2], y = T[3];
    out[0] = tw[0];
    out[1] = (y * tw[0]) + (c * tw[1]) - (s * tw[2]);
    out[2] = -(x * tw[0]) + (s * tw[1]) + (c * tw[2]);
}

/* integrateTwist, rigid2d.cpp:294-328. */
void orc_integrate_twist(const double tw[3], double T_out[4])
{
    if (tw[0] == 0) {
        tf_translation(tw[1], tw[2], T_out);
        return;
    }
    double T_sb[4], T_ss[4], T_bs[4], tmp[4];
    tf_translation(tw[2] / tw[0], -(tw[1] / tw[0]), T_sb);   /* :310-312 */
    tf_rotation(tw[0], T_ss);                                /* :317 */
    tf_inv(T_sb, T_bs);                                      /* :323 */
    tf_mul(T_bs, T_ss, tmp);                                 /* :325, operator* is left-associative */
    tf_mul(tmp, T_sb, T_out);
}

/* dd = {wheelBase, wheelRad, x, y, th, thL, thR}  (diff_drive.hpp members, ctor diff_drive.cpp:20-29) */
void orc_dd_convert_twist(const double dd[7], const double tw[3], double u_out[2]) /* diff_drive.cpp:66-78 */
{
    const double d = dd[0] / 2;
    const double r = dd[1];
    const double omg = tw[0];
    const double vbx = tw[1];
    u_out[0] = (-(d / r) * omg) + (vbx / r);
    u_out[1] = ((d / r) * omg) + (vbx / r);
}

void orc_dd_get_twist(const double dd[7], double thLnew, double thRnew, double tw_out[3]) /* :80-110 */
{
    const double dUL = thLnew - dd[5];
    const double dUR = thRnew - dd[6];
    tw_out[0] = (dd[1] / dd[0]) * (dUR - dUL);
    tw_out[1] = (dd[1] / 2) * (dUL + dUR);
    tw_out[2] = 0.0;
}

void orc_dd_step(double dd[7], double thLnew, double thRnew) /* DiffDrive::operator(), :111-146 */
{
    double twb[3], Tbb[4], dqb[3], adj[4], dq[3];
    orc_dd_get_twist(dd, thLnew, thRnew, twb);   /* same arithmetic as :114-121 */
    orc_integrate_twist(twb, Tbb);               /* :124 */
    dqb[0] = atan(Tbb[1] / Tbb[0]);              /* :129 */
    dqb[1] = Tbb[2];
    dqb[2] = Tbb[3];
    tf_rotation(dd[4], adj);                     /* :134 */
    orc_transform_twist(adj, dqb, dq);           /* :137 */
    dd[4] += dq[0];                              /* :140-144; th is never normalised */
    dd[2] += dq[1];
    dd[3] += dq[2];
    dd[5] = thLnew;
    dd[6] = thRnew;
}

/* ------------------------------------------------------------------ dense kernels */

/* C(m x n) = A(m x kk) * op(B), op(B) = B (kk x n) or B^T (B is n x kk).  Every C(i,j) is the sum over
 * k = 0..kk-1 in ascending order starting from 0.0, multiply and add rounded separately. */
static void gemm(int m, int n, int kk, const double* A, int lda, const double* B, int ldb, int transB,
                 double* C, int ldc)
{
    const int nth = orc_get_threads();
#pragma omp parallel for schedule(static) num_threads(nth)
    for (int jb = 0; jb < n; jb += 4) {
        const int nb = (n - jb) < 4 ? (n - jb) : 4;
        double* c0 = C + (size_t)(jb + 0) * ldc;
        double* c1 = C + (size_t)(jb + (nb > 1 ? 1 : 0)) * ldc;
        double* c2 = C + (size_t)(jb + (nb > 2 ? 2 : 0)) * ldc;
        double* c3 = C + (size_t)(jb + (nb > 3 ? 3 : 0)) * ldc;
        for (int c = 0; c < nb; ++c)
            memset(C + (size_t)(jb + c) * ldc, 0, sizeof(double) * (size_t)m);
        for (int k = 0; k < kk; ++k) {
            const double* a = A + (size_t)k * lda;
            double b[4];
            for (int c = 0; c < 4; ++c) {
                const int j = jb + (c < nb ? c : 0);
                b[c] = transB ? B[j + (size_t)k * ldb] : B[k + (size_t)j * ldb];
            }
            if (nb == 4) {
                for (int i = 0; i < m; ++i) {
                    const double ai = a[i];
                    c0[i] = fma(ai, b[0], c0[i]);
                    c1[i] = fma(ai, b[1], c1[i]);
                    c2[i] = fma(ai, b[2], c2[i]);
                    c3[i] = fma(ai, b[3], c3[i]);
                }
            } else {
                for (int c = 0; c < nb; ++c) {
                    double* cc = C + (size_t)(jb + c) * ldc;
                    for (int i = 0; i < m; ++i) cc[i] = fma(a[i], b[c], cc[i]);
                }
            }
        }
    }
}

/* Armadillo 9.800 auxlib::inv_noalias_tinymat, N = 2 (what inv()/.i() take at slam_library.cpp:231,270):
 * closed form when |det| >= epsilon, otherwise the general LAPACK route (restated here as 2x2 LU with
 * partial pivoting, getrf + getri); exactly singular -> inv() throws std::runtime_error. */
static int inv2(const double X[4], double out[4])
{
    const double a = X[0], c = X[1], b = X[2], d = X[3]; /* col-major: X(0,0) X(1,0) X(0,1) X(1,1) */
    const double det = (a * d) - (b * c);
    if (fabs(det) >= DBL_EPSILON) {
        out[0] = d / det;
        out[2] = -b / det;
        out[1] = -c / det;
        out[3] = a / det;
        return ORC_OK;
    }
    /* LU, partial pivoting */
    double r0[2] = { a, b }, r1[2] = { c, d };
    int swap = 0;
    if (fabs(r1[0]) > fabs(r0[0])) { double t0 = r0[0], t1 = r0[1]; r0[0] = r1[0]; r0[1] = r1[1]; r1[0] = t0; r1[1] = t1; swap = 1; }
    if (r0[0] == 0.0) return ORC_E_SINGULAR;
    const double l = r1[0] / r0[0];
    const double u11 = r1[1] - l * r0[1];
    if (u11 == 0.0 || isnan(u11)) return ORC_E_SINGULAR;
    /* solve (PA) Y = I column by column, then undo the row swap on the columns of the inverse */
    double inv[4];
    for (int col = 0; col < 2; ++col) {
        double e0 = col == 0 ? 1.0 : 0.0, e1 = col == 1 ? 1.0 : 0.0;
        const double y1 = e1 - l * e0;
        const double x1 = y1 / u11;
        const double x0 = (e0 - r0[1] * x1) / r0[0];
        inv[0 + 2 * col] = x0;
        inv[1 + 2 * col] = x1;
    }
    if (swap) { /* A^-1 = (PA)^-1 P : swap columns */
        out[0] = inv[2]; out[1] = inv[3]; out[2] = inv[0]; out[3] = inv[1];
    } else {
        memcpy(out, inv, sizeof(inv));
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ slam_library */

void orc_cartesian2polar(double x, double y, double out[2]) /* slam_library.cpp:16-22 */
{
    /* pow(x,2) + pow(y,2): written as exact products (what gcc folds pow(.,2) to). */
    out[0] = sqrt((x * x) + (y * y));
    out[1] = orc_normalize_angle(atan2(y, x));
}

orc_ekf* orc_create(const double robot[3], const double* map, int n_landmarks, const double Q[9],
                    const double R[4]) /* ctor slam_library.cpp:39-63 + initCov :24-33 */
{
    if (n_landmarks < 0) return NULL;
    orc_ekf* e = (orc_ekf*)calloc(1, sizeof(orc_ekf));
    if (!e) return NULL;
    e->n = n_landmarks;
    e->len = 3 + 2 * n_landmarks;
    e->seen = 0;
    e->mode = ORC_DENSE;
    e->state = (double*)malloc(sizeof(double) * (size_t)e->len);
    e->P = (double*)calloc((size_t)e->len * (size_t)e->len, sizeof(double));
    if (!e->state || !e->P) { orc_destroy(e); return NULL; }
    memcpy(e->Q, Q, sizeof(e->Q));
    memcpy(e->R, R, sizeof(e->R));
    e->state[0] = robot[0];
    e->state[1] = robot[1];
    e->state[2] = robot[2];
    for (int i = 3; i < e->len; ++i) e->state[i] = map[i - 3];
    for (int i = 3; i < e->len; ++i) e->P[i + (size_t)i * e->len] = INT_MAX; /* :28-31 */
    return e;
}

void orc_destroy(orc_ekf* e)
{
    if (!e) return;
    free(e->state);
    free(e->P);
    free(e);
}

void orc_set_mode(orc_ekf* e, int mode) { e->mode = mode; }
int orc_len(const orc_ekf* e) { return e->len; }
int orc_n(const orc_ekf* e) { return e->n; }
int orc_seen(const orc_ekf* e) { return e->seen; }
void orc_set_seen(orc_ekf* e, int seen) { e->seen = seen; }
double* orc_state(orc_ekf* e) { return e->state; }
double* orc_cov(orc_ekf* e) { return e->P; }

/* predictEstimate, slam_library.cpp:71-94 */
static void predict_estimate(orc_ekf* e, double dth, double dx)
{
    double dq_th, dq_x, dq_y;
    const double theta = e->state[0];
    if (dth == 0.0) {
        dq_th = 0.0;
        dq_x = dx * cos(theta);
        dq_y = dx * sin(theta);
    } else {
        dq_th = dth;
        dq_x = -(dx / dth) * sin(theta) + (dx / dth) * sin(theta + dth);
        dq_y = (dx / dth) * cos(theta) - (dx / dth) * cos(theta + dth);
    }
    e->state[0] += dq_th;
    e->state[1] += dq_x;
    e->state[2] += dq_y;
}

/* The two non-zeros of B in getA, slam_library.cpp:127-148 -- evaluated at the ALREADY ADVANCED heading. */
static void motion_jacobian(const orc_ekf* e, double dth, double dx, double* a1, double* a2)
{
    const double theta = e->state[0];
    if (dth == 0) {
        *a1 = -dx * sin(theta);
        *a2 = dx * cos(theta);
    } else {
        *a1 = -(dx / dth) * cos(theta) + (dx / dth) * cos(theta + dth);
        *a2 = -(dx / dth) * sin(theta) + (dx / dth) * sin(theta + dth);
    }
}

/* P <- F P F^T + Qbar, dense, with the reference's five L x L temporaries (slam_library.cpp:96-125). */
static int propagate_dense(orc_ekf* e, const double* F)
{
    const int L = e->len;
    const size_t LL = (size_t)L * (size_t)L;
    double* Qbar = (double*)calloc(LL, sizeof(double));      /* expanded_process_noise :110-125 */
    double* T = (double*)malloc(sizeof(double) * LL);
    double* U = (double*)malloc(sizeof(double) * LL);
    if (!Qbar || !T || !U) { free(Qbar); free(T); free(U); return ORC_E_ARG; }
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) Qbar[i + (size_t)j * L] = e->Q[i + 3 * j];
    gemm(L, L, L, F, L, e->P, L, 0, T, L);                   /* A * covariance            :104 */
    gemm(L, L, L, T, L, F, L, 1, U, L);                      /* (A * covariance) * A.t()       */
    const int nth = orc_get_threads();
#pragma omp parallel for schedule(static) num_threads(nth)
    for (long long idx = 0; idx < (long long)LL; ++idx) e->P[idx] = U[idx] + Qbar[idx]; /* + Q_bar */
    free(Qbar); free(T); free(U);
    return ORC_OK;
}

int orc_predict_dense(orc_ekf* e, const double* F) { return propagate_dense(e, F); }

/* propagateUncertainty, slam_library.cpp:96-108 */
static void propagate_uncertainty(orc_ekf* e, double dth, double dx)
{
    const int L = e->len;
    double a1, a2;
    motion_jacobian(e, dth, dx, &a1, &a2);
    if (e->mode == ORC_DENSE) {
        double* A = (double*)calloc((size_t)L * (size_t)L, sizeof(double)); /* getA: I + B */
        for (int i = 0; i < L; ++i) A[i + (size_t)i * L] = 1.0 + 0.0;
        A[1 + 0 * (size_t)L] = 0.0 + a1;
        A[2 + 0 * (size_t)L] = 0.0 + a2;
        propagate_dense(e, A);
        free(A);
        return;
    }
    /* Structured: the same sums with the exactly-zero terms dropped, same ascending-k order.
     * T = A P changes rows 1,2;  U = T A^T changes columns 1,2;  + Qbar touches the 3x3 corner. */
    double* P = e->P;
    for (int j = 0; j < L; ++j) {
        double* col = P + (size_t)j * L;
        const double p0 = col[0];
        col[1] = a1 * p0 + 1.0 * col[1];    /* k=0 then k=1 */
        col[2] = a2 * p0 + 1.0 * col[2];    /* k=0 then k=2 */
    }
    double* c0 = P;
    double* c1 = P + (size_t)1 * L;
    double* c2 = P + (size_t)2 * L;
    for (int i = 0; i < L; ++i) {
        const double t0 = c0[i];
        c1[i] = t0 * a1 + c1[i] * 1.0;
        c2[i] = t0 * a2 + c2[i] * 1.0;
    }
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) P[i + (size_t)j * L] = P[i + (size_t)j * L] + e->Q[i + 3 * j];
}

void orc_predict(orc_ekf* e, double dth, double dx, double dy) /* slam_library.cpp:65-69 */
{
    (void)dy; /* Twist2D::dy is never read by the filter */
    predict_estimate(e, dth, dx);
    propagate_uncertainty(e, dth, dx);
}

void orc_measurement(const double* s, int j, double out[2]) /* computeTheoreticalMeasurement :150-160 */
{
    const double mx = s[3 + 2 * (j - 1)] - s[1];
    const double my = s[4 + 2 * (j - 1)] - s[2];
    orc_cartesian2polar(mx, my, out);
    out[1] = orc_normalize_angle(out[1] - s[0]);
}

/* The nine non-zeros of linearizedMeasurementModel (:162-186) in compact form:
 * Hc[r + 2*q] = H(r, set[q]),  set = {0, 1, 2, c, c+1},  c = 3 + 2(j-1). */
static void jacobian_compact(const double* s, int j, double Hc[10])
{
    const double dx = s[3 + 2 * (j - 1)] - s[1];
    const double dy = s[4 + 2 * (j - 1)] - s[2];
    const double d = (dx * dx) + (dy * dy);
    Hc[0] = 0.0;            Hc[1] = -1;               /* column 0 */
    Hc[2] = -dx / sqrt(d);  Hc[3] = dy / d;           /* column 1 */
    Hc[4] = -dy / sqrt(d);  Hc[5] = -dx / d;          /* column 2 */
    Hc[6] = dx / sqrt(d);   Hc[7] = -dy / d;          /* column c */
    Hc[8] = dy / sqrt(d);   Hc[9] = dx / d;           /* column c+1 */
}

void orc_jacobian(const double* s, int len, int j, double* H) /* :162-186, dense 2 x len */
{
    double Hc[10];
    jacobian_compact(s, j, Hc);
    memset(H, 0, sizeof(double) * 2 * (size_t)len);
    const int c = 3 + 2 * (j - 1);
    const int set[5] = { 0, 1, 2, c, c + 1 };
    for (int q = 0; q < 5; ++q) {
        H[0 + 2 * (size_t)set[q]] = Hc[0 + 2 * q];
        H[1 + 2 * (size_t)set[q]] = Hc[1 + 2 * q];
    }
}

/* psi = H P H^T + R for landmark j linearised at state s (used by update and associate).
 * Structured evaluation; the dense twin lives in update_dense/associate dense branch. */
static void innovation_cov_structured(const orc_ekf* e, const double Hc[10], int c, double S[4])
{
    const int L = e->len;
    const int set[5] = { 0, 1, 2, c, c + 1 };
    double HPs[2][5]; /* (H P)(r, set[q]) */
    for (int q = 0; q < 5; ++q)
        for (int r = 0; r < 2; ++r) {
            double acc = 0.0;
            for (int q2 = 0; q2 < 5; ++q2) acc = fma(Hc[r + 2 * q2], e->P[set[q2] + (size_t)set[q] * L], acc);
            HPs[r][q] = acc;
        }
    for (int s2 = 0; s2 < 2; ++s2)
        for (int r = 0; r < 2; ++r) {
            double acc = 0.0;
            for (int q = 0; q < 5; ++q) acc = fma(HPs[r][q], Hc[s2 + 2 * q], acc);
            S[r + 2 * s2] = acc + e->R[r + 2 * s2];
        }
}

int orc_init_landmark(orc_ekf* e, double r, double phi, int id) /* :255-261 */
{
    if (id < 1 || id > e->n) return ORC_E_BOUNDS;
    double* s = e->state;
    s[3 + 2 * (id - 1)] = s[1] + r * cos(phi + s[0]);
    s[4 + 2 * (id - 1)] = s[2] + r * sin(phi + s[0]);
    return ORC_OK;
}

int orc_update(orc_ekf* e, double r, double phi, int id) /* update, :263-282 (tw unused there) */
{
    if (id < 1 || id > e->n) return ORC_E_BOUNDS;
    const int L = e->len;
    const int c = 3 + 2 * (id - 1);
    const int set[5] = { 0, 1, 2, c, c + 1 };
    double* s = e->state;
    double* P = e->P;
    double zhat[2], Hc[10], S[4], Sinv[4];
    orc_measurement(s, id, zhat);          /* :265 */
    jacobian_compact(s, id, Hc);           /* :268 */

    double* K = (double*)malloc(sizeof(double) * 2 * (size_t)L);   /* L x 2 */
    if (!K) return ORC_E_ARG;

    if (e->mode == ORC_DENSE) {
        double* H = (double*)malloc(sizeof(double) * 2 * (size_t)L);
        double* HP = (double*)malloc(sizeof(double) * 2 * (size_t)L);
        double* PHt = (double*)malloc(sizeof(double) * 2 * (size_t)L);
        orc_jacobian(s, L, id, H);
        gemm(2, L, L, H, 2, P, L, 0, HP, 2);          /* H * covariance                 :270 */
        gemm(2, 2, L, HP, 2, H, 2, 1, S, 2);          /* (H * covariance) * trans(H)         */
        for (int q = 0; q < 4; ++q) S[q] = S[q] + e->R[q];  /* + sensor_noise                */
        int rc = inv2(S, Sinv);                       /* inv(...)                            */
        if (rc) { free(H); free(HP); free(PHt); free(K); return rc; }
        gemm(L, 2, L, P, L, H, 2, 1, PHt, L);         /* covariance * trans(H)               */
        gemm(L, 2, 2, PHt, L, Sinv, 2, 0, K, L);      /* (...) * inv(...)                    */
        /* state += K * z_diff ; normalise heading      :272-276 */
        const double dz[2] = { r - zhat[0], phi - zhat[1] };
        for (int i = 0; i < L; ++i) {
            double acc = 0.0;
            acc = fma(K[i], dz[0], acc);
            acc = fma(K[i + (size_t)L], dz[1], acc);
            s[i] += acc;
        }
        s[0] = orc_normalize_angle(s[0]);
        /* covariance = (eye - K*H) * covariance         :279 */
        double* M = (double*)malloc(sizeof(double) * (size_t)L * (size_t)L);
        double* Pn = (double*)malloc(sizeof(double) * (size_t)L * (size_t)L);
        gemm(L, L, 2, K, L, H, 2, 0, M, L);           /* K * H */
        const int nth = orc_get_threads();
#pragma omp parallel for schedule(static) num_threads(nth)
        for (int k = 0; k < L; ++k)
            for (int i = 0; i < L; ++i) M[i + (size_t)k * L] = (i == k ? 1.0 : 0.0) - M[i + (size_t)k * L];
        gemm(L, L, L, M, L, P, L, 0, Pn, L);
        memcpy(P, Pn, sizeof(double) * (size_t)L * (size_t)L);
        free(H); free(HP); free(PHt); free(M); free(Pn); free(K);
        return ORC_OK;
    }

    /* ---- structured: identical sums over the non-zero columns {0,1,2,c,c+1}, ascending ---- */
    innovation_cov_structured(e, Hc, c, S);
    int rc = inv2(S, Sinv);
    if (rc) { free(K); return rc; }
    for (int i = 0; i < L; ++i) {
        double ph[2];
        for (int rr = 0; rr < 2; ++rr) {             /* (P H^T)(i, rr) */
            double acc = 0.0;
            for (int q = 0; q < 5; ++q) acc = fma(P[i + (size_t)set[q] * L], Hc[rr + 2 * q], acc);
            ph[rr] = acc;
        }
        for (int s2 = 0; s2 < 2; ++s2) {             /* K(i, s2) = sum_r PHt(i,r) Sinv(r,s2) */
            double acc = 0.0;
            acc = fma(ph[0], Sinv[0 + 2 * s2], acc);
            acc = fma(ph[1], Sinv[1 + 2 * s2], acc);
            K[i + (size_t)s2 * L] = acc;
        }
    }
    const double dz[2] = { r - zhat[0], phi - zhat[1] };
    for (int i = 0; i < L; ++i) {
        double acc = 0.0;
        acc = fma(K[i], dz[0], acc);
        acc = fma(K[i + (size_t)L], dz[1], acc);
        s[i] += acc;
    }
    s[0] = orc_normalize_angle(s[0]);

    /* M(i, set[q]) = delta - (K(i,0) H(0,set[q]) + K(i,1) H(1,set[q])); all other columns of M are delta. */
    double* Mc = (double*)malloc(sizeof(double) * 5 * (size_t)L);
    double* Rw = (double*)malloc(sizeof(double) * 5 * (size_t)L);   /* old rows set[q] of P */
    for (int i = 0; i < L; ++i)
        for (int q = 0; q < 5; ++q) {
            double kh = 0.0;
            kh = fma(K[i], Hc[0 + 2 * q], kh);
            kh = fma(K[i + (size_t)L], Hc[1 + 2 * q], kh);
            Mc[q + 5 * (size_t)i] = (i == set[q] ? 1.0 : 0.0) - kh;
        }
    for (int j = 0; j < L; ++j)
        for (int q = 0; q < 5; ++q) Rw[q + 5 * (size_t)j] = P[set[q] + (size_t)j * L];
    const int nth = orc_get_threads();
#pragma omp parallel for schedule(static) num_threads(nth)
    for (int j = 0; j < L; ++j) {
        double* col = P + (size_t)j * L;
        const double* rw = Rw + 5 * (size_t)j;
        for (int i = 0; i < L; ++i) {
            const double* m = Mc + 5 * (size_t)i;
            const int in_set = (i < 3) || (i == c) || (i == c + 1);
            const double pij = col[i];
            /* ascending k over {0,1,2} U {i} U {c,c+1}; M(i,i) = 1 exactly when i is outside the set */
            double acc = 0.0;
            acc = fma(m[0], rw[0], acc);
            acc = fma(m[1], rw[1], acc);
            acc = fma(m[2], rw[2], acc);
            if (!in_set && i < c) acc = fma(1.0, pij, acc);
            acc = fma(m[3], rw[3], acc);
            acc = fma(m[4], rw[4], acc);
            if (!in_set && i > c + 1) acc = fma(1.0, pij, acc);
            col[i] = acc;
        }
    }
    free(Mc); free(Rw); free(K);
    return ORC_OK;
}

int orc_associate(orc_ekf* e, double r, double phi, int* id_out, double* d_out) /* :188-253 */
{
    const double min_threshold = 0.01;  /* :193 */
    const double max_threshold = 60;    /* :194 */
    const int L = e->len;
    if (d_out) for (int k = 0; k < e->seen; ++k) d_out[k] = NAN;
    if (e->seen == 0) {                 /* :197-200 */
        e->seen++;
        *id_out = e->seen;
        return ORC_OK;
    }
    /* :204-207 temp = state with hypothetical landmark seen+1; with a full map the write is out of
     * bounds and Armadillo throws std::logic_error. */
    if (3 + 2 * e->seen + 1 >= L) return ORC_E_BOUNDS;
    double* temp = (double*)malloc(sizeof(double) * (size_t)L);
    memcpy(temp, e->state, sizeof(double) * (size_t)L);
    temp[3 + 2 * e->seen] = temp[1] + r * cos(phi + temp[0]);
    temp[4 + 2 * e->seen] = temp[2] + r * sin(phi + temp[0]);

    double* H = NULL; double* HP = NULL;
    if (e->mode == ORC_DENSE) {
        H = (double*)malloc(sizeof(double) * 2 * (size_t)L);
        HP = (double*)malloc(sizeof(double) * 2 * (size_t)L);
    }
    int rc = ORC_OK;
    int result = 0;
    for (int k = 1; k < e->seen + 1; ++k) {      /* :209 */
        double psi[4], psi_inv[4], zhat[2], Hc[10];
        if (e->mode == ORC_DENSE) {
            orc_jacobian(temp, L, k, H);                       /* :212 */
            gemm(2, L, L, H, 2, e->P, L, 0, HP, 2);            /* :215 */
            gemm(2, 2, L, HP, 2, H, 2, 1, psi, 2);
            for (int q = 0; q < 4; ++q) psi[q] = psi[q] + e->R[q];
        } else {
            jacobian_compact(temp, k, Hc);
            innovation_cov_structured(e, Hc, 3 + 2 * (k - 1), psi);
        }
        orc_measurement(temp, k, zhat);                        /* :218 */
        const double dz[2] = { r - zhat[0], phi - zhat[1] };   /* :229, bearing difference NOT wrapped */
        rc = inv2(psi, psi_inv);
        if (rc) break;
        /* (dz^T * psi^-1) * dz                                   :231 */
        double w[2];
        for (int s2 = 0; s2 < 2; ++s2) {
            double acc = 0.0;
            acc = fma(dz[0], psi_inv[0 + 2 * s2], acc);
            acc = fma(dz[1], psi_inv[1 + 2 * s2], acc);
            w[s2] = acc;
        }
        double mahalanobis = 0.0;
        mahalanobis = fma(w[0], dz[0], mahalanobis);
        mahalanobis = fma(w[1], dz[1], mahalanobis);
        if (d_out) d_out[k - 1] = mahalanobis;
        if (mahalanobis < min_threshold) { result = k; break; }                                   /* :238-241 */
        else if ((mahalanobis > min_threshold) && (mahalanobis < max_threshold)) { result = -1; break; } /* :243-246 */
    }
    free(temp); free(H); free(HP);
    if (rc) return rc;
    if (result == 0) {                   /* :251-252 */
        e->seen++;
        result = e->seen;
    }
    *id_out = result;
    return ORC_OK;
}

int orc_tick(orc_ekf* e, double dd[7], double thL, double thR, const double* tw_override, int m,
             const double* mx, const double* my, const int* known_ids, int total_landmarks, int* ids_out)
{
    /* slam.cpp:250-251 -- `seen_landmarks` is cached at the top of the loop body, before predict. */
    const int seen_cached = e->seen;
    double tw[3];
    if (tw_override) {
        tw[0] = tw_override[0]; tw[1] = tw_override[1]; tw[2] = tw_override[2];
    } else {
        orc_dd_get_twist(dd, thL, thR, tw);        /* slam.cpp:264 */
        orc_dd_step(dd, thL, thR);                 /* slam.cpp:265 */
    }
    orc_predict(e, tw[0], tw[1], tw[2]);           /* slam.cpp:269 */
    if (ids_out) for (int i = 0; i < m; ++i) ids_out[i] = 0;
    for (int i = 0; i < m; ++i) {                  /* slam.cpp:279 */
        double z[2];
        int id;
        orc_cartesian2polar(mx[i], my[i], z);      /* :286 */
        if (known_ids) {
            /* Known association (benchmark configs 2-4): the id comes with the marker; `seen` tracks the
             * highest id met so far, which is what associateLandmark would have counted. */
            id = known_ids[i];
            if (id > e->seen) e->seen = id;
        } else {
            int rc = orc_associate(e, z[0], z[1], &id, NULL);   /* :291 */
            if (rc) return rc;
        }
        if (ids_out) ids_out[i] = id;
        if (id > seen_cached) {                    /* :295-297 */
            int rc = orc_init_landmark(e, z[0], z[1], id);
            if (rc) return rc;
        } else if (id < 0) {                       /* :298-300 */
            continue;
        } else if (id > total_landmarks) {         /* :301-316 */
            break;
        }
        int rc = orc_update(e, z[0], z[1], id);    /* :318 */
        if (rc) return rc;
    }
    return ORC_OK;
}
