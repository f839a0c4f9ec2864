// circle_fit.hip -- batched landmark extraction for gfx950: SURVEY.md section 8f row f3.
//
// Reference: nuslam/src/circle_fit_library.cpp -- circleFit (:15-134, the "hyper-accurate" algebraic circle fit) and
// classifyCluster (:208-250).  The reference fits one lidar cluster at a time on the CPU through Armadillo's
// svd / eig_sym / solve; here every cluster of a batch (all scans of all Monte-Carlo filters) is fitted by one wave:
// the n x 4 data matrix Z lives in a per-cluster scratch strip in HBM (it is L2-resident: n <= 360 points), the
// one-sided Jacobi SVD of Z runs with wave-wide dot products, and the 4 x 4 tail (Y = V diag(s) V^T, Q = Y H^-1 Y,
// symmetric eigenproblem, 4 x 4 solve) is evaluated redundantly by every lane.  Latency / compute bound, not HBM.
#include "../../include/nuslam_hip.h"

#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <vector>

namespace {

__device__ inline double wave_sum(double x)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}

// cyclic Jacobi for a symmetric 4x4 (column-major, destroyed): eigenvalues w, eigenvectors in the columns of E
__device__ inline void eig_sym4(double* A, double w[4], double E[16])
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
                for (int k = 0; k < 4; ++k) {
                    const double akp = A[k + 4 * p], akq = A[k + 4 * q];
                    A[k + 4 * p] = c * akp - sn * akq;
                    A[k + 4 * q] = sn * akp + c * akq;
                }
                for (int k = 0; k < 4; ++k) {
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

__device__ inline int solve4(const double* M, const double b[4], double x[4])
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

// one wave per cluster; 4 clusters per 256-thread workgroup
__global__ __launch_bounds__(256) void k_circle_fit(int n_clusters, const int* __restrict__ offsets,
                                                    const double* __restrict__ xs, const double* __restrict__ ys,
                                                    double* __restrict__ work, double* __restrict__ out, int* __restrict__ status)
{
    const int lane = threadIdx.x & 63;
    const int cl = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cl >= n_clusters) return;
    const int p0 = offsets[cl], n = offsets[cl + 1] - p0;
    if (n < 4) {                                                    // circle_fit_library.cpp:73-77
        if (lane == 0) { status[cl] = 1; out[3 * cl] = out[3 * cl + 1] = out[3 * cl + 2] = 0.0; }
        return;
    }
    const double* X = xs + p0;
    const double* Y_ = ys + p0;
    double* Z = work + (size_t)4 * p0;                              // n x 4, column-major, leading dimension n
    // centroid (:21-25), shifted data and z_bar (:28-47)
    double sx = 0, sy = 0;
    for (int i = lane; i < n; i += 64) { sx += X[i] / n; sy += Y_[i] / n; }
    const double x_hat = wave_sum(sx), y_hat = wave_sum(sy);
    double sz = 0;
    for (int i = lane; i < n; i += 64) {
        const double px = X[i] - x_hat, py = Y_[i] - y_hat;
        const double z = (px * px) + (py * py);
        sz += z / n;
        Z[i] = z; Z[i + (size_t)n] = px; Z[i + (size_t)2 * n] = py; Z[i + (size_t)3 * n] = 1.0;
    }
    const double z_bar = wave_sum(sz);
    // one-sided Jacobi SVD of Z (svd(U, s, V, Z), :64-67); each lane re-reads only what it wrote itself
    double V[16];
    for (int i = 0; i < 16; ++i) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                double* zp = Z + (size_t)p * n;
                double* zq = Z + (size_t)q * n;
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = lane; i < n; i += 64) {
                    const double a = zp[i], b = zq[i];
                    alpha += a * a; beta += b * b; gamma += a * b;
                }
                alpha = wave_sum(alpha); beta = wave_sum(beta); gamma = wave_sum(gamma);
                if (gamma == 0.0) continue;
                const double rel = fabs(gamma) / sqrt(alpha * beta);
                if (rel > off) off = rel;
                if (rel < 1e-16) continue;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = lane; i < n; i += 64) {
                    const double a = zp[i], b = zq[i];
                    zp[i] = c * a - sn * b;
                    zq[i] = sn * a + c * b;
                }
                for (int i = 0; i < 4; ++i) {
                    const double a = V[i + 4 * p], b = V[i + 4 * q];
                    V[i + 4 * p] = c * a - sn * b;
                    V[i + 4 * q] = sn * a + c * b;
                }
            }
        if (off < 1e-15) break;
    }
    double s[4];
    for (int k = 0; k < 4; ++k) {
        double a = 0;
        const double* z = Z + (size_t)k * n;
        for (int i = lane; i < n; i += 64) a += z[i] * z[i];
        s[k] = sqrt(wave_sum(a));
    }
    for (int a = 0; a < 3; ++a)
        for (int b = a + 1; b < 4; ++b)
            if (s[b] > s[a]) {
                double t = s[a]; s[a] = s[b]; s[b] = t;
                for (int i = 0; i < 4; ++i) { t = V[i + 4 * a]; V[i + 4 * a] = V[i + 4 * b]; V[i + 4 * b] = t; }
            }
    // the 4x4 tail, identical on every lane
    double A[4];
    int st = 0;
    if (s[3] < 1e-12) {                                             // :79-81
        for (int i = 0; i < 4; ++i) A[i] = V[i + 4 * 3];
    } else {
        double Hinv[16], Ym[16], T[16], Qm[16];
        for (int i = 0; i < 16; ++i) Hinv[i] = (i % 5 == 0) ? 1.0 : 0.0;    // :57-61
        Hinv[0] = 0.0; Hinv[0 + 4 * 3] = 0.5; Hinv[3 + 4 * 0] = 0.5; Hinv[3 + 4 * 3] = -2 * z_bar;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double acc = 0.0;
                for (int k = 0; k < 4; ++k) acc += V[i + 4 * k] * s[k] * V[j + 4 * k];
                Ym[i + 4 * j] = acc;                                // Y = V diag(s) V^T, :83
            }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double acc = 0.0;
                for (int k = 0; k < 4; ++k) acc += Ym[i + 4 * k] * Hinv[k + 4 * j];
                T[i + 4 * j] = acc;
            }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double acc = 0.0;
                for (int k = 0; k < 4; ++k) acc += T[i + 4 * k] * Ym[k + 4 * j];
                Qm[i + 4 * j] = acc;                                // Q = Y Hinv Y, :84
            }
        for (int i = 0; i < 4; ++i)
            for (int j = i + 1; j < 4; ++j) { const double m = 0.5 * (Qm[i + 4 * j] + Qm[j + 4 * i]); Qm[i + 4 * j] = m; Qm[j + 4 * i] = m; }
        double w[4], E[16];
        eig_sym4(Qm, w, E);                                         // :88
        int eig_index = 0;                                          // :91-101
        double eig_max = INT_MAX;
        for (int i = 0; i < 4; ++i)
            if (w[i] > 0 && w[i] < eig_max) { eig_index = i; eig_max = w[i]; }
        double Astar[4];
        for (int i = 0; i < 4; ++i) Astar[i] = E[i + 4 * eig_index];
        if (solve4(Ym, Astar, A)) st = 2;                           // :103
    }
    if (lane == 0) {
        if (st == 0) {
            const double a = -A[1] / (2 * A[0]);                    // :107-110
            const double b = -A[2] / (2 * A[0]);
            const double R2 = ((A[1] * A[1]) + (A[2] * A[2]) - 4 * A[0] * A[3]) / (4 * (A[0] * A[0]));
            out[3 * cl] = a + x_hat; out[3 * cl + 1] = b + y_hat; out[3 * cl + 2] = sqrt(R2);
        } else { out[3 * cl] = out[3 * cl + 1] = out[3 * cl + 2] = 0.0; }
        status[cl] = st;
    }
}

// classifyCluster, :208-250 -- one wave per cluster
__global__ __launch_bounds__(256) void k_classify(int n_clusters, const int* __restrict__ offsets,
                                                  const double* __restrict__ xs, const double* __restrict__ ys,
                                                  int* __restrict__ is_circle, double* __restrict__ std_dev)
{
    const double PI = 3.14159265358979323846;
    const int lane = threadIdx.x & 63;
    const int cl = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cl >= n_clusters) return;
    const int p0 = offsets[cl], n = offsets[cl + 1] - p0;
    if (n < 3) { if (lane == 0) { is_circle[cl] = 0; std_dev[cl] = 0.0; } return; }
    const double* X = xs + p0;
    const double* Yv = ys + p0;
    const double p2x = X[0], p2y = Yv[0], p3x = X[n - 1], p3y = Yv[n - 1];
    const int cnt = n - 2;
    double m = 0;
    for (int i = 1 + lane; i < n - 1; i += 64) {
        const double num = p2y * (X[i] - p3x) + Yv[i] * (p3x - p2x) + p3y * (p2x - X[i]);
        const double den = (p2x - X[i]) * (X[i] - p3x) + (p2y - Yv[i]) * (Yv[i] - p3y);
        m += (((double)180 / PI) * atan2(num, den)) / cnt;
    }
    const double mean = wave_sum(m);
    double sd = 0;
    for (int i = 1 + lane; i < n - 1; i += 64) {
        const double num = p2y * (X[i] - p3x) + Yv[i] * (p3x - p2x) + p3y * (p2x - X[i]);
        const double den = (p2x - X[i]) * (X[i] - p3x) + (p2y - Yv[i]) * (Yv[i] - p3y);
        const double ang = ((double)180 / PI) * atan2(num, den);
        sd += (ang - mean) * (ang - mean);
    }
    sd = sqrt(wave_sum(sd) / cnt);
    if (lane == 0) { std_dev[cl] = sd; is_circle[cl] = sd < 10 ? 1 : 0; }
}

#define CHK(expr) do { if ((expr) != hipSuccess) { rc = NUSLAM_E_HIP; goto done; } } while (0)

} // namespace

extern "C" {

int nuslam_circle_fit_batch(int n_clusters, const int* offsets, const double* xs, const double* ys, double* centre_x,
                            double* centre_y, double* radius, int* status, int* is_circle, double* angle_std_dev,
                            int device, double* kernel_ms)
{
    if (n_clusters < 0 || !offsets || !xs || !ys || !centre_x || !centre_y || !radius || !status) return NUSLAM_E_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); return NUSLAM_E_NODEV; }
    if (device < 0 || device >= count) return NUSLAM_E_ARG;
    if (n_clusters == 0) return NUSLAM_OK;
    const int npts = offsets[n_clusters];
    for (int c = 0; c < n_clusters; ++c)
        if (offsets[c + 1] < offsets[c]) return NUSLAM_E_ARG;
    int rc = NUSLAM_OK;
    int *d_off = nullptr, *d_status = nullptr, *d_circ = nullptr;
    double *d_x = nullptr, *d_y = nullptr, *d_work = nullptr, *d_out = nullptr, *d_sd = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::vector<double> h_out((size_t)3 * n_clusters);
    const size_t pb = sizeof(double) * (size_t)(npts > 0 ? npts : 1);
    CHK(hipSetDevice(device));
    CHK(hipMalloc(&d_off, sizeof(int) * (n_clusters + 1)));
    CHK(hipMalloc(&d_x, pb)); CHK(hipMalloc(&d_y, pb)); CHK(hipMalloc(&d_work, 4 * pb));
    CHK(hipMalloc(&d_out, sizeof(double) * 3 * n_clusters));
    CHK(hipMalloc(&d_status, sizeof(int) * n_clusters));
    CHK(hipMalloc(&d_circ, sizeof(int) * n_clusters));
    CHK(hipMalloc(&d_sd, sizeof(double) * n_clusters));
    CHK(hipMemcpy(d_off, offsets, sizeof(int) * (n_clusters + 1), hipMemcpyHostToDevice));
    if (npts > 0) {
        CHK(hipMemcpy(d_x, xs, sizeof(double) * npts, hipMemcpyHostToDevice));
        CHK(hipMemcpy(d_y, ys, sizeof(double) * npts, hipMemcpyHostToDevice));
    }
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL(k_circle_fit, dim3((n_clusters + 3) / 4), dim3(256), 0, nullptr, n_clusters, (const int*)d_off,
                       (const double*)d_x, (const double*)d_y, d_work, d_out, d_status);
    hipLaunchKernelGGL(k_classify, dim3((n_clusters + 3) / 4), dim3(256), 0, nullptr, n_clusters, (const int*)d_off,
                       (const double*)d_x, (const double*)d_y, d_circ, d_sd);
    CHK(hipGetLastError());
    CHK(hipEventRecord(e1, nullptr));
    CHK(hipEventSynchronize(e1));
    if (kernel_ms) { float ms = 0.f; CHK(hipEventElapsedTime(&ms, e0, e1)); *kernel_ms = ms; }
    CHK(hipMemcpy(h_out.data(), d_out, sizeof(double) * 3 * n_clusters, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(status, d_status, sizeof(int) * n_clusters, hipMemcpyDeviceToHost));
    if (is_circle) CHK(hipMemcpy(is_circle, d_circ, sizeof(int) * n_clusters, hipMemcpyDeviceToHost));
    if (angle_std_dev) CHK(hipMemcpy(angle_std_dev, d_sd, sizeof(double) * n_clusters, hipMemcpyDeviceToHost));
    for (int c = 0; c < n_clusters; ++c) { centre_x[c] = h_out[3 * c]; centre_y[c] = h_out[3 * c + 1]; radius[c] = h_out[3 * c + 2]; }
done:
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    void* ptrs[] = { d_off, d_x, d_y, d_work, d_out, d_status, d_circ, d_sd };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    return rc;
}

} // extern "C"
