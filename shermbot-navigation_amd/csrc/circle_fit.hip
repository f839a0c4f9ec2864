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
// cluster cl owns points [offsets[cl], ends ? ends[cl] : offsets[cl + 1])
__global__ __launch_bounds__(256) void k_circle_fit(int n_clusters, const int* __restrict__ offsets,
                                                    const int* __restrict__ ends,
                                                    const double* __restrict__ xs, const double* __restrict__ ys,
                                                    double* __restrict__ work, double* __restrict__ out, int* __restrict__ status)
{
    const int lane = threadIdx.x & 63;
    const int cl = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cl >= n_clusters) return;
    const int p0 = offsets[cl], n = (ends ? ends[cl] : offsets[cl + 1]) - p0;
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
                                                  const int* __restrict__ ends,
                                                  const double* __restrict__ xs, const double* __restrict__ ys,
                                                  int* __restrict__ is_circle, double* __restrict__ std_dev)
{
    const double PI = 3.14159265358979323846;
    const int lane = threadIdx.x & 63;
    const int cl = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cl >= n_clusters) return;
    const int p0 = offsets[cl], n = (ends ? ends[cl] : offsets[cl + 1]) - p0;
    if (n < 3) { if (lane == 0) { is_circle[cl] = 0; std_dev[cl] = __builtin_nan(""); } return; }   // sqrt(0/0): NaN < 10 is false (:243-249)
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

// ---------------------------------------------------------------- scans -> markers (the landmarks node on the device)
// circle_fit::clusterPoints, circle_fit_library.cpp:136-206, one lane per 360-ray scan (a sequential walk: each step
// decides from the previous one), with the reference's quirks kept: the wrap-around point goes to cluster 0, an
// unfinished cluster at the end of the walk is lost, and the erase loop (:197-204) skips the element after each one
// it erases.  Surviving clusters are written contiguously, in push order, into the scan's 360-point region:
// cluster c of scan s owns points [cbeg[s * kMaxScanClusters + c], cend[...]) of px / py.
constexpr int kMaxScanClusters = 64;

__global__ __launch_bounds__(64) void k_scan_clusters(int n_scans, const float* __restrict__ scans, double min_range,
                                                      double max_range, double* __restrict__ px, double* __restrict__ py,
                                                      int* __restrict__ cbeg, int* __restrict__ cend,
                                                      int* __restrict__ tmp_i, double* __restrict__ tmp_d,
                                                      int* __restrict__ overflow)
{
    const double PI = 3.14159265358979323846;
    const double threshold = 0.04;
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n_scans) return;
    const float* ranges = scans + (size_t)s * 360;
    int* cl_of = tmp_i + (size_t)s * 1080;                // cluster of the k-th emitted point
    int* sizes = cl_of + 360;                             // points per pushed cluster
    int* alive = cl_of + 720;                             // the vector of clusters during the erase loop
    double* ex = tmp_d + (size_t)s * 720;                 // emitted points, in emission order
    double* ey = ex + 360;
    for (int k = 0; k < 360; ++k) sizes[k] = 0;
    int curr = 0, npts = 0, nclusters = 0, open = 0;
    while (curr < 360) {
        const float rc = ranges[curr];
        if (((double)rc > max_range) || ((double)rc < min_range)) { curr += 1; continue; }     // :148-152
        const int next = (curr + 1) % 360;
        const double cd = rc, nd = ranges[next];
        const double x = (double)rc * cos((PI / (double)180) * curr);                          // :161-162
        const double y = (double)rc * sin((PI / (double)180) * curr);
        if (fabs(cd - nd) < threshold) {
            if (next < curr) {                            // wrap-around: the point joins the first cluster (:169-172)
                if (nclusters > 0) { ex[npts] = x; ey[npts] = y; cl_of[npts] = 0; sizes[0]++; npts++; }
            } else {
                ex[npts] = x; ey[npts] = y; cl_of[npts] = nclusters; sizes[nclusters]++; npts++;
                open = 1;
                curr += 1;
            }
        } else {
            ex[npts] = x; ey[npts] = y; cl_of[npts] = nclusters; sizes[nclusters]++; npts++;
            nclusters++;
            open = 0;
            curr += 1;
        }
        if (next < curr) break;
    }
    if (open)                                             // the open cluster is never pushed: its points are lost
        for (int k = 0; k < npts; ++k) if (cl_of[k] == nclusters) cl_of[k] = -1;
    // erase loop :197-204: erasing element i shifts the rest down while i still advances (the next one is skipped)
    int cnt = nclusters;
    for (int i = 0; i < cnt; ++i) alive[i] = i;
    for (int i = 0; i < cnt; ++i)
        if (sizes[alive[i]] < 3) {
            for (int j = i; j < cnt - 1; ++j) alive[j] = alive[j + 1];
            cnt--;
        }
    // gather the survivors, cluster by cluster, points in emission order
    const int base = s * 360;
    double* X = px + (size_t)base;
    double* Y = py + (size_t)base;
    int* cb = cbeg + (size_t)s * kMaxScanClusters;
    int* ce = cend + (size_t)s * kMaxScanClusters;
    int kept = 0;
    for (int c = 0; c < kMaxScanClusters; ++c) {
        cb[c] = base + kept;
        if (c < cnt) {
            const int id = alive[c];
            for (int k = 0; k < npts; ++k)
                if (cl_of[k] == id) { X[kept] = ex[k]; Y[kept] = ey[k]; kept++; }
        }
        ce[c] = base + kept;
    }
    if (cnt > kMaxScanClusters) atomicAdd(overflow + 0, 1);       // [0]: cluster table too small for this scan
}

// the node's loop body, nuslam/src/landmarks.cpp:82-108, per scan: clusters in order; keep a cluster if classifyCluster
// says circle, circleFit returned a marker (id >= 0) and its radius is at most 1; its centre becomes the next marker.
__global__ __launch_bounds__(64) void k_scan_markers(int n_scans, int m, const int* __restrict__ is_circle,
                                                     const int* __restrict__ status, const double* __restrict__ fit,
                                                     double* __restrict__ mx, double* __restrict__ my, int* __restrict__ ids,
                                                     unsigned long long* __restrict__ empty, int* __restrict__ overflow)
{
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n_scans) return;
    int slot = 0;
    for (int c = 0; c < kMaxScanClusters; ++c) {
        const int cl = s * kMaxScanClusters + c;
        if (!is_circle[cl]) continue;                                            // :86
        if (status[cl] != 0) continue;                                           // marker.id < 0, :91-93
        if (fit[3 * cl + 2] > 1) continue;                                       // marker.scale.x / 2 > 1, :95-97
        if (slot >= m) { atomicAdd(overflow + 1, 1); break; }       // [1]: more accepted markers than slots
        mx[(size_t)s * m + slot] = fit[3 * cl];
        my[(size_t)s * m + slot] = fit[3 * cl + 1];
        ids[(size_t)s * m + slot] = slot + 1;                                    // marker.id = running index (:103); > 0 = present
        slot++;
    }
    if (slot < m) atomicAdd(empty, (unsigned long long)(m - slot));
    for (; slot < m; ++slot) { mx[(size_t)s * m + slot] = 0.0; my[(size_t)s * m + slot] = 0.0; ids[(size_t)s * m + slot] = -1; }
}

#define CHK(expr) do { if ((expr) != hipSuccess) { rc = NUSLAM_E_HIP; goto done; } } while (0)

} // namespace

namespace nuslam {

// Scans (device, [n_scans][360] float) -> marker slots (device, [n_scans][m]): k_scan_clusters, k_classify,
// k_circle_fit, k_scan_markers on `stream`.  overflow_out[0]: scans with more than kMaxScanClusters surviving clusters,
// overflow_out[1]: scans with more than m accepted markers (the surplus is dropped in both cases).  Returns a nuslam_status.
int scan_to_markers(hipStream_t stream, const float* d_scans, int n_scans, double min_range, double max_range, int m,
                    double* d_mx, double* d_my, int* d_ids, unsigned long long* d_empty, int* overflow_out)
{
    int rc = NUSLAM_OK;
    const size_t S = (size_t)n_scans, NC = S * kMaxScanClusters;
    double *px = nullptr, *py = nullptr, *tmp_d = nullptr, *work = nullptr, *fit = nullptr, *sd = nullptr;
    int *cbeg = nullptr, *cend = nullptr, *tmp_i = nullptr, *status = nullptr, *circ = nullptr, *ovf = nullptr;
    int h_ovf[2] = { 0, 0 };
    CHK(hipMalloc(&px, sizeof(double) * S * 360)); CHK(hipMalloc(&py, sizeof(double) * S * 360));
    CHK(hipMalloc(&tmp_d, sizeof(double) * S * 720)); CHK(hipMalloc(&tmp_i, sizeof(int) * S * 1080));
    CHK(hipMalloc(&work, sizeof(double) * 4 * S * 360));
    CHK(hipMalloc(&cbeg, sizeof(int) * NC)); CHK(hipMalloc(&cend, sizeof(int) * NC));
    CHK(hipMalloc(&fit, sizeof(double) * 3 * NC)); CHK(hipMalloc(&sd, sizeof(double) * NC));
    CHK(hipMalloc(&status, sizeof(int) * NC)); CHK(hipMalloc(&circ, sizeof(int) * NC));
    CHK(hipMalloc(&ovf, 2 * sizeof(int)));
    CHK(hipMemsetAsync(ovf, 0, 2 * sizeof(int), stream));
    hipLaunchKernelGGL(k_scan_clusters, dim3((n_scans + 63) / 64), dim3(64), 0, stream, n_scans, d_scans, min_range, max_range,
                       px, py, cbeg, cend, tmp_i, tmp_d, ovf);
    hipLaunchKernelGGL(k_classify, dim3(((int)NC + 3) / 4), dim3(256), 0, stream, (int)NC, (const int*)cbeg, (const int*)cend,
                       (const double*)px, (const double*)py, circ, sd);
    hipLaunchKernelGGL(k_circle_fit, dim3(((int)NC + 3) / 4), dim3(256), 0, stream, (int)NC, (const int*)cbeg, (const int*)cend,
                       (const double*)px, (const double*)py, work, fit, status);
    hipLaunchKernelGGL(k_scan_markers, dim3((n_scans + 63) / 64), dim3(64), 0, stream, n_scans, m, (const int*)circ,
                       (const int*)status, (const double*)fit, d_mx, d_my, d_ids, d_empty, ovf);
    CHK(hipGetLastError());
    CHK(hipMemcpyAsync(h_ovf, ovf, 2 * sizeof(int), hipMemcpyDeviceToHost, stream));
    CHK(hipStreamSynchronize(stream));
    if (overflow_out) { overflow_out[0] = h_ovf[0]; overflow_out[1] = h_ovf[1]; }
done:
    void* ptrs[] = { px, py, tmp_d, tmp_i, work, cbeg, cend, fit, sd, status, circ, ovf };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    return rc;
}

} // namespace nuslam

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
                       (const int*)nullptr, (const double*)d_x, (const double*)d_y, d_work, d_out, d_status);
    hipLaunchKernelGGL(k_classify, dim3((n_clusters + 3) / 4), dim3(256), 0, nullptr, n_clusters, (const int*)d_off,
                       (const int*)nullptr, (const double*)d_x, (const double*)d_y, d_circ, d_sd);
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
