// ekf_kernels.h -- the HIP kernels of the EKF-SLAM hot path for CDNA4 (gfx950, wave64).
//
// Data layout in HBM (per filter b of a batch of B; a single filter is B = 1):
//   P      covariance, COLUMN-major like arma::mat, leading dimension ld = roundup(len, 32) elements so that
//          every column starts on a 128-byte line and 16-byte vector accesses never straddle columns;
//          rows [len, ld) are zero padding.  Element type T = double or float; all arithmetic is fp64.
//          Two buffers: update reads one and writes the other (ping-pong), predict works in place.
//   state  fp64, ld entries, double-buffered (kernels read s_in, write s_out; the host flips the pair).
//   ctrl   {seen, seen cached at tick start, break flag, latched status}, double-buffered the same way.
//
// Kernels (one stream per handle, launched back to back, no host round trip inside a tick):
//   k_predict    slam_library.cpp:65-148   O(len) -- A = I + B has two non-zeros, so A P A^T + Qbar only touches
//                                          rows/cols 1,2 and the 3x3 corner; bit-identical to the dense product
//                                          evaluated in ascending-k order.
//   k_associate  slam_library.cpp:188-253  all seen candidates in parallel, min-index reduction.
//   k_update     slam_library.cpp:263-282  + the caller's decision chain slam.cpp:295-316: z_hat, H, S, S^-1, K,
//                                          state correction and P <- (I - K H) P in ONE streaming pass: each
//                                          element of P is read once and written once (2 len^2 w bytes) -- the
//                                          HBM-bound kernel.
#pragma once
#include "ekf_device.h"

namespace nuslam {

// 16 bytes of consecutive rows of one column: what one lane moves per global_load/store_dwordx4
template <typename T> struct alignas(16) Pack16 { T v[16 / sizeof(T)]; };

// 16-byte streaming store (global_store_dwordx4 ... nt): the written tile is not read again by this kernel, and the
// next kernel's readers sit on other XCDs (other L2s), so the lines have no business staying dirty in this L2 until
// the end-of-kernel write-back.
template <typename T>
__device__ inline Pack16<T> load_stream(const T* p)
{
    typedef T vt __attribute__((ext_vector_type(16 / sizeof(T))));
#ifdef NUSLAM_NO_STREAM
    const vt y = *reinterpret_cast<const vt*>(p);
#else
    const vt y = __builtin_nontemporal_load(reinterpret_cast<const vt*>(p));
#endif
    Pack16<T> x;
#pragma unroll
    for (int e = 0; e < (int)(16 / sizeof(T)); ++e) x.v[e] = y[e];
    return x;
}
template <typename T>
__device__ inline void store_stream(T* p, const Pack16<T>& x)
{
    typedef T vt __attribute__((ext_vector_type(16 / sizeof(T))));
    vt y;
#pragma unroll
    for (int e = 0; e < (int)(16 / sizeof(T)); ++e) y[e] = x.v[e];
#ifdef NUSLAM_NO_STREAM
    *reinterpret_cast<vt*>(p) = y;
#else
    __builtin_nontemporal_store(y, reinterpret_cast<vt*>(p));
#endif
}

constexpr int kStatusBounds = 2;    // NUSLAM_E_BOUNDS
constexpr int kStatusSingular = 3;  // NUSLAM_E_SINGULAR

// ------------------------------------------------------------------------------------------------ init
template <typename T>
__global__ __launch_bounds__(256) void k_init(View v, const double* __restrict__ robot, const double* __restrict__ map,
                                               T* __restrict__ P, double* __restrict__ s0, double* __restrict__ s1,
                                               int* __restrict__ c0, int* __restrict__ c1)
{
    // ctor slam_library.cpp:39-63 and initCov :24-33 (P was zero-filled by hipMemsetAsync)
    const int b = blockIdx.z;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= v.ld) return;
    double val = 0.0;
    if (t < 3) val = robot ? robot[3 * b + t] : 0.0;
    else if (t < v.L) val = map ? map[(size_t)b * 2 * v.n + (t - 3)] : 0.0;
    s0[(size_t)b * v.ld + t] = val;
    s1[(size_t)b * v.ld + t] = val;
    if (t >= 3 && t < v.L) P[(size_t)b * v.p_stride + (size_t)t * v.ld + t] = (T)2147483647.0; // INT_MAX, :30
    if (t < C_WORDS) { c0[b * C_WORDS + t] = 0; c1[b * C_WORDS + t] = 0; }
    if (t < 2) v.akey[2 * b + t] = kNoKey;
}

// A = I (len x len, leading dimension ld, padding rows zero) for the dense-Jacobian predict that forms getA on the device
template <typename T>
__global__ __launch_bounds__(256) void k_fill_identity(int L, int ld, T* __restrict__ F)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)ld * L) return;
    const int i = (int)(e % ld), j = (int)(e / ld);
    F[e] = (i == j) ? (T)1 : (T)0;
}

// ------------------------------------------------------------------------------------------------ predict
// STATE_ONLY: predictEstimate + the tick bookkeeping only; the covariance is then propagated by the dense MFMA
// path (nuslam_ekf_predict_dense) instead of the two-non-zero shortcut below.
template <typename T, bool STATE_ONLY>
__global__ __launch_bounds__(256) void k_predict(View v, TwistArg tw, T* __restrict__ P, int bookkeeping, T* __restrict__ Fa)
{
    // Fa (STATE_ONLY, may be null): the resident dense Jacobian of filter 0, already I everywhere else: getA's two entries
    // A(1,0), A(2,0) (slam_library.cpp:133-146, at the heading AFTER predictEstimate) are written into it here, so that the
    // two MFMA products that follow compute the reference's A P A^T with this tick's A
    // bookkeeping == 0: the control words are carried by the chain stream (overlapped runs), this kernel leaves them alone
    const int b = blockIdx.z;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const double* s = v.s_in + (size_t)b * v.ld;
    double* so = v.s_out + (size_t)b * v.ld;
    const double dth = tw.tw ? tw.tw[b * tw.stride + tw.off + 0] : tw.dth0;
    const double dx = tw.tw ? tw.tw[b * tw.stride + tw.off + 1] : tw.dx0;

    // predictEstimate, slam_library.cpp:71-94
    const double theta = s[0];
    const MotionStep ms = motion_step(theta, dth, dx);                 // predictEstimate :71-94, getA :127-148
    const double dq_x = ms.dq_x, dq_y = ms.dq_y, th1 = ms.th1, a1 = ms.a1, a2 = ms.a2;
    if (t < v.ld) so[t] = t == 0 ? th1 : t == 1 ? s[1] + dq_x : t == 2 ? s[2] + dq_y : s[t];

    if (STATE_ONLY) {
        if (t == 0 && Fa && b == 0) { Fa[1] = (T)a1; Fa[2] = (T)a2; }
        if (t == 0 && bookkeeping) {
            const int* ci = v.c_in + b * C_WORDS;
            int* co = v.c_out + b * C_WORDS;
            co[C_SEEN] = ci[C_SEEN]; co[C_SEEN_CACHED] = ci[C_SEEN]; co[C_BRK] = 0; co[C_STATUS] = ci[C_STATUS];
        }
        return;
    }
    // propagateUncertainty, slam_library.cpp:96-108:  T = A P (rows 1,2),  U = T A^T (cols 1,2),  + Qbar
    T* Pb = P + (size_t)b * v.p_stride;
    const int ld = v.ld;
    if (t == 0) {
        double p[3][3], tt[3][3], u[3][3];
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) p[i][j] = (double)Pb[(size_t)j * ld + i];
        for (int j = 0; j < 3; ++j) {
            tt[0][j] = p[0][j];
            tt[1][j] = a1 * p[0][j] + p[1][j];
            tt[2][j] = a2 * p[0][j] + p[2][j];
        }
        for (int i = 0; i < 3; ++i) {
            u[i][0] = tt[i][0];
            u[i][1] = tt[i][0] * a1 + tt[i][1];
            u[i][2] = tt[i][0] * a2 + tt[i][2];
        }
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) Pb[(size_t)j * ld + i] = (T)(u[i][j] + v.Q[i + 3 * j]);
        // slam.cpp:250-251: the caller caches seen_landmarks at the top of the loop body; a new tick also
        // clears the `break` of the previous marker loop.
        if (bookkeeping) {
            const int* ci = v.c_in + b * C_WORDS;
            int* co = v.c_out + b * C_WORDS;
            co[C_SEEN] = ci[C_SEEN];
            co[C_SEEN_CACHED] = ci[C_SEEN];
            co[C_BRK] = 0;
            co[C_STATUS] = ci[C_STATUS];
        }
    } else if (t >= 3 && t < v.L) {
        T* col = Pb + (size_t)t * ld;               // column role: rows 1,2 of column t
        const double p0 = (double)col[0];
        const double p1 = (double)col[1];
        const double p2 = (double)col[2];
        col[1] = (T)(a1 * p0 + p1);
        col[2] = (T)(a2 * p0 + p2);
        const double t0 = (double)Pb[t];            // row role: columns 1,2 of row t (rows >= 3: T == P)
        Pb[(size_t)1 * ld + t] = (T)(t0 * a1 + (double)Pb[(size_t)1 * ld + t]);
        Pb[(size_t)2 * ld + t] = (T)(t0 * a2 + (double)Pb[(size_t)2 * ld + t]);
    }
}

// ------------------------------------------------------------------------------------------------ associate
__device__ inline int wave_min(int x)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const int y = __shfl_xor(x, m, 64);
        x = y < x ? y : x;
    }
    return x;
}

// associateLandmark, slam_library.cpp:188-253.  One lane per candidate landmark, one wave per workgroup, so the
// candidates of one filter spread over ceil(seen/64) CUs.  The reference walks k = 1..seen and stops at the first k
// whose distance is < 0.01 (match) or in (0.01, 60) (gray zone): that is the minimum k over a per-candidate
// predicate -> wave min, then one atomicMin per wave on the filter's key slot.  The key is decoded by the consumer
// (k_update in a tick, k_associate_finish for the stand-alone entry point).
template <typename T>
__global__ __launch_bounds__(64) void k_associate(View v, ObsArg o, const T* __restrict__ P)
{
    const int b = blockIdx.y;
    const int* ci = v.c_in + b * C_WORDS;
    const int seen = ci[C_SEEN];
    if (ci[C_BRK] || seen == 0 || seen >= v.n) return;        // decided without looking at any candidate
    if ((int)blockIdx.x * 64 >= seen) return;
    const int k = blockIdx.x * 64 + threadIdx.x + 1;
    const double* s = v.s_in + (size_t)b * v.ld;
    int key = kNoKey;
    if (k <= seen) {
        double r, phi;
        fetch_obs(o, b, r, phi);
        const double th = s[0], x = s[1], y = s[2];
        const T* Pb = P + (size_t)b * v.p_stride;
        const double min_threshold = 0.01;   // :193
        const double max_threshold = 60;     // :194
        const int c = 3 + 2 * (k - 1);
        const double lx = s[c], ly = s[c + 1];
        double Hc[10], psi[4], psi_inv[4], zr, zb;
        const int set[5] = { 0, 1, 2, c, c + 1 };
        jacobian_compact(x, y, lx, ly, Hc);                    // :212
        innovation_cov<T>(Pb, v.ld, set, Hc, v.R, psi);        // :215
        measurement(th, x, y, lx, ly, zr, zb);                 // :218
        const double dz0 = r - zr, dz1 = phi - zb;             // :229 (bearing difference not wrapped)
        int code = -1;
        if (inv2(psi, psi_inv)) code = 2;                      // psi_k.i() would throw
        else {
            double w0 = 0.0, w1 = 0.0, d = 0.0;                // (dz^T psi^-1) dz, :231
            w0 = fma(dz0, psi_inv[0], w0); w0 = fma(dz1, psi_inv[1], w0);
            w1 = fma(dz0, psi_inv[2], w1); w1 = fma(dz1, psi_inv[3], w1);
            d = fma(w0, dz0, d); d = fma(w1, dz1, d);
            if (d < min_threshold) code = 0;                                   // :238
            else if ((d > min_threshold) && (d < max_threshold)) code = 1;     // :243
        }
        if (code >= 0) key = k * 4 + code;
    }
    key = wave_min(key);
    if (threadIdx.x == 0 && key != kNoKey) atomicMin(v.akey + 2 * b + v.aslot, key);
}

// Stand-alone associateLandmark(): decode the key, publish the id, count a new landmark, re-arm the other slot.
// host_word (may be null; single-filter API): pinned host memory the answer is ALSO written to -- {id, latched status, expired
// device-side waits} of filter 0 -- so that the synchronising caller reads it after its stream synchronize without two more
// blocking copies (the reference's associateLandmark returns the id to the host: one round trip per marker is the API's).
__global__ void k_associate_finish(View v, int* __restrict__ host_word, const int* __restrict__ expired)
{
    const int b = blockIdx.x;
    const int* ci = v.c_in + b * C_WORDS;
    int* co = v.c_out + b * C_WORDS;
    const Assoc a = decode_association(v.n, ci[C_SEEN], ci[C_BRK], ci[C_STATUS], v.akey[2 * b + v.aslot]);
    v.cur_id[b] = a.id;
    co[C_SEEN] = a.new_seen; co[C_SEEN_CACHED] = ci[C_SEEN_CACHED]; co[C_BRK] = ci[C_BRK]; co[C_STATUS] = a.new_status;
    v.akey[2 * b + v.aslot] = kNoKey;         // consumed
    v.akey[2 * b + (v.aslot ^ 1)] = kNoKey;
    if (host_word && b == 0) {
        host_word[0] = a.id;
        host_word[1] = a.new_status;
        host_word[2] = expired ? *expired : 0;
    }
}

// ------------------------------------------------------------------------------------------------ init landmark
// initializeLandmark, slam_library.cpp:255-261 (stand-alone entry point; inside a tick it is folded into prepare)
__global__ void k_init_landmark(View v, ObsArg o, double* __restrict__ s_cur)
{
    const int b = blockIdx.x;
    double* s = s_cur + (size_t)b * v.ld;
    double r, phi;
    fetch_obs(o, b, r, phi);
    const int id = o.ids ? o.ids[b * o.stride + o.off] : o.id0;
    const double th = s[0], x = s[1], y = s[2];
    s[3 + 2 * (id - 1)] = x + r * cos(phi + th);
    s[4 + 2 * (id - 1)] = y + r * sin(phi + th);
}

// ------------------------------------------------------------------------------------------------ update
// k_update lives in ekf_update.h (included at the end of this header).

// ------------------------------------------------------------------------------------------------ batch statistics
// trace(P) per filter: one workgroup per filter, fixed reduction tree (deterministic).
template <typename T>
__global__ __launch_bounds__(256) void k_trace(View v, const T* __restrict__ P, double* __restrict__ tr)
{
    const int b = blockIdx.x;
    const T* Pb = P + (size_t)b * v.p_stride;
    __shared__ double sh[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < v.L; i += 256) acc = acc + (double)Pb[(size_t)i * v.ld + i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] = sh[threadIdx.x] + sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) tr[b] = sh[0];
}

// Per-filter Monte-Carlo terms against the simulated truth (generated traces keep the true pose of every tick):
// e = (wrap(theta_hat - theta), x_hat - x, y_hat - y) and NEES = e^T Ppose^-1 e with Ppose = P[0:3, 0:3] (3x3 inverse
// by cofactors, fp64).  pe[4 b + {0,1,2}] = e^2, pe[4 b + 3] = NEES.  truth == nullptr: zeros.
template <typename T>
__global__ __launch_bounds__(64) void k_pose_error(View v, const T* __restrict__ P, const double* __restrict__ s_cur,
                                                   const double* __restrict__ truth, long long truth_stride,
                                                   long long truth_off, double* __restrict__ pe)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= v.B) return;
    double* o = pe + 4 * (size_t)b;
    if (truth == nullptr) { o[0] = o[1] = o[2] = o[3] = 0.0; return; }
    const double* tp = truth + (size_t)b * truth_stride + truth_off;     // (theta, x, y) after the last tick run
    const double* s = s_cur + (size_t)b * v.ld;
    const double e0 = normalize_angle(s[0] - tp[0]), e1 = s[1] - tp[1], e2 = s[2] - tp[2];
    const T* Pb = P + (size_t)b * v.p_stride;
    double a[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) a[i][j] = (double)Pb[(size_t)j * v.ld + i];
    const double c00 = a[1][1] * a[2][2] - a[1][2] * a[2][1];
    const double c01 = a[1][2] * a[2][0] - a[1][0] * a[2][2];
    const double c02 = a[1][0] * a[2][1] - a[1][1] * a[2][0];
    const double det = a[0][0] * c00 + a[0][1] * c01 + a[0][2] * c02;
    const double i00 = c00 / det, i01 = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) / det, i02 = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) / det;
    const double i10 = c01 / det, i11 = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) / det, i12 = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) / det;
    const double i20 = c02 / det, i21 = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) / det, i22 = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) / det;
    const double q0 = i00 * e0 + i01 * e1 + i02 * e2;
    const double q1 = i10 * e0 + i11 * e1 + i12 * e2;
    const double q2 = i20 * e0 + i21 * e1 + i22 * e2;
    o[0] = e0 * e0; o[1] = e1 * e1; o[2] = e2 * e2;
    o[3] = e0 * q0 + e1 * q1 + e2 * q2;
}

// out = { sum_b state[i] (L), sum_b state[i]^2 (L), sum_b (est - truth)^2 of the pose (3), sum_b NEES, sum_b trace(P), B }
//  -- 2 L + 6 doubles, every sum accumulated in filter order (deterministic).
__global__ __launch_bounds__(256) void k_stats(View v, const double* __restrict__ s_cur, const double* __restrict__ tr,
                                               const double* __restrict__ pe, double* __restrict__ out)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < v.L) {
        double a = 0.0, a2 = 0.0;
        for (int b = 0; b < v.B; ++b) {
            const double x = s_cur[(size_t)b * v.ld + t];
            a = a + x;
            a2 = a2 + x * x;
        }
        out[t] = a;
        out[v.L + t] = a2;
    }
    if (t < 4) {
        double a = 0.0;
        for (int b = 0; b < v.B; ++b) a = a + pe[4 * (size_t)b + t];
        out[2 * v.L + t] = a;
    }
    if (t == 4) {
        double a = 0.0;
        for (int b = 0; b < v.B; ++b) a = a + tr[b];
        out[2 * v.L + 4] = a;
        out[2 * v.L + 5] = (double)v.B;
    }
}

} // namespace nuslam

#include "ekf_update.h"
#include "ekf_update2.h"
#include "ekf_tick.h"
#include "ekf_rank.h"
#include "ekf_fused.h"
#include "ekf_da.h"
#include "ekf_deferred.h"
