// ekf_device.h -- device-side math of the EKF-SLAM hot path (gfx950).
//
// Every function states the reference lines it computes (paths relative to the reference checkout).
// This translation unit is compiled with -ffp-contract=off: each multiply and add is rounded on its own,
// in the order written, which is also how the CPU oracle is built -- the two sides then differ only in
// libm (sin/cos/atan2), never in the covariance algebra.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>

namespace nuslam {

// ctrl words per filter (double-buffered, see View)
enum { C_SEEN = 0, C_SEEN_CACHED = 1, C_BRK = 2, C_STATUS = 3, C_WORDS = 4 };
// how k_update resolves the landmark id and whether the slam.cpp decision chain applies
enum { MODE_FORCE = 0, MODE_KNOWN = 1, MODE_DA = 2 };

struct View {
    int n, L, ld, B;
    const double* s_in;   // [B][ld]   state before this kernel
    double* s_out;        // [B][ld]   state after  (host flips the pair after every state-writing kernel)
    const int* c_in;      // [B][C_WORDS]
    int* c_out;
    long long p_stride;   // elements between consecutive filters' covariance (ld * L)
    int* cur_id;          // [B] id resolved by the stand-alone associate entry point
    int* akey;            // [B][2] association keys: min over candidates of 4k + outcome, INT_MAX = none (two slots,
    int aslot;            //        used alternately: the consumer of slot s re-arms slot s^1)
    int* id_log;          // [B][log_stride] resolved id per observation of the current tick (may be null)
    int log_stride;
    double* dump;         // kTickDump doubles nobody reads: where lanes of the strip kernels that own nothing store (null until the
                          // tick pipeline's buffers exist).  A pointer of its own, not "behind filter B's strips": a launch may
                          // cover a sub-range of the handle's filters
    double Q[9];          // 3x3 column-major process noise   (slam_library.hpp:27)
    double R[4];          // 2x2 column-major sensor noise    (slam_library.hpp:28)
};

constexpr int kNoKey = 0x7fffffff;

// Turn the reduced association key into associateLandmark's return value (slam_library.cpp:197-200, 206-207,
// 238-252).  outcome code: 0 = match (d < 0.01), 1 = gray zone (0.01 < d < 60), 2 = psi singular.
struct Assoc { int id, new_seen, new_status; };
__device__ inline Assoc decode_association(int n, int seen, int brk, int status, int key)
{
    Assoc a;
    a.id = 0; a.new_seen = seen; a.new_status = status;
    if (brk) { a.id = 0; }                                               // marker loop already left (slam.cpp:315)
    else if (seen == 0) { a.new_seen = 1; a.id = 1; }                    // :197-200
    else if (seen >= n) { a.id = -1; if (status == 0) a.new_status = 2; } // :206-207 out-of-bounds write -> NUSLAM_E_BOUNDS
    else if (key == kNoKey) { a.new_seen = seen + 1; a.id = a.new_seen; } // :251-252
    else {
        const int code = key & 3;
        if (code == 0) a.id = key >> 2;
        else if (code == 1) a.id = -1;
        else { a.id = -1; if (status == 0) a.new_status = 3; }           // NUSLAM_E_SINGULAR
    }
    return a;
}

// One observation per filter: either inline (single-filter API) or from a trace resident in HBM.
struct ObsArg {
    const double* a;     // marker x (cartesian != 0) or range; null -> inline a0
    const double* b;     // marker y or bearing
    const int* ids;      // known 1-based ids; null -> inline id0 (MODE_FORCE / MODE_KNOWN) or cur_id (MODE_DA)
    long long stride;    // per-filter stride in elements (0 = every filter reads the same trace)
    long long off;       // element offset of this observation
    double a0, b0;
    int id0;
    int cartesian;
    int log_slot;        // index into id_log, or -1
};

struct TwistArg {
    const double* tw;    // (dth, dx) pairs; null -> inline
    long long stride, off;
    double dth0, dx0;
};

// rigid2d::normalize_angle, rigid2d/src/rigid2d.cpp:9-13, is atan2(sin(rad), cos(rad)): the representative of rad in
// (-pi, pi].  Here the same value by range reduction: rad - k 2pi with 2pi as a two-constant sum (hi + lo, fma), the
// identity for |rad| <= pi.  It agrees with glibc's atan2(sin, cos) to <= 1 ulp -- the class of difference the device's
// own libm already had -- including the edges (+-pi map to themselves, pi + 1 ulp to -pi, 3 pi to +pi; pinned against the
// reference-generated vectors of tests/golden/rigid2d_ref.npz by tests/test_gpu_more.py through nuslam_device_normalize_angle),
// and it takes ~15 instructions where sincos + atan2 take ~450: three of these wraps sat on the serial chain of every
// correction (the heading after K nu, :276; the two of z_hat, :20 and :157).  Beyond 1e6 rad, where k 2pi_lo loses bits,
// the libm form is kept.
__device__ inline double normalize_angle(double rad)
{
    constexpr double kPi = 3.141592653589793, kTwoPiHi = 6.283185307179586, kTwoPiLo = 2.4492935982947064e-16;
    constexpr double kInvTwoPi = 0.15915494309189535;
    const double a = fabs(rad);
    if (a <= kPi) return rad;
    if (!(a <= 1.0e6)) {
        double sn, cs;
        sincos(rad, &sn, &cs);
        return atan2(sn, cs);
    }
    const double k = rint(rad * kInvTwoPi);
    double r = fma(-k, kTwoPiHi, rad);
    r = fma(-k, kTwoPiLo, r);
    // k was rounded from a rounded quotient: r may sit one period off by a hair beyond +-pi
    if (r > kPi) r = fma(-1.0, kTwoPiLo, r - kTwoPiHi);
    else if (r < -kPi) r = fma(1.0, kTwoPiLo, r + kTwoPiHi);
    return r;
}

// predictEstimate (slam_library.cpp:71-94) and getA's two entries (:127-148, at the heading AFTER predictEstimate) for one filter;
// every lane of a full wave calls it with the same arguments.  The reference's eight sines and cosines are TWO library calls
// with four lanes side by side (theta, theta + dth, th1, th1 + dth) instead of eight in a row -- a wave issues ~one fp64
// instruction per 6.6 cycles however many lanes take part, and the eight calls were 2.4 us at the head of every tick.  Each value is
// the library's for that argument; the expressions around them are the reference's, term for term.
struct MotionStep { double dq_th, dq_x, dq_y, th1, a1, a2; };
__device__ inline double wave_bcast(double val, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(val), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(val), src);
    return __hiloint2double(hi, lo);
}
__device__ inline MotionStep motion_step(double theta, double dth, double dx)
{
    MotionStep m;
    m.dq_th = dth == 0.0 ? 0.0 : dth;
    m.th1 = theta + m.dq_th;
    const int k = threadIdx.x & 3;
    const double ang = k == 0 ? theta : k == 1 ? theta + dth : k == 2 ? m.th1 : m.th1 + dth;
    const double sv = sin(ang), cv = cos(ang);
    const double s_th = wave_bcast(sv, 0), c_th = wave_bcast(cv, 0);          // sin / cos of theta
    const double s_th_d = wave_bcast(sv, 1), c_th_d = wave_bcast(cv, 1);      // ... of theta + dth
    const double s_t1 = wave_bcast(sv, 2), c_t1 = wave_bcast(cv, 2);          // ... of th1
    const double s_t1_d = wave_bcast(sv, 3), c_t1_d = wave_bcast(cv, 3);      // ... of th1 + dth
    if (dth == 0.0) {
        m.dq_x = dx * c_th;
        m.dq_y = dx * s_th;
        m.a1 = -dx * s_t1;
        m.a2 = dx * c_t1;
    } else {
        m.dq_x = -(dx / dth) * s_th + (dx / dth) * s_th_d;
        m.dq_y = (dx / dth) * c_th - (dx / dth) * c_th_d;
        m.a1 = -(dx / dth) * c_t1 + (dx / dth) * c_t1_d;
        m.a2 = -(dx / dth) * s_t1 + (dx / dth) * s_t1_d;
    }
    return m;
}

__device__ inline void cartesian2polar(double x, double y, double& r, double& b) // slam_library.cpp:16-22
{
    r = sqrt((x * x) + (y * y));
    b = normalize_angle(atan2(y, x));
}

// The raw marker of filter `bidx`, loaded UNCONDITIONALLY (the address falls back to `safe`, any readable double,
// when the marker travels inline): a kernel that wants the value while its tile loads are still in flight must issue
// this load before them in straight-line code -- vmcnt retires in order, and a load inside a branch makes the
// compiler's scoreboard merge conservative (s_waitcnt vmcnt(0)).
__device__ inline void load_obs_raw(const double* oa, const double* ob, long long stride, long long off, double a0,
                                    double b0, int bidx, const double* safe, double& a, double& b)
{
    const double* pa = oa ? oa + (bidx * stride + off) : safe;
    const double* pb = ob ? ob + (bidx * stride + off) : safe;
    const double la = *pa, lb = *pb;
    a = oa ? la : a0;
    b = ob ? lb : b0;
}
__device__ inline void obs_polar(const ObsArg& o, double a, double b, double& r, double& phi)
{
    if (o.cartesian) cartesian2polar(a, b, r, phi);        // slam.cpp:286
    else { r = a; phi = b; }
}

__device__ inline void fetch_obs(const ObsArg& o, int bidx, double& r, double& phi)
{
    double a = o.a ? o.a[bidx * o.stride + o.off] : o.a0;
    double b = o.b ? o.b[bidx * o.stride + o.off] : o.b0;
    if (o.cartesian) cartesian2polar(a, b, r, phi);        // slam.cpp:286
    else { r = a; phi = b; }
}

// computeTheoreticalMeasurement with the five state entries it reads, slam_library.cpp:150-160
__device__ inline void measurement(double th, double x, double y, double lx, double ly, double& zr, double& zb)
{
    const double mx = lx - x;
    const double my = ly - y;
    cartesian2polar(mx, my, zr, zb);
    zb = normalize_angle(zb - th);
}

// linearizedMeasurementModel, slam_library.cpp:162-186: the nine non-zeros, Hc[r + 2q] = H(r, set[q]),
// set = {0, 1, 2, c, c+1}
__device__ inline void jacobian_compact(double x, double y, double lx, double ly, double Hc[10])
{
    const double dx = lx - x;
    const double dy = ly - y;
    const double d = (dx * dx) + (dy * dy);
    const double sd = sqrt(d);
    // four divisions; the other four entries are exact negations ((-a)/b == -(a/b) in IEEE arithmetic)
    const double xs = dx / sd, ys = dy / sd, xd = dx / d, yd = dy / d;
    Hc[0] = 0.0;  Hc[1] = -1.0;
    Hc[2] = -xs;  Hc[3] = yd;
    Hc[4] = -ys;  Hc[5] = -xd;
    Hc[6] = xs;   Hc[7] = -yd;
    Hc[8] = ys;   Hc[9] = xd;
}

// inv() / .i() of a 2x2 as Armadillo 9.800 evaluates it (auxlib::inv_noalias_tinymat; general LU when
// |det| < epsilon).  Column-major in and out.  Returns 0, or 1 when exactly singular.
__device__ inline int inv2(const double X[4], double out[4])
{
    const double a = X[0], c = X[1], b = X[2], d = X[3];
    const double det = (a * d) - (b * c);
    if (fabs(det) >= DBL_EPSILON) {
        out[0] = d / det;
        out[2] = -b / det;
        out[1] = -c / det;
        out[3] = a / det;
        return 0;
    }
    double r00 = a, r01 = b, r10 = c, r11 = d;
    bool swp = false;
    if (fabs(r10) > fabs(r00)) { double t0 = r00, t1 = r01; r00 = r10; r01 = r11; r10 = t0; r11 = t1; swp = true; }
    if (r00 == 0.0) return 1;
    const double l = r10 / r00;
    const double u11 = r11 - l * r01;
    if (u11 == 0.0 || u11 != u11) return 1;
    double inv[4];
    for (int col = 0; col < 2; ++col) {
        const double e0 = col == 0 ? 1.0 : 0.0, e1 = col == 1 ? 1.0 : 0.0;
        const double y1 = e1 - l * e0;
        const double x1 = y1 / u11;
        const double x0 = (e0 - r01 * x1) / r00;
        inv[0 + 2 * col] = x0;
        inv[1 + 2 * col] = x1;
    }
    if (swp) { out[0] = inv[2]; out[1] = inv[3]; out[2] = inv[0]; out[3] = inv[1]; }
    else { out[0] = inv[0]; out[1] = inv[1]; out[2] = inv[2]; out[3] = inv[3]; }
    return 0;
}

// psi = H P H^T + R from the 5x5 block of P at rows/cols set (slam_library.cpp:215,270), ascending-k sums.
template <typename T>
__device__ inline void innovation_cov(const T* __restrict__ Pb, int ld, const int set[5], const double Hc[10],
                                      const double R[4], double S[4])
{
    double HPs[2][5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        double p[5];
#pragma unroll
        for (int q2 = 0; q2 < 5; ++q2) p[q2] = (double)Pb[(size_t)set[q] * ld + set[q2]];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int q2 = 0; q2 < 5; ++q2) acc = fma(Hc[r + 2 * q2], p[q2], acc);
            HPs[r][q] = acc;
        }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 5; ++q) acc = fma(HPs[r][q], Hc[s2 + 2 * q], acc);
            S[r + 2 * s2] = acc + R[r + 2 * s2];
        }
}

} // namespace nuslam
