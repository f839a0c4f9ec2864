// ekf_sim.h -- the simulator as an on-device Monte-Carlo trace generator.
//
// One iteration of nuturtlesim/src/tube_world.cpp:509-533 per filter and tick, written straight into the resident
// trace the tick kernels replay (twists as DiffDrive::getTwist hands them to the slam node, slam.cpp:264; markers in
// the robot frame as set_rel_markers publishes them, tube_world.cpp:270-329), so B Monte-Carlo worlds need no host
// involvement and nothing crosses PCIe per tick.
//
// Departures from the reference (deliberate; the oracle, oracle/sim_oracle.c, states the same ones):
//   - ideal loop period dt instead of ros::Time differences (:520-521);
//   - every random draw is a pure function of (seed, filter, tick, stream, index): Philox4x32-10 + Box-Muller instead
//     of one process-wide std::mt19937 (:49-62) -- filters can be generated in any order on any number of GPUs;
//   - out-of-range tubes are dropped (the reference only flags them DELETE, :300-307), at most the m nearest are kept,
//     in tube order; unused slots carry id -1, which the slam node's chain skips (slam.cpp:298-300).
//
// Two kernels: k_sim_path walks one filter's true pose through all ticks (a serial recurrence: one lane per filter;
// the collision loop over the tubes is the only O(n) part); k_sim_markers is one workgroup per (tick, filter) doing
// the range gate, the nearest-m selection by rank counting and the marker transform for all tubes in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/nuslam_hip.h"

namespace nuslam {

// ---------------------------------------------------------------- rigid2d on the device
struct Tf { double c, s, x, y; };                               // rigid2d::Transform2D, rigid2d.hpp:171-175

__device__ inline Tf tf_make(double x, double y, double rad) { return Tf{ cos(rad), sin(rad), x, y }; }   // rigid2d.cpp:170-176
__device__ inline Tf tf_inv(const Tf& T)                        // rigid2d.cpp:187-196
{
    return Tf{ T.c, -T.s, (-T.x * T.c) + (-T.y * T.s), (T.x * T.s) + (-T.y * T.c) };
}
__device__ inline Tf tf_mul(const Tf& L, const Tf& R)           // rigid2d.cpp:198-209
{
    return Tf{ (L.c * R.c) - (L.s * R.s), (L.s * R.c) + (L.c * R.s), (L.c * R.x) - (L.s * R.y) + L.x,
               (L.s * R.x) + (L.c * R.y) + L.y };
}
__device__ inline void tf_point(const Tf& T, double x, double y, double& ox, double& oy)   // rigid2d.cpp:178-185
{
    ox = (x * T.c) + (y * (-T.s)) + T.x;
    oy = (x * T.s) + (y * T.c) + T.y;
}
__device__ inline Tf integrate_twist(double dth, double dx, double dy)   // rigid2d.cpp:294-328
{
    if (dth == 0) return Tf{ 1.0, 0.0, dx, dy };
    const Tf T_sb{ 1.0, 0.0, dy / dth, -(dx / dth) };           // :310-312
    const Tf T_ss{ cos(dth), sin(dth), 0.0, 0.0 };              // :317
    return tf_mul(tf_mul(tf_inv(T_sb), T_ss), T_sb);            // :323-325
}

struct DiffDriveDev {                                           // rigid2d::DiffDrive, diff_drive.hpp members
    double base, rad, x, y, th, thL, thR;
    __device__ void convert_twist(double dth, double dx, double& uL, double& uR) const   // diff_drive.cpp:66-78
    {
        const double d = base / 2;
        uL = (-(d / rad) * dth) + (dx / rad);
        uR = ((d / rad) * dth) + (dx / rad);
    }
    __device__ void get_twist(double nL, double nR, double& dth, double& dx) const       // :80-110
    {
        const double dUL = nL - thL, dUR = nR - thR;
        dth = (rad / base) * (dUR - dUL);
        dx = (rad / 2) * (dUL + dUR);
    }
    __device__ void step(double nL, double nR)                                           // operator(), :111-146
    {
        double dth, dx;
        get_twist(nL, nR, dth, dx);
        const Tf Tbb = integrate_twist(dth, dx, 0.0);                                    // :124
        const double q0 = atan(Tbb.s / Tbb.c), q1 = Tbb.x, q2 = Tbb.y;                   // :129
        const double c = cos(th), s = sin(th);                                           // :134, adjoint of a pure rotation
        th += q0;                                                                        // :137-144
        x += (0.0 * q0) + (c * q1) - (s * q2);
        y += -(0.0 * q0) + (s * q1) + (c * q2);
        thL = nL; thR = nR;
    }
};

// ---------------------------------------------------------------- Philox4x32-10 (Salmon et al., SC'11)
__device__ inline void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                     unsigned out[4])
{
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const unsigned long long p0 = 0xD2511F53ull * c0;
        const unsigned long long p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        const unsigned n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        const unsigned n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ inline double u53(unsigned hi, unsigned lo)
{
    const double k = (double)(((unsigned long long)(hi >> 5) << 26) | (unsigned long long)(lo >> 6));
    return (k + 0.5) * (1.0 / 9007199254740992.0);
}

enum { SIM_STREAM_TWIST = 0, SIM_STREAM_SLIP = 1, SIM_STREAM_MARKER = 2 };

__device__ inline void normal_pair(unsigned long long seed, unsigned filter, unsigned tick, unsigned stream,
                                   unsigned idx, double& z0, double& z1)
{
    unsigned r[4];
    philox4x32_10(filter, tick, stream, idx, (unsigned)(seed & 0xffffffffu), (unsigned)(seed >> 32), r);
    const double u1 = u53(r[0], r[1]), u2 = u53(r[2], r[3]);
    const double rad = sqrt(-2.0 * log(u1));
    const double ang = 6.283185307179586476925286766559 * u2;
    z0 = rad * cos(ang);
    z1 = rad * sin(ang);
}

struct SimArg {
    nuslam_sim_params p;
    const double* lm;          // 2 * n_world: x0, y0, x1, y1, ...
    const double* cmd;         // ticks x (dth, dx)
    int n_world, ticks, m, B;
    unsigned long long seed;
    unsigned first_filter;     // global index of filter 0 of this batch (its shard offset)
    double* tw;                // [B][ticks][2]
    double* mx; double* my;    // [B][ticks][m]
    int* ids;                  // [B][ticks][m]
    double* truth;             // [B][ticks][3]  th, x, y after the tick
    unsigned long long* empty; // number of unused marker slots (all filters, all ticks)
    unsigned* raw;             // optional: philox words of (filter b, tick 0, stream 0, idx 0) for the bit-exact check
};

// one lane per filter: the serial walk of tube_world.cpp:509-533 (+ slam.cpp:264-265 for the twist the filter sees)
__global__ __launch_bounds__(64) void k_sim_path(SimArg a)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= a.B) return;
    const unsigned f = a.first_filter + (unsigned)b;
    const nuslam_sim_params& p = a.p;
    DiffDriveDev turtle{ p.wheel_base, p.wheel_radius, 0.0, 0.0, 0.0, 0.0, 0.0 };       // :487-491
    double oL = 0.0, oR = 0.0;                                                           // the odometry model's wheel angles
    double jL = 0.0, jR = 0.0;                                                           // :497-498
    const double slip_mean = (p.slip_min + p.slip_max) / 2;                              // :480
    const double slip_var = p.slip_max - slip_mean;                                      // :481
    if (a.raw) {
        unsigned r[4];
        philox4x32_10(f, 0u, 0u, 0u, (unsigned)(a.seed & 0xffffffffu), (unsigned)(a.seed >> 32), r);
        for (int q = 0; q < 4; ++q) a.raw[4 * b + q] = r[q];
    }
    for (int t = 0; t < a.ticks; ++t) {
        double z0, z1;
        normal_pair(a.seed, f, (unsigned)t, SIM_STREAM_TWIST, 0u, z0, z1);               // :177-189
        const double des_dth = a.cmd[2 * t + 0] + p.twist_noise * z0;
        const double des_dx = a.cmd[2 * t + 1] + p.twist_noise * z1;
        for (int i = 0; i < a.n_world; ++i) {                                            // check_collision :371-389
            const double dx = a.lm[2 * i] - turtle.x;
            const double dy = a.lm[2 * i + 1] - turtle.y;
            const double d = sqrt((dx * dx) + (dy * dy));
            if (d <= (p.tube_radius + p.robot_radius)) {
                const double move_x = dy / d;
                const double move_y = -dx / d;
                turtle.x += move_x / 50;                                                 // changeConfig, diff_drive.cpp:154-159
                turtle.y += move_y / 50;
            }
        }
        double uL, uR;
        turtle.convert_twist(des_dth, des_dx, uL, uR);                                   // :512
        jL += uL * p.dt;                                                                 // :520-521
        jR += uR * p.dt;
        normal_pair(a.seed, f, (unsigned)t, SIM_STREAM_SLIP, 0u, z0, z1);
        const double slipL = slip_mean + slip_var * z0;
        const double slipR = slip_mean + slip_var * z1;
        turtle.step(jL + uL * slipL, jR + uR * slipR);                                   // :526-527
        // what the slam node computes from the joint angles, slam.cpp:264-265
        const double dUL = jL - oL, dUR = jR - oR;
        const size_t o = (size_t)b * a.ticks + t;
        a.tw[2 * o + 0] = (p.wheel_radius / p.wheel_base) * (dUR - dUL);
        a.tw[2 * o + 1] = (p.wheel_radius / 2) * (dUL + dUR);
        oL = jL; oR = jR;
        a.truth[3 * o + 0] = turtle.th; a.truth[3 * o + 1] = turtle.x; a.truth[3 * o + 2] = turtle.y;
    }
}

// one workgroup per (tick, filter): set_rel_markers, tube_world.cpp:270-329, with the range gate applied and the
// m nearest kept.  LDS: the n_world distances and keep flags (n_world <= kSimMaxWorld).
constexpr int kSimMaxWorld = 8192;

__global__ __launch_bounds__(256) void k_sim_markers(SimArg a)
{
    const int t = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x;
    const unsigned f = a.first_filter + (unsigned)b;
    const nuslam_sim_params& p = a.p;
    const int n = a.n_world, m = a.m;
    extern __shared__ double sh_dist[];                          // n doubles, then n bytes of flags
    unsigned char* sh_keep = reinterpret_cast<unsigned char*>(sh_dist + n);
    __shared__ int sh_count;
    const size_t o = (size_t)b * a.ticks + t;
    const double th = a.truth[3 * o + 0], x = a.truth[3 * o + 1], y = a.truth[3 * o + 2];
    if (tid == 0) sh_count = 0;
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < n; i += 256) {
        const double dx = a.lm[2 * i] - x;
        const double dy = a.lm[2 * i + 1] - y;
        const double d = sqrt((dx * dx) + (dy * dy));            // :299
        bool inr = (p.max_range <= 0.0) || !(d > p.max_range);   // :300
        // extension (fov > 0): the tube must lie within +-fov of the robot's heading -- (tube - robot) . heading >= cos(fov) d
        if (p.fov > 0.0) inr = inr && ((cos(th) * dx) + (sin(th) * dy) >= cos(p.fov) * d) && !(d < p.min_range);
        sh_dist[i] = inr ? d : INFINITY;                         // (a gated tube takes no part in the ranking below)
        sh_keep[i] = inr ? 1 : 0;
        mine += inr ? 1 : 0;
    }
    if (mine) atomicAdd(&sh_count, mine);
    __syncthreads();
    const int in_range = sh_count;
    __syncthreads();
    if (in_range > m) {                                          // keep the m nearest; ties go to the lower tube index
        for (int i = tid; i < n; i += 256) {
            if (!sh_keep[i]) continue;
            const double di = sh_dist[i];
            int rank = 0;
            for (int j = 0; j < n; ++j) {
                const double dj = sh_dist[j];
                rank += (dj < INFINITY && (dj < di || (dj == di && j < i))) ? 1 : 0;
            }
            if (rank >= m) sh_keep[i] = 2;                       // own flag only; the rank loop reads distances
        }
        __syncthreads();
    }
    const Tf T_tw = tf_inv(tf_make(x, y, th));                   // :274-276
    const int kept = in_range < m ? in_range : m;
    for (int i = tid; i < n; i += 256) {
        if (sh_keep[i] != 1) continue;
        int slot = 0;
        for (int j = 0; j < i; ++j) slot += (sh_keep[j] == 1) ? 1 : 0;
        if (slot >= m) continue;
        double qx, qy, z0, z1;
        tf_point(T_tw, a.lm[2 * i], a.lm[2 * i + 1], qx, qy);                            // :310
        normal_pair(a.seed, f, (unsigned)t, SIM_STREAM_MARKER, (unsigned)i, z0, z1);
        a.mx[o * m + slot] = qx + p.tube_var + p.marker_sigma * z0;                      // :311
        a.my[o * m + slot] = qy + p.tube_var + p.marker_sigma * z1;                      // :312
        a.ids[o * m + slot] = i + 1;
    }
    for (int slot = kept + tid; slot < m; slot += 256) {
        a.mx[o * m + slot] = 0.0;
        a.my[o * m + slot] = 0.0;
        a.ids[o * m + slot] = -1;
    }
    if (tid == 0 && kept < m) atomicAdd(a.empty, (unsigned long long)(m - kept));
}

// ---------------------------------------------------------------- the lidar
// simulate_lidar_scanner, nuturtlesim/src/tube_world.cpp:405-471, as it stands (the author's "still need to fix lidar
// function" included): for every tube, 54 one-degree rays around round(rad2deg(atan2(yt - y1, xt - x1))) with
// (x1, y1) = robot - tube, the ray / circle intersection in the tube's frame (:430-452; `dy / fabs(dy)` is NaN for a
// horizontal ray, and such a ray then never updates a range), the scan index (i - int(rad2deg(theta))) mod 360,
// ranges stored as float and initialised to max_scan_range + 1.  No walls, no noise, no minimum range: the reference
// function has none.  One workgroup per (tick, filter); the (tube, ray) pairs are spread over the lanes and the
// reference's sequential "keep the smaller" becomes a 64-bit unsigned minimum in LDS (non-negative doubles order
// like their bit patterns), converted to float once at the end.
__global__ __launch_bounds__(256) void k_sim_scan(SimArg a, float* __restrict__ scans)
{
    const double PI = 3.14159265358979323846;
    const int t = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x;
    const nuslam_sim_params& p = a.p;
    __shared__ unsigned long long sh_min[360];
    const size_t o = (size_t)b * a.ticks + t;
    const double th = a.truth[3 * o + 0], x = a.truth[3 * o + 1], y = a.truth[3 * o + 2];
    const double max_scan = p.lidar_max_range;
    const float init = (float)(max_scan + 1);                                   // :416
    for (int k = tid; k < 360; k += 256) sh_min[k] = (unsigned long long)__double_as_longlong((double)init);
    __syncthreads();
    const int shift = (int)(((double)180 / PI) * th);                           // int(rad2deg(theta)), :458
    const long long items = (long long)a.n_world * 54;
    for (long long it = tid; it < items; it += 256) {
        const int tube = (int)(it / 54), ray = (int)(it % 54);
        const double xt = a.lm[2 * tube], yt = a.lm[2 * tube + 1];
        const double x1 = x - xt, y1 = y - yt;                                   // :423-424
        const int tube_angle = (int)round(((double)180 / PI) * atan2(yt - y1, xt - x1));   // :426
        const int i = tube_angle - 27 + ray;                                     // :428
        const double x2 = x1 + max_scan * cos((PI / (double)180) * i);
        const double y2 = y1 + max_scan * sin((PI / (double)180) * i);
        const double dx = x2 - x1, dy = y2 - y1;
        const double dr = sqrt((dx * dx) + (dy * dy));
        const double det = x1 * y2 - x2 * y1;
        const double r2 = p.tube_radius * p.tube_radius;
        const double dis = (r2 * (dr * dr)) - (det * det);
        double distance;
        if (fabs(dis) < 1e-5) {                                                  // :438-442
            const double ix = (det * dy) / (dr * dr);
            const double iy = -(det * dx) / (dr * dr);
            distance = sqrt(((ix - x1) * (ix - x1)) + ((iy - y1) * (iy - y1)));
        } else if (dis > 0) {                                                    // :443-453
            const double root = sqrt((r2 * (dr * dr)) - (det * det));
            const double ix1 = ((det * dy) + ((dy / fabs(dy)) * dx * root)) / (dr * dr);
            const double iy1 = (-(det * dx) + fabs(dy) * root) / (dr * dr);
            const double d1 = sqrt(((ix1 - x1) * (ix1 - x1)) + ((iy1 - y1) * (iy1 - y1)));
            const double ix2 = ((det * dy) - ((dy / fabs(dy)) * dx * root)) / (dr * dr);
            const double iy2 = (-(det * dx) - fabs(dy) * root) / (dr * dr);
            const double d2 = sqrt(((ix2 - x1) * (ix2 - x1)) + ((iy2 - y1) * (iy2 - y1)));
            distance = (d2 < d1) ? d2 : d1;                                      // std::min(dist1, dist2)
        } else {
            distance = max_scan + 1;                                             // :455
        }
        int ind = (i - shift) % 360;                                             // :458
        if (ind < 0) ind += 360;
        if (distance < (double)init)                                             // NaN never passes, as in `distance < ranges[ind]`
            atomicMin(&sh_min[ind], (unsigned long long)__double_as_longlong(distance));
    }
    __syncthreads();
    for (int k = tid; k < 360; k += 256)
        scans[o * 360 + k] = (float)__longlong_as_double((long long)sh_min[k]);  // :461-463 store as float
}

} // namespace nuslam
