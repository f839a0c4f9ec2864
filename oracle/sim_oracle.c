/*
 * sim_oracle.c -- CPU ORACLE for the on-device Monte-Carlo trace generator.
 *
 * TEST INFRASTRUCTURE ONLY (see nuslam_oracle.h for the rules).
 *
 * What it restates: one iteration of the simulator's loop, nuturtlesim/src/tube_world.cpp:509-533, with the
 * pieces it calls -- the noisy commanded twist (twist_callback :177-189), the collision slide (check_collision
 * :371-389, DiffDrive::changeConfig diff_drive.cpp:154-159), DiffDrive::convertTwist (diff_drive.cpp:66-78), the
 * joint-angle integration (:520-521), the wheel-slip update of the true robot (:480-483, :526-527,
 * DiffDrive::operator() diff_drive.cpp:111-146) and the relative markers (set_rel_markers :270-329) -- followed
 * by what the slam node makes of the joint angles (DiffDrive::getTwist, slam.cpp:264-265).
 *
 * Where this generator departs from the reference, on purpose:
 *   - time: the loop period is the ideal dt = 1/frequency, not ros::Time differences (:520-521);
 *   - randomness: the reference draws from one process-wide std::mt19937 through std::normal_distribution
 *     (:49-62), whose output is implementation-defined; here every draw is a pure function of
 *     (seed, filter, tick, stream, index) through Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11) and
 *     Box-Muller, so B filters can be generated in any order on any number of devices;
 *   - markers: the reference publishes every tube and only flags the out-of-range ones DELETE (:300-307; the
 *     slam node never looks at the flag, slam.cpp:279); here out-of-range tubes are dropped, at most the m
 *     nearest are kept (in tube order, as the reference iterates them) and unused slots carry id -1, which the
 *     slam node's `id < 0 -> continue` (slam.cpp:298-300) skips; max_range <= 0 keeps the reference's
 *     "every tube every tick".  marker_sigma adds Gaussian noise on top of the constant tube_var offset.
 *
 * PARITY PINNING: Philox is pinned by the Random123 known-answer vectors (tests/test_sim.py); the rigid2d /
 * DiffDrive steps are the ones pinned in nuslam_oracle.c; the loop itself has no golden output in the reference
 * (it needs ROS to run) -- "parity unpinned" for the trace as a whole, it is this project's own generator.
 */
#include "nuslam_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* ------------------------------------------------------------------ Philox4x32-10 */
void orc_philox4x32_10(const unsigned ctr[4], const unsigned key[2], unsigned out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 53-bit uniform in (0, 1) from two words */
static double u53(uint32_t hi, uint32_t lo)
{
    const double k = (double)(((uint64_t)(hi >> 5) << 26) | (uint64_t)(lo >> 6));   /* < 2^53, exact */
    return (k + 0.5) * (1.0 / 9007199254740992.0);
}

/* two independent N(0,1) draws for (seed, filter, tick, stream, idx): Box-Muller on one Philox block */
void orc_sim_normal_pair(unsigned long long seed, unsigned filter, unsigned tick, unsigned stream, unsigned idx,
                         double z[2])
{
    const unsigned ctr[4] = { filter, tick, stream, idx };
    const unsigned key[2] = { (unsigned)(seed & 0xffffffffu), (unsigned)(seed >> 32) };
    unsigned r[4];
    orc_philox4x32_10(ctr, key, r);
    const double u1 = u53(r[0], r[1]), u2 = u53(r[2], r[3]);
    const double rad = sqrt(-2.0 * log(u1));
    const double ang = 6.283185307179586476925286766559 * u2;
    z[0] = rad * cos(ang);
    z[1] = rad * sin(ang);
}

enum { STREAM_TWIST = 0, STREAM_SLIP = 1, STREAM_MARKER = 2 };

long long orc_simulate(const orc_sim_params* p, const double* landmarks, int n, const double* cmd, int ticks, int m,
                       unsigned long long seed, unsigned filter, double* tw, double* mx, double* my, int* ids,
                       double* truth, double* joints)
{
    if (!p || !landmarks || !cmd || n < 0 || ticks < 1 || m < 0) return -1;
    /* the true robot and the slam node's odometry model both start at rest at the origin (:487-491, slam.cpp:240) */
    double turtle[7] = { p->wheel_base, p->wheel_radius, 0.0, 0.0, 0.0, 0.0, 0.0 };
    double odom[7] = { p->wheel_base, p->wheel_radius, 0.0, 0.0, 0.0, 0.0, 0.0 };
    double jL = 0.0, jR = 0.0;                                                      /* :497-498 */
    const double slip_mean = (p->slip_min + p->slip_max) / 2;                       /* :480 */
    const double slip_var = p->slip_max - slip_mean;                                /* :481 */
    double* dist = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    unsigned char* keep = (unsigned char*)malloc((size_t)(n > 0 ? n : 1));
    long long empty = 0;

    for (int t = 0; t < ticks; ++t) {
        double z[2];
        /* twist_callback :177-189 */
        orc_sim_normal_pair(seed, filter, (unsigned)t, STREAM_TWIST, 0, z);
        double desired[3];
        desired[0] = cmd[2 * t + 0] + p->twist_noise * z[0];
        desired[1] = cmd[2 * t + 1] + p->twist_noise * z[1];
        desired[2] = 0.0;
        /* check_collision :371-389 (the pose moves inside the loop, later tubes see the moved pose) */
        for (int i = 0; i < n; ++i) {
            const double dx = landmarks[2 * i] - turtle[2];
            const double dy = landmarks[2 * i + 1] - turtle[3];
            const double d = sqrt((dx * dx) + (dy * dy));
            if (d <= (p->tube_radius + p->robot_radius)) {
                const double move_x = dy / d;
                const double move_y = -dx / d;
                turtle[2] += move_x / 50;                                          /* changeConfig */
                turtle[3] += move_y / 50;
            }
        }
        /* :512-521 */
        double u[2];
        orc_dd_convert_twist(turtle, desired, u);
        jL += u[0] * p->dt;
        jR += u[1] * p->dt;
        /* :526-527 */
        orc_sim_normal_pair(seed, filter, (unsigned)t, STREAM_SLIP, 0, z);
        const double slipL = slip_mean + slip_var * z[0];
        const double slipR = slip_mean + slip_var * z[1];
        orc_dd_step(turtle, jL + u[0] * slipL, jR + u[1] * slipR);
        /* the slam node's side: slam.cpp:264-265 */
        double twb[3];
        orc_dd_get_twist(odom, jL, jR, twb);
        odom[5] = jL; odom[6] = jR;                 /* operator() stores the angles; its pose is not needed here */
        if (tw) { tw[2 * t] = twb[0]; tw[2 * t + 1] = twb[1]; }
        if (joints) { joints[2 * t] = jL; joints[2 * t + 1] = jR; }
        if (truth) { truth[3 * t] = turtle[4]; truth[3 * t + 1] = turtle[2]; truth[3 * t + 2] = turtle[3]; }

        /* set_rel_markers :270-329 */
        double T_wt[4], T_tw[4];
        orc_tf_make(turtle[2], turtle[3], turtle[4], T_wt);                         /* :274-275 */
        orc_tf_inv(T_wt, T_tw);                                                     /* :276 */
        int in_range = 0;
        for (int i = 0; i < n; ++i) {
            const double dx = landmarks[2 * i] - turtle[2];
            const double dy = landmarks[2 * i + 1] - turtle[3];
            dist[i] = sqrt((dx * dx) + (dy * dy));                                  /* :299 */
            keep[i] = (p->max_range <= 0.0 || !(dist[i] > p->max_range)) ? 1 : 0;   /* :300 */
            /* extension (fov > 0): within +-fov of the heading -- (tube - robot) . heading >= cos(fov) d */
            if (p->fov > 0.0 && (!((cos(turtle[4]) * dx) + (sin(turtle[4]) * dy) >= cos(p->fov) * dist[i]) || dist[i] < p->min_range)) keep[i] = 0;
            if (!keep[i]) dist[i] = INFINITY;        /* a gated tube takes no part in the ranking below */
            in_range += keep[i];
        }
        if (in_range > m) {                          /* keep the m nearest; ties go to the lower tube index */
            for (int i = 0; i < n; ++i) {
                if (!keep[i]) continue;
                int rank = 0;
                for (int j = 0; j < n; ++j) {
                    if (dist[j] < INFINITY && (dist[j] < dist[i] || (dist[j] == dist[i] && j < i))) ++rank;
                }
                if (rank >= m) keep[i] = 2;          /* in range but not kept */
            }
        }
        int slot = 0;
        for (int i = 0; i < n && slot < m; ++i) {
            if (keep[i] != 1) continue;
            double q[2];
            orc_tf_point(T_tw, landmarks[2 * i], landmarks[2 * i + 1], q);          /* :310 */
            orc_sim_normal_pair(seed, filter, (unsigned)t, STREAM_MARKER, (unsigned)i, z);
            if (mx) mx[(size_t)t * m + slot] = q[0] + p->tube_var + p->marker_sigma * z[0];   /* :311 */
            if (my) my[(size_t)t * m + slot] = q[1] + p->tube_var + p->marker_sigma * z[1];   /* :312 */
            if (ids) ids[(size_t)t * m + slot] = i + 1;
            ++slot;
        }
        for (; slot < m; ++slot) {
            if (mx) mx[(size_t)t * m + slot] = 0.0;
            if (my) my[(size_t)t * m + slot] = 0.0;
            if (ids) ids[(size_t)t * m + slot] = -1;
            ++empty;
        }
    }
    free(dist);
    free(keep);
    return empty;
}

/* ------------------------------------------------------------------ the lidar
 * simulate_lidar_scanner, nuturtlesim/src/tube_world.cpp:405-471, as it stands (the author's "still need to fix lidar
 * function" included): for every tube, 54 one-degree rays around round(rad2deg(atan2(yt - y1, xt - x1))) with
 * (x1, y1) = robot - tube -- not the bearing of the tube, but that is what the node computes --, the ray/circle
 * intersection in the tube's frame (:430-452; `dy / fabs(dy)` is NaN for a horizontal ray and such a ray then never
 * updates a range), the scan index (i - int(rad2deg(theta))) mod 360, ranges stored as float and initialised to
 * max_scan_range + 1.  No walls, no noise, no minimum range: the reference function has none either. */
void orc_sim_scan(const double* landmarks, int n, double tube_radius, double max_scan_range, double x, double y,
                  double th, float ranges[360])
{
    const double PI = 3.14159265358979323846;
    for (int k = 0; k < 360; ++k) ranges[k] = (float)(max_scan_range + 1);          /* :416 */
    for (int t = 0; t < n; ++t) {
        const double xt = landmarks[2 * t], yt = landmarks[2 * t + 1];
        const double x1 = x - xt, y1 = y - yt;                                       /* :423-424 */
        const int tube_angle = (int)round(((double)180 / PI) * atan2(yt - y1, xt - x1));   /* :426 */
        for (int i = tube_angle - 27; i < tube_angle + 27; ++i) {                    /* :428 */
            const double x2 = x1 + max_scan_range * cos((PI / (double)180) * i);
            const double y2 = y1 + max_scan_range * sin((PI / (double)180) * i);
            const double dx = x2 - x1, dy = y2 - y1;
            const double dr = sqrt((dx * dx) + (dy * dy));
            const double det = x1 * y2 - x2 * y1;
            const double dis = ((tube_radius * tube_radius) * (dr * dr)) - (det * det);
            double distance;
            if (fabs(dis) < 1e-5) {                                                  /* :438-442 */
                const double ix = (det * dy) / (dr * dr);
                const double iy = -(det * dx) / (dr * dr);
                distance = sqrt(((ix - x1) * (ix - x1)) + ((iy - y1) * (iy - y1)));
            } else if (dis > 0) {                                                    /* :443-453 */
                const double root = sqrt(((tube_radius * tube_radius) * (dr * dr)) - (det * det));
                const double ix1 = ((det * dy) + ((dy / fabs(dy)) * dx * root)) / (dr * dr);
                const double iy1 = (-(det * dx) + fabs(dy) * root) / (dr * dr);
                const double d1 = sqrt(((ix1 - x1) * (ix1 - x1)) + ((iy1 - y1) * (iy1 - y1)));
                const double ix2 = ((det * dy) - ((dy / fabs(dy)) * dx * root)) / (dr * dr);
                const double iy2 = (-(det * dx) - fabs(dy) * root) / (dr * dr);
                const double d2 = sqrt(((ix2 - x1) * (ix2 - x1)) + ((iy2 - y1) * (iy2 - y1)));
                distance = (d2 < d1) ? d2 : d1;                                      /* std::min(dist1, dist2) */
            } else {
                distance = max_scan_range + 1;                                       /* :455 */
            }
            int ind = (i - (int)(((double)180 / PI) * th)) % 360;                    /* :458 */
            if (ind < 0) ind += 360;
            if (distance < ranges[ind]) ranges[ind] = (float)distance;               /* :461-463 */
        }
    }
}

