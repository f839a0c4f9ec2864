// node_loop_rate.cpp -- api_rate.cpp plus what the node does at the TOP of every iteration (slam.cpp:247-251): broadcast_map2odom_tf()
// reads getStateVector() (:184) and the loop copies it again (:250) -- a getter, hence a synchronising read of the whole state
// vector, once per tick.  The rate the unchanged node would see with its own loop top: slam_library::ExtendedKalman driven call by call exactly
// as nuslam/src/slam.cpp:250-319 drives it (getSeenLandmarks, predict, then per marker cartesian2polar ->
// associateLandmark -> initializeLandmark / skip / stop -> update), every call crossing the C ABI on its own -- no
// nuslam_ekf_tick, no resident trace.  bench.py runs it as a child process and quotes the figure as `api_driven`.
//
// stdin : n ticks m known(0|1) q r      (landmarks, timed ticks, markers per tick, known ids?, diagonal of Q, of R)
//         n lines "x y id"              (the map-initialising markers, applied untimed)
//         per tick "dth dx" and m lines "x y id"      (id is ignored when known == 0)
// stdout: one JSON object
#include <chrono>
#include <cstdio>
#include <iostream>
#include <vector>

#include "nuslam/slam_library.hpp"

int main()
{
    using namespace slam_library;
    int n, ticks, m, known;
    double q, r;
    if (!(std::cin >> n >> ticks >> m >> known >> q >> r)) return 2;
    std::vector<double> wx(n), wy(n);
    std::vector<int> wid(n);
    for (int i = 0; i < n; ++i) std::cin >> wx[i] >> wy[i] >> wid[i];
    std::vector<double> dth(ticks), dx(ticks), mx((size_t)ticks * m), my((size_t)ticks * m);
    std::vector<int> ids((size_t)ticks * m);
    for (int t = 0; t < ticks; ++t) {
        std::cin >> dth[t] >> dx[t];
        for (int i = 0; i < m; ++i) std::cin >> mx[(size_t)t * m + i] >> my[(size_t)t * m + i] >> ids[(size_t)t * m + i];
    }
    if (!std::cin) return 2;

    colvec robot_state(3), map_state(2 * n);
    mat Q(3, 3), R(2, 2);
    for (int i = 0; i < 3; ++i) Q(i, i) = q;
    for (int i = 0; i < 2; ++i) R(i, i) = r;
    ExtendedKalman ekf;
    ekf = ExtendedKalman(robot_state, map_state, Q, R);                  // slam.cpp:81,157
    const int total = n;
    try {
        rigid2d::Twist2D tw0;
        tw0.dth = 0.0; tw0.dx = 0.0; tw0.dy = 0.0;
        ekf.tick(tw0, wx, wy, wid, total, false);                        // the map, untimed
        ekf.sync();
        long long updates = 0, associations = 0;
        double tf_sum = 0.0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < ticks; ++t) {
            {                                                            // broadcast_map2odom_tf(), slam.cpp:175-210
                const colvec& se = ekf.getStateVector();                 // :184
                const double odom[3] = { 0.0, 0.0, 0.0 }, pose[3] = { se(0), se(1), se(2) };
                double tf[3];
                nuslam_map_to_odom(odom, pose, tf);
                tf_sum += tf[0] + tf[1] + tf[2];
            }
            const colvec& state_estimation = ekf.getStateVector();       // :250 (the same tick: the mirror is still valid)
            tf_sum += state_estimation(0);
            const int seen_landmarks = ekf.getSeenLandmarks();           // slam.cpp:251
            rigid2d::Twist2D twist;
            twist.dth = dth[t]; twist.dx = dx[t]; twist.dy = 0.0;
            ekf.predict(twist);                                          // :269
            for (int i = 0; i < m; ++i) {                                // :279
                colvec z_i = cartesian2polar(mx[(size_t)t * m + i], my[(size_t)t * m + i]);   // :286
                int id;
                if (known) id = ids[(size_t)t * m + i];
                else { id = ekf.associateLandmark(z_i); ++associations; }                       // :291
                if (id > seen_landmarks) ekf.initializeLandmark(z_i, id);                       // :295-297
                else if (id < 0) continue;                                                      // :298-300
                else if (id > total) break;                                                     // :301-316
                ekf.update(twist, z_i, id);                                                     // :318
                ++updates;
            }
        }
        ekf.sync();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("{\"ticks\": %d, \"updates\": %lld, \"associations\": %lld, \"seconds\": %.6f, \"updates_per_s\": %.1f, "
                    "\"ticks_per_s\": %.1f, \"seen\": %d, \"tf_checksum\": %.6f}\n", ticks, updates, associations, dt, updates / dt, ticks / dt,
                    ekf.getSeenLandmarks(), tf_sum);
    } catch (const std::exception& e) {
        std::printf("{\"error\": \"%s\"}\n", e.what());
        return 3;
    }
    return 0;
}
