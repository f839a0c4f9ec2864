// replay_slam_loop.cpp -- drives the host-side slam_library::ExtendedKalman exactly the way the reference's slam
// node does (nuslam/src/slam.cpp:231-319): DiffDrive::getTwist -> DiffDrive::operator() -> predict -> per marker
// cartesian2polar -> associateLandmark -> initializeLandmark / skip / stop -> update.
//
// stdin : n total_landmarks ticks m wheel_base wheel_radius, then per tick: thL thR followed by m (x y) pairs
// stdout: per tick "T seen th x y" and "M x y yaw" (map -> odom, slam.cpp:175-210), at the end "S <len state values>" and "P <len*len covariance, column-major>"
#include <cmath>
#include <cstdio>
#include <iostream>
#include <vector>

#include "nuslam/slam_library.hpp"

int main()
{
    using namespace slam_library;
    int n, total, ticks, m;
    double base, rad;
    if (!(std::cin >> n >> total >> ticks >> m >> base >> rad)) return 2;

    colvec robot_state(3), map_state(2 * n);
    mat Q(3, 3), R(2, 2);
    for (int i = 0; i < 3; ++i) Q(i, i) = 0.1;        // nuslam/config/slam_params.yaml:2-3
    for (int i = 0; i < 2; ++i) R(i, i) = 1e-3;

    ExtendedKalman extended_kalman_filter;                                     // slam.cpp:81
    extended_kalman_filter = ExtendedKalman(robot_state, map_state, Q, R);     // slam.cpp:157
    rigid2d::DiffDrive odom_model(base, rad, 0.0, 0.0, 0.0, 0.0, 0.0);         // slam.cpp:240

    try {
        for (int t = 0; t < ticks; ++t) {
            double thL, thR;
            std::cin >> thL >> thR;
            std::vector<double> xs(m), ys(m);
            for (int i = 0; i < m; ++i) std::cin >> xs[i] >> ys[i];

            const int seen_landmarks = extended_kalman_filter.getSeenLandmarks();   // slam.cpp:251
            rigid2d::Twist2D twist = odom_model.getTwist(thL, thR);                 // :264
            odom_model(thL, thR);                                                   // :265
            extended_kalman_filter.predict(twist);                                  // :269
            for (int i = 0; i < m; ++i) {                                           // :279
                colvec z_i = cartesian2polar(xs[i], ys[i]);                         // :286
                int id = extended_kalman_filter.associateLandmark(z_i);             // :291
                if (id > seen_landmarks) extended_kalman_filter.initializeLandmark(z_i, id);   // :295-297
                else if (id < 0) continue;                                          // :298-300
                else if (id > total) break;                                         // :301-316
                extended_kalman_filter.update(twist, z_i, id);                      // :318
            }
            const colvec& s = extended_kalman_filter.getStateVector();
            std::printf("T %d %.17g %.17g %.17g\n", extended_kalman_filter.getSeenLandmarks(), s(0), s(1), s(2));
            // the map -> odom transform the node broadcasts at the top of its next iteration, slam.cpp:175-210
            rigid2d::Vector2D v;
            v.x = odom_model.getX(); v.y = odom_model.getY();
            const rigid2d::Transform2D T_ob(v, odom_model.getTh());
            v.x = s(1); v.y = s(2);
            const rigid2d::Transform2D T_mb(v, s(0));
            const rigid2d::Transform2D T_mo = T_mb * T_ob.inv();
            std::printf("M %.17g %.17g %.17g\n", T_mo.getX(), T_mo.getY(),
                        rigid2d::normalize_angle(std::asin(T_mo.getSinTh())));
        }
    } catch (const std::logic_error& e) {
        std::printf("E logic_error %s\n", e.what());
        return 3;
    } catch (const std::runtime_error& e) {
        std::printf("E runtime_error %s\n", e.what());
        return 4;
    }
    ExtendedKalman copy = extended_kalman_filter;          // value semantics: the copy owns its own device state
    const colvec& s = copy.getStateVector();
    const mat& P = copy.getCovariance();
    std::printf("S");
    for (std::size_t i = 0; i < s.n_elem; ++i) std::printf(" %.17g", s(i));
    std::printf("\nP");
    for (std::size_t i = 0; i < P.n_elem; ++i) std::printf(" %.17g", P(i));
    std::printf("\n");
    return 0;
}
