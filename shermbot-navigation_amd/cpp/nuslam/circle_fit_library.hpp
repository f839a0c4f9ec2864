// circle_fit_library.hpp -- host-side mirror of the reference's landmark-extraction library
// (nuslam/include/nuslam/circle_fit_library.hpp:18-28, nuslam/src/circle_fit_library.cpp) over the C ABI.
//
// clusterPoints stays on the host (a sequential walk over 360 rays, control logic); circleFit and classifyCluster run
// on the GPU, one wave per cluster.  The reference signatures use ROS message types (geometry_msgs::Point,
// visualization_msgs::Marker); when those headers are available they are used unchanged, otherwise the two small
// structs below carry the same fields the library touches (Point x/y/z; Marker id, pose.position, scale).
// fitClusters() is the batched entry point a caller with many clusters (all scans of all Monte-Carlo filters) wants.
#ifndef NUSLAM_HIP_CIRCLE_FIT_LIBRARY_HPP
#define NUSLAM_HIP_CIRCLE_FIT_LIBRARY_HPP

#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "nuslam_hip.h"
#include "rigid2d/rigid2d.hpp"

#if defined(__has_include)
#if __has_include(<geometry_msgs/Point.h>) && __has_include(<visualization_msgs/Marker.h>)
#include <geometry_msgs/Point.h>
#include <visualization_msgs/Marker.h>
#define NUSLAM_HIP_WITH_ROS_MSGS 1
#endif
#endif

namespace circle_fit
{
#ifdef NUSLAM_HIP_WITH_ROS_MSGS
    typedef geometry_msgs::Point Point;
    typedef visualization_msgs::Marker Marker;
#else
    struct Point { double x = 0, y = 0, z = 0; };
    struct Marker
    {
        int id = 0;
        std::string ns;
        struct { Point position; } pose;
        struct { double x = 0, y = 0, z = 0; } scale;
    };
#endif

    struct Fit { double x, y, radius; bool is_circle; double angle_std_dev; int status; };

    /// all clusters in one launch: circleFit (:15-134) + classifyCluster (:208-250) per cluster
    inline std::vector<Fit> fitClusters(const std::vector<std::vector<Point>>& clusters, int device = 0)
    {
        const int n = static_cast<int>(clusters.size());
        std::vector<int> off(n + 1, 0), status(n), circ(n);
        std::vector<double> xs, ys, cx(n), cy(n), r(n), sd(n);
        for (int c = 0; c < n; ++c) {
            off[c + 1] = off[c] + static_cast<int>(clusters[c].size());
            for (const auto& p : clusters[c]) { xs.push_back(p.x); ys.push_back(p.y); }
        }
        if (xs.empty()) { xs.push_back(0); ys.push_back(0); }
        const int rc = nuslam_circle_fit_batch(n, off.data(), xs.data(), ys.data(), cx.data(), cy.data(), r.data(),
                                               status.data(), circ.data(), sd.data(), device, nullptr);
        if (rc != NUSLAM_OK) throw std::runtime_error(std::string("circle_fit: ") + nuslam_strerror(rc));
        std::vector<Fit> out(n);
        for (int c = 0; c < n; ++c) out[c] = Fit{ cx[c], cy[c], r[c], circ[c] != 0, sd[c], status[c] };
        return out;
    }

    /// cylinder marker of one cluster (circle_fit_library.cpp:15-134); marker.id = -1 for fewer than four points
    inline Marker circleFit(std::vector<Point> data)
    {
        const Fit f = fitClusters({ data })[0];
        Marker marker;
        if (f.status == 1) { marker.id = -1; return marker; }      // :73-77
        marker.ns = "real";
        marker.pose.position.x = f.x;
        marker.pose.position.y = f.y;
        marker.pose.position.z = 0.25;
        marker.scale.x = 2 * f.radius;                               // :124-126
        marker.scale.y = 2 * f.radius;
        marker.scale.z = 0.5;
        return marker;
    }

    /// circle / not circle by the spread of the inscribed angles (circle_fit_library.cpp:208-250)
    inline bool classifyCluster(std::vector<Point> cluster) { return fitClusters({ cluster })[0].is_circle; }

    /// group the rays of one 360-degree scan into clusters (circle_fit_library.cpp:136-206), host side
    inline std::vector<std::vector<Point>> clusterPoints(std::vector<float> ranges, double minRange, double maxRange)
    {
        std::vector<std::vector<Point>> clusters;
        std::vector<Point> current_cluster;
        const double threshold = 0.04;
        int curr_angle = 0;
        while (curr_angle < 360) {
            if ((ranges[curr_angle] > maxRange) || (ranges[curr_angle] < minRange)) { curr_angle += 1; continue; }
            const int next_angle = (curr_angle + 1) % 360;
            const double curr_dist = ranges[curr_angle], next_dist = ranges[next_angle];
            Point point;
            point.x = ranges[curr_angle] * std::cos(rigid2d::deg2rad(curr_angle));
            point.y = ranges[curr_angle] * std::sin(rigid2d::deg2rad(curr_angle));
            if (std::fabs(curr_dist - next_dist) < threshold) {
                if (next_angle < curr_angle) {                       // 359 -> 0 wrap-around joins the first cluster
                    if (!clusters.empty()) clusters[0].push_back(point);
                } else {
                    current_cluster.push_back(point);
                    curr_angle += 1;
                }
            } else {
                current_cluster.push_back(point);
                clusters.push_back(current_cluster);
                current_cluster.clear();
                curr_angle += 1;
            }
            if (next_angle < curr_angle) break;
        }
        for (std::size_t i = 0; i < clusters.size(); i++)            // :197-204 (erase shifts, the index still advances)
            if (clusters[i].size() < 3) clusters.erase(clusters.begin() + i);
        return clusters;
    }
}

#endif
