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

    /// Group the rays of one 360-degree scan into clusters (behaviour of circle_fit_library.cpp:136-206), host side.
    /// Worked on ray indices: a cluster is the set of in-range rays in [first, last] (+ ray 359 when the scan wraps into
    /// the first cluster); points are only materialised for the clusters that survive.  The reference's observable
    /// quirks are kept: a cluster closes after ray a when |r[a] - r[(a+1) % 360]| >= 0.04 whether or not ray a+1 is in
    /// range; a wrapping ray 359 joins cluster 0 and whatever was still open is lost; the discard pass removes a
    /// small cluster and then exempts the one that follows it (:197-204 erase while the index advances).
    inline std::vector<std::vector<Point>> clusterPoints(std::vector<float> ranges, double minRange, double maxRange)
    {
        constexpr int kRays = 360;
        constexpr double kGap = 0.04;
        struct Span { int first, last, members; bool takes_wrap; };
        auto usable = [&](int a) { return !(ranges[a] > maxRange) && !(ranges[a] < minRange); };
        auto joined = [&](int a) {
            return std::fabs(static_cast<double>(ranges[a]) - static_cast<double>(ranges[(a + 1) % kRays])) < kGap;
        };

        // pass 1: spans of ray indices
        std::vector<Span> spans;
        int open_first = -1, open_members = 0;
        for (int a = 0; a < kRays; ++a) {
            if (!usable(a)) continue;
            const bool link = joined(a);
            if (a == kRays - 1 && link) {                            // wrap-around: ray 359 belongs to the first cluster
                if (!spans.empty()) { spans[0].takes_wrap = true; spans[0].members += 1; }
                open_first = -1;                                     // the open span is never closed: dropped
                break;
            }
            if (open_first < 0) { open_first = a; open_members = 0; }
            open_members += 1;
            if (!link) {
                spans.push_back(Span{ open_first, a, open_members, false });
                open_first = -1;
            }
        }

        // pass 2: the discard rule, as a keep mask
        std::vector<char> keep(spans.size(), 1);
        bool exempt = false;
        for (std::size_t k = 0; k < spans.size(); ++k) {
            if (exempt) { exempt = false; continue; }
            if (spans[k].members < 3) { keep[k] = 0; exempt = true; }
        }

        // pass 3: points of the survivors
        auto point_of = [&](int a) {
            Point p;
            p.x = ranges[a] * std::cos(rigid2d::deg2rad(a));
            p.y = ranges[a] * std::sin(rigid2d::deg2rad(a));
            return p;
        };
        std::vector<std::vector<Point>> clusters;
        for (std::size_t k = 0; k < spans.size(); ++k) {
            if (!keep[k]) continue;
            std::vector<Point> pts;
            pts.reserve(static_cast<std::size_t>(spans[k].members));
            for (int a = spans[k].first; a <= spans[k].last; ++a)
                if (usable(a)) pts.push_back(point_of(a));
            if (spans[k].takes_wrap) pts.push_back(point_of(kRays - 1));
            clusters.push_back(std::move(pts));
        }
        return clusters;
    }
}

#endif
