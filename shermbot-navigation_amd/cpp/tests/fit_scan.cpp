// fit_scan.cpp -- drives the host-side circle_fit mirror the way the reference's landmarks node does
// (nuslam/src/landmarks.cpp:60-110): clusterPoints on one 360-ray scan, then classifyCluster / circleFit per cluster
// (here: all clusters in one GPU launch through fitClusters, plus the single-cluster entry points on the first one).
// stdin: 360 ranges.  stdout: one line per cluster "C n is_circle x y radius", then "M id x y scale" for circleFit().
#include <cstdio>
#include <iostream>
#include <vector>

#include "nuslam/circle_fit_library.hpp"

int main()
{
    std::vector<float> ranges(360);
    for (auto& r : ranges)
        if (!(std::cin >> r)) return 2;
    const auto clusters = circle_fit::clusterPoints(ranges, 0.05, 1.0);     // nuturtlesim/config/scan_params.yaml:1-4
    const auto fits = circle_fit::fitClusters(clusters);
    for (std::size_t c = 0; c < clusters.size(); ++c)
        std::printf("C %zu %d %.17g %.17g %.17g\n", clusters[c].size(), fits[c].is_circle ? 1 : 0, fits[c].x, fits[c].y,
                    fits[c].radius);
    if (!clusters.empty()) {
        const circle_fit::Marker m = circle_fit::circleFit(clusters[0]);
        std::printf("M %d %.17g %.17g %.17g %d\n", m.id, m.pose.position.x, m.pose.position.y, m.scale.x,
                    circle_fit::classifyCluster(clusters[0]) ? 1 : 0);
    }
    return 0;
}
