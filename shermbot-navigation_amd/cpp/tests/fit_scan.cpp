// fit_scan.cpp -- drives the host-side circle_fit mirror the way the reference's landmarks node does
// (nuslam/src/landmarks.cpp:60-110): clusterPoints on one 360-ray scan, then classifyCluster / circleFit per cluster
// (here: all clusters in one GPU launch through fitClusters, plus the single-cluster entry points on the first one).
// `fit_scan --clusters`: clusterPoints only, host-side, for any number of scans (CPU test).
// stdin: 360 ranges.  stdout: one line per cluster "C n is_circle x y radius", then "M id x y scale" for circleFit().
#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

#include "nuslam/circle_fit_library.hpp"

int main(int argc, char** argv)
{
    std::vector<float> ranges(360);
    if (argc > 1 && std::string(argv[1]) == "--clusters") {
        // host-only mode (no GPU call): any number of scans on stdin; per scan "S n_clusters", then per cluster
        // "K n" followed by n lines "x y"
        while (true) {
            for (auto& r : ranges)
                if (!(std::cin >> r)) return 0;
            const auto cl = circle_fit::clusterPoints(ranges, 0.05, 1.0);
            std::printf("S %zu\n", cl.size());
            for (const auto& c : cl) {
                std::printf("K %zu\n", c.size());
                for (const auto& p : c) std::printf("%.17g %.17g\n", p.x, p.y);
            }
        }
    }
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
