"""Landmark extraction front end: circleFit / classifyCluster / clusterPoints.
The oracle is PINNED by the reference's own known answers (nuslam/tests/circle_tests.cpp:38-40, 67-69); the GPU path is
compared with the oracle on seeded clusters."""
import numpy as np
import pytest

import _oracle as O


def arc(cx, cy, r, a0, a1, n, noise=0.0, seed=0):
    rng = np.random.default_rng(seed)
    a = np.linspace(a0, a1, n)
    return cx + r * np.cos(a) + noise * rng.normal(size=n), cy + r * np.sin(a) + noise * rng.normal(size=n)


def test_reference_known_answers():
    # nuslam/tests/circle_tests.cpp:13-40 -- six points -> centre (4.615482, 2.807354), R 4.827575 (Catch Approx: 1.2e-5)
    st, x, y, r = O.circle_fit([1, 2, 5, 7, 9, 3], [7, 6, 8, 7, 5, 7])
    assert st == 0
    assert x == pytest.approx(4.615482, rel=1.2e-5) and y == pytest.approx(2.807354, rel=1.2e-5)
    assert r == pytest.approx(4.827575, rel=1.2e-5)
    # :44-69 -- four points -> centre (0.4908357, -22.15212), R 22.17979
    st, x, y, r = O.circle_fit([-1, -0.3, 0.3, 1], [0, -0.06, 0.1, 0])
    assert st == 0
    assert x == pytest.approx(0.4908357, rel=1.2e-5) and y == pytest.approx(-22.15212, rel=1.2e-5)
    assert r == pytest.approx(22.17979, rel=1.2e-5)


def test_too_few_points_and_exact_circle():
    assert O.circle_fit([0, 1, 2], [0, 1, 0])[0] == 1                 # marker.id = -1, circle_fit_library.cpp:73-77
    xs, ys = arc(0.7, -0.2, 0.0381, 0.3, 2.9, 25)                     # tube radius of tube_world_params.yaml
    st, x, y, r = O.circle_fit(xs, ys)                                # noise-free: sigma_4 -> 0, the :79-81 branch or close to it
    assert st == 0 and abs(x - 0.7) < 1e-9 and abs(y + 0.2) < 1e-9 and abs(r - 0.0381) < 1e-9


def test_noisy_arcs_recover_the_circle():
    for k in range(20):
        rng = np.random.default_rng(100 + k)
        cx, cy, r = rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.03, 0.3)
        xs, ys = arc(cx, cy, r, 0.2, 2.6, 12 + k, noise=1e-4, seed=k)
        st, x, y, rr = O.circle_fit(xs, ys)
        assert st == 0 and abs(x - cx) < 5e-3 and abs(y - cy) < 5e-3 and abs(rr - r) < 5e-3


def test_classify_cluster():
    # points of a circle seen from its chord: constant inscribed angle -> a circle; a straight wall is not
    xs, ys = arc(0.0, 0.0, 0.05, 0.1, 3.0, 15)
    ok, sd = O.classify_cluster(xs, ys)
    assert ok and sd < 1e-6
    rng = np.random.default_rng(3)
    xs = np.linspace(0, 1, 15) + 0.02 * rng.normal(size=15)
    ys = 0.5 + 0.05 * rng.normal(size=15)
    ok, sd = O.classify_cluster(xs, ys)
    assert not ok and sd > 10


def scan_with_tubes(tubes, max_range=1.0):
    """360-ray scan of circular tubes around the origin (the lidar model of nuturtlesim/src/tube_world.cpp:405-471)."""
    r = np.full(360, 3.5, dtype=np.float32)
    for (cx, cy, rad) in tubes:
        for a in range(360):
            d = np.array([np.cos(np.deg2rad(a)), np.sin(np.deg2rad(a))])
            b = d @ np.array([cx, cy])
            disc = b * b - (cx * cx + cy * cy - rad * rad)
            if disc >= 0 and b - np.sqrt(disc) > 0:
                r[a] = min(r[a], b - np.sqrt(disc))
    return r


def test_cluster_points_then_fit():
    tubes = [(0.5, 0.2, 0.0381), (-0.3, 0.4, 0.0381), (0.1, -0.6, 0.0381)]
    ranges = scan_with_tubes(tubes)
    clusters = O.cluster_points(ranges, 0.05, 1.0)
    assert len(clusters) == 3
    found = []
    for xs, ys in clusters:
        assert len(xs) >= 3
        if len(xs) >= 4:
            st, x, y, r = O.circle_fit(xs, ys)
            assert st == 0
            found.append((x, y, r))
    for (cx, cy, rad) in tubes:
        assert any(abs(x - cx) < 0.02 and abs(y - cy) < 0.02 for x, y, _ in found)


def test_cluster_points_quirks():
    # every ray out of range -> no clusters; a 2-point blob is discarded (:197-204)
    assert O.cluster_points(np.full(360, 3.5, dtype=np.float32), 0.05, 1.0) == []
    r = np.full(360, 3.5, dtype=np.float32)
    r[10:12] = 0.5
    assert O.cluster_points(r, 0.05, 1.0) == []
    r[100:106] = 0.6
    cl = O.cluster_points(r, 0.05, 1.0)
    assert len(cl) == 1 and len(cl[0][0]) == 6


@pytest.mark.gpu
def test_gpu_batch_matches_oracle(hip):
    clusters = [([1, 2, 5, 7, 9, 3], [7, 6, 8, 7, 5, 7]), ([-1, -0.3, 0.3, 1], [0, -0.06, 0.1, 0]), ([0, 1, 2], [0, 1, 0])]
    for k in range(200):
        rng = np.random.default_rng(500 + k)
        n = int(rng.integers(4, 90 if k % 10 else 360))
        cx, cy, r = rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.03, 0.3)
        clusters.append(arc(cx, cy, r, 0.1, rng.uniform(1.5, 5.5), n, noise=10 ** rng.uniform(-5, -2.5), seed=k))
    out = hip.circle_fit_batch(clusters)
    # the reference's known answers through the GPU path
    assert out["cx"][0] == pytest.approx(4.615482, rel=1.2e-5) and out["radius"][0] == pytest.approx(4.827575, rel=1.2e-5)
    assert out["cy"][1] == pytest.approx(-22.15212, rel=1.2e-5) and out["radius"][1] == pytest.approx(22.17979, rel=1.2e-5)
    assert out["status"][2] == 1
    worst = 0.0
    for c, (xs, ys) in enumerate(clusters):
        st, x, y, r = O.circle_fit(xs, ys)
        assert out["status"][c] == st
        if st == 0:
            e = max(abs(out["cx"][c] - x), abs(out["cy"][c] - y), abs(out["radius"][c] - r)) / max(abs(r), 1e-3)
            worst = max(worst, e)
        if len(xs) >= 3:
            ok, sd = O.classify_cluster(xs, ys)
            assert abs(out["angle_std"][c] - sd) <= 1e-9 * max(1.0, sd)
            if abs(sd - 10) > 1e-6:
                assert bool(out["is_circle"][c]) == ok
    print("circle fit: worst relative deviation GPU vs oracle %.2e over %d clusters" % (worst, len(clusters)))
    assert worst < 1e-7


@pytest.mark.gpu
def test_gpu_empty_and_large_batch(hip):
    assert hip.circle_fit_batch([])["cx"].size == 0
    clusters = [arc(0.1 * (k % 7), -0.05 * (k % 5), 0.04 + 0.001 * (k % 11), 0.3, 2.8, 20 + k % 40, noise=1e-4, seed=k)
                for k in range(6144)]                    # 1024 filters x 6 tubes
    out = hip.circle_fit_batch(clusters)
    assert (out["status"] == 0).all() and np.isfinite(out["radius"]).all()
    assert np.abs(out["radius"] - np.array([0.04 + 0.001 * (k % 11) for k in range(6144)])).max() < 2e-3
    print("6144 clusters: %.3f ms on the device -> %.2f M fits/s" % (out["kernel_ms"], 6144 / out["kernel_ms"] / 1e3))


@pytest.mark.gpu
def test_cpp_host_mirror_of_circle_fit(hip):
    """cpp/tests/fit_scan: circle_fit::clusterPoints (host C++) + fitClusters / circleFit / classifyCluster (GPU)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "shermbot-navigation_amd", "cpp", "tests", "fit_scan")
    assert os.path.exists(exe), "build it first: make -C shermbot-navigation_amd/cpp"
    tubes = [(0.5, 0.2, 0.0381), (-0.3, 0.4, 0.0381), (0.1, -0.6, 0.0381)]
    ranges = scan_with_tubes(tubes)
    out = subprocess.run([exe], input=" ".join("%.9g" % r for r in ranges) + "\n", capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = [l.split() for l in out.stdout.splitlines()]
    ref = O.cluster_points(ranges, 0.05, 1.0)
    crow = [r for r in rows if r[0] == "C"]
    assert len(crow) == len(ref) == 3
    for r, (xs, ys) in zip(crow, ref):
        assert int(r[1]) == len(xs)
        st, x, y, rad = O.circle_fit(xs, ys)
        ok, _ = O.classify_cluster(xs, ys)
        assert int(r[2]) == int(ok)
        assert abs(float(r[3]) - x) < 1e-9 and abs(float(r[4]) - y) < 1e-9 and abs(float(r[5]) - rad) < 1e-9
    m = next(r for r in rows if r[0] == "M")
    st, x, y, rad = O.circle_fit(*ref[0])
    assert abs(float(m[2]) - x) < 1e-9 and abs(float(m[4]) - 2 * rad) < 1e-9      # scale.x is the diameter (:124)


def _ragged_scan(seed):
    """A 360-ray scan with everything the cluster walk reacts to: out-of-range rays inside runs, runs of one and two
    rays (discarded, and exempting their successor), a run through the 359 -> 0 seam, jumps right at the 0.04 gap."""
    rng = np.random.default_rng(seed)
    r = np.full(360, 2.0, dtype=np.float32)
    a = int(rng.integers(0, 20))
    while a < 360:
        n = int(rng.choice([1, 1, 2, 2, 3, 4, 7, 15, 30]))
        base = rng.uniform(0.06, 0.98)
        seg = base + np.cumsum(rng.choice([0.0, 0.01, -0.01, 0.039, 0.041, -0.05], size=n, p=[.5, .2, .2, .04, .03, .03]))
        hi = min(360, a + n)
        r[a:hi] = seg[:hi - a].astype(np.float32)
        a = hi + int(rng.choice([0, 0, 1, 1, 2, 5, 12]))
    if seed % 3 == 0:
        r[355:360] = 0.5; r[0:4] = 0.5          # a tube across the seam
    if seed % 5 == 0:
        r[rng.integers(0, 360, size=6)] = 1.2   # out-of-range rays dropped into runs
    return r


def test_host_cluster_points_matches_oracle_on_ragged_scans():
    """cpp/nuslam/circle_fit_library.hpp::clusterPoints (index spans) against the oracle's restatement of the
    reference walk (circle_fit_library.cpp:136-206), point for point, incl. the wrap-around and erase-loop quirks."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "shermbot-navigation_amd", "cpp", "tests", "fit_scan")
    assert os.path.exists(exe), "build it first: make -C shermbot-navigation_amd/cpp"
    scans = [_ragged_scan(s) for s in range(60)]
    scans.append(np.full(360, 2.0, dtype=np.float32))                  # nothing in range
    scans.append(np.full(360, 0.5, dtype=np.float32))                  # one run that never closes: everything is lost
    txt = "\n".join(" ".join("%.9g" % x for x in sc) for sc in scans) + "\n"
    out = subprocess.run([exe, "--clusters"], input=txt, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.split("\n")
    pos, small_seen, wrap_seen = 0, 0, 0
    for sc in scans:
        ref = O.cluster_points(sc, 0.05, 1.0)
        assert lines[pos].split() == ["S", str(len(ref))], (lines[pos], len(ref))
        pos += 1
        for xs, ys in ref:
            k = lines[pos].split(); pos += 1
            assert k[0] == "K" and int(k[1]) == len(xs)
            small_seen += len(xs) < 3
            got = np.array([[float(v) for v in lines[pos + i].split()] for i in range(len(xs))]).reshape(-1, 2)
            pos += len(xs)
            assert np.array_equal(got[:, 0], xs) and np.array_equal(got[:, 1], ys)
            wrap_seen += len(xs) > 0 and abs(np.arctan2(ys[-1], xs[-1]) + np.deg2rad(1)) < 1e-6
    assert small_seen > 0 and wrap_seen > 0, "the scans must exercise the exempted small clusters and the seam"


def test_classify_degenerate_clusters_oracle():
    """1- and 2-point clusters survive clusterPoints' erase loop; the reference's classifyCluster then divides 0 by
    angles.size() == 0: NaN, not a circle (circle_fit_library.cpp:236-249)."""
    for n in (1, 2):
        ok, sd = O.classify_cluster(np.arange(n) * 0.01 + 0.3, np.zeros(n) + 0.1)
        assert not ok and np.isnan(sd)


@pytest.mark.gpu
def test_classify_degenerate_clusters_gpu(hip):
    cl = [(np.array([0.3]), np.array([0.1])), (np.array([0.3, 0.31]), np.array([0.1, 0.1])),
          arc(0.5, 0.2, 0.04, 0.3, 2.8, 12)]
    out = hip.circle_fit_batch(cl)
    assert list(out["is_circle"]) == [False, False, True]
    assert np.isnan(out["angle_std"][0]) and np.isnan(out["angle_std"][1])
    assert list(out["status"][:2]) == [1, 1]                       # circleFit: fewer than four points
