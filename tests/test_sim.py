"""The Monte-Carlo trace generator and the map->odom output of the tick protocol (f1).

CPU part: the oracle's Philox against the Random123 known answers, its Transform2D algebra against vectors produced by
the reference's own rigid2d (tests/golden/tf_ref.npz), and the generator's semantics (nuturtlesim/src/tube_world.cpp).
GPU part: the HIP generator against the oracle, and the filters run from a device-made trace against the same trace
loaded from the host.
"""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


# Random123 (Salmon et al., SC'11) kat_vectors for philox4x32 with 10 rounds: counter, key -> output
PHILOX_KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox_known_answers():
    for ctr, key, want in PHILOX_KAT:
        assert tuple(O.philox(ctr, key)) == want


def test_normal_pairs_are_standard_normal_and_keyed():
    z = np.array([O.sim_normal_pair(7, 3, t, 2, i) for t in range(64) for i in range(64)]).ravel()
    assert abs(z.mean()) < 0.05 and abs(z.std() - 1.0) < 0.05
    assert abs(np.mean(z ** 4) - 3.0) < 0.3
    a = O.sim_normal_pair(7, 3, 5, 2, 9)
    assert np.array_equal(a, O.sim_normal_pair(7, 3, 5, 2, 9))
    for other in ((8, 3, 5, 2, 9), (7, 4, 5, 2, 9), (7, 3, 6, 2, 9), (7, 3, 5, 1, 9), (7, 3, 5, 2, 10)):
        assert not np.array_equal(a, O.sim_normal_pair(*other))
    # seeds above 32 bits reach the second key word
    assert not np.array_equal(O.sim_normal_pair(1, 0, 0, 0, 0), O.sim_normal_pair(1 + (1 << 32), 0, 0, 0, 0))


def test_transform_algebra_matches_reference_golden_vectors():
    """tests/golden/tf_ref.npz holds outputs of the reference's own Transform2D: bit-exact."""
    g = np.load(os.path.join(GOLD, "tf_ref.npz"))
    L = O.lib()
    for i in range(g["a"].shape[0]):
        Ta = np.zeros(4); Tb = np.zeros(4); out = np.zeros(4); pt = np.zeros(2)
        L.orc_tf_make(*g["a"][i], _p(Ta))
        L.orc_tf_make(*g["b"][i], _p(Tb))
        L.orc_tf_inv(_p(Ta), _p(out))
        assert np.array_equal(out, g["inv"][i])
        L.orc_tf_mul(_p(Ta), _p(Tb), _p(out))
        assert np.array_equal(out, g["mul"][i])
        L.orc_tf_point(_p(Ta), g["pt"][i, 0], g["pt"][i, 1], _p(pt))
        assert np.array_equal(pt, g["point"][i])
        assert np.array_equal(O.map_to_odom(g["a"][i], g["b"][i, [2, 0, 1]]), g["m2o"][i])


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (no /root/reference here)")
def test_map_to_odom_matches_reference_live():
    rng = np.random.default_rng(5)
    R = O.ref()
    for _ in range(200):
        odom = rng.normal(size=3) * 2
        st = rng.normal(size=3) * 2
        want = np.zeros(3)
        R.ref_map_to_odom(_p(odom), _p(st), _p(want))
        assert np.array_equal(O.map_to_odom(odom, st), want)


def test_map_to_odom_is_identity_when_filter_agrees_with_odometry():
    out = O.map_to_odom([0.3, -0.2, 0.7], [0.7, 0.3, -0.2])
    assert np.allclose(out, 0.0, atol=1e-15)


TUBES = np.array([[0.5, 0.5], [-0.5, -0.5], [1.0, 1.0], [-1.0, -1.0], [-0.75, 0.75], [0.75, -0.75]])  # tube_world_params.yaml:4-9


def _cmd(T, dth=0.4, dx=0.1):
    c = np.zeros((T, 2)); c[:, 0] = dth; c[:, 1] = dx
    return c


def test_simulator_without_noise_reproduces_the_commanded_twist():
    """slip = 1 exactly and no twist noise: the odometry twist is cmd * dt and truth follows DiffDrive exactly."""
    p = O.SimParams(slip_min=1.0, slip_max=1.0, max_range=0.0, tube_var=0.0)
    T = 30
    cmd = _cmd(T)
    cmd[10:15, 0] = 0.0                                  # straight stretch: the dth == 0 branch
    s = O.simulate(p, TUBES + 5.0, cmd, m=6, seed=1)     # tubes far away: no collision
    assert s["empty"] == 0 and np.array_equal(s["ids"], np.tile(np.arange(1, 7), (T, 1)))
    assert np.allclose(s["tw"], cmd * p.dt, rtol=1e-9, atol=1e-15)
    # markers: every tube every tick, T_tw(tube) exactly (tube_var = 0, marker_sigma = 0)
    th, x, y = s["truth"][-1]
    d = TUBES + 5.0 - np.array([x, y])
    want_x = np.cos(th) * d[:, 0] + np.sin(th) * d[:, 1]
    assert np.allclose(s["mx"][-1], want_x, atol=1e-12)
    # truth = what DiffDrive::operator() integrates from (joint + u * slip) -- with slip 1 that is joint + u
    dd = np.array([p.wheel_base, p.wheel_radius, 0, 0, 0, 0, 0], dtype=np.float64)
    L = O.lib()
    jprev = np.zeros(2)
    for t in range(T):
        u = (s["joints"][t] - jprev) / p.dt
        L.orc_dd_step(_p(dd), s["joints"][t, 0] + u[0], s["joints"][t, 1] + u[1])
        jprev = s["joints"][t]
    assert np.allclose(dd[[4, 2, 3]], s["truth"][-1], atol=1e-9)


def test_simulator_range_gate_nearest_m_and_empty_slots():
    p = O.SimParams(max_range=1.0)
    s = O.simulate(p, TUBES, _cmd(300, 0.05, 0.5), m=3, seed=2)      # drives out of everyone's range
    for t in range(300):
        th, x, y = s["truth"][t]
        dist = np.hypot(TUBES[:, 0] - x, TUBES[:, 1] - y)
        inr = np.nonzero(dist <= 1.0)[0]
        keep = sorted(inr[np.argsort(dist[inr], kind="stable")[:3]])
        got = s["ids"][t]
        assert list(got[got > 0]) == [k + 1 for k in keep]
        assert np.all(got[len(keep):] == -1) and np.all(s["mx"][t, len(keep):] == 0.0)
        # the reference's constant offset (tube_world.cpp:311-312)
        for slot, k in enumerate(keep):
            bx = np.cos(th) * (TUBES[k, 0] - x) + np.sin(th) * (TUBES[k, 1] - y)
            assert abs(s["mx"][t, slot] - (bx + 0.001)) < 1e-12
    assert s["empty"] == int(np.sum(s["ids"] == -1)) > 0


def test_simulator_collision_slides_along_the_tangent():
    """Driving straight at a tube: once within tube_radius + robot_radius the pose is pushed sideways (:371-389)."""
    p = O.SimParams(slip_min=1.0, slip_max=1.0, max_range=0.0)
    tube = np.array([[1.0, 0.0]])
    s = O.simulate(p, tube, _cmd(250, 0.0, 0.2), m=1, seed=3)
    # the first tick jumps by one second's worth of wheel motion: the reference feeds joint + u * slip (an angle
    # plus a velocity times a factor, tube_world.cpp:526-527) to DiffDrive::operator(); kept as is
    assert abs(s["truth"][0, 1] - 0.2 * (1.0 + 0.02)) < 1e-12
    y = s["truth"][:, 2]
    assert np.all(y[:50] == 0.0) and np.abs(y).max() > 0.01
    free = O.simulate(p, tube + 50.0, _cmd(200, 0.0, 0.2), m=1, seed=3)
    assert np.all(free["truth"][:, 2] == 0.0)


def test_simulator_filters_are_independent_streams():
    p = O.SimParams(twist_noise=0.01, marker_sigma=0.002)
    a = O.simulate(p, TUBES, _cmd(20), m=6, seed=9, filt=0)
    b = O.simulate(p, TUBES, _cmd(20), m=6, seed=9, filt=1)
    a2 = O.simulate(p, TUBES, _cmd(20), m=6, seed=9, filt=0)
    assert np.array_equal(a["tw"], a2["tw"]) and np.array_equal(a["mx"], a2["mx"])
    assert not np.array_equal(a["tw"], b["tw"])
    # slip noise N(0.95, 0.05) (tube_world.cpp:480-483): truth drifts from the commanded arc but stays near it
    assert 1e-5 < np.abs(a["truth"] - b["truth"]).max() < 0.5


# ------------------------------------------------------------------ the C ABI's host helper (no GPU needed)
def test_capi_map_to_odom_matches_reference_golden_vectors():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), "..", "shermbot-navigation_amd"))
    import nuslam_hip as H
    g = np.load(os.path.join(GOLD, "tf_ref.npz"))
    for i in range(g["a"].shape[0]):
        assert np.array_equal(H.map_to_odom(g["a"][i], g["b"][i, [2, 0, 1]]), g["m2o"][i])


# ------------------------------------------------------------------ GPU: the HIP generator against the oracle
Q = np.diag([0.1, 0.1, 0.1])
R = np.diag([1e-3, 1e-3])


def _same_params(hip, **kw):
    return hip.SimParams(**kw), O.SimParams(**kw)


@pytest.mark.gpu
def test_device_philox_known_answers(hip):
    for ctr, key, want in PHILOX_KAT:
        assert tuple(hip.philox(ctr, key)) == want
    rng = np.random.default_rng(1)
    for _ in range(16):
        ctr = rng.integers(0, 2 ** 32, 4); key = rng.integers(0, 2 ** 32, 2)
        assert hip.philox(ctr, key) == O.philox(ctr, key)


def _check_trace(hip, B, lm, cmd, m, seed, first=0, **kw):
    ph, po = _same_params(hip, **kw)
    n = lm.shape[0]
    b = hip.Batch(B, n, Q, R)
    empty = b.simulate(ph, lm, cmd, m, seed, first_filter=first)
    total = 0
    for f in range(B):
        got = b.get_trace(f)
        want = O.simulate(po, lm, cmd, m, seed, filt=first + f)
        total += want["empty"]
        assert np.array_equal(got["ids"], want["ids"])
        # continuous outputs: libm's sin/cos/log differ from the device's by an ulp or so, nothing else does
        assert np.allclose(got["tw"], want["tw"], atol=1e-13, rtol=1e-12)
        assert np.allclose(got["truth"], want["truth"], atol=1e-12, rtol=0)
        assert np.allclose(got["mx"], want["mx"], atol=1e-12, rtol=0)
        assert np.allclose(got["my"], want["my"], atol=1e-12, rtol=0)
    assert empty == total
    return b


@pytest.mark.gpu
def test_device_trace_matches_oracle_reference_world(hip):
    """The reference's six tubes and parameters (tube_world_params.yaml), plus twist / marker noise."""
    cmd = _cmd(120, 0.3, 0.15)
    cmd[40:50, 0] = 0.0
    _check_trace(hip, 5, TUBES, cmd, 6, seed=12345)                                   # reference behaviour, range gate
    _check_trace(hip, 3, TUBES, cmd, 3, seed=7, twist_noise=0.02, marker_sigma=0.003)  # nearest-3, noisy
    _check_trace(hip, 2, TUBES, cmd, 6, seed=(5 << 32) + 9, max_range=0.0)            # every tube every tick


@pytest.mark.gpu
def test_device_trace_matches_oracle_collision_and_large_world(hip):
    _check_trace(hip, 2, np.array([[1.0, 0.0]]), _cmd(250, 0.0, 0.2), 1, seed=3, slip_min=1.0, slip_max=1.0, max_range=0.0)
    import sys
    from nuslam_hip import synth
    lm = synth.make_landmarks(1000)
    _check_trace(hip, 2, lm, _cmd(12, 0.5, 0.3), 16, seed=11, marker_sigma=0.001, max_range=0.0)   # nearest 16 of 1000
    _check_trace(hip, 2, lm, _cmd(12, 0.5, 0.3), 16, seed=11, marker_sigma=0.001, max_range=0.8)


@pytest.mark.gpu
def test_device_trace_is_shard_invariant(hip):
    """Filter b of a shard starting at first_filter draws the streams of global filter first_filter + b."""
    ph = hip.SimParams(twist_noise=0.01, marker_sigma=0.002)
    cmd = _cmd(30)
    whole = hip.Batch(4, 6, Q, R)
    whole.simulate(ph, TUBES, cmd, 4, 99)
    part = hip.Batch(2, 6, Q, R)
    part.simulate(ph, TUBES, cmd, 4, 99, first_filter=2)
    for f in range(2):
        a, b = whole.get_trace(2 + f), part.get_trace(f)
        for k in ("tw", "mx", "my", "ids", "truth"):
            assert np.array_equal(a[k], b[k])


@pytest.mark.gpu
def test_filters_run_from_device_trace_equal_host_loaded_trace(hip):
    """The generated trace is the resident trace: replaying it gives bit for bit what loading the same arrays gives,
    and the oracle's filter fed the same arrays agrees; empty slots (id -1) are skipped as slam.cpp:298-300 does."""
    ph = hip.SimParams(marker_sigma=0.002, max_range=1.2)
    B, n, m, T = 3, 6, 4, 60
    cmd = _cmd(T, 0.25, 0.3)
    dev = hip.Batch(B, n, Q, R)
    empty = dev.simulate(ph, TUBES, cmd, m, 4242)
    traces = [dev.get_trace(f) for f in range(B)]
    assert empty > 0 and any((t["ids"] == -1).any() for t in traces)
    host = hip.Batch(B, n, Q, R)
    host.load_trace(np.stack([t["tw"] for t in traces]), np.stack([t["mx"] for t in traces]),
                    np.stack([t["my"] for t in traces]), np.stack([t["ids"] for t in traces]))
    dev.run(0, T); host.run(0, T)
    for f in range(B):
        assert np.array_equal(dev.state(f), host.state(f)) and np.array_equal(dev.cov(f), host.cov(f))
        assert dev.seen(f) == host.seen(f)
    # against the oracle: landmarks are first seen in tube order here?  not necessarily -- known ids above `seen`
    # initialise, exactly as the tick protocol does, so the oracle's tick handles the same trace
    o = O.OracleEKF(np.zeros(3), np.zeros(2 * n), Q, R, O.ORC_STRUCTURED)
    tr = traces[1]
    for t in range(T):
        o.tick(tw=np.array([tr["tw"][t, 0], tr["tw"][t, 1], 0.0]), mx=tr["mx"][t], my=tr["my"][t], known_ids=tr["ids"][t])
    assert o.seen == dev.seen(1)
    assert np.allclose(dev.state(1), o.state, atol=5e-3, rtol=0)       # cold start: INT_MAX conditioning


@pytest.mark.gpu
def test_device_trace_data_association_with_range_gate(hip):
    """Unknown association on a gated trace (the reference's own configuration: six tubes, max_range 1 m): slots
    without a marker are markers the node never received -- associateLandmark is not called for them.  The oracle is
    driven with only the markers that exist; ids, `seen` and the estimate must agree."""
    ph, po = _same_params(hip, marker_sigma=0.0005, max_range=1.2, slip_min=1.0, slip_max=1.0)
    B, n, m, T = 2, 8, 6, 50
    cmd = _cmd(T, 0.3, 0.25)
    Qs = np.diag([1e-4, 1e-4, 1e-4])                   # the well-conditioned process noise of SURVEY 8d (matches happen)
    b = hip.Batch(B, n, Qs, R)
    empty = b.simulate(ph, TUBES, cmd, m, 77, known_ids=False)
    assert empty > 0
    b.run(0, T, total_landmarks=n)
    assert b.status()[1] == 0
    for f in range(B):
        want = O.simulate(po, TUBES, cmd, m, 77, filt=f)
        o = O.OracleEKF(np.zeros(3), np.zeros(2 * n), Qs, R, O.ORC_STRUCTURED)
        for t in range(T):
            keep = want["ids"][t] > 0
            o.tick(tw=np.array([want["tw"][t, 0], want["tw"][t, 1], 0.0]), mx=want["mx"][t][keep], my=want["my"][t][keep])
        assert b.seen(f) == o.seen and 1 <= o.seen <= 6
        assert np.allclose(b.state(f), o.state, atol=5e-3, rtol=0)   # cold start: INT_MAX conditioning


@pytest.mark.gpu
def test_get_trace_reads_back_a_host_loaded_trace(hip):
    from nuslam_hip import synth
    tr = synth.make_trace(6, 5, 3, landmarks=TUBES)
    b = hip.Batch(2, 6, Q, R)
    b.load_trace(tr.tw[:, :2], tr.mx, tr.my, tr.ids, bcast=True)
    got = b.get_trace(1)                                   # broadcast: every filter reads the one trace
    assert np.array_equal(got["tw"], tr.tw[:, :2]) and np.array_equal(got["mx"], tr.mx)
    assert np.array_equal(got["ids"], tr.ids) and got["truth"] is None


# ------------------------------------------------------------------ the lidar and the landmarks node's chain
def test_lidar_oracle_sees_the_tubes_in_range():
    """simulate_lidar_scanner (tube_world.cpp:405-471) as restated: from the origin the two tubes within 1 m show up
    as dips of the right depth, everything else reads max_range + 1; the landmarks chain recovers their centres."""
    r = O.sim_scan(TUBES, 0.0381, 1.0, (0.0, 0.0, 0.0))
    assert r.dtype == np.float32 and np.all(r[np.r_[0:40, 60:120, 150:200]] == np.float32(2.0))
    d = np.hypot(0.5, 0.5) - 0.0381
    assert abs(float(r[45]) - d) < 1e-6 and abs(float(r[225]) - d) < 1e-6
    ms = O.scan_markers(r, 0.05, 1.0)
    assert len(ms) == 2
    assert np.allclose(ms[0], (0.5, 0.5), atol=1e-6) and np.allclose(ms[1], (-0.5, -0.5), atol=1e-6)
    # robot frame: turn the robot by 90 degrees and the tube that was at 45 degrees shows up at -45 (= 315)
    r2 = O.sim_scan(TUBES, 0.0381, 1.0, (np.pi / 2, 0.0, 0.0))
    assert abs(float(r2[315]) - d) < 1e-6


@pytest.mark.gpu
def test_device_lidar_pipeline_matches_oracle(hip):
    """k_sim_scan -> k_scan_clusters -> k_classify -> k_circle_fit -> k_scan_markers against the oracle's restatement
    of tube_world.cpp:405-471 and nuslam/src/landmarks.cpp:63, 82-108, scan by scan and marker by marker."""
    kw = dict(lidar=1.0, slip_min=1.0, slip_max=1.0)
    ph, po = _same_params(hip, **kw)
    B, n, m, T = 3, 8, 6, 40
    cmd = _cmd(T, 0.35, 0.3)
    b = hip.Batch(B, n, np.diag([1e-4] * 3), R)
    empty = b.simulate(ph, TUBES, cmd, m, 31, known_ids=False)
    total_empty = 0
    for f in range(B):
        want = O.simulate(po, TUBES, cmd, m, 31, filt=f)
        got = b.get_trace(f)
        assert np.allclose(got["truth"], want["truth"], atol=1e-12, rtol=0)
        for t in range(T):
            scan_o = O.sim_scan(TUBES, po.tube_radius, po.lidar_max_range, want["truth"][t])
            scan_g = b.get_scan(f, t)
            assert np.allclose(scan_g, scan_o, atol=2e-6, rtol=0), (f, t)
            ms = O.scan_markers(scan_o, po.lidar_min_range, po.lidar_max_range)
            present = got["ids"][t] > 0
            assert present.sum() == len(ms), (f, t, got["ids"][t], ms)
            assert np.array_equal(got["ids"][t][present], np.arange(1, len(ms) + 1))   # packed to the front
            for k, (cx, cy) in enumerate(ms):
                assert abs(got["mx"][t, k] - cx) < 1e-7 and abs(got["my"][t, k] - cy) < 1e-7
            total_empty += m - len(ms)
    assert empty == total_empty and empty > 0


@pytest.mark.gpu
def test_slam_from_device_lidar_markers(hip):
    """The whole reference chain on the device -- simulator, lidar, landmark extraction, EKF-SLAM with unknown data
    association -- against the same chain in the oracle."""
    kw = dict(lidar=1.0, slip_min=1.0, slip_max=1.0)
    ph, po = _same_params(hip, **kw)
    n, m, T = 8, 6, 60
    cmd = _cmd(T, 0.3, 0.25)
    Qs = np.diag([1e-4, 1e-4, 1e-4])
    b = hip.Batch(2, n, Qs, R)
    b.simulate(ph, TUBES, cmd, m, 5, known_ids=False)
    b.run(0, T, total_landmarks=n)
    assert b.status()[1] == 0
    want = O.simulate(po, TUBES, cmd, m, 5, filt=1)
    o = O.OracleEKF(np.zeros(3), np.zeros(2 * n), Qs, R, O.ORC_STRUCTURED)
    for t in range(T):
        scan = O.sim_scan(TUBES, po.tube_radius, po.lidar_max_range, want["truth"][t])
        ms = O.scan_markers(scan, po.lidar_min_range, po.lidar_max_range)
        o.tick(tw=np.array([want["tw"][t, 0], want["tw"][t, 1], 0.0]), mx=np.array([p[0] for p in ms]),
               my=np.array([p[1] for p in ms]))
    assert b.seen(1) == o.seen and o.seen >= 2
    assert np.allclose(b.state(1), o.state, atol=5e-3, rtol=0)            # cold start: INT_MAX conditioning
