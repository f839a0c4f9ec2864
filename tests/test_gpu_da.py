"""Unknown data association as a tick pipeline (csrc/ekf_da.h: k_da_begin -> one k_da_step per marker -> k_tick_apply,
ONE pass over P per tick) against the per-correction path it replaces (k_associate + k_update per marker): the tracked
rows / columns / diagonal blocks go through the same floating-point operations in the same order as the full
covariance would, so every association verdict (match, new landmark, gray zone), the resolved ids, `seen`, the state
and the covariance must agree BIT FOR BIT -- from a cold start (every landmark a first sighting), warm, with markers in
the gray zone, the same landmark twice in a tick, more markers than one round of 16, fp32 storage, batches, and traces
with empty marker slots.  (Against the oracle the default path is exercised by test_gpu_parity / test_gpu_baseline_sizes.)

Statements about the EXACT chain (PASS_EXACT): the default pass over P, the rank-2m update on the matrix cores, is the
same algebra re-associated and is held to a tolerance (tests/test_gpu_rank.py); the verdicts it must still reproduce.""" 
import numpy as np
import pytest

from nuslam_hip import synth

pytestmark = pytest.mark.gpu
Q, R = synth.Q_DEFAULT, synth.R_DEFAULT


@pytest.fixture(params=[1, 2], ids=["resident-round", "launch-per-marker"])
def pipe(request):
    """tick mode 1: one resident launch per round (workgroups meet at a device counter between corrections);
    mode 2: k_da_begin + one k_da_step launch per marker (what large batches take)"""
    return request.param


def pair_of_filters(hip, n, dtype=0, Qm=Q, pipe=1):
    a = hip.EKF(np.zeros(3), np.zeros(2 * n), Qm, R, dtype=dtype)
    b = hip.EKF(np.zeros(3), np.zeros(2 * n), Qm, R, dtype=dtype)
    a.as_batch().set_tick_mode(pipe)
    a.as_batch().set_pass_variant(hip.PASS_EXACT)
    b.as_batch().set_tick_mode(0)               # k_associate + k_update per marker
    return a, b


def same(a, b):
    return (np.array_equal(a.state, b.state, equal_nan=True) and np.array_equal(a.cov, b.cov, equal_nan=True)
            and a.seen == b.seen and a.status() == b.status())


@pytest.mark.parametrize("n,m,dtype,sigma", [(12, 5, 0, 1e-3), (12, 1, 0, 1e-3), (40, 16, 0, 1e-3), (40, 16, 1, 1e-3),
                                             (70, 37, 0, 1e-3), (35, 16, 0, None)])
def test_da_pipeline_equals_per_correction_kernels_cold_start(hip, pipe, n, m, dtype, sigma):
    """sigma None: the simulator's marker noise (sqrt(1e-3) m), where gray-zone verdicts and spurious new landmarks occur."""
    T = 10
    tr = synth.make_trace(n, T, m, straight_every=3, noise_sigma=sigma)
    a, b = pair_of_filters(hip, n, dtype, pipe=pipe)
    verdicts = []
    for t in range(T):
        try:
            ia = a.tick(tr.tw[t], tr.mx[t], tr.my[t])
            ea = None
        except hip.NuslamError as e:
            ia, ea = None, e.code
        try:
            ib = b.tick(tr.tw[t], tr.mx[t], tr.my[t])
            eb = None
        except hip.NuslamError as e:
            ib, eb = None, e.code
        assert ea == eb, (t, ea, eb)
        if ia is not None:
            assert np.array_equal(ia, ib), (t, ia, ib)
            verdicts.append(ia)
        assert same(a, b), "tick %d" % t
        if ea is not None:
            break
    v = np.concatenate(verdicts)
    print("n=%d m=%d: %d matches/new, %d gray-zone, seen %d" % (n, m, (v > 0).sum(), (v < 0).sum(), a.seen))
    assert (v > 0).sum() > 0


def test_da_pipeline_new_landmark_gray_zone_and_resighting_in_one_tick(hip, pipe):
    n, n_world, m, T = 30, 24, 8, 5
    Qs = np.diag([1e-4, 1e-4, 1e-4])
    lm = synth.make_landmarks(n_world)
    tr = synth.make_trace(n_world, T, m, landmarks=lm, noise_sigma=1e-4)
    bx, by, wid = synth.warmup_observations(lm, noise_sigma=1e-4)
    a, b = pair_of_filters(hip, n, 0, Qs, pipe)
    for f in (a, b):
        f.tick(np.zeros(3), bx, by, known_ids=wid)
    assert same(a, b) and a.seen == n_world
    mx, my = tr.mx.copy(), tr.my.copy()
    mx[1, 3], my[1, 3] = 20.0, 1.0              # far from every landmark: a new one ...
    mx[1, 6], my[1, 6] = 20.0, 1.0              # ... and seen again in the SAME tick (id > cached: initialised again, slam.cpp:295)
    mx[2, 2] += 0.05                            # the gray zone
    mx[3, 5], my[3, 5] = mx[3, 0], my[3, 0]     # the same landmark twice in one tick
    got = []
    for t in range(T):
        ia = a.tick(tr.tw[t], mx[t], my[t])
        ib = b.tick(tr.tw[t], mx[t], my[t])
        assert np.array_equal(ia, ib), (t, ia, ib)
        assert same(a, b), "tick %d" % t
        got.append(ia)
    got = np.array(got)
    assert got[1, 3] == n_world + 1 and got[1, 6] == n_world + 1 and got[2, 2] == -1 and got[3, 5] == got[3, 0]


def test_da_pipeline_full_map_latches_bounds(hip, pipe):
    """seen == n and an unmatched marker: associateLandmark would write past the map (slam_library.cpp:206-207)."""
    n, m = 6, 4
    lm = synth.make_landmarks(n)
    tr = synth.make_trace(n, 2, m, landmarks=lm, noise_sigma=1e-4)
    bx, by, wid = synth.warmup_observations(lm, noise_sigma=1e-4)
    a, b = pair_of_filters(hip, n, pipe=pipe)
    codes = []
    for f in (a, b):
        f.tick(np.zeros(3), bx, by, known_ids=wid)
        try:
            f.tick(tr.tw[0], tr.mx[0], tr.my[0])
            codes.append(0)
        except hip.NuslamError as e:
            codes.append(e.code)
    assert codes[0] == codes[1]
    assert np.array_equal(a.state, b.state) and np.array_equal(a.cov, b.cov) and a.seen == b.seen


@pytest.mark.parametrize("B,n,m,dtype", [(3, 20, 6, 0), (5, 33, 16, 0), (2, 16, 16, 1)])
def test_da_pipeline_batch_run(hip, pipe, B, n, m, dtype):
    T = 8
    traces = [synth.make_trace(n, T, m, seed=700 + k, noise_sigma=1e-3, straight_every=4) for k in range(B)]
    tw = np.stack([t.tw[:, :2] for t in traces]); mx = np.stack([t.mx for t in traces]); my = np.stack([t.my for t in traces])
    out, stats = [], []
    for mode in (pipe, 0):
        bt = hip.Batch(B, n, Q, R, dtype=dtype)
        bt.set_tick_mode(mode)
        bt.set_pass_variant(hip.PASS_EXACT)
        bt.load_trace(tw, mx, my, None)
        bt.run(0, T)
        stats.append(bt.status())
        out.append([(bt.state(k), bt.cov(k), bt.seen(k)) for k in range(B)])
    assert stats[0] == stats[1], stats          # (fp32 storage: a spurious new landmark can fill the map -- in both paths)
    for k in range(B):
        assert np.array_equal(out[0][k][0], out[1][k][0]) and np.array_equal(out[0][k][1], out[1][k][1])
        assert out[0][k][2] == out[1][k][2] and out[0][k][2] > 0


def test_da_pipeline_device_trace_with_empty_marker_slots(hip, pipe):
    """A generated trace with a range gate: marker slots without a marker (presence word < 0) are not associated."""
    import nuslam_hip as nh
    B, n, m, T = 4, 24, 8, 12
    lm = synth.make_landmarks(n)
    cmd = np.zeros((T, 2)); cmd[:, 0] = 0.1; cmd[:, 1] = 0.05
    sim = nh.SimParams(marker_sigma=1e-3, max_range=1.2)
    out = []
    for mode in (pipe, 0):
        bt = hip.Batch(B, n, Q, R)
        bt.set_tick_mode(mode)
        bt.set_pass_variant(hip.PASS_EXACT)
        bt.simulate(sim, lm, cmd, m, 4321, first_filter=0, known_ids=False)
        bt.run(0, T)
        assert bt.status() == (-1, 0)
        out.append([(bt.state(k), bt.cov(k), bt.seen(k)) for k in range(B)])
    for k in range(B):
        assert np.array_equal(out[0][k][0], out[1][k][0]) and np.array_equal(out[0][k][1], out[1][k][1])
        assert out[0][k][2] == out[1][k][2]
    assert 0 < out[0][0][2] < n, "the range gate must leave some landmarks unseen"


def test_da_pipeline_n1000(hip, pipe):
    """BASELINE configs[4] size: 3 ticks x 16 markers over 998 seen landmarks (34 workgroups per step), one new landmark,
    one gray-zone marker: bitwise the per-correction path."""
    n, n_world, m, T = 1000, 998, 16, 3
    Qs = np.diag([1e-4, 1e-4, 1e-4])
    lm = synth.make_landmarks(n_world)
    tr = synth.make_trace(n_world, T, m, landmarks=lm, noise_sigma=1e-4)
    bx, by, wid = synth.warmup_observations(lm, noise_sigma=1e-4)
    a, b = pair_of_filters(hip, n, 0, Qs, pipe)
    for f in (a, b):
        f.tick(np.zeros(3), bx, by, known_ids=wid, want_ids=False)
    mx, my = tr.mx.copy(), tr.my.copy()
    mx[1, 7], my[1, 7] = 20.0, 1.0
    mx[2, 11] += 0.05
    for t in range(T):
        ia = a.tick(tr.tw[t], mx[t], my[t])
        ib = b.tick(tr.tw[t], mx[t], my[t])
        assert np.array_equal(ia, ib), (t, ia, ib)
        assert a.seen == b.seen and np.array_equal(a.state, b.state), "tick %d" % t
    assert ia is not None and np.array_equal(a.cov, b.cov)
    assert a.seen == 999


def test_da_pipeline_long_run_n1000(hip):
    """1500 ticks x 16 markers at N = 1000 through nuslam_batch_run: the resident round kernel (24 000 meets of 34
    workgroups), the launch-per-marker variant and the per-correction kernels end in the same bits, no wait expires."""
    n, n_world, m, T = 1000, 998, 16, 1500
    Qs = np.diag([1e-4, 1e-4, 1e-4])
    lm = synth.make_landmarks(n_world)
    tr = synth.make_trace(n_world, T, m, landmarks=lm, noise_sigma=1e-4)
    bx, by, wid = synth.warmup_observations(lm, noise_sigma=1e-4)
    res = []
    for mode in (1, 2, 0):
        ekf = hip.EKF(np.zeros(3), np.zeros(2 * n), Qs, R)
        ekf.tick(np.zeros(3), bx, by, known_ids=wid, want_ids=False)
        bt = ekf.as_batch()
        bt.set_tick_mode(mode)
        bt.set_pass_variant(hip.PASS_EXACT)
        bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, None, bcast=True)
        for t0 in range(0, T, 500):
            bt.run(t0, t0 + 500)
        assert bt.status() == (-1, 0)
        res.append((ekf.state, ekf.cov, ekf.seen))
    for r in res[1:]:
        assert np.array_equal(res[0][0], r[0]) and np.array_equal(res[0][1], r[1]) and res[0][2] == r[2]
