"""GPU parity, edge cases: degenerate sizes, padding boundaries, the caller's break / skip branches, latched device
errors, empty ticks."""
import numpy as np
import pytest

import _oracle as O
from nuslam_hip import synth

pytestmark = pytest.mark.gpu
Q, R = synth.Q_DEFAULT, synth.R_DEFAULT


def entry_rel_err(a, b):
    floor = 1e-12 * max(np.abs(b).max(), 1e-300)
    return (np.abs(a - b) / np.maximum(np.abs(b), floor)).max()


def test_no_landmarks(hip):
    """n = 0: the state is the pose, P is 3x3 (len = 3, one 32-row padded column block)."""
    g = hip.EKF(np.array([0.1, 0.2, 0.3]), np.zeros(0), Q, R)
    o = O.OracleEKF(np.array([0.1, 0.2, 0.3]), np.zeros(0), Q, R)
    for dth, dx in ((0.05, 0.1), (0.0, 0.2), (-0.3, 0.05)):
        g.predict(dth, dx)
        o.predict(dth, dx)
    assert g.len == 3 and entry_rel_err(g.state, o.state) < 1e-14 and entry_rel_err(g.cov, o.cov) < 1e-13
    g.tick([0.01, 0.02, 0.0], [], [], known_ids=[], want_ids=False)          # a tick without markers is a predict
    o.tick(tw=[0.01, 0.02, 0.0])
    assert entry_rel_err(g.state, o.state) < 1e-14
    with pytest.raises(hip.NuslamError):
        g.update(1.0, 0.0, 1)


@pytest.mark.parametrize("n", [14, 15, 30, 31, 62, 63, 64, 127])
def test_padding_boundaries(hip, n):
    """len = 3 + 2n straddles the 32-element column padding, the 128-row wave tile and the 64-column workgroup tile."""
    tr = synth.make_trace(n, 3, min(n, 7))
    o = O.OracleEKF(np.zeros(3), np.zeros(2 * n), Q, R, O.ORC_STRUCTURED)
    bx, by, ids = synth.warmup_observations(tr.landmarks)
    o.tick(tw=np.zeros(3), mx=bx, my=by, known_ids=ids)
    g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    g.restore(o.state, o.cov, n)
    r, b = tr.polar()
    # the very last landmark touches the last two rows / columns of P
    o.update(r[0][0], b[0][0], n)
    g.update(r[0][0], b[0][0], n)
    assert np.array_equal(g.cov, o.cov)
    for t in range(tr.ticks):
        o.tick(tw=tr.tw[t], mx=tr.mx[t], my=tr.my[t], known_ids=tr.ids[t])
        g.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t], want_ids=False)
    assert entry_rel_err(g.state, o.state) < 1e-8 and entry_rel_err(g.cov, o.cov) < 1e-6


def test_skip_and_break_branches(hip):
    """slam.cpp:298-316: a negative id skips the marker, an id above total_landmarks leaves the marker loop."""
    n = 8
    tr = synth.make_trace(n, 1, 6)
    o = O.OracleEKF(np.zeros(3), np.zeros(2 * n), Q, R, O.ORC_STRUCTURED)
    bx, by, ids = synth.warmup_observations(tr.landmarks)
    o.tick(tw=np.zeros(3), mx=bx, my=by, known_ids=ids)
    g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    g.restore(o.state, o.cov, n)
    ids = np.array([3, -1, 5, 7, 2, 4], dtype=np.int32)      # total_landmarks = 6: id 7 breaks the loop
    io = o.tick(tw=tr.tw[0], mx=tr.mx[0], my=tr.my[0], known_ids=ids, total_landmarks=6)
    ig = g.tick(tr.tw[0], tr.mx[0], tr.my[0], known_ids=ids, total_landmarks=6)
    assert io.tolist() == [3, -1, 5, 7, 0, 0]
    assert ig.tolist() == [3, -1, 5, 7, 0, 0]                 # 0 = not reached
    assert entry_rel_err(g.state, o.state) < 1e-10 and entry_rel_err(g.cov, o.cov) < 1e-9
    # the break flag is per tick: the next tick processes markers again
    io = o.tick(tw=tr.tw[0], mx=tr.mx[0][:3], my=tr.my[0][:3], known_ids=ids[:3], total_landmarks=6)
    ig = g.tick(tr.tw[0], tr.mx[0][:3], tr.my[0][:3], known_ids=ids[:3], total_landmarks=6)
    assert ig.tolist() == io.tolist() == [3, -1, 5]
    assert entry_rel_err(g.cov, o.cov) < 1e-9


def test_out_of_range_id_in_a_tick_is_latched(hip):
    n = 5
    g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    P0 = g.cov
    with pytest.raises(hip.NuslamError) as ei:                 # want_ids synchronises and reports the latched status
        g.tick([0.0, 0.0, 0.0], [0.5], [0.1], known_ids=[9])
    assert ei.value.code == hip.E_BOUNDS
    assert g.status(clear=True) == hip.E_BOUNDS and g.status() == 0
    assert np.isfinite(g.cov).all()
    assert np.array_equal(g.cov[3:, 3:], P0[3:, 3:])            # the bad marker changed nothing in the map block


def test_singular_innovation_covariance_is_latched(hip):
    """R = 0 and P = 0 make S = H P H^T + R exactly singular: Armadillo's inv() throws std::runtime_error at
    slam_library.cpp:270; here the correction is a no-op and NUSLAM_E_SINGULAR is latched."""
    n = 3
    Rz = np.zeros((2, 2))
    g = hip.EKF(np.zeros(3), np.array([1.0, 0.5, -0.4, 0.8, 0.2, -0.9]), Q, Rz)
    o = O.OracleEKF(np.zeros(3), np.array([1.0, 0.5, -0.4, 0.8, 0.2, -0.9]), Q, Rz)
    z = np.zeros(g.len); z[:] = g.state
    g.restore(z, np.zeros((g.len, g.len)), n)
    o.restore(z, np.zeros((g.len, g.len)), n)
    with pytest.raises(O.OracleError) as eo:
        o.update(1.0, 0.3, 2)
    assert eo.value.code == O.ORC_E_SINGULAR
    g.update(1.0, 0.3, 2)
    with pytest.raises(hip.NuslamError) as eg:
        g.sync()
    assert eg.value.code == hip.E_SINGULAR
    assert np.array_equal(g.state, z) and not g.cov.any()
    g.status(clear=True)
    g.sync()


def test_clone_and_restore_of_fp32(hip):
    n = 9
    g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R, dtype=hip.F32)
    rng = np.random.default_rng(4)
    P = rng.normal(size=(g.len, g.len)).astype(np.float32).astype(np.float64)   # exactly representable in fp32
    s = rng.normal(size=g.len)
    g.restore(s, P, 4)
    c = g.clone()
    assert np.array_equal(c.cov, P) and np.array_equal(c.state, s) and c.seen == 4


def test_many_markers_per_tick_and_repeated_ids(hip):
    """m > n: the same landmark corrected several times in one tick, in order."""
    n, m = 4, 11
    rng = np.random.default_rng(8)
    tr = synth.make_trace(n, 2, n)
    o = O.OracleEKF(np.zeros(3), np.zeros(2 * n), Q, R, O.ORC_STRUCTURED)
    bx, by, ids = synth.warmup_observations(tr.landmarks)
    o.tick(tw=np.zeros(3), mx=bx, my=by, known_ids=ids)
    g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    g.restore(o.state, o.cov, n)
    pick = rng.integers(0, n, size=m)
    mx, my, kid = tr.mx[0][pick], tr.my[0][pick], tr.ids[0][pick]
    o.tick(tw=tr.tw[0], mx=mx, my=my, known_ids=kid)
    g.tick(tr.tw[0], mx, my, known_ids=kid, want_ids=False)
    assert entry_rel_err(g.state, o.state) < 1e-9 and entry_rel_err(g.cov, o.cov) < 1e-8


def test_poisoned_batch_comes_back_only_when_every_filter_was_restored(hip):
    """An expired device-side wait (NUSLAM_E_SYNC) poisons the whole handle: the run went on from a hand-off that never arrived.
    restore() of ONE filter of a batch must not re-admit the others (they still hold what the code itself declares invalid)."""
    B, n, m, T = 3, 12, 5, 4
    tr = synth.make_trace(n, T, m)
    bt = hip.Batch(B, n, synth.Q_DEFAULT, synth.R_DEFAULT)
    bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, tr.ids, bcast=True)
    bt.run(0, 2)
    snaps = [(bt.state(b), bt.cov(b), bt.seen(b)) for b in range(B)]
    bt.inject_fault(1)                                                  # as if a hand-off between workgroups had expired
    assert bt.status()[1] == hip.E_SYNC
    with pytest.raises(hip.NuslamError) as e:
        bt.run(2, 3)
    assert e.value.code == hip.E_SYNC
    bt.status(clear=True)
    bt.restore(0, *snaps[0])
    with pytest.raises(hip.NuslamError) as e:                           # filters 1 and 2 are still what the failed run left
        bt.run(2, 3)
    assert e.value.code == hip.E_SYNC
    bt.restore(1, *snaps[1])
    bt.restore(2, *snaps[2])
    bt.run(2, T)                                                        # every filter restored: the handle ticks again
    assert bt.status() == (-1, 0)
    ref = hip.Batch(B, n, synth.Q_DEFAULT, synth.R_DEFAULT)
    ref.load_trace(tr.tw[:, :2], tr.mx, tr.my, tr.ids, bcast=True)
    ref.run(0, T)
    for b in range(B):
        assert np.array_equal(bt.state(b), ref.state(b)) and np.array_equal(bt.cov(b), ref.cov(b))


def test_restore_clears_what_a_diverged_run_left_in_the_strips(hip):
    """The rank-2m pass reads all 32 factor rows of the K / V strips every round and masks the unused ones by multiplication: a NaN a
    diverged run left in a slot the next round does not rewrite (a skipped marker, fewer markers than a round) would spread over all of
    P (0 * NaN).  restore() re-zeroes the filter's strips."""
    n, m = 40, 16
    tr = synth.make_wellposed_trace(n, 3, m) if False else synth.make_trace(n, 3, m)
    f = hip.EKF(np.zeros(3), np.zeros(2 * n), synth.Q_DEFAULT, synth.R_DEFAULT)
    f.as_batch().set_tick_mode(4)                                       # the pass as k_tick_rank, a launch of its own
    bx, by, wid = synth.warmup_observations(tr.landmarks)
    f.tick(np.zeros(3), bx, by, known_ids=wid, want_ids=False)
    f.tick(tr.tw[0], tr.mx[0], tr.my[0], known_ids=tr.ids[0], want_ids=False)
    snap = f.snapshot()
    f.as_batch().inject_fault(2)                                        # NaN in every strip slot
    f.restore(*snap)
    ids = tr.ids[1].copy()
    ids[3] = -1                                                         # a slot the round leaves unwritten
    f.tick(tr.tw[1], tr.mx[1][:10], tr.my[1][:10], known_ids=ids[:10], want_ids=False)      # and only ten markers of sixteen
    assert np.isfinite(f.cov).all() and np.isfinite(f.state).all() and f.status() == 0
