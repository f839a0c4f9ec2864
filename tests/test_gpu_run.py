"""nuslam_batch_run on ONE filter with known ids as ONE launch (csrc/ekf_fused.h, k_run_fused: the roles of k_tick_fused looping over the
run's ticks, the covariance resident in the pass workgroups' registers in between, only the next tick's rows / columns exported)
against the same run with a launch per tick (tick mode 5): every element takes the same k-ordered fma chain, the predict role, the
chain and the strips the same arithmetic -- identical bits after every run, whatever its length (odd / even: the result ends in either
P buffer), with skipped markers, a landmark twice in a tick, fewer markers than a round, fp32 storage, runs back to back, and a run that
holds a first sighting (it steps aside to a launch per tick)."""
import numpy as np
import pytest

import _oracle as O
from nuslam_hip import synth

pytestmark = pytest.mark.gpu
Q, R = synth.Q_DEFAULT, synth.R_DEFAULT
EXACT_WHEELS = dict(dL=0.3125, dR=0.375)


def make(hip, n, dtype, mode, lm):
    f = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R, dtype=dtype)
    bt = f.as_batch()
    bt.set_tick_mode(mode)
    bx, by, wid = synth.warmup_observations(lm)
    bt.load_trace(np.zeros((1, 2)), bx[None, :], by[None, :], wid[None, :], bcast=True)
    bt.run(0, 1)
    return f, bt


@pytest.mark.parametrize("n,m,dtype", [(1000, 16, 0), (200, 16, 0), (200, 7, 0), (300, 16, 1), (40, 16, 0), (120, 1, 0)])
def test_run_as_one_launch_same_bits_as_a_launch_per_tick(hip, n, m, dtype):
    T = 13
    lm = synth.make_landmarks(n)
    tr = synth.make_wellposed_trace(n, T, m, landmarks=lm, straight_every=4) if n >= 200 else synth.make_trace(n, T, m, landmarks=lm, straight_every=4, **EXACT_WHEELS)
    ids = tr.ids.copy()
    ids[2, 1 % m] = -1                                # update() not called for a marker
    if m > 2:
        ids[4, 2] = ids[4, 0]                         # the same landmark twice in a tick
        ids[9, m - 1] = -1
    out = []
    for mode in (1, 5):                               # 1: the default (the run as one launch), 5: a launch per tick
        f, bt = make(hip, n, dtype, mode, lm)
        bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, ids, bcast=True)
        bt.profile(True)
        snaps = []
        bt.run(0, 5)                                  # odd: the covariance ends in the other buffer
        snaps.append((f.state.copy(), f.cov.copy(), f.seen))
        bt.run(5, 11)                                 # even
        snaps.append((f.state.copy(), f.cov.copy(), f.seen))
        bt.run(11, 12)                                # a single tick: a launch of its own either way
        bt.run(12, 13)
        snaps.append((f.state.copy(), f.cov.copy(), f.seen))
        _, launches = bt.profile_read(hip.K_TICK_CHAIN)
        bt.profile(False)
        assert bt.status() == (-1, 0)
        out.append((snaps, launches))
    assert out[0][1] == 4 and out[1][1] == 13, (out[0][1], out[1][1])          # two runs as one launch each + two single ticks
    for k, ((s0, p0, n0), (s1, p1, n1)) in enumerate(zip(out[0][0], out[1][0])):
        assert np.isfinite(p0).all()
        assert np.array_equal(s0, s1), ("state", k, np.abs(s0 - s1).max())
        assert np.array_equal(p0, p1), ("covariance", k, np.abs(p0 - p1).max(), np.argwhere(p0 != p1)[:8].tolist())
        assert n0 == n1


def test_a_run_that_may_hold_a_first_sighting_takes_a_launch_per_tick(hip):
    n, m, T = 60, 16, 6
    lm = synth.make_landmarks(n)
    tr = synth.make_trace(n, T, m, landmarks=lm, straight_every=3, **EXACT_WHEELS)
    out = []
    for mode in (1, 5):
        f = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
        bt = f.as_batch()
        bt.set_tick_mode(mode)
        bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, tr.ids, bcast=True)      # cold: every landmark is a first sighting somewhere in the run
        bt.run(0, T)
        assert bt.status() == (-1, 0)
        out.append((f.state.copy(), f.cov.copy(), f.seen))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]


def test_run_as_one_launch_against_the_oracle(hip):
    """The one-launch run against the CPU oracle (structured mode) on a well-posed trace: N = 200, 30 ticks x 16, <= 1e-6 per entry."""
    n, m, T = 200, 16, 30
    lm = synth.make_landmarks(n)
    tr = synth.make_wellposed_trace(n, T, m, landmarks=lm)
    bx, by, wid = synth.warmup_observations(lm)
    o = O.OracleEKF(np.zeros(3), np.zeros(2 * n), Q, R, O.ORC_STRUCTURED)
    o.tick(tw=np.zeros(3), mx=bx, my=by, known_ids=wid)
    # (from the oracle's own post-initialisation snapshot, as tests/test_gpu_depth.py: the corrections that cancel the INT_MAX diagonal are
    # ill-conditioned -- DESIGN.md section 4)
    f = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    f.restore(o.state.copy(), o.cov.copy(), o.seen)
    bt = f.as_batch()
    bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, tr.ids, bcast=True)
    bt.profile(True)
    bt.run(0, T)
    _, launches = bt.profile_read(hip.K_TICK_CHAIN)
    bt.profile(False)
    assert launches == 1 and bt.status() == (-1, 0)
    for t in range(T):
        o.tick(tw=tr.tw[t], mx=tr.mx[t], my=tr.my[t], known_ids=tr.ids[t])
    es = float((np.abs(f.state - o.state) / np.maximum(np.abs(o.state), 1e-12 * np.abs(o.state).max())).max())
    ep = float((np.abs(f.cov - o.cov) / np.maximum(np.abs(o.cov), 1e-12 * np.abs(o.cov).max())).max())
    print("one-launch run vs oracle, tick %d: state %.2e cov %.2e" % (T, es, ep))
    assert es < 1e-6 and ep < 1e-6 and f.seen == o.seen


def test_a_long_run_goes_in_stretches(hip):
    """More ticks than one launch takes (4096): the run goes in stretches of one launch each -- same bits as a launch per tick."""
    n, m, T = 40, 16, 4100
    lm = synth.make_landmarks(n)
    tr = synth.make_trace(n, 60, m, landmarks=lm, straight_every=4, **EXACT_WHEELS)
    reps = (T + 59) // 60
    tw = np.tile(tr.tw[:, :2], (reps, 1))[:T]; mx = np.tile(tr.mx, (reps, 1))[:T]; my = np.tile(tr.my, (reps, 1))[:T]
    ids = np.tile(tr.ids, (reps, 1))[:T]
    out = []
    for mode in (1, 5):
        f, bt = make(hip, n, 0, mode, lm)
        bt.load_trace(tw, mx, my, ids, bcast=True)
        bt.profile(True)
        bt.run(0, T)
        _, launches = bt.profile_read(hip.K_TICK_CHAIN)
        bt.profile(False)
        out.append((f.state.copy(), f.cov.copy(), f.seen, launches, bt.status()))
    assert out[0][3] == 2 and out[1][3] == T
    assert out[0][4] == out[1][4]
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]
