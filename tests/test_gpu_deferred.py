"""GPU parity of the opt-in deferred-application mode (csrc/ekf_deferred.h): the reference's update re-associated as
P_j = P_0 - sum K_i (H_i P_{i-1}), applied once per tick.  Not bit-identical to the eager path (different
association of the same products), so the assertions are tolerances: tight against the eager kernel, the
north-star 1e-6 against the oracle."""
import numpy as np
import pytest

import _oracle as O
from nuslam_hip import synth

pytestmark = pytest.mark.gpu
Q, R = synth.Q_DEFAULT, synth.R_DEFAULT


def entry_rel_err(a, b):
    floor = 1e-12 * np.abs(b).max()
    return (np.abs(a - b) / np.maximum(np.abs(b), floor)).max()


def warm(hip, n, dtype=0, deferred=False):
    lm = synth.make_landmarks(n)
    o = O.OracleEKF(np.zeros(3), np.zeros(2 * n), Q, R, O.ORC_STRUCTURED)
    bx, by, ids = synth.warmup_observations(lm)
    o.tick(tw=np.zeros(3), mx=bx, my=by, known_ids=ids)
    g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R, dtype=dtype)
    g.restore(o.state, o.cov, n)
    if deferred:
        g.set_deferred(True)
    return o, g, lm


@pytest.mark.parametrize("n,T,m", [(10, 30, 10), (50, 10, 16), (40, 6, 37)])
def test_deferred_matches_eager_and_oracle_warm(hip, n, T, m):
    """m = 37 > 16 pending factors forces flushes inside a tick."""
    o, gd, lm = warm(hip, n, deferred=True)
    _, ge, _ = warm(hip, n)
    tr = synth.make_trace(n, T, m, landmarks=lm)
    worst = 0.0
    for t in range(T):
        o.tick(tw=tr.tw[t], mx=tr.mx[t], my=tr.my[t], known_ids=tr.ids[t])
        gd.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t], want_ids=False)
        ge.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t], want_ids=False)
        if t % 3 == 0:                                  # reading P mid-run forces a flush: must not disturb anything
            worst = max(worst, entry_rel_err(gd.cov, ge.cov))
    es, ep = entry_rel_err(gd.state, ge.state), entry_rel_err(gd.cov, ge.cov)
    os_, op = entry_rel_err(gd.state, o.state), entry_rel_err(gd.cov, o.cov)
    print("deferred n=%d m=%d: vs eager state %.1e cov %.1e (mid-run %.1e); vs oracle state %.1e cov %.1e"
          % (n, m, es, ep, worst, os_, op))
    assert es < 1e-8 and ep < 1e-7 and worst < 1e-7
    assert os_ < 1e-6 and op < 1e-6
    assert gd.seen == ge.seen == o.seen and gd.status() == 0


def test_deferred_cold_start(hip):
    """From the constructor: landmark initialisation inside the factor form (INT_MAX in P_0)."""
    n = 10
    tr = synth.make_trace(n, 20, n)
    gd = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    gd.set_deferred(True)
    ge = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    for t in range(tr.ticks):
        gd.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t], want_ids=False)
        ge.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t], want_ids=False)
    es = np.abs(gd.state - ge.state).max()
    ep = np.linalg.norm(gd.cov - ge.cov) / np.linalg.norm(ge.cov)
    print("deferred cold start vs eager: |dstate| %.2e ||dP||_F %.2e" % (es, ep))
    assert es < 5e-3 and ep < 1e-4            # the cold-start tolerance of test_gpu_parity.py::test_trajectory_cold
    assert np.isfinite(gd.cov).all() and gd.seen == ge.seen


def test_deferred_plain_update_and_gray_zone_skip(hip):
    o, gd, lm = warm(hip, 8, deferred=True)
    _, ge, _ = warm(hip, 8)
    tr = synth.make_trace(8, 2, 5, landmarks=lm)
    r, b = tr.polar()
    for g in (gd, ge):
        g.predict(tr.tw[0][0], tr.tw[0][1])
        for i in range(5):
            g.update(r[0, i], b[0, i], int(tr.ids[0, i]))
        ids = tr.ids[1].copy()
        ids[2] = -1                                      # a skipped marker is a zero factor
        g.tick(tr.tw[1], tr.mx[1], tr.my[1], known_ids=ids, want_ids=False)
    assert entry_rel_err(gd.cov, ge.cov) < 1e-8 and entry_rel_err(gd.state, ge.state) < 1e-9
    c = gd.clone()                                       # clone flushes first and carries the applied covariance
    assert np.array_equal(c.cov, gd.cov)


def test_deferred_fp32_and_batch(hip):
    n, T, m, B = 30, 4, 9, 3
    lm = synth.make_landmarks(n)
    tr = synth.make_trace(n, T, m, landmarks=lm)
    bx, by, ids = synth.warmup_observations(lm)
    res = {}
    for dtype in (hip.F64, hip.F32):
        for deferred in (False, True):
            bt = hip.Batch(B, n, Q, R, dtype=dtype)
            bt.load_trace(np.zeros((1, 2)), bx[None, :], by[None, :], ids[None, :], bcast=True)
            bt.run(0, 1)                                 # eager map initialisation in both cases
            bt.set_deferred(deferred)
            bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, tr.ids, bcast=True)
            bt.run(0, T)
            res[(dtype, deferred)] = (bt.state(B - 1), bt.cov(B - 1))
            assert bt.status() == (-1, 0)
    s64, p64 = res[(hip.F64, False)]
    assert entry_rel_err(res[(hip.F64, True)][1], p64) < 1e-8
    sd, pd = res[(hip.F32, True)]
    assert np.abs(sd - s64).max() < 1e-3 and np.abs(pd - p64).max() / np.abs(p64).max() < 1e-3


def test_deferred_full_size(hip):
    """BASELINE configs[1] size: N = 1000, 2 ticks x 16 corrections, deferred vs eager."""
    n, m, T = 1000, 16, 2
    lm = synth.make_landmarks(n)
    tr = synth.make_trace(n, T, m, landmarks=lm)
    bx, by, ids = synth.warmup_observations(lm)
    out = []
    for deferred in (False, True):
        g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
        g.tick(np.zeros(3), bx, by, known_ids=ids, want_ids=False)
        g.set_deferred(deferred)
        for t in range(T):
            g.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t], want_ids=False)
        out.append((g.state, g.cov))
    es, ep = entry_rel_err(out[1][0], out[0][0]), entry_rel_err(out[1][1], out[0][1])
    print("deferred N=1000 vs eager: state %.1e cov %.1e" % (es, ep))
    assert es < 1e-8 and ep < 1e-6


def test_deferred_with_data_association_falls_back_to_flush(hip):
    """Deferred mode covers known association only: an associateLandmark forces a flush first and that marker's
    correction takes the eager kernel, so a data-association trace gives the eager result."""
    n, T, m = 12, 10, 4
    tr = synth.make_trace(n, T, m, noise_sigma=1e-3)
    ge = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    ge.as_batch().set_pass_variant(hip.PASS_EXACT)      # (bit equality below is a statement about the exact chain)
    gd = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    gd.set_deferred(True)
    for t in range(T):
        ie = ge.tick(tr.tw[t], tr.mx[t], tr.my[t])
        idd = gd.tick(tr.tw[t], tr.mx[t], tr.my[t])
        assert np.array_equal(ie, idd)
    assert np.array_equal(gd.state, ge.state) and np.array_equal(gd.cov, ge.cov) and gd.seen == ge.seen
