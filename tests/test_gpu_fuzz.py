"""Model-based fuzz: random sequences of C-ABI calls (predict / update / tick / associate / initializeLandmark /
clone / restore, eager and deferred) against the same sequence on the oracle, from a warm snapshot.
Every covariance produced by a single eager update must stay bit-identical; everything else within 1e-6."""
import numpy as np
import pytest

import _oracle as O
from nuslam_hip import synth

pytestmark = pytest.mark.gpu
Q, R = synth.Q_DEFAULT, synth.R_DEFAULT


def entry_rel_err(a, b):
    floor = 1e-12 * max(np.abs(b).max(), 1e-300)
    return (np.abs(a - b) / np.maximum(np.abs(b), floor)).max()


@pytest.mark.parametrize("seed", range(6))
def test_random_call_sequences(hip, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(3, 40))
    lm = synth.make_landmarks(n, seed=seed)
    o = O.OracleEKF(np.zeros(3), np.zeros(2 * n), Q, R, O.ORC_STRUCTURED)
    bx, by, ids = synth.warmup_observations(lm, seed=seed)
    o.tick(tw=np.zeros(3), mx=bx, my=by, known_ids=ids)
    g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    g.restore(o.state, o.cov, n)
    deferred = False
    tr = synth.make_trace(n, 60, min(n, 6), landmarks=lm, seed=seed)
    r, b = tr.polar()
    t = 0
    for step in range(40):
        op = rng.choice(["predict", "update", "tick", "tick", "clone", "restore", "toggle", "init", "sync"])
        if op == "predict":
            dth = float(rng.choice([0.0, rng.normal(0, 0.05)]))
            dx = float(rng.normal(0.01, 0.01))
            o.predict(dth, dx); g.predict(dth, dx)
        elif op == "update":
            i = int(rng.integers(0, tr.m))
            j = int(tr.ids[t, i])
            eager_bits = not deferred
            if eager_bits:
                g.restore(o.state, g.cov, o.seen)          # identical state going in: the covariance must match bit for bit
                pg = g.cov
                assert entry_rel_err(pg, o.cov) < 1e-6
                g.restore(o.state, o.cov, o.seen)
            o.update(r[t, i], b[t, i], j); g.update(r[t, i], b[t, i], j)
            if eager_bits:
                assert np.array_equal(g.cov, o.cov), "step %d: single eager update not bit-identical" % step
        elif op == "tick":
            ids = tr.ids[t].copy()
            if rng.random() < 0.3:
                ids[int(rng.integers(0, tr.m))] = -1        # a marker the caller skips
            o.tick(tw=tr.tw[t], mx=tr.mx[t], my=tr.my[t], known_ids=ids)
            g.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=ids, want_ids=bool(rng.random() < 0.5))
            t = (t + 1) % tr.ticks
        elif op == "clone":
            g = g.clone()                                   # the copy carries on; the original is dropped
            if deferred:
                g.set_deferred(True)
        elif op == "restore":
            g.restore(o.state, o.cov, o.seen)
        elif op == "toggle":
            deferred = not deferred
            g.set_deferred(deferred)
        elif op == "init":
            j = int(rng.integers(1, n + 1))
            rr, bb = float(rng.uniform(0.2, 1.0)), float(rng.uniform(-1, 1))
            o.init_landmark(rr, bb, j); g.init_landmark(rr, bb, j)
        elif op == "sync":
            g.sync()
        es, ep = entry_rel_err(g.state, o.state), entry_rel_err(g.cov, o.cov)
        assert es < 1e-6 and ep < 1e-6, "seed %d step %d (%s): state %.2e cov %.2e" % (seed, step, op, es, ep)
        assert g.seen == o.seen
    assert g.status() == 0
