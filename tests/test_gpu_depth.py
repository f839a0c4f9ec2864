"""Parity AT DEPTH, through the path the bench times (SURVEY 8d configs[1] and configs[3]).

The loop followed is the slam node's (nuslam/src/slam.cpp:269-318: predict, then update() per marker) replayed by
`nuslam_batch_run` on a resident trace -- with the chain of tick t+1 running ahead on its own stream (the bench's default
for one filter) and on one stream -- against the oracle's structured mode (asserted bit-equal to its dense mode in
tests/test_oracle.py) from the oracle's own post-initialisation snapshot:

  configs[1]  N = 1000 fp64, T = 200 ticks x 16 corrections; per-entry relative error with SURVEY 8(d)'s floor
              (|a - b| / max(|b|, 1e-12 max|P|)) asserted <= 1e-6 at ticks {1, 2, 10, 50, 200}; the growth is printed.
  configs[3]  four filters of the 1024 x N = 200 batch on device-generated traces over T = 100 ticks.

Every pass variant is held to the same bar: the exact chain (bit-identical to the per-correction kernel) and the
rank-2m pass on the matrix cores (P -= [K_1..K_m][V_1;..;V_m], the same algebra re-associated).

The traces are the WELL-POSED ones (synth.make_wellposed_trace; for the device-made traces a +-2 rad field of view and no
straight ticks).  On the all-around trace of SURVEY 8(d) as first written the 1e-6 contract cannot hold at depth for ANY
pair of implementations, the oracle against itself included: the reference does not wrap the bearing innovation
(slam_library.cpp:272), so landmarks behind the robot produce 2 pi innovations in nearly every tick and the sign of
each is a discontinuity; and a 'straight' tick whose dth rounds to 4.6e-17 instead of 0 takes the arc branch
(slam_library.cpp:77), where the last bit of the heading decides between a displacement of 0 and of 2.4 cm.  Measured
there: every GPU path (exact chain included) and the perturbed oracle all leave 1e-6 between ticks 10 and 50
(tests/test_trace_conditioning.py pins that on the CPU)."""
import numpy as np
import pytest

import _oracle as O
from nuslam_hip import synth

pytestmark = pytest.mark.gpu
Q, R = synth.Q_DEFAULT, synth.R_DEFAULT
TOL = 1e-6


def entry_rel_err(a, ref, scale=None):
    a, ref = np.asarray(a), np.asarray(ref)
    floor = 1e-12 * (np.abs(ref).max() if scale is None else scale)
    return float((np.abs(a - ref) / np.maximum(np.abs(ref), floor)).max())


def pass_variants(hip):
    """(name, set_pass_variant argument) of every pass over P this build has"""
    out = [("exact-chain", hip.PASS_EXACT)]
    if hasattr(hip, "PASS_RANK"):
        out.insert(0, ("rank-2m-mfma", hip.PASS_RANK))
    return out


def test_config1_n1000_t200_through_batch_run(hip):
    n, m, T = 1000, 16, 200
    checkpoints = (1, 2, 10, 50, 200)
    lm = synth.make_landmarks(n)
    tr = synth.make_wellposed_trace(n, T, m, landmarks=lm)
    bx, by, wid = synth.warmup_observations(lm)
    O.set_threads(O.usable_cpus())
    o = O.OracleEKF(np.zeros(3), np.zeros(2 * n), Q, R, O.ORC_STRUCTURED)
    o.tick(tw=np.zeros(3), mx=bx, my=by, known_ids=wid)
    snap = (o.state.copy(), o.cov.copy(), o.seen)
    runs = []
    for name, variant in pass_variants(hip):
        for overlap in (True, False):
            g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
            g.restore(*snap)
            bt = g.as_batch()
            bt.set_pass_variant(variant)
            bt.set_overlap(overlap)
            bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, tr.ids, bcast=True)
            runs.append(("%s, %s" % (name, "chain of tick t+1 overlapped" if overlap else "one stream"), g, bt, []))
    t_at = 0
    for cp in checkpoints:
        for t in range(t_at, cp):
            o.tick(tw=tr.tw[t], mx=tr.mx[t], my=tr.my[t], known_ids=tr.ids[t])
        for _, g, bt, errs in runs:
            bt.run(t_at, cp)           # (a one-tick segment runs on one stream whatever the setting: nothing to run ahead of)
            assert bt.status() == (-1, 0)
            errs.append((entry_rel_err(g.state, o.state), entry_rel_err(g.cov, o.cov),
                         float(np.linalg.norm(g.cov - o.cov) / np.linalg.norm(o.cov))))
            assert g.seen == o.seen
        t_at = cp
    O.set_threads(1)
    for name, _, _, errs in runs:
        print("N=1000 fp64 vs oracle, %s:" % name)
        for cp, (es, ep, ef) in zip(checkpoints, errs):
            print("   tick %3d: state %.2e   covariance per entry %.2e   Frobenius %.2e" % (cp, es, ep, ef))
    for name, _, _, errs in runs:
        for cp, (es, ep, ef) in zip(checkpoints, errs):
            assert es < TOL and ep < TOL, "%s: tick %d: state %.2e covariance %.2e" % (name, cp, es, ep)
    # the two stream orders of one pass variant are the same arithmetic
    for k in range(0, len(runs), 2):
        a, b = runs[k][1], runs[k + 1][1]
        assert np.array_equal(a.state, b.state) and np.array_equal(a.cov, b.cov), runs[k][0]
    # size-independent properties after 3200 corrections, each pass variant beside the oracle: the reference never
    # symmetrises P (no Joseph form) and the snapshot's own asymmetry -- left by the INT_MAX cancellation of the map's
    # initialisation, ~5e-6 of max|P| -- is carried along: the GPU must carry the SAME asymmetry as the oracle; P stays
    # positive semi-definite to rounding; its trace agrees with the oracle's
    Po = o.cov
    scale = np.abs(Po).max()
    asym_o = float(np.abs(Po - Po.T).max() / scale)
    for k in range(0, len(runs), 2):
        Pg = runs[k][1].cov
        asym = float(np.abs(Pg - Pg.T).max() / scale)
        lam_min = float(np.linalg.eigvalsh(0.5 * (Pg + Pg.T))[0])
        dtrace = abs(np.trace(Pg) - np.trace(Po)) / np.trace(Po)
        print("%s: max|P - P^T| / max|P| = %.1e (oracle %.1e), smallest eigenvalue %.2e (max|P| %.2e), trace vs oracle %.1e"
              % (runs[k][0].split(",")[0], asym, asym_o, lam_min, scale, dtrace))
        assert abs(asym - asym_o) < 1e-9 * max(asym_o, 1e-7) + 1e-12
        assert lam_min > -1e-10 * scale
        assert dtrace < 1e-10


def test_config3_batch_n200_t100_four_filters(hip):
    """The batch workload's own horizon (T = 100): filters 0, 341, 682, 1023 of 1024 Monte-Carlo trials on the traces
    the device made for them, against the oracle driven with exactly those traces."""
    B, n, m, T = 1024, 200, 16, 100
    checkpoints = (1, 10, 50, 100)
    probe = (0, 341, 682, 1023)
    lm = synth.make_landmarks(n)
    bx, by, wid = synth.warmup_observations(lm)
    uL, uR = 0.30 * 50, 0.36 * 50
    cmd = np.zeros((T, 2))
    cmd[:, 0] = (synth.WHEEL_RADIUS / synth.WHEEL_BASE) * (uR - uL)
    cmd[:, 1] = (synth.WHEEL_RADIUS / 2) * (uL + uR)
    sim = hip.SimParams(marker_sigma=float(np.sqrt(1e-3)), max_range=0.0, fov=synth.FOV_DEFAULT, min_range=synth.MIN_RANGE_DEFAULT)
    O.set_threads(O.usable_cpus())
    for name, variant in pass_variants(hip):
        bt = hip.Batch(B, n, Q, R)
        bt.set_pass_variant(variant)
        bt.load_trace(np.zeros((1, 2)), bx[None, :], by[None, :], wid[None, :], bcast=True)
        bt.run(0, 1)
        snap = {b: (bt.state(b), bt.cov(b), bt.seen(b)) for b in probe}
        bt.simulate(sim, lm, cmd, m, 12345, first_filter=0, known_ids=True)
        orcs = {}
        for b in probe:
            o = O.OracleEKF(np.zeros(3), np.zeros(2 * n), Q, R, O.ORC_STRUCTURED)
            o.restore(*snap[b])
            orcs[b] = (o, bt.get_trace(b))
        t_at, rows = 0, []
        for cp in checkpoints:
            bt.run(t_at, cp)
            assert bt.status() == (-1, 0)
            worst = (0.0, 0.0)
            for b in probe:
                o, trb = orcs[b]
                for t in range(t_at, cp):
                    tw = np.array([trb["tw"][t, 0], trb["tw"][t, 1], 0.0])
                    o.tick(tw=tw, mx=trb["mx"][t], my=trb["my"][t], known_ids=trb["ids"][t])
                worst = (max(worst[0], entry_rel_err(bt.state(b), o.state)), max(worst[1], entry_rel_err(bt.cov(b), o.cov)))
                assert bt.seen(b) == o.seen
            rows.append(worst)
            t_at = cp
        print("1024 x N=200 vs oracle (filters %s), %s:" % (probe, name))
        for cp, (es, ep) in zip(checkpoints, rows):
            print("   tick %3d: state %.2e   covariance per entry %.2e" % (cp, es, ep))
        for cp, (es, ep) in zip(checkpoints, rows):
            assert es < TOL and ep < TOL, "%s: tick %d: state %.2e covariance %.2e" % (name, cp, es, ep)
    O.set_threads(1)


# ------------------------------------------------------------------------------------------- configs[4] at depth
def test_config4_unknown_association_n1000_t100_at_depth(hip):
    """associateLandmark in front of every correction (slam_library.cpp:188-253 in the loop slam.cpp:279-318), N = 1000,
    T = 100 ticks x 16 markers on the bench's da1000 input (well-posed trace, 1e-4 m marker noise, Q = diag(1e-4), the map
    never full), on the DEFAULT path -- k_da_round (resident round) or k_da_begin + k_da_step per marker, each followed by the
    rank-2m pass k_tick_rank -- against the oracle's structured mode driven marker by marker:

      * the id every marker resolves to and `seen` equal the oracle's at EVERY tick (ids come from nuslam_ekf_tick, which runs
        the same kernels as nuslam_batch_run; a second filter goes through nuslam_batch_run in four runs and must equal the first
        bit for bit at the checkpoints);
      * every Mahalanobis distance a decision depends on keeps a >= 1e-6 relative margin from the 0.01 / 60 thresholds
        (slam_library.cpp:193-194), so rounding cannot legitimately flip a verdict -- asserted on the oracle's distances;
      * state / covariance within 1e-6 per entry at ticks {1, 10, 50, 100};
      * after tick 10: new landmarks (first sightings: the round goes through the exact chain on the device) re-observed
        later, and gray-zone markers (skipped).
    The served round of the class API (tests/test_gpu_lazy.py) runs the same kernel."""
    n, n_world, m, T = 1000, 990, 16, 100
    checkpoints = (1, 10, 50, 100)
    Qs = np.diag([1e-4, 1e-4, 1e-4])
    lm = synth.make_landmarks(n_world)
    tr = synth.make_wellposed_trace(n_world, T, m, landmarks=lm, noise_sigma=1e-4)
    bx, by, wid = synth.warmup_observations(lm, noise_sigma=1e-4)
    mx, my = tr.mx.copy(), tr.my.copy()
    # new landmarks far outside the grid at ticks 15, 40, 70 (ids 991, 992, 993), each re-observed three ticks later from the true
    # pose; gray-zone markers (a known landmark seen 5 cm off) at ticks 20, 45, 80
    news = {}
    for j, t0 in enumerate((15, 40, 70)):
        slot = 3 + j
        mx[t0, slot], my[t0, slot] = 18.0 + 3.0 * j, 1.0 - 2.0 * j
        th, x, y = tr.truth[t0]
        w = np.array([x + np.cos(th) * mx[t0, slot] - np.sin(th) * my[t0, slot], y + np.sin(th) * mx[t0, slot] + np.cos(th) * my[t0, slot]])
        th, x, y = tr.truth[t0 + 3]
        d = w - np.array([x, y])
        mx[t0 + 3, slot + 6], my[t0 + 3, slot + 6] = np.cos(th) * d[0] + np.sin(th) * d[1], -np.sin(th) * d[0] + np.cos(th) * d[1]
        news[t0] = (slot, n_world + 1 + j)
    for t0 in (20, 45, 80):
        mx[t0, 9] += 0.05

    O.set_threads(O.usable_cpus())
    o = O.OracleEKF(np.zeros(3), np.zeros(2 * n), Qs, R, O.ORC_STRUCTURED)
    o.tick(tw=np.zeros(3), mx=bx, my=by, known_ids=wid)
    assert o.seen == n_world
    snap = (o.state.copy(), o.cov.copy(), o.seen)
    # the oracle, marker by marker (slam.cpp:269-318 by hand), recording ids, `seen`, the decision margins and the checkpoints
    ids_o = np.zeros((T, m), dtype=np.int64)
    seen_o = np.zeros(T, dtype=np.int64)
    margins = []
    ref = {}
    for t in range(T):
        o.predict(tr.tw[t][0], tr.tw[t][1])
        cached = o.seen
        for i in range(m):
            z = O.cartesian2polar(mx[t, i], my[t, i])
            k, dk = o.associate(z[0], z[1], want_d=True)
            dk = dk[~np.isnan(dk)]
            if dk.size:
                margins.append(float(min(np.min(np.abs(dk - 0.01) / 0.01), np.min(np.abs(dk - 60.0) / 60.0))))
            ids_o[t, i] = k
            if k > cached:
                o.init_landmark(z[0], z[1], k)
            elif k < 0:
                continue
            o.update(z[0], z[1], k)
        seen_o[t] = o.seen
        if t + 1 in checkpoints:
            ref[t + 1] = (o.state.copy(), o.cov.copy())
    O.set_threads(1)
    assert min(margins) >= 1e-6, "a candidate distance sits on a threshold (%.2e): the trace proves nothing" % min(margins)
    for t0, (slot, new_id) in news.items():
        assert ids_o[t0, slot] == new_id and ids_o[t0 + 3, slot + 6] == new_id, (t0, ids_o[t0], ids_o[t0 + 3])
    assert all(ids_o[t0, 9] == -1 for t0 in (20, 45, 80)), [ids_o[t0, 9] for t0 in (20, 45, 80)]
    assert (ids_o > 0).mean() > 0.75 and seen_o[-1] == n_world + 3

    for name, mode in (("resident round", 1), ("launch per marker", 2)):
        g = hip.EKF(np.zeros(3), np.zeros(2 * n), Qs, R)                # tick by tick: the ids
        g.restore(*snap)
        g.as_batch().set_tick_mode(mode)
        h = hip.EKF(np.zeros(3), np.zeros(2 * n), Qs, R)                # nuslam_batch_run on the resident trace: the timed entry point
        h.restore(*snap)
        hb = h.as_batch()
        hb.set_tick_mode(mode)
        hb.load_trace(tr.tw[:, :2], mx, my, None, bcast=True)
        done = 0
        worst = {}
        for t in range(T):
            idg = g.tick(tr.tw[t], mx[t], my[t])
            assert np.array_equal(idg, ids_o[t]), "%s, tick %d: oracle %s gpu %s" % (name, t, ids_o[t], idg)
            assert g.seen == seen_o[t], (name, t)
            if t + 1 in checkpoints:
                hb.run(done, t + 1)
                done = t + 1
                gs, gP = g.state, g.cov
                assert np.array_equal(h.state, gs) and np.array_equal(h.cov, gP) and h.seen == g.seen, \
                    "%s: nuslam_batch_run and nuslam_ekf_tick part ways by tick %d" % (name, t + 1)
                os_, oP = ref[t + 1]
                es, ep = entry_rel_err(gs, os_), entry_rel_err(gP, oP)
                worst[t + 1] = (es, ep)
                assert es <= TOL and ep <= TOL, "%s, tick %d: state %.2e cov %.2e" % (name, t + 1, es, ep)
        assert g.status() == 0 and h.status() == 0
        print("N=1000 unknown association, %s + rank-2m pass, T=%d x %d: ids and seen equal the oracle's at every tick (%d matches, "
              "%d new, %d gray-zone; threshold margin %.1e); state / cov vs oracle: %s"
              % (name, T, m, (ids_o > 0).sum(), len(news), (ids_o < 0).sum(), min(margins),
                 ", ".join("t%d %.1e / %.1e" % (k, v[0], v[1]) for k, v in sorted(worst.items()))))
        g.close(); h.close()
