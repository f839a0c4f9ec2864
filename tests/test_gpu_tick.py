"""The tick pipeline (csrc/ekf_tick.h: k_tick_chain -> k_tick_panels -> k_tick_apply, ONE pass over P per tick) against
the one-pass-per-correction kernel it replaces: the same floating-point operations on every element in the same order,
so state, covariance, `seen`, resolved ids and latched status must agree BIT FOR BIT -- through landmark
initialisation (INT_MAX diagonal), skipped markers (id < 0), the break of the marker loop (id > total_landmarks), bad
ids, repeated ids inside one tick, fewer and more markers than one round of 16, fp32 storage and batches with
per-filter ids.  (Against the oracle the default path is exercised by every other GPU test.)

These are statements about the EXACT chain (nuslam_batch_set_pass_variant PASS_EXACT / PASS_EXACT_PLAIN); the default pass,
the rank-2m update on the matrix cores, is the same algebra re-associated and is held to a tolerance instead
(tests/test_gpu_rank.py, tests/test_gpu_depth.py) -- except that the two stream orders of a run must agree bit for bit
for EVERY pass variant (test_overlapped_run_is_bit_identical).""" 
import numpy as np
import pytest

import _oracle as O
from nuslam_hip import synth

pytestmark = pytest.mark.gpu
Q, R = synth.Q_DEFAULT, synth.R_DEFAULT


def pair_of_filters(hip, n, dtype=0, mode=1):
    """mode 1: the tick pipeline as the library launches it for one filter (chain and strips in ONE launch, k_tick_front);
    mode 3: chain and strips as two launches (what batches and overlapped runs take)"""
    a = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R, dtype=dtype)
    b = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R, dtype=dtype)
    a.as_batch().set_tick_mode(mode)
    a.as_batch().set_pass_variant(hip.PASS_EXACT)
    bb = b.as_batch()
    bb.set_tick_mode(0)
    bb.set_pairing(False)                       # one k_update launch per correction: the reference arithmetic
    return a, b


def same(a, b):
    return np.array_equal(a.state, b.state) and np.array_equal(a.cov, b.cov) and a.seen == b.seen and a.status() == b.status()


@pytest.mark.parametrize("mode", [1, 3], ids=["one-launch", "two-launches"])
@pytest.mark.parametrize("n,m,dtype", [(10, 10, 0), (10, 3, 0), (40, 16, 0), (40, 16, 1), (60, 37, 0), (6, 1, 0)])
def test_tick_pipeline_equals_per_correction_kernels_cold_start(hip, n, m, dtype, mode):
    tr = synth.make_trace(n, 6, m, straight_every=3)
    a, b = pair_of_filters(hip, n, dtype, mode)
    for t in range(tr.ticks):
        ia = a.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t])
        ib = b.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t])
        assert np.array_equal(ia, ib)
        assert same(a, b), "tick %d" % t


@pytest.mark.parametrize("mode", [1, 3], ids=["one-launch", "two-launches"])
def test_tick_pipeline_decision_chain_edges(hip, mode):
    """skip (id -1), break (id > total_landmarks: the rest of the tick is dropped), out-of-range id (latched
    NUSLAM_E_BOUNDS), the same landmark twice in one tick, a first sighting followed by a re-sighting in one tick."""
    n, m = 12, 8
    tr = synth.make_trace(n, 5, m)
    a, b = pair_of_filters(hip, n, 0, mode)
    cases = []
    ids = tr.ids[0].copy(); ids[2] = -1; ids[5] = ids[1]; cases.append((ids, n))            # skip + duplicate (second one: re-sighting)
    ids = tr.ids[1].copy(); ids[3] = -1; ids[4] = ids[0]; cases.append((ids, n))
    ids = tr.ids[2].copy(); cases.append((ids, 5))                                          # total_landmarks = 5: ids > 5 break the loop
    ids = tr.ids[3].copy(); ids[1] = ids[0]; ids[2] = ids[0]; cases.append((ids, n))        # three times the same landmark
    for t, (ids, total) in enumerate(cases):
        ia = a.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=ids, total_landmarks=total)
        ib = b.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=ids, total_landmarks=total)
        assert np.array_equal(ia, ib), (t, ia, ib)
        assert same(a, b), "case %d" % t
    ids = tr.ids[4].copy(); ids[2] = n + 3                                                  # bad id: both latch E_BOUNDS
    for f in (a, b):
        with pytest.raises(hip.NuslamError) as ei:
            f.tick(tr.tw[4], tr.mx[4], tr.my[4], known_ids=ids)
        assert ei.value.code == hip.E_BOUNDS
    assert np.array_equal(a.state, b.state) and np.array_equal(a.cov, b.cov) and a.seen == b.seen


def test_tick_pipeline_batch_with_per_filter_ids(hip):
    n, m, T, B = 30, 16, 4, 5
    traces = [synth.make_trace(n, T, m, seed=300 + k) for k in range(B)]
    tw = np.stack([t.tw[:, :2] for t in traces]); mx = np.stack([t.mx for t in traces])
    my = np.stack([t.my for t in traces]); ids = np.stack([t.ids for t in traces])
    out = []
    for mode in (1, 0):
        bt = hip.Batch(B, n, Q, R)
        bt.set_tick_mode(mode)
        bt.set_pass_variant(hip.PASS_EXACT)
        if mode == 0:
            bt.set_pairing(False)
        bt.load_trace(tw, mx, my, ids)
        bt.run(0, T)
        assert bt.status() == (-1, 0)
        out.append([(bt.state(k), bt.cov(k), bt.seen(k)) for k in range(B)])
    for k in range(B):
        assert np.array_equal(out[0][k][0], out[1][k][0]) and np.array_equal(out[0][k][1], out[1][k][1])
        assert out[0][k][2] == out[1][k][2]


def test_tick_pipeline_n1000_matches_oracle_and_pairs(hip):
    """BASELINE configs[1] size: 2 ticks x 16 corrections from the oracle's post-initialisation snapshot: the tick
    pipeline == the pair kernel == the single kernel bitwise, and within 1e-6 per entry of the oracle."""
    n, m, T = 1000, 16, 2
    lm = synth.make_landmarks(n)
    tr = synth.make_trace(n, T, m, landmarks=lm)
    bx, by, wid = synth.warmup_observations(lm)
    O.set_threads(O.usable_cpus())
    o = O.OracleEKF(np.zeros(3), np.zeros(2 * n), Q, R, O.ORC_STRUCTURED)
    o.tick(tw=np.zeros(3), mx=bx, my=by, known_ids=wid)
    fs = []
    for mode, pairing in ((1, True), (3, True), (0, True), (0, False)):
        g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
        g.restore(o.state.copy(), o.cov.copy(), o.seen)
        bt = g.as_batch()
        bt.set_tick_mode(mode)
        bt.set_pass_variant(hip.PASS_EXACT)
        bt.set_pairing(pairing)
        fs.append(g)
    for t in range(T):
        o.tick(tw=tr.tw[t], mx=tr.mx[t], my=tr.my[t], known_ids=tr.ids[t])
        for g in fs:
            g.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t], want_ids=False)
    O.set_threads(1)
    P0 = fs[0].cov
    for g in fs[1:]:
        assert np.array_equal(fs[0].state, g.state) and np.array_equal(P0, g.cov)
    rel = lambda x, ref: float((np.abs(x - ref) / np.maximum(np.abs(ref), 1e-12 * np.abs(ref).max())).max())
    es, ep = rel(fs[0].state, o.state), rel(P0, o.cov)
    print("N=1000 tick pipeline: == pair kernel == single kernel bitwise; vs oracle state %.2e cov %.2e" % (es, ep))
    assert es < 1e-6 and ep < 1e-6


@pytest.mark.parametrize("variant", ["rank", "exact", "exact_plain"])
@pytest.mark.parametrize("B,n,m,dtype,cold", [(1, 40, 16, 0, False), (3, 30, 5, 0, False), (4, 20, 16, 1, False), (2, 12, 8, 0, True),
                                                (96, 10, 6, 0, True),      # (more chains than a quarter of the CUs: a one-wave kernel waits for them)
                                                (1, 12, 3, 0, False),      # (a small filter, few markers: the hand-offs' margins are a few us)
                                                (1, 1000, 16, 0, False)])  # BASELINE configs[1] size: the bench's default path
def test_overlapped_run_is_bit_identical(hip, B, n, m, dtype, cold, variant):
    """nuslam_batch_run with the chain of tick t+1 running ahead on its own stream (it forms its starting block from
    tick t's strips itself: tick_carry) against the one-stream order: same state, covariance, seen, status -- warm, through first
    sightings (cold), with skipped markers and a straight tick (dth == 0) in the trace, fp64 and fp32 storage."""
    T = 7
    traces = [synth.make_trace(n, T, m, seed=400 + k, straight_every=3) for k in range(B)]
    tw = np.stack([t.tw[:, :2] for t in traces]); mx = np.stack([t.mx for t in traces])
    my = np.stack([t.my for t in traces]); ids = np.stack([t.ids for t in traces]).copy()
    ids[:, 2, 1] = -1                                            # a skipped marker in tick 2
    if m > 2:
        ids[:, 4, 2] = ids[:, 4, 0]                              # the same landmark twice in tick 4
    res = []
    for overlap in (True, False):
        bt = hip.Batch(B, n, Q, R, dtype=dtype)
        bt.set_tick_mode(1)
        bt.set_pass_variant({"rank": hip.PASS_RANK, "exact": hip.PASS_EXACT, "exact_plain": hip.PASS_EXACT_PLAIN}[variant])
        bt.set_overlap(overlap)
        if not cold:
            bx, by, wid = synth.warmup_observations(traces[0].landmarks)
            bt.load_trace(np.zeros((1, 2)), bx[None, :], by[None, :], wid[None, :], bcast=True)
            bt.run(0, 1)
        if B == 1:
            bt.load_trace(tw[0], mx[0], my[0], ids[0], bcast=True)
        else:
            bt.load_trace(tw, mx, my, ids)
        bt.run(0, 3)                                             # two runs: the hand-off chain restarts from memory
        bt.run(3, T)
        assert bt.status() == (-1, 0)
        res.append([(bt.state(k), bt.cov(k), bt.seen(k)) for k in range(B)])
    for k in range(B):
        assert np.array_equal(res[0][k][0], res[1][k][0]), "state of filter %d" % k
        assert np.array_equal(res[0][k][1], res[1][k][1]), "covariance of filter %d" % k
        assert res[0][k][2] == res[1][k][2]


@pytest.mark.parametrize("variant", ["rank", "exact"])
def test_overlapped_run_falls_back_when_the_streams_do_not_run_side_by_side(hip, variant):
    """Overlap robustness: with the chain "stream" forced to be the handle's own (set_overlap(2)) nothing can run beside the
    handle's kernels -- what a profiler's counter pass or a serialising environment does to two real streams.  The
    handle's probe must notice, the run must take the one-stream order, and state / covariance / status must equal the
    one-stream run bit for bit (no NUSLAM_E_SYNC, no poisoned handle)."""
    n, m, T = 40, 16, 7
    lm = synth.make_landmarks(n)
    tr = synth.make_trace(n, T, m, landmarks=lm, straight_every=3, dL=0.3125, dR=0.375)
    bx, by, wid = synth.warmup_observations(lm)
    res = []
    for overlap in (2, False):
        bt = hip.Batch(1, n, Q, R)
        bt.set_tick_mode(1)
        bt.set_pass_variant(hip.PASS_RANK if variant == "rank" else hip.PASS_EXACT)
        bt.set_overlap(overlap)
        bt.load_trace(np.zeros((1, 2)), bx[None, :], by[None, :], wid[None, :], bcast=True)
        bt.run(0, 1)
        bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, tr.ids, bcast=True)
        bt.run(0, 4)
        bt.run(4, T)
        assert bt.status() == (-1, 0)
        res.append((bt.state(0), bt.cov(0), bt.seen(0)))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]
