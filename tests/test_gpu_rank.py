"""The rank-2m pass over P (csrc/ekf_rank.h: P <- P - [K_1 .. K_m][V_1; ..; V_m] on v_mfma_f64_16x16x4_f64, the default of a
tick pipeline) against the exact chain it replaces (k_tick_apply: (I - K H) P entry by entry as the oracle writes it,
slam_library.cpp:279) -- the same algebra re-associated, so the two agree to rounding, not bit for bit:

  * rounds that hold a first sighting are routed to the exact chain on the device: a cold start stays BIT-identical while
    landmarks are being initialised, and only then drifts apart by rounding;
  * skipped markers (id < 0), the break of the marker loop, the same landmark twice, fewer and more markers than a round
    of 16, fp32 storage, batches with per-filter ids, unknown data association (verdicts must be equal);
  * every tile shape of the kernel computes every element by the same k-ordered fma chain: identical bits whatever the
    tiling, and for a filter inside a batch as for the same filter alone.
Against the oracle the rank-2m pass is held to the 1e-6 contract at depth in tests/test_gpu_depth.py."""
import numpy as np
import pytest

from nuslam_hip import synth

pytestmark = pytest.mark.gpu
Q, R = synth.Q_DEFAULT, synth.R_DEFAULT
# wheel increments that are exact binary fractions: a straight tick then has dth == 0.0 exactly.  (With 0.30 / 0.36 it comes
# out as ~1e-16, takes the arc branch, slam_library.cpp:77, and turns the 1e-13 by which two correct filters differ into
# millimetres in one predict -- tests/test_trace_conditioning.py.)
EXACT_WHEELS = dict(dL=0.3125, dR=0.375)


def entry_rel_err(a, ref):
    a, ref = np.asarray(a), np.asarray(ref)
    return float((np.abs(a - ref) / np.maximum(np.abs(ref), 1e-12 * np.abs(ref).max())).max())


def pair(hip, n, dtype=0, Qm=Q, mode=1):
    a = hip.EKF(np.zeros(3), np.zeros(2 * n), Qm, R, dtype=dtype)
    b = hip.EKF(np.zeros(3), np.zeros(2 * n), Qm, R, dtype=dtype)
    for f, variant in ((a, hip.PASS_RANK), (b, hip.PASS_EXACT)):
        bt = f.as_batch()
        bt.set_tick_mode(mode)
        bt.set_pass_variant(variant)
    return a, b


@pytest.mark.parametrize("n,m", [(10, 10), (10, 3), (40, 16), (60, 37), (6, 1)])
def test_cold_start_is_exact_while_landmarks_appear_then_within_rounding(hip, n, m):
    """Rounds that cancel an INT_MAX diagonal take the exact chain in both filters: tick 0 (every landmark new) must agree
    bit for bit.  After it the rank-2m pass differs by rounding (~1e-16 of max|P|) -- until the NEXT new landmark appears:
    that correction is ill-conditioned (the oracle itself moves by 3e-5 when an input moves by one
    ulp) and amplifies the 1e-13 the states differ by then to ~1e-7 in its covariance, exact chain or not.  So rounding
    level is asserted up to that tick, and only sanity after it."""
    T = 12
    tr = synth.make_trace(n, T, m, straight_every=3, **EXACT_WHEELS)
    assert len(set(tr.ids[0].tolist())) == m          # tick 0: m distinct landmarks, all first sightings, in every round
    a, b = pair(hip, n)
    known = set()
    clean = True                                      # no landmark has appeared since tick 0
    worst = 0.0
    for t in range(T):
        ia = a.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t])
        ib = b.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t])
        assert np.array_equal(ia, ib) and a.seen == b.seen
        if t == 0:                                    # every round held a first sighting: the exact chain ran, same bits
            assert np.array_equal(a.cov, b.cov) and np.array_equal(a.state, b.state)
        elif any(int(i) not in known for i in tr.ids[t]):
            clean = False
        known.update(int(i) for i in tr.ids[t])
        Pa, Pb = a.cov, b.cov
        fin = np.abs(Pb) < 1e9
        scale = np.abs(Pb[fin]).max()
        ds, dp = float(np.abs(a.state - b.state).max()), float(np.abs(Pa - Pb)[fin].max() / scale)
        if clean:
            worst = max(worst, ds, dp)
            assert ds < 1e-10 and dp < 1e-13, "tick %d: state %.2e cov %.2e" % (t, ds, dp)
        assert np.array_equal(Pa[~fin], Pb[~fin]) and np.isfinite(a.state).all()
    print("n=%d m=%d: rank-2m vs exact chain from a cold start, until the next new landmark: worst abs %.2e" % (n, m, worst))


@pytest.mark.parametrize("n,m,dtype", [(10, 10, 0), (40, 16, 0), (60, 37, 0), (200, 16, 1), (200, 16, 0)])
def test_warm_trajectory_within_rounding_of_the_exact_chain(hip, n, m, dtype):
    T = 20
    lm = synth.make_landmarks(n)
    # (fp32 storage differs by 1e-7 per rounding: only on a trace that keeps the reference algorithm itself well-conditioned
    # -- limited field of view, tests/test_trace_conditioning.py -- do two correct fp32 filters stay together for 20 ticks)
    tr = (synth.make_wellposed_trace(n, T, m, landmarks=lm, straight_every=4) if n >= 200
          else synth.make_trace(n, T, m, landmarks=lm, straight_every=4, **EXACT_WHEELS))
    bx, by, wid = synth.warmup_observations(lm)
    a, b = pair(hip, n, dtype)
    for f in (a, b):
        f.tick(np.zeros(3), bx, by, known_ids=wid, want_ids=False)      # first sightings: exact chain in both
    assert np.array_equal(a.cov, b.cov) and np.array_equal(a.state, b.state)
    ids = tr.ids.copy()
    ids[3, 1] = -1                       # a skipped marker
    if m > 2:
        ids[5, 2] = ids[5, 0]            # the same landmark twice in one tick
    for t in range(T):
        total = 5 if t == 7 else n       # tick 7: ids > 5 break the marker loop (slam.cpp:301-316)
        ia = a.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=ids[t], total_landmarks=total)
        ib = b.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=ids[t], total_landmarks=total)
        assert np.array_equal(ia, ib)
    assert a.status() == 0 and b.status() == 0 and a.seen == b.seen
    es, ep = entry_rel_err(a.state, b.state), entry_rel_err(a.cov, b.cov)
    scale = np.abs(b.cov).max()
    ea = float(np.abs(a.cov - b.cov).max() / scale)
    print("n=%d m=%d %s: rank-2m vs exact chain after %d warm ticks: state %.2e, cov per entry %.2e, max|dP|/max|P| %.2e"
          % (n, m, "fp32" if dtype else "fp64", T, es, ep, ea))
    if dtype == 0:
        assert es < 1e-9 and ep < 1e-7 and ea < 1e-12
    else:
        # fp32 storage: the exact chain rounds to fp32 after every correction, the rank-2m pass once per round; the bounds are
        # the ones the fp32 path is held to against the fp64 oracle, in absolute terms
        assert float(np.abs(a.state - b.state).max()) < 1e-3 and ea < 1e-4


def test_batch_with_per_filter_ids_and_tile_shapes(hip):
    """Filter b of a batch == the same filter alone, bit for bit, and every tile shape of the kernel gives the same bits."""
    n, m, T, B = 30, 16, 6, 11
    lm = synth.make_landmarks(n)
    traces = [synth.make_trace(n, T, m, seed=300 + k, landmarks=lm, **EXACT_WHEELS) for k in range(B)]
    tw = np.stack([t.tw[:, :2] for t in traces]); mx = np.stack([t.mx for t in traces])
    my = np.stack([t.my for t in traces]); ids = np.stack([t.ids for t in traces])
    bx, by, wid = synth.warmup_observations(lm)
    out = {}
    for dtype in (hip.F64, hip.F32):
        for variant in (hip.PASS_RANK, 10, 11, 12, 13, hip.PASS_EXACT):
            bt = hip.Batch(B, n, Q, R, dtype=dtype)
            bt.set_pass_variant(variant)
            bt.load_trace(np.zeros((1, 2)), bx[None, :], by[None, :], wid[None, :], bcast=True)
            bt.run(0, 1)
            bt.load_trace(tw, mx, my, ids)
            bt.run(0, T)
            assert bt.status() == (-1, 0)
            out[(dtype, variant)] = [(bt.state(k), bt.cov(k)) for k in range(B)]
        ref = out[(dtype, hip.PASS_RANK)]
        for variant in (10, 11, 12, 13):
            for k in range(B):
                assert np.array_equal(ref[k][0], out[(dtype, variant)][k][0]) and np.array_equal(ref[k][1], out[(dtype, variant)][k][1]), (dtype, variant, k)
        ex = out[(dtype, hip.PASS_EXACT)]
        for k in range(B):
            ea = np.abs(ref[k][1] - ex[k][1]).max() / np.abs(ex[k][1]).max()
            assert ea < (1e-12 if dtype == hip.F64 else 1e-4), (dtype, k, ea)
        # one filter of the batch alone (B = 1 takes the other workgroup -> tile mapping)
        k = 7
        g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R, dtype=dtype)
        g.tick(np.zeros(3), bx, by, known_ids=wid, want_ids=False)
        for t in range(T):
            g.tick(traces[k].tw[t], traces[k].mx[t], traces[k].my[t], known_ids=traces[k].ids[t], want_ids=False)
        assert np.array_equal(g.state, ref[k][0]) and np.array_equal(g.cov, ref[k][1])


@pytest.mark.parametrize("n,B,dtype", [(30, 1100, 0), (200, 150, 0), (30, 1100, 1)])
def test_batch_strips_a_lane_per_index_same_bits_as_the_quads(hip, n, B, dtype):
    """Enough filters to fill the chip four times over, a round the host proves free of first sightings: the strips are formed by
    k_tick_strips_lane (a lane per index and role, the plan through the scalar cache) -- against k_tick_panels<T, 64> (variant 20:
    quads, the plan in LDS) on the same trace: per-filter ids, a skipped marker, the same landmark twice, a bad id, fewer markers
    than a round.  Identical state and covariance in every filter."""
    m, T, K = 16, 5, 8
    lm = synth.make_landmarks(n)
    traces = [synth.make_trace(n, T, m, seed=500 + k, landmarks=lm, **EXACT_WHEELS) for k in range(K)]
    pick = np.arange(B) % K
    tw = np.stack([t.tw[:, :2] for t in traces])[pick]; mx = np.stack([t.mx for t in traces])[pick]
    my = np.stack([t.my for t in traces])[pick]; ids = np.stack([t.ids for t in traces])[pick].copy()
    ids[1::3, 1, 4] = -1                              # a marker update() is not called for
    ids[2::5, 2, 7] = ids[2::5, 2, 3]                 # the same landmark twice in a round
    ids[4::7, 3, 0] = n + 5                           # an id outside the map: NUSLAM_E_BOUNDS latched, nothing applied
    bx, by, wid = synth.warmup_observations(lm)
    out = []
    for variant in (hip.PASS_RANK, 20):
        bt = hip.Batch(B, n, Q, R, dtype=dtype)
        bt.set_pass_variant(variant)
        bt.load_trace(np.zeros((1, 2)), bx[None, :], by[None, :], wid[None, :], bcast=True)
        bt.run(0, 1)
        bt.load_trace(tw, mx, my, ids)
        bt.run(0, T)
        bt.load_trace(tw[:, :2], mx[:, :2, :5], my[:, :2, :5], ids[:, :2, :5])      # five markers per tick
        bt.run(0, 2)
        out.append((bt.status(), [(bt.state(k), bt.cov(k)) for k in range(0, B, max(1, B // 64))]))
    assert out[0][0] == out[1][0]
    for (s0, p0), (s1, p1) in zip(out[0][1], out[1][1]):
        assert np.isfinite(p0).all()
        assert np.array_equal(s0, s1) and np.array_equal(p0, p1)


@pytest.mark.parametrize("n,B,dtype", [(200, 600, 0), (30, 1300, 1)])
def test_groups_of_filters_on_their_own_streams_same_bits(hip, n, B, dtype):
    """nuslam_batch_run of a large batch as 1, 2, 3, 4 groups of filters on streams of their own (the default picks by size): the same
    kernels on the same per-filter data -- identical state, covariance and statistics; a getter between two runs waits for every stream."""
    m, T, K = 16, 6, 8
    lm = synth.make_landmarks(n)
    traces = [synth.make_trace(n, T, m, seed=700 + k, landmarks=lm, **EXACT_WHEELS) for k in range(K)]
    pick = np.arange(B) % K
    tw = np.stack([t.tw[:, :2] for t in traces])[pick]; mx = np.stack([t.mx for t in traces])[pick]
    my = np.stack([t.my for t in traces])[pick]; ids = np.stack([t.ids for t in traces])[pick].copy()
    ids[1::3, 1, 4] = -1
    bx, by, wid = synth.warmup_observations(lm)
    out = []
    for groups in (1, 2, 3, 4, -1, 12):
        bt = hip.Batch(B, n, Q, R, dtype=dtype)
        bt.set_interleave(groups)
        bt.load_trace(np.zeros((1, 2)), bx[None, :], by[None, :], wid[None, :], bcast=True)
        bt.run(0, 1)
        bt.load_trace(tw, mx, my, ids)
        bt.run(0, 3)
        mid = bt.state(B - 1).copy()                   # (a filter of the last group, in mid-run)
        bt.run(3, T)
        assert bt.status() == (-1, 0)
        out.append((mid, bt.stats(), [(bt.state(k), bt.cov(k)) for k in range(0, B, max(1, B // 48))]))
    for o in out[1:]:
        assert np.array_equal(out[0][0], o[0]) and np.array_equal(out[0][1], o[1])
        for (s0, p0), (s1, p1) in zip(out[0][2], o[2]):
            assert np.array_equal(s0, s1) and np.array_equal(p0, p1)


@pytest.mark.parametrize("mode", [1, 2], ids=["resident-round", "launch-per-marker"])
def test_unknown_association_same_verdicts_covariance_within_rounding(hip, mode):
    """associateLandmark in front of every correction (slam_library.cpp:188-253): the verdicts are threshold decisions on
    distances that differ by ~1e-13 between the two passes; on this trace none sits on a threshold, so ids and `seen` agree."""
    n, n_world, m, T = 40, 36, 16, 12
    Qs = np.diag([1e-4, 1e-4, 1e-4])
    lm = synth.make_landmarks(n_world)
    tr = synth.make_trace(n_world, T, m, landmarks=lm, noise_sigma=1e-4, straight_every=5, **EXACT_WHEELS)
    a, b = pair(hip, n, 0, Qs, mode)
    new_seen = []
    for t in range(T):
        ia = a.tick(tr.tw[t], tr.mx[t], tr.my[t])
        ib = b.tick(tr.tw[t], tr.mx[t], tr.my[t])
        assert np.array_equal(ia, ib), (t, ia, ib)
        assert a.seen == b.seen
        new_seen.append(a.seen)
    assert new_seen[-1] > new_seen[0] > 0 and a.status() == 0 and b.status() == 0     # landmarks kept appearing
    es, ep = entry_rel_err(a.state, b.state), entry_rel_err(a.cov, b.cov)
    print("unknown association, rank-2m vs exact chain: state %.2e cov %.2e" % (es, ep))
    assert es < 1e-6 and np.abs(a.cov - b.cov).max() / np.abs(b.cov).max() < 1e-9


@pytest.mark.parametrize("n,m,dtype,cold", [(40, 16, 0, False), (12, 5, 0, True), (60, 37, 0, False), (30, 16, 1, False)])
def test_rank_form_strips_same_bits_in_every_launch_form(hip, n, m, dtype, cold):
    """With the rank-2m pass the strips carry their panels in rank form in rounds without a first sighting (ekf_tick.h).  Which form
    a filter's round takes must not depend on the launch form it runs in: chain and strips as ONE launch (k_tick_front: the
    strips take the form the host can prove) against chain, strips, pass as separate launches (k_tick_panels decides from the
    round's own flags) -- warm, through first sightings (cold: the host cannot prove anything, the one-launch form steps aside),
    with a skipped marker, a repeated landmark, more markers than a round, fp32 storage: identical state, covariance, seen."""
    T = 6
    tr = synth.make_trace(n, T, m, seed=77, straight_every=3, **EXACT_WHEELS)
    ids = tr.ids.copy()
    ids[2, 1 % m] = -1
    if m > 2:
        ids[4, 2] = ids[4, 0]
    res = []
    for mode in (1, 3):                       # 1: one launch (the default), 3: separate launches
        f = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R, dtype=dtype)
        bt = f.as_batch()
        bt.set_tick_mode(mode)
        bt.set_pass_variant(hip.PASS_RANK)
        if not cold:
            bx, by, wid = synth.warmup_observations(tr.landmarks)
            bt.load_trace(np.zeros((1, 2)), bx[None, :], by[None, :], wid[None, :], bcast=True)
            bt.run(0, 1)
        bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, ids, bcast=True)
        bt.run(0, T)
        assert bt.status() == (-1, 0)
        res.append((f.state.copy(), f.cov.copy(), f.seen))
    assert np.array_equal(res[0][0], res[1][0]), "state"
    assert np.array_equal(res[0][1], res[1][1]), "covariance"
    assert res[0][2] == res[1][2]


@pytest.mark.parametrize("n,m,dtype", [(1000, 16, 0), (200, 16, 0), (200, 7, 0), (300, 16, 1), (40, 16, 0)])
def test_fused_tick_same_bits_as_the_separate_launches(hip, n, m, dtype):
    """One filter's known-id tick as ONE launch (csrc/ekf_fused.h: predict || chain || strips || the rank-2m pass, whose tile loads
    run under the serial chain and whose k-steps follow the strips entry by entry) against the same tick with the pass as a launch
    of its own behind k_tick_front (tick mode 4, round 3's default) and with chain / strips / pass as three launches (mode 3):
    every element takes k_tick_rank's k-ordered fma chain in all three -- identical bits, tick after tick, with skipped markers,
    fewer markers than a round, fp32 storage, and a first sighting in mid-run (that round steps aside to the two-launch form)."""
    T = 12
    lm = synth.make_landmarks(n)
    tr = synth.make_wellposed_trace(n, T, m, landmarks=lm, straight_every=4) if n >= 200 else synth.make_trace(n, T, m, straight_every=4, **EXACT_WHEELS)
    bx, by, wid = synth.warmup_observations(lm)
    fs = []
    for mode in (1, 4, 3):
        f = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R, dtype=dtype)
        f.as_batch().set_tick_mode(mode)
        # the map without its last landmark: that one appears at tick 6 as a first sighting
        f.tick(np.zeros(3), bx[:-1], by[:-1], known_ids=wid[:-1], want_ids=False)
        fs.append(f)
    for t in range(T):
        ids = tr.ids[t].copy()
        mx, my = tr.mx[t].copy(), tr.my[t].copy()
        if t == 3:
            ids[2] = -1                                                 # a skipped marker
        if t == 6:
            ids[1] = n; mx[1], my[1] = bx[-1], by[-1]                   # landmark n for the first time
        for f in fs:
            f.tick(tr.tw[t], mx, my, known_ids=ids, want_ids=False)
        if t in (0, 3, 6, 7, T - 1):
            s0, P0 = fs[0].state, fs[0].cov
            for f in fs[1:]:
                assert np.array_equal(s0, f.state) and np.array_equal(P0, f.cov), "tick %d" % t
    assert all(f.status() == 0 for f in fs) and len({f.seen for f in fs}) == 1
