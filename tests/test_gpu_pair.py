"""k_update2 (two consecutive corrections in one pass over P) must be BIT-IDENTICAL to two k_update launches -- state,
covariance, seen, status -- for every storage type, for repeated / adjacent / extreme landmark ids, for odd marker
counts, and it must step aside (fall back to single launches) whenever a marker is not a plain correction."""
import numpy as np
import pytest

import _oracle as O
from nuslam_hip import synth

pytestmark = pytest.mark.gpu
Q, R = synth.Q_DEFAULT, synth.R_DEFAULT


GROUPS = [2]              # k_update2 (the J-corrections-per-pass kernel k_updatej, slower than pairs, was retired in round 4)


def expected_launches(m, group):
    if group == 4:
        return m // 4 + (m % 4) // 2, m % 2
    return m // 2, m % 2


def warm_pair_of_filters(hip, n, dtype=0, seed=12345, group=2):
    lm = synth.make_landmarks(n, seed)
    o = O.OracleEKF(np.zeros(3), np.zeros(2 * n), Q, R, O.ORC_STRUCTURED)
    bx, by, ids = synth.warmup_observations(lm, seed=seed)
    o.tick(tw=np.zeros(3), mx=bx, my=by, known_ids=ids)
    gs = []
    for pairing in (True, False):
        g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R, dtype=dtype)
        g.restore(o.state, o.cov, n)
        g.as_batch().set_tick_mode(0)                    # these tests are about the per-pair kernels
        g.as_batch().set_pairing(group if pairing else 0)
        gs.append(g)
    return o, gs[0], gs[1], lm


def kernel_counts(g, fn):
    bt = g.as_batch()
    bt.profile(True)
    fn()
    g.sync()
    n2 = bt.profile_read(g_hip.K_UPDATE2)[1]
    n1 = bt.profile_read(g_hip.K_UPDATE)[1]
    bt.profile(False)
    return n1, n2


g_hip = None


@pytest.fixture(autouse=True)
def _bind(hip):
    global g_hip
    g_hip = hip


@pytest.mark.parametrize("group", GROUPS)
@pytest.mark.parametrize("n,m,dtype", [(2, 2, 0), (3, 3, 1), (10, 10, 0), (37, 16, 0), (64, 7, 0), (40, 16, 1), (127, 9, 0),
                                       (300, 8, 0)])
def test_pair_equals_two_singles_bitwise(hip, n, m, dtype, group):
    o, gp, gs, lm = warm_pair_of_filters(hip, n, dtype, group=group)
    tr = synth.make_trace(n, 5, m, landmarks=lm)
    for t in range(tr.ticks):
        n1, n2 = kernel_counts(gp, lambda: gp.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t], want_ids=False))
        assert (n2, n1) == expected_launches(m, group)           # groups, pairs, plus one single for an odd count
        gs.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t], want_ids=False)
        o.tick(tw=tr.tw[t], mx=tr.mx[t], my=tr.my[t], known_ids=tr.ids[t])
        assert np.array_equal(gp.state, gs.state), "tick %d: state differs" % t
        assert np.array_equal(gp.cov, gs.cov), "tick %d: covariance differs" % t
    assert gp.seen == gs.seen == o.seen and gp.status() == 0
    if dtype == 0:
        err = (np.abs(gp.cov - o.cov) / np.maximum(np.abs(o.cov), 1e-12 * np.abs(o.cov).max())).max()
        assert err < 1e-6


@pytest.mark.parametrize("group", GROUPS)
def test_pair_special_id_patterns(hip, group):
    """same landmark twice, neighbouring landmarks (overlapping index sets), the first and the last landmark."""
    n = 12
    o, gp, gs, lm = warm_pair_of_filters(hip, n, group=group)
    tr = synth.make_trace(n, 1, n, landmarks=lm)
    by_id = {int(i): k for k, i in enumerate(tr.ids[0])}
    for ids in ([3, 3], [4, 5], [5, 4], [1, n], [n, 1], [n, n], [1, 2, 1, 2], [7, 8, 9], [5, 5, 5, 5], [1, n, 2, n - 1],
                [3, 4, 3, 4, 6], [n, n - 1, n - 2, n - 3, 1, 2, 3, 4]):
        k = [by_id[i] for i in ids]
        mx, my = tr.mx[0][k], tr.my[0][k]
        gp.tick([0.01, 0.005, 0.0], mx, my, known_ids=ids, want_ids=False)
        gs.tick([0.01, 0.005, 0.0], mx, my, known_ids=ids, want_ids=False)
        o.tick(tw=[0.01, 0.005, 0.0], mx=mx, my=my, known_ids=ids)
        assert np.array_equal(gp.state, gs.state) and np.array_equal(gp.cov, gs.cov), ids
    assert np.abs(gp.state - o.state).max() < 1e-9


def test_pair_logs_ids_and_handles_two_marker_ticks(hip):
    n = 9
    o, gp, gs, lm = warm_pair_of_filters(hip, n)
    tr = synth.make_trace(n, 3, 2, landmarks=lm)
    for t in range(3):
        ip = gp.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t])
        isg = gs.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t])
        assert np.array_equal(ip, isg) and np.array_equal(ip, tr.ids[t])
        assert np.array_equal(gp.cov, gs.cov)


def test_pairing_steps_aside(hip):
    """a skipped marker, an uninitialised landmark, an id above total_landmarks or data association: single launches."""
    n = 10
    o, gp, gs, lm = warm_pair_of_filters(hip, n)
    tr = synth.make_trace(n, 4, 6, landmarks=lm)
    ids = tr.ids[0].copy(); ids[2] = -1
    n1, n2 = kernel_counts(gp, lambda: gp.tick(tr.tw[0], tr.mx[0], tr.my[0], known_ids=ids, want_ids=False))
    assert n2 == 0 and n1 == 6
    gs.tick(tr.tw[0], tr.mx[0], tr.my[0], known_ids=ids, want_ids=False)
    assert np.array_equal(gp.cov, gs.cov)
    n1, n2 = kernel_counts(gp, lambda: gp.tick(tr.tw[1], tr.mx[1], tr.my[1], known_ids=tr.ids[1], total_landmarks=3, want_ids=False))
    assert n2 == 0
    gs.tick(tr.tw[1], tr.mx[1], tr.my[1], known_ids=tr.ids[1], total_landmarks=3, want_ids=False)
    assert np.array_equal(gp.cov, gs.cov) and np.array_equal(gp.state, gs.state)
    # a filter that has seen only 4 landmarks: markers of landmarks 5.. initialise -> no pairing in that tick
    g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    g.restore(o.state, o.cov, 4)
    n1, n2 = kernel_counts(g, lambda: g.tick(tr.tw[2], tr.mx[2], tr.my[2], known_ids=tr.ids[2], want_ids=False))
    assert n2 == (0 if (tr.ids[2] > 4).any() else 3)
    # data association never pairs, and invalidates the host's mirror of `seen` for later ticks
    gp.restore(gs.state, gs.cov, n - 1)            # one free slot: associateLandmark indexes out of bounds on a full map
    n1, n2 = kernel_counts(gp, lambda: gp.tick(tr.tw[3], tr.mx[3], tr.my[3], want_ids=False))
    assert n2 == 0
    n1, n2 = kernel_counts(gp, lambda: gp.tick(tr.tw[3], tr.mx[3], tr.my[3], known_ids=tr.ids[3], want_ids=False))
    assert n2 == 0


def test_pair_batch_and_full_size(hip):
    n, m, T, B = 200, 16, 2, 3
    tr = synth.make_trace(n, T, m)
    bx, by, ids = synth.warmup_observations(tr.landmarks)
    res = []
    for pairing in (4, False, 2):
        bt = hip.Batch(B, n, Q, R)
        bt.set_tick_mode(0)
        bt.set_pairing(pairing)
        bt.load_trace(np.zeros((1, 2)), bx[None, :], by[None, :], ids[None, :], bcast=True)
        bt.run(0, 1)
        bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, tr.ids, bcast=True)
        bt.run(0, T)
        res.append((bt.state(B - 1), bt.cov(B - 1)))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert np.array_equal(res[2][0], res[1][0]) and np.array_equal(res[2][1], res[1][1])
    # N = 1000 (BASELINE configs[1])
    n = 1000
    lm = synth.make_landmarks(n)
    tr = synth.make_trace(n, 2, 16, landmarks=lm)
    bx, by, ids = synth.warmup_observations(lm)
    out = []
    for pairing in (4, False):
        g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
        g.as_batch().set_tick_mode(0)
        g.as_batch().set_pairing(pairing)
        g.tick(np.zeros(3), bx, by, known_ids=ids, want_ids=False)
        for t in range(2):
            g.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t], want_ids=False)
        out.append((g.state, g.cov))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_pair_with_per_filter_ids_in_resident_trace(hip):
    """B Monte-Carlo filters, each replaying its OWN trace (different landmark ids per filter): once every filter's
    map is initialised the host pairs the corrections, the kernel reads each filter's ids from the resident trace,
    and the result is bit for bit what single launches give.  While some filter still has a landmark to initialise
    (the cold first ticks) the host must not pair."""
    B, n, m, T = 3, 20, 6, 8
    lm = synth.make_landmarks(n)
    traces = [synth.make_trace(n, T, m, seed=100 + b, landmarks=lm) for b in range(B)]
    for b in range(B):                                   # make the id patterns differ between the filters
        traces[b].ids[:] = np.roll(traces[b].ids, b, axis=1)
        traces[b].mx[:] = np.roll(traces[b].mx, b, axis=1); traces[b].my[:] = np.roll(traces[b].my, b, axis=1)
    tw = np.stack([t.tw[:, :2] for t in traces]); mx = np.stack([t.mx for t in traces]); my = np.stack([t.my for t in traces])
    ids = np.stack([t.ids for t in traces])
    assert not np.array_equal(ids[0], ids[1])
    bx, by, wid = synth.warmup_observations(lm)
    res = []
    for pairing in (True, False):
        bt = hip.Batch(B, n, Q, R)
        bt.set_tick_mode(0)
        bt.set_pairing(pairing)
        bt.load_trace(tw, mx, my, ids)
        bt.profile(True)
        bt.run(0, 1)                                     # cold: landmarks get initialised, never paired
        bt.sync()
        assert bt.profile_read(hip.K_UPDATE2)[1] == 0 and bt.profile_read(hip.K_UPDATE)[1] == m
        bt.run(1, 3)
        bt.profile(False)
        # finish initialising every landmark in every filter, then replay the rest of the per-filter traces
        bt2 = hip.Batch(B, n, Q, R)
        bt2.set_tick_mode(0)
        bt2.set_pairing(pairing)
        bt2.load_trace(np.zeros((1, 2)), bx[None], by[None], wid[None], bcast=True)
        bt2.run(0, 1)
        bt2.load_trace(tw, mx, my, ids)
        bt2.profile(True)
        bt2.run(0, T)
        bt2.sync()
        n2, n1 = bt2.profile_read(hip.K_UPDATE2)[1], bt2.profile_read(hip.K_UPDATE)[1]
        assert (n2, n1) == ((T * m // 2, 0) if pairing else (0, T * m))
        bt2.profile(False)
        res.append([(bt2.state(b), bt2.cov(b), bt2.seen(b)) for b in range(B)] +
                   [(bt.state(b), bt.cov(b), bt.seen(b)) for b in range(B)])
        assert bt2.status()[1] == 0
    for (s1, c1, k1), (s2, c2, k2) in zip(*res):
        assert np.array_equal(s1, s2) and np.array_equal(c1, c2) and k1 == k2
    # and the filters really are different trials
    assert not np.array_equal(res[0][0][0], res[0][1][0])


@pytest.mark.parametrize("dtype", [0, 1])
def test_four_wave_groups_when_the_grid_exceeds_one_generation(hip, dtype):
    """More workgroups than CUs (300 filters): the host picks the 4-wave variants of both sweep kernels.  Paired and
    unpaired runs must still agree bit for bit, and filter b must equal the single-filter path (8-wave variant)."""
    B, n, m, T = 300, 10, 6, 3
    lm = synth.make_landmarks(n)
    tr = synth.make_trace(n, T, m, landmarks=lm)
    bx, by, wid = synth.warmup_observations(lm)
    res = []
    for pairing in (True, False):
        bt = hip.Batch(B, n, Q, R, dtype=dtype)
        bt.set_tick_mode(0)
        bt.set_pairing(pairing)
        bt.load_trace(np.zeros((1, 2)), bx[None], by[None], wid[None], bcast=True)
        bt.run(0, 1)
        bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, tr.ids, bcast=True)
        bt.run(0, T)
        res.append([(bt.state(b), bt.cov(b)) for b in (0, 149, 299)])
        assert bt.status()[1] == 0
    for (s1, c1), (s2, c2) in zip(*res):
        assert np.array_equal(s1, s2) and np.array_equal(c1, c2)
    g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R, dtype=dtype)
    g.as_batch().set_pass_variant(hip.PASS_EXACT)      # (its ticks run as the tick pipeline: bit equality is the exact chain's)
    g.tick(np.zeros(3), bx, by, known_ids=wid, want_ids=False)
    for t in range(T):
        g.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t], want_ids=False)
    assert np.array_equal(g.state, res[0][1][0]) and np.array_equal(g.cov, res[0][1][1])
