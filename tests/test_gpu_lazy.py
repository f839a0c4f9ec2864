"""Lazy ticks: the class API driven call by call, as nuslam/src/slam.cpp:250-319 drives slam_library::ExtendedKalman
(getSeenLandmarks -> predict -> per marker [initializeLandmark] / skip / stop -> update), against the explicit one-call
tick entry (nuslam_ekf_tick_ex) on the same inputs.  The recorded calls reach the device as the SAME kernels
(k_tick_front + k_tick_rank; the exact chain for rounds with a first sighting), so every comparison here is bitwise:
state, covariance, `seen`, latched status -- with a getter in mid-tick, a copy in mid-tick, first sightings, a gray-zone
skip (id < 0: the caller `continue`s, slam.cpp:298-300) and the `id > total` break (:301-316).
"""
import numpy as np
import pytest

from nuslam_hip import synth

pytestmark = pytest.mark.gpu
Q, R = synth.Q_DEFAULT, synth.R_DEFAULT
EXACT_WHEELS = dict(dL=0.3125, dR=0.375)


def polar(hip, mx, my):
    z = np.array([hip.cartesian2polar(float(x), float(y)) for x, y in zip(mx, my)])
    return z[:, 0].copy(), z[:, 1].copy()


def drive_like_the_node(f, tw, r, phi, ids, total, seen, peek=None):
    """slam.cpp:250-319 with known ids, one C-ABI call per call of the node.  peek(i): called in front of marker i.
    `seen`: the node's seen_landmarks of :251.  With KNOWN ids nothing in the reference ever counts landmarks (only
    associateLandmark does); the explicit known-id tick counts them on the device (csrc/ekf_update.h resolve(): "what
    associateLandmark would have counted"), so the caller passes that handle's count -- as cpp/tests/api_rate.cpp, whose
    map-initialising call is an explicit tick, reads it back from the filter.  Returns the number of update() calls."""
    updates = 0
    f.predict(tw[0], tw[1])                                             # :269
    for i in range(len(ids)):
        if peek is not None:
            peek(i)
        k = int(ids[i])
        if k > seen:
            f.init_landmark(r[i], phi[i], k)                            # :295-297
        elif k < 0:
            continue                                                    # :298-300
        elif k > total:
            break                                                       # :301-316
        f.update(r[i], phi[i], k)                                       # :318
        updates += 1
    return updates


def same(a, b):
    sa, sb = a.state, b.state
    Pa, Pb = a.cov, b.cov
    assert np.array_equal(sa, sb), "state differs by %.3e" % np.abs(sa - sb).max()
    assert np.array_equal(Pa, Pb), "covariance differs by %.3e" % np.abs(Pa - Pb).max()
    assert a.status() == b.status()


@pytest.mark.parametrize("n,m,T", [(40, 16, 8), (60, 37, 4), (1000, 16, 6)])
def test_lazy_calls_equal_the_explicit_tick_bit_for_bit(hip, n, m, T):
    """Cold start (tick 0: every landmark a first sighting), then warm ticks; with 37 markers a tick is three rounds."""
    tr = synth.make_trace(n, T, m, straight_every=3, **EXACT_WHEELS)
    a = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)                     # lazy (the default): call by call
    b = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)                     # one nuslam_ekf_tick_ex per tick
    for t in range(T):
        r, phi = polar(hip, tr.mx[t], tr.my[t])
        drive_like_the_node(a, tr.tw[t], r, phi, tr.ids[t], n, b.seen)
        b.tick_ex(tr.tw[t], r, phi, polar=True, known_ids=tr.ids[t], total_landmarks=n, want_ids=False)
        if t in (0, 1, T - 1):
            same(a, b)
    same(a, b)


def test_getter_and_copy_in_mid_tick(hip):
    """A getter between two update() calls flushes what was recorded so far (predict + the first corrections); the rest of the
    tick follows without a second predict.  The explicit counterpart is tick_ex on the first markers, then tick_ex without a
    twist on the others.  A copy taken in mid-tick carries exactly the flushed part."""
    n, m, T = 100, 16, 5
    tr = synth.make_wellposed_trace(n, T, m, straight_every=4)
    a = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    b = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    # initialise the map through one explicit tick on both (every landmark once)
    lm = tr.landmarks
    ids0 = np.arange(1, n + 1, dtype=np.int32)
    for f in (a, b):
        f.tick((0.0, 0.0), lm[:, 0], lm[:, 1], known_ids=ids0, want_ids=False)
    same(a, b)
    cut = 7
    copies = {}
    for t in range(T):
        r, phi = polar(hip, tr.mx[t], tr.my[t])

        def peek(i):
            if i == cut:
                copies["state"] = a.state                               # a getter in mid-tick
                if t == 2:
                    copies["clone"] = a.clone()                         # ... and a copy
        drive_like_the_node(a, tr.tw[t], r, phi, tr.ids[t], n, b.seen, peek=peek)
        b.tick_ex(tr.tw[t], r[:cut], phi[:cut], polar=True, known_ids=tr.ids[t][:cut], total_landmarks=n, want_ids=False)
        if t == 2:
            c = copies["clone"]
            same(c, b)                                                  # the copy holds predict + the first `cut` corrections
            c.close()
        assert np.array_equal(copies["state"], b.state)
        b.tick_ex(None, r[cut:], phi[cut:], polar=True, known_ids=tr.ids[t][cut:], total_landmarks=n, want_ids=False)
        same(a, b)


def test_skip_break_and_first_sighting_inside_a_warm_tick(hip):
    n, m = 40, 16
    tr = synth.make_trace(n, 6, m, straight_every=3, **EXACT_WHEELS)
    a = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    b = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    broke = False
    for t in range(6):
        ids = tr.ids[t].copy()
        total = n
        if t == 3:
            ids[4] = -1                                                 # gray zone: skipped by the caller
        if t == 4:
            # capacity below some ids that were seen before: the first such marker stops the loop (slam.cpp:301-316; a NEW
            # landmark above capacity is initialised and corrected all the same, :295 comes first)
            total = int(np.sort(ids)[m // 2])
        r, phi = polar(hip, tr.mx[t], tr.my[t])
        ups = drive_like_the_node(a, tr.tw[t], r, phi, ids, total, b.seen)
        broke = broke or (t == 4 and ups < m)
        b.tick_ex(tr.tw[t], r, phi, polar=True, known_ids=ids, total_landmarks=total, want_ids=False)
        sa, sb = a.state, b.state
        assert np.array_equal(sa, sb) and np.array_equal(a.cov, b.cov), "tick %d" % t
    assert broke, "the trace never took the break"
    # `seen`: the class API leaves it to associateLandmark, as the reference does (slam_library.cpp:188-253)
    assert a.seen == 0 and b.seen > 0


def test_lazy_matches_per_call_kernels_within_rounding_and_few_calls_bitwise(hip):
    """With lazy off every call launches its own kernels (k_predict, k_update: the oracle's arithmetic bit for bit); lazy on
    re-associates a tick's corrections as one rank-2m update: rounding-level agreement.  Fewer than four corrections take the
    per-call kernels either way: bitwise."""
    n, m, T = 100, 16, 6
    tr = synth.make_wellposed_trace(n, T, m, straight_every=4)
    a = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    b = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    b.set_lazy(False)
    lm = tr.landmarks
    ids0 = np.arange(1, n + 1, dtype=np.int32)
    for f in (a, b):
        f.tick((0.0, 0.0), lm[:, 0], lm[:, 1], known_ids=ids0, want_ids=False)
    for t in range(T):
        r, phi = polar(hip, tr.mx[t], tr.my[t])
        k = 3 if t < 2 else m
        for f in (a, b):
            drive_like_the_node(f, tr.tw[t], r[:k], phi[:k], tr.ids[t][:k], n, n)
        if t < 2:
            assert np.array_equal(a.state, b.state) and np.array_equal(a.cov, b.cov)
    Pa, Pb = a.cov, b.cov
    assert np.abs(a.state - b.state).max() < 1e-10
    assert np.abs(Pa - Pb).max() / np.abs(Pb).max() < 1e-12


def test_bad_id_is_reported_by_the_call_itself(hip):
    f = hip.EKF(np.zeros(3), np.zeros(10), Q, R)
    f.predict(0.01, 0.02)
    with pytest.raises(hip.NuslamError) as e:
        f.update(1.0, 0.1, 6)
    assert e.value.code == hip.E_BOUNDS
    with pytest.raises(hip.NuslamError) as e:
        f.init_landmark(1.0, 0.1, 0)
    assert e.value.code == hip.E_BOUNDS
    f.sync()
    assert f.status() == 0


# ------------------------------------------------------------------------------------------------ unknown association
def drive_with_association(f, tw, r, phi, total):
    """slam.cpp:250-319 as written: associateLandmark in front of every marker; returns the ids it answered."""
    seen = f.seen                                                       # :251
    f.predict(tw[0], tw[1])                                             # :269
    ids = []
    for i in range(len(r)):
        k = f.associate(r[i], phi[i])                                   # :291
        ids.append(k)
        if k > seen:
            f.init_landmark(r[i], phi[i], k)                            # :295-297
        elif k < 0:
            continue                                                    # :298-300
        elif k > total:
            break                                                       # :301-316
        f.update(r[i], phi[i], k)                                       # :318
    return ids


def served_pair(hip, n, Qm, exact):
    a = hip.EKF(np.zeros(3), np.zeros(2 * n), Qm, R)                    # lazy: the tick is a round served to the host
    b = hip.EKF(np.zeros(3), np.zeros(2 * n), Qm, R)
    b.set_lazy(False)                                                   # k_associate + k_update per call
    if exact:
        a.as_batch().set_pass_variant(hip.PASS_EXACT)
    return a, b


@pytest.mark.parametrize("n,m,sigma,exact,park_us", [(12, 5, 1e-3, True, 0), (40, 16, 1e-3, True, 0), (70, 37, 1e-3, True, 0),
                                                     (35, 16, None, True, 0), (40, 16, 1e-3, False, 0), (40, 16, 1e-3, True, 5)])
def test_served_round_equals_per_call_kernels_cold_start(hip, n, m, sigma, exact, park_us):
    """The class API with associateLandmark in the loop: every verdict (match, new landmark, gray zone), `seen`, and -- with the
    exact chain as the pass -- every bit of state and covariance are the per-call kernels' (k_associate + k_update).  park_us: the
    served round closes itself after that many microseconds without a call -- from Python that is every call -- and is re-opened:
    same results."""
    T = 8
    tr = synth.make_trace(n, T, m, straight_every=3, noise_sigma=sigma)
    a, b = served_pair(hip, n, Q, exact)
    if park_us:
        a.set_lazy(park_us)
    nv = 0
    for t in range(T):
        r, phi = polar(hip, tr.mx[t], tr.my[t])
        ia = drive_with_association(a, tr.tw[t], r, phi, n)
        ib = drive_with_association(b, tr.tw[t], r, phi, n)
        assert ia == ib, (t, ia, ib)
        nv += len(ia)
        assert a.seen == b.seen
        if exact:
            assert np.array_equal(a.state, b.state) and np.array_equal(a.cov, b.cov), "tick %d" % t
        else:
            Pa, Pb = a.cov, b.cov
            fin = np.abs(Pb) < 1e9
            assert np.array_equal(Pa[~fin], Pb[~fin])
    assert nv > 0 and a.status() == b.status()


def test_served_round_warm_map_gray_zone_new_landmark_and_full_map(hip):
    n, n_world, m, T = 30, 24, 8, 6
    Qs = np.diag([1e-4, 1e-4, 1e-4])
    lm = synth.make_landmarks(n_world)
    tr = synth.make_trace(n_world, T, m, landmarks=lm, noise_sigma=1e-4)
    bx, by, wid = synth.warmup_observations(lm, noise_sigma=1e-4)
    a, b = served_pair(hip, n, Qs, True)
    for f in (a, b):
        f.tick(np.zeros(3), bx, by, known_ids=wid)
    kinds = set()
    for t in range(T):
        mx, my = tr.mx[t].copy(), tr.my[t].copy()
        if t == 2:
            mx[3] += 0.08                                               # off a landmark by 8 cm: gray zone
        if t == 3:
            mx[5], my[5] = 7.0, -6.5                                    # far from everything: a new landmark
        r, phi = polar(hip, mx, my)
        ia = drive_with_association(a, tr.tw[t], r, phi, n)
        ib = drive_with_association(b, tr.tw[t], r, phi, n)
        assert ia == ib, (t, ia, ib)
        kinds.update("gray" if k < 0 else ("new" if k > n_world else "match") for k in ia)
        assert a.seen == b.seen
        assert np.array_equal(a.state, b.state) and np.array_equal(a.cov, b.cov), "tick %d" % t
    assert kinds == {"gray", "new", "match"}, kinds
    # two update() calls after one associateLandmark, an update() with another z than the marker associated, a getter in mid-tick
    r, phi = polar(hip, tr.mx[T - 1], tr.my[T - 1])
    got = []
    for f in (a, b):
        f.predict(0.0, 0.0)
        k0 = f.associate(r[0], phi[0])
        f.update(r[0], phi[0], int(tr.ids[T - 1][0]))
        f.update(r[1], phi[1], int(tr.ids[T - 1][1]))
        k2 = f.associate(r[2], phi[2])
        f.update(r[2] + 1e-3, phi[2], int(tr.ids[T - 1][2]))
        _ = f.state
        k3 = f.associate(r[3], phi[3])
        f.update(r[3], phi[3], int(tr.ids[T - 1][3]))
        got.append((k0, k2, k3))
    assert got[0] == got[1]
    assert np.array_equal(a.state, b.state) and np.array_equal(a.cov, b.cov)


def test_served_round_full_map_is_reported_from_inside_the_call(hip):
    n = 6
    a = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    lm = synth.make_landmarks(n)
    bx, by, wid = synth.warmup_observations(lm)
    a.tick(np.zeros(3), bx, by, known_ids=wid)                          # seen == n: the map is full
    a.predict(0.0, 0.01)
    with pytest.raises(hip.NuslamError) as e:
        a.associate(3.0, 0.5)                                           # slam_library.cpp:206-207 indexes out of bounds
    assert e.value.code == hip.E_BOUNDS
    a.sync()
    assert a.status() == 0 and a.seen == n


def test_state_getter_from_the_host_mirror_equals_the_device_copy(hip):
    """getStateVector() at the top of the caller's loop (slam.cpp:184,250): the launch that forms a tick's state also writes it
    into mapped host memory, workgroup by workgroup, and the getter waits for their stamps instead of synchronising the stream and
    copying.  A clone copies the DEVICE buffers and reads them back the plain way: both must hold the same bits, tick after tick,
    with known ids (the fused launch) and with associateLandmark in the loop (the served round)."""
    n, m, T = 200, 16, 6
    lm = synth.make_landmarks(n - 1)
    tr = synth.make_wellposed_trace(n - 1, T, m, landmarks=lm, noise_sigma=1e-4)
    bx, by, wid = synth.warmup_observations(lm, noise_sigma=1e-4)
    Qs = np.diag([1e-4, 1e-4, 1e-4])
    for known in (True, False):
        f = hip.EKF(np.zeros(3), np.zeros(2 * n), Qs, R)
        f.tick(np.zeros(3), bx, by, known_ids=wid, want_ids=False)
        for t in range(T):
            r, phi = polar(hip, tr.mx[t], tr.my[t])
            if known:
                drive_like_the_node(f, tr.tw[t], r, phi, tr.ids[t], n, n - 1)
            else:
                drive_with_association(f, tr.tw[t], r, phi, n)
            s = f.state                                                 # through the mirror
            c = f.clone()
            assert np.array_equal(s, c.state), "tick %d (%s ids)" % (t, "known" if known else "unknown")
            c.close()
        assert f.status() == 0
        f.close()


def test_a_tick_without_association_after_a_tick_with_it(hip):
    """A caller whose last tick associated gets its next served round opened by predict() already.  If that tick then brings known
    ids instead (update() without associateLandmark), the round -- which has not scanned anything -- is ended and the updates are
    recorded the lazy way: nothing may be lost.  Against the per-call kernels (lazy off) with the exact pass: bit for bit."""
    n, n_world, m, T = 30, 24, 8, 6
    Qs = np.diag([1e-4, 1e-4, 1e-4])
    lm = synth.make_landmarks(n_world)
    tr = synth.make_trace(n_world, T, m, landmarks=lm, noise_sigma=1e-4)
    bx, by, wid = synth.warmup_observations(lm, noise_sigma=1e-4)
    a, b = served_pair(hip, n, Qs, True)
    for f in (a, b):
        f.tick(np.zeros(3), bx, by, known_ids=wid)
    for t in range(T):
        r, phi = polar(hip, tr.mx[t], tr.my[t])
        for f in (a, b):
            if t % 2 == 0:
                drive_with_association(f, tr.tw[t], r, phi, n)
            else:                                                       # known ids, fewer than four and more than four recorded calls
                k = 3 if t == 1 else m
                drive_like_the_node(f, tr.tw[t], r[:k], phi[:k], tr.ids[t][:k], n, n_world)
        assert np.array_equal(a.state, b.state) and np.array_equal(a.cov, b.cov), "tick %d" % t
    assert a.seen == b.seen and a.status() == b.status() == 0
