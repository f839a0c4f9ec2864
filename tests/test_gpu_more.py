"""GPU parity, part 2: dense MFMA predict, fp32 storage, per-filter batch traces, the C++ host mirror."""
import os
import subprocess

import numpy as np
import pytest

import _oracle as O
from nuslam_hip import synth

pytestmark = pytest.mark.gpu
Q, R = synth.Q_DEFAULT, synth.R_DEFAULT
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def warm_oracle(n, mode=O.ORC_STRUCTURED, seed=12345):
    lm = synth.make_landmarks(n, seed)
    o = O.OracleEKF(np.zeros(3), np.zeros(2 * n), Q, R, mode)
    bx, by, ids = synth.warmup_observations(lm, seed=seed)
    o.tick(tw=np.zeros(3), mx=bx, my=by, known_ids=ids)
    return o, lm


# ------------------------------------------------------------------ dense predict (MFMA)
@pytest.mark.parametrize("n", [3, 30, 100])
def test_dense_predict_identity_and_permutation_exact(hip, n):
    """F = I leaves P + Qbar exactly; a permutation F only moves entries: both are exact in any summation order,
    and P is not symmetric to the last bit, so a transposed operand or a swapped C/D lane map cannot hide."""
    o, _ = warm_oracle(n)
    L = o.len
    rng = np.random.default_rng(5)
    P0 = o.cov.copy() + 1e-3 * rng.normal(size=(L, L))      # clearly asymmetric
    g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    g.restore(o.state, P0, n)
    g.predict_dense(np.eye(L))
    Qb = np.zeros((L, L)); Qb[:3, :3] = Q
    assert np.array_equal(g.cov, P0 + Qb)
    perm = rng.permutation(L)
    F = np.zeros((L, L)); F[np.arange(L), perm] = 1.0
    g.restore(o.state, P0, n)
    g.predict_dense(F)
    assert np.array_equal(g.cov, P0[np.ix_(perm, perm)] + Qb)
    assert np.array_equal(g.state, o.state)                  # the state is not touched


@pytest.mark.parametrize("n", [10, 100, 333])
def test_dense_predict_matches_oracle_fp64(hip, n):
    o, _ = warm_oracle(n, O.ORC_DENSE)
    L = o.len
    rng = np.random.default_rng(11)
    F = np.eye(L) + 0.05 * rng.normal(size=(L, L)) / np.sqrt(L)
    g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    g.restore(o.state, o.cov, n)
    O.set_threads(O.usable_cpus())
    o.predict_dense(F)
    O.set_threads(1)
    g.predict_dense(F)
    err = np.abs(g.cov - o.cov).max() / np.abs(o.cov).max()
    print("dense predict fp64 n=%d: max|dP|/max|P| = %.2e" % (n, err))
    assert np.array_equal(g.cov, o.cov)      # v_mfma_f64_16x16x4_f64 accumulates in ascending k like the oracle's FMA chain: same bits


def test_dense_predict_fp32(hip):
    """fp32 storage + exact-f32 MFMA (v_mfma_f32_32x32x2_f32): tolerance stated against the fp64 oracle."""
    n = 100
    o, _ = warm_oracle(n, O.ORC_DENSE)
    L = o.len
    rng = np.random.default_rng(12)
    F = np.eye(L) + 0.05 * rng.normal(size=(L, L)) / np.sqrt(L)
    g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R, dtype=hip.F32)
    g.restore(o.state, o.cov, n)
    o.predict_dense(F)
    g.predict_dense(F)
    err = np.abs(g.cov - o.cov).max() / np.abs(o.cov).max()
    print("dense predict fp32: max|dP|/max|P| = %.2e" % err)
    assert err < 5e-6


# ------------------------------------------------------------------ fp32 covariance storage
def test_fp32_storage_trajectory(hip):
    """BASELINE config 3 stores P in fp32 (state and all O(len) arithmetic stay fp64).  Against the fp64 oracle
    from a warm snapshot: state 1e-3, covariance 1e-3 of max|P| (1e-6 is not reachable in fp32)."""
    n, T, m = 40, 10, 8
    o, lm = warm_oracle(n)
    tr = synth.make_trace(n, T, m, landmarks=lm)
    g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R, dtype=hip.F32)
    g.restore(o.state, o.cov, n)
    for t in range(T):
        o.tick(tw=tr.tw[t], mx=tr.mx[t], my=tr.my[t], known_ids=tr.ids[t])
        g.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t], want_ids=False)
    es = np.abs(g.state - o.state).max()
    ep = np.abs(g.cov - o.cov).max() / np.abs(o.cov).max()
    print("fp32 storage: |dstate| %.2e, |dP|/max|P| %.2e" % (es, ep))
    assert es < 1e-3 and ep < 1e-3


def test_fp32_cold_start_initialises(hip):
    """INT_MAX on the diagonal rounds to 2^31 in fp32; the first update must still produce a finite, small variance."""
    n = 6
    tr = synth.make_trace(n, 2, n)
    g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R, dtype=hip.F32)
    o = O.OracleEKF(np.zeros(3), np.zeros(2 * n), Q, R)
    for t in range(2):
        g.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t], want_ids=False)
        o.tick(tw=tr.tw[t], mx=tr.mx[t], my=tr.my[t], known_ids=tr.ids[t])
    assert np.isfinite(g.cov).all() and np.diag(g.cov).max() < 1.0
    assert np.abs(g.state - o.state).max() < 1e-2


# ------------------------------------------------------------------ batch with per-filter traces
def test_batch_per_filter_traces(hip):
    """Monte-Carlo trials: every filter gets its own trace; filter b must equal a single filter fed trace b,
    bit for bit, and the batch statistics must be the filter-ordered sums."""
    n, m, T, B = 12, 5, 5, 4
    traces = [synth.make_trace(n, T, m, seed=100 + b) for b in range(B)]
    bt = hip.Batch(B, n, Q, R)
    tw = np.stack([t.tw[:, :2] for t in traces]); mx = np.stack([t.mx for t in traces])
    my = np.stack([t.my for t in traces]); ids = np.stack([t.ids for t in traces])
    bt.load_trace(tw, mx, my, ids)
    bt.run(0, T)
    states = []
    for b in range(B):
        g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
        for t in range(T):
            g.tick(traces[b].tw[t], traces[b].mx[t], traces[b].my[t], known_ids=traces[b].ids[t], want_ids=False)
        assert np.array_equal(bt.state(b), g.state) and np.array_equal(bt.cov(b), g.cov) and bt.seen(b) == g.seen
        states.append(g.state)
    st = bt.stats()
    acc = np.zeros(g.len); acc2 = np.zeros(g.len)
    for s in states:
        acc = acc + s
        acc2 = acc2 + s * s
    assert np.array_equal(st[:g.len], acc) and np.array_equal(st[g.len:2 * g.len], acc2)
    assert bt.status() == (-1, 0)


def test_batch_data_association(hip):
    n, m, T, B = 8, 3, 12, 3
    traces = [synth.make_trace(n, T, m, seed=200 + b, noise_sigma=1e-3) for b in range(B)]
    bt = hip.Batch(B, n, Q, R)
    bt.load_trace(np.stack([t.tw[:, :2] for t in traces]), np.stack([t.mx for t in traces]),
                  np.stack([t.my for t in traces]), None)
    bt.run(0, T)
    for b in range(B):
        o = O.OracleEKF(np.zeros(3), np.zeros(2 * n), Q, R, O.ORC_STRUCTURED)
        for t in range(T):
            o.tick(tw=traces[b].tw[t], mx=traces[b].mx[t], my=traces[b].my[t])
        assert bt.seen(b) == o.seen
        assert np.abs(bt.state(b) - o.state).max() < 1e-4


# ------------------------------------------------------------------ the C++ host mirror
def test_cpp_host_mirror_replays_slam_loop(hip):
    """cpp/tests/replay_slam_loop drives slam_library::ExtendedKalman (C++ class over the C ABI) and
    rigid2d::DiffDrive (host C++) exactly like nuslam/src/slam.cpp:231-319, with unknown data association."""
    exe = os.path.join(ROOT, "shermbot-navigation_amd", "cpp", "tests", "replay_slam_loop")
    assert os.path.exists(exe), "build it first: make -C shermbot-navigation_amd/cpp"
    g = np.load(os.path.join(ROOT, "tests", "golden", "ekf_oracle.npz"))
    lm = np.array([[0.5, 0.5], [-0.5, -0.5], [1.0, 1.0], [-1.0, -1.0], [-0.75, 0.75], [0.75, -0.75]])
    n, T, m = 8, 25, 3
    tr = synth.make_trace(n, T, m, landmarks=lm, noise_sigma=2e-3)
    assert np.array_equal(tr.mx, g["da_mx"])               # the same trace the DA fixture was made from
    lines = ["%d %d %d %d %.17g %.17g" % (n, n, T, m, synth.WHEEL_BASE, synth.WHEEL_RADIUS)]
    for t in range(T):
        lines.append("%.17g %.17g " % (tr.thL[t], tr.thR[t]) + " ".join("%.17g %.17g" % (tr.mx[t, i], tr.my[t, i]) for i in range(m)))
    out = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    # oracle, driven from the same wheel angles
    o = O.OracleEKF(np.zeros(3), np.zeros(2 * n), Q, R, O.ORC_STRUCTURED)
    dd = O.dd_new(synth.WHEEL_BASE, synth.WHEEL_RADIUS)
    rows = [l.split() for l in out.stdout.splitlines()]
    trows = [r for r in rows if r[0] == "T"]
    mrows = [r for r in rows if r[0] == "M"]
    for t in range(T):
        ids = o.tick(dd=dd, thL=tr.thL[t], thR=tr.thR[t], mx=tr.mx[t], my=tr.my[t])
        # map -> odom (slam.cpp:175-210): the host DiffDrive is bit-equal to the oracle's, so from the printed pose
        # estimate the oracle's algebra must give the printed transform
        want = O.map_to_odom(dd[[2, 3, 4]], [float(x) for x in trows[t][2:5]])
        assert np.allclose([float(x) for x in mrows[t][1:4]], want, atol=1e-14, rtol=0)
        assert np.array_equal(ids, g["da_ids"][t])
        assert int(trows[t][1]) == o.seen
        assert np.allclose([float(x) for x in trows[t][2:5]], o.state[:3], atol=1e-4, rtol=0)
    S = np.array([float(x) for x in next(r for r in rows if r[0] == "S")[1:]])
    P = np.array([float(x) for x in next(r for r in rows if r[0] == "P")[1:]]).reshape((o.len, o.len), order="F")
    assert np.allclose(S, o.state, atol=1e-4, rtol=0)
    assert np.linalg.norm(P - o.cov) / np.linalg.norm(o.cov) < 1e-4


# ------------------------------------------------------------------ dense predict inside the tick
def test_tick_with_resident_dense_jacobian(hip):
    """use_dense_predict: predict() advances the state and propagates P with the resident dense Jacobian on the
    matrix cores.  With F = the reference's own A = I + B for that twist it must reproduce the ordinary tick."""
    n, m = 20, 6
    o, lm = warm_oracle(n)
    tr = synth.make_trace(n, 1, m, landmarks=lm)
    s0, P0 = o.state.copy(), o.cov.copy()
    dth, dx = tr.tw[0][0], tr.tw[0][1]
    th1 = s0[0] + dth                                   # getA is evaluated at the advanced heading (:66-67,129)
    r = dx / dth
    F = np.eye(o.len)
    F[1, 0] = -r * np.cos(th1) + r * np.cos(th1 + dth)
    F[2, 0] = -r * np.sin(th1) + r * np.sin(th1 + dth)
    g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    g.restore(s0, P0, n)
    g.predict_dense(F)                                  # stages F (and applies it once)
    g.restore(s0, P0, n)
    g.use_dense_predict(True)
    g.tick(tr.tw[0], tr.mx[0], tr.my[0], known_ids=tr.ids[0], want_ids=False)
    o.tick(tw=tr.tw[0], mx=tr.mx[0], my=tr.my[0], known_ids=tr.ids[0])
    assert np.abs(g.state - o.state).max() < 1e-12
    assert np.abs(g.cov - o.cov).max() / np.abs(o.cov).max() < 1e-12


# ------------------------------------------------------------------ size-independent properties at BASELINE sizes
@pytest.mark.parametrize("n,dtype,tol", [(1000, 0, 1e-5), (5000, 1, 2e-4)])
def test_full_size_properties(hip, n, dtype, tol):
    """BASELINE configs[1] / [2] sizes, where a dense oracle run is out of reach: properties the EKF algebra
    guarantees in exact arithmetic -- symmetry of P, non-negative diagonal, trace(P) never grows under a
    correction, a correction with zero innovation leaves the state alone, restore/get round-trips.
    Symmetry tolerance: the reference never symmetrises P and the INT_MAX cold start leaves ~1e-7 relative
    asymmetry in fp64 (the reference's own P is 5e-7 asymmetric after step 0); that absolute
    residue stays while max|P| shrinks, hence 1e-5."""
    m = 16
    lm = synth.make_landmarks(n)
    tr = synth.make_trace(n, 3, m, landmarks=lm)
    bx, by, ids = synth.warmup_observations(lm)
    g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R, dtype=dtype)
    g.tick(np.zeros(3), bx, by, known_ids=ids, want_ids=False)      # initialise every landmark
    assert g.seen == n and g.status() == 0
    P = g.cov
    scale = np.abs(P).max()
    assert np.isfinite(P).all() and np.diag(P).min() >= 0
    assert np.abs(P - P.T).max() / scale < tol
    r, b = tr.polar()
    tr_prev = np.trace(P)
    for t in range(tr.ticks):
        g.predict(tr.tw[t][0], tr.tw[t][1])
        tr_prev = np.trace(g.cov)
        for i in range(m):
            g.update(r[t, i], b[t, i], int(tr.ids[t, i]))
        tr_now = np.trace(g.cov)
        assert tr_now <= tr_prev * (1 + 1e-6)
        tr_prev = tr_now
    P = g.cov
    assert np.abs(P - P.T).max() / np.abs(P).max() < tol and np.diag(P).min() >= -tol * np.abs(P).max()
    # zero innovation: measuring exactly what the filter expects changes no state entry beyond rounding
    s = g.state
    j = int(tr.ids[0, 0])
    z = hip.measurement(s, j)
    g.update(z[0], z[1], j)
    assert np.abs(g.state - s).max() < 1e-9
    # snapshot / restore round trip (fp32 storage rounds P once)
    s, P, seen = g.state, g.cov, g.seen
    g.restore(s, P, seen)
    assert np.array_equal(g.state, s) and np.array_equal(g.cov, P) and g.seen == seen


def test_batch_equals_single_at_config4_size(hip):
    n, m, T, B = 200, 16, 2, 6
    tr = synth.make_trace(n, T, m)
    bx, by, ids = synth.warmup_observations(tr.landmarks)
    g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    g.tick(np.zeros(3), bx, by, known_ids=ids, want_ids=False)
    for t in range(T):
        g.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t], want_ids=False)
    bt = hip.Batch(B, n, Q, R)
    bt.load_trace(np.zeros((1, 2)), bx[None, :], by[None, :], ids[None, :], bcast=True)
    bt.run(0, 1)
    bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, tr.ids, bcast=True)
    bt.run(0, T)
    for b in (0, B - 1):
        assert np.array_equal(bt.state(b), g.state) and np.array_equal(bt.cov(b), g.cov)


def test_device_normalize_angle_matches_the_reference(hip):
    """rigid2d::normalize_angle on the device is a two-constant range reduction, not atan2(sin, cos) (csrc/ekf_device.h):
    it must give the reference's values -- the reference-generated vectors of tests/golden/rigid2d_ref.npz (the compiled
    rigid2d.cpp) and glibc's atan2(sin, cos) -- to <= 2 ulp, the edges included: +-pi, one ulp either side of them, 3 pi,
    -7.5, multiples of 2 pi, tiny and zero arguments, 25 000 random angles, and beyond 1e6 rad (libm form kept there)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rigid2d_ref.npz"))
    pi = np.pi
    edge = np.array([pi, -pi, np.nextafter(pi, 4.0), np.nextafter(pi, 0.0), np.nextafter(-pi, -4.0), np.nextafter(-pi, 0.0),
                     3 * pi, -3 * pi, -7.5, 7.5, 2 * pi, -2 * pi, 4 * pi, 6.0, -6.0, 100.0, -100.0, 1e-300, 0.0, -0.0, 5 * pi,
                     999999.0, 1.0e6, 1234567.0, -9.87654321e8])
    rng = np.random.default_rng(11)
    rnd = np.concatenate([rng.uniform(-2 * pi, 2 * pi, 20000), rng.uniform(-40, 40, 5000)])

    def ulps(x, ref):
        d = np.abs(x - ref)
        return np.where(d == 0, 0.0, d / np.spacing(np.maximum(np.abs(x), np.abs(ref))))

    got = hip.device_normalize_angle(g["ang"])
    u_gold = ulps(got, g["norm"]).max()
    for a in (edge, rnd):
        ref = np.arctan2(np.sin(a), np.cos(a))
        got = hip.device_normalize_angle(a)
        assert ulps(got, ref).max() <= 2.0, (a[ulps(got, ref).argmax()], got[ulps(got, ref).argmax()], ref[ulps(got, ref).argmax()])
    e = hip.device_normalize_angle(edge[:6])
    assert e[0] == pi and e[1] == -pi                     # +-pi (the fp64 values lie inside (-pi, pi]) map to themselves
    assert abs(e[2] + pi) < 1e-15 and abs(e[4] - pi) < 1e-15      # one ulp beyond: the other end
    print("device normalize_angle vs the compiled reference's vectors: %.1f ulp worst" % u_gold)
    assert u_gold <= 2.0


def test_dense_predict_with_device_jacobian_is_the_reference_predict_fp64(hip):
    """nuslam_ekf_use_dense_predict(h, 2): A = I + B(theta', twist) (slam_library.cpp:127-148) is formed on the device
    every tick and P <- A P A^T + Qbar runs as two dense products on the matrix cores.  fp64: bit-identical to the
    O(len) shortcut k_predict (both are the dense product's k-ordered sums, the extra terms exact zeros), over a
    trajectory with corrections in between, a straight tick (dth == 0) included."""
    n, m, T = 50, 8, 6
    lm = synth.make_landmarks(n)
    tr = synth.make_trace(n, T, m, landmarks=lm, straight_every=3, dL=0.3125, dR=0.375)
    bx, by, wid = synth.warmup_observations(lm)
    fs = []
    for dense in (2, 0):
        g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
        g.as_batch().set_pass_variant(hip.PASS_EXACT)
        g.tick(np.zeros(3), bx, by, known_ids=wid, want_ids=False)
        g.use_dense_predict(dense)
        for t in range(T):
            g.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t], want_ids=False)
        assert g.status() == 0
        fs.append((g.state, g.cov))
    assert np.array_equal(fs[0][0], fs[1][0]) and np.array_equal(fs[0][1], fs[1][1])
