"""Seeded synthetic odometry + range-bearing traces for the EKF-SLAM hot path.

The world follows the reference's simulator where it specifies one: TurtleBot3 wheel constants
(nuturtle_description/config/diff_params.yaml:2-3), marker positions reported in the robot frame with
additive noise of variance tube_var = 1e-3 (nuturtlesim/src/tube_world.cpp:270-329,
nuturtlesim/config/tube_world_params.yaml:11), Q = diag(0.1), R = diag(1e-3)
(nuslam/config/slam_params.yaml:2-3).  Landmarks are a jittered grid (pitch 0.5 m) around the origin; the
robot drives a gentle left arc with every 25th tick straight so the `dth == 0.0` branch
(nuslam/src/slam_library.cpp:77,135) is exercised.  RNG: numpy PCG64, seed 12345 unless stated.

Pure numpy; no dependency on the oracle or on the HIP library.
"""
import numpy as np

WHEEL_RADIUS = 0.033
WHEEL_BASE = 0.16
Q_DEFAULT = np.diag([0.1, 0.1, 0.1])
R_DEFAULT = np.diag([1e-3, 1e-3])


def make_landmarks(n, seed=12345, pitch=0.5, jitter=0.05):
    rng = np.random.default_rng(seed)
    side = int(np.ceil(np.sqrt(n)))
    g = (np.arange(side) - (side - 1) / 2.0) * pitch
    gx, gy = np.meshgrid(g, g, indexing="xy")
    pts = np.stack([gx.ravel(), gy.ravel()], axis=1)
    # keep the n grid points closest to the origin (stable order: by distance, then index)
    order = np.argsort(np.hypot(pts[:, 0], pts[:, 1]) + 1e-9 * np.arange(pts.shape[0]), kind="stable")
    pts = pts[order[:n]]
    pts = pts + rng.normal(0.0, jitter, size=pts.shape)
    return pts


class Trace:
    """ticks x (twist, wheel angles, m markers in the robot frame, 1-based ids)."""

    def __init__(self, landmarks, thL, thR, tw, mx, my, ids, truth):
        self.landmarks = landmarks
        self.thL, self.thR = thL, thR          # cumulative wheel angles per tick
        self.tw = tw                           # (T, 3) body twists dth, dx, dy
        self.mx, self.my = mx, my              # (T, m)
        self.ids = ids                         # (T, m) int32, 1-based landmark ids
        self.truth = truth                     # (T, 3) th, x, y after the tick

    @property
    def ticks(self):
        return self.tw.shape[0]

    @property
    def m(self):
        return self.mx.shape[1]

    def polar(self):
        """range / bearing exactly as cartesian2polar (slam_library.cpp:16-22) would form them."""
        r = np.sqrt(self.mx * self.mx + self.my * self.my)
        b = np.arctan2(self.my, self.mx)
        b = np.arctan2(np.sin(b), np.cos(b))
        return r, b


MIN_RANGE_DEFAULT = 0.2   # m: nearer landmarks are not reported in the well-posed traces
FOV_DEFAULT = 2.0    # rad: half-angle of the synthetic sensor's field of view in the well-posed traces (see make_trace)


def make_trace(n, ticks, m, seed=12345, dL=0.30, dR=0.36, straight_every=25, noise_sigma=None,
               landmarks=None, fov=None, min_range=0.0):
    """Known-association trace: each tick observes the m landmarks nearest to the true pose.

    fov (rad): only landmarks whose true bearing lies within +-fov are visible (the m nearest of those).  The reference's
    update() subtracts the two bearings without wrapping the difference (slam_library.cpp:272): a landmark behind the
    robot, measured at +pi - e and predicted at -pi + e, yields an innovation of ~2 pi and throws the pose estimate by a
    radian -- on an all-around trace (fov=None) that happens in nearly every tick, the reference filter itself never
    tracks the truth, and the sign of such a wrap is a discontinuity that amplifies a one-ulp difference to O(1) within
    ~50 ticks.  A limited field of view keeps every bearing clear of the +-pi cut and the reference algorithm in its
    working regime; parity at depth (tests/test_gpu_depth.py) and the bench run on such traces.  min_range (with fov):
    landmarks nearer than this are not reported either -- at 5 cm the 3 cm marker noise alone can carry a bearing across
    the cut."""
    rng = np.random.default_rng(seed + 1)
    lm = make_landmarks(n, seed) if landmarks is None else np.asarray(landmarks, dtype=np.float64)
    m = min(m, n)
    sigma = np.sqrt(1e-3) if noise_sigma is None else noise_sigma
    th, x, y = 0.0, 0.0, 0.0
    aL = aR = 0.0
    thL = np.zeros(ticks); thR = np.zeros(ticks)
    tw = np.zeros((ticks, 3)); truth = np.zeros((ticks, 3))
    mx = np.zeros((ticks, m)); my = np.zeros((ticks, m)); ids = np.zeros((ticks, m), dtype=np.int32)
    for t in range(ticks):
        if straight_every and (t + 1) % straight_every == 0:
            uL = uR = 0.5 * (dL + dR)
        else:
            uL, uR = dL, dR
        aL_new, aR_new = aL + uL, aR + uR
        # take the differences the way DiffDrive::getTwist does (diff_drive.cpp:83-89) so dth == 0.0 exactly
        # when the increments are equal
        dth = (WHEEL_RADIUS / WHEEL_BASE) * ((aR_new - aR) - (aL_new - aL))
        dx = (WHEEL_RADIUS / 2) * ((aL_new - aL) + (aR_new - aR))
        aL, aR = aL_new, aR_new
        if dth == 0.0:
            x += dx * np.cos(th); y += dx * np.sin(th)
        else:
            rr = dx / dth
            x += -rr * np.sin(th) + rr * np.sin(th + dth)
            y += rr * np.cos(th) - rr * np.cos(th + dth)
            th += dth
        thL[t], thR[t] = aL, aR
        tw[t] = (dth, dx, 0.0)
        truth[t] = (th, x, y)
        d = lm - np.array([x, y])
        c, s = np.cos(th), np.sin(th)
        d2 = d[:, 0] ** 2 + d[:, 1] ** 2
        if fov is not None:
            bearing = np.arctan2(-s * d[:, 0] + c * d[:, 1], c * d[:, 0] + s * d[:, 1])
            d2 = np.where((np.abs(bearing) <= fov) & (d2 >= min_range * min_range), d2, np.inf)
            assert np.isfinite(d2).sum() >= m, "fewer than m landmarks inside the field of view"
        near = np.argsort(d2, kind="stable")[:m]
        bx = c * d[near, 0] + s * d[near, 1]
        by = -s * d[near, 0] + c * d[near, 1]
        mx[t] = bx + rng.normal(0.0, sigma, size=m)
        my[t] = by + rng.normal(0.0, sigma, size=m)
        ids[t] = near + 1
    return Trace(lm, thL, thR, tw, mx, my, ids, truth)


def make_wellposed_trace(n, ticks, m, **kw):
    """make_trace on which the reference algorithm itself is well-conditioned, so that two correct implementations stay
    together for hundreds of ticks (the oracle against itself with one input moved by one ulp: 1e-9 at tick 200, where
    the plain trace has lost all digits by tick 50 -- tests/test_trace_conditioning.py):
      * a +-FOV_DEFAULT field of view and a minimum range: no bearing near the +-pi cut, whose un-wrapped difference
        (slam_library.cpp:272) is a 2 pi discontinuity;
      * wheel increments that are exact binary fractions (5/16, 3/8, straight ticks 11/32): the accumulated wheel angles
        are exact, so a straight tick has dth == 0.0 exactly and takes the straight branch (slam_library.cpp:77).  With
        0.30 / 0.36 the accumulated angles round, a 'straight' tick comes out with dth = 4.6e-17, takes the arc branch,
        and dx/dth * (sin(th + dth) - sin(th)) is 0 or 2.4 cm depending on the last bit of the heading."""
    kw.setdefault("fov", FOV_DEFAULT)
    kw.setdefault("min_range", MIN_RANGE_DEFAULT)
    kw.setdefault("dL", 0.3125)
    kw.setdefault("dR", 0.375)
    return make_trace(n, ticks, m, **kw)


def warmup_observations(landmarks, pose=(0.0, 0.0, 0.0), seed=12345, noise_sigma=None):
    """One observation of every landmark from `pose` (th, x, y): initialises the whole map (config 2/3 warm-up)."""
    rng = np.random.default_rng(seed + 2)
    sigma = np.sqrt(1e-3) if noise_sigma is None else noise_sigma
    th, x, y = pose
    d = np.asarray(landmarks) - np.array([x, y])
    c, s = np.cos(th), np.sin(th)
    bx = c * d[:, 0] + s * d[:, 1] + rng.normal(0.0, sigma, size=d.shape[0])
    by = -s * d[:, 0] + c * d[:, 1] + rng.normal(0.0, sigma, size=d.shape[0])
    ids = np.arange(1, d.shape[0] + 1, dtype=np.int32)
    return bx, by, ids
