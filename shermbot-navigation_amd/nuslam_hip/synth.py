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


def make_trace(n, ticks, m, seed=12345, dL=0.30, dR=0.36, straight_every=25, noise_sigma=None,
               landmarks=None):
    """Known-association trace: each tick observes the m landmarks nearest to the true pose."""
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
        near = np.argsort(d[:, 0] ** 2 + d[:, 1] ** 2, kind="stable")[:m]
        c, s = np.cos(th), np.sin(th)
        bx = c * d[near, 0] + s * d[near, 1]
        by = -s * d[near, 0] + c * d[near, 1]
        mx[t] = bx + rng.normal(0.0, sigma, size=m)
        my[t] = by + rng.normal(0.0, sigma, size=m)
        ids[t] = near + 1
    return Trace(lm, thL, thR, tw, mx, my, ids, truth)


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
