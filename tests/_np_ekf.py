"""A second, independent restatement of the reference EKF in numpy (dense matrices, BLAS products), used only to
cross-check the C oracle: two restatements written separately from the same source lines
(nuslam/src/slam_library.cpp) must agree to rounding.  TEST INFRASTRUCTURE."""
import numpy as np


def wrap(a):                      # rigid2d.cpp:9-13
    return np.arctan2(np.sin(a), np.cos(a))


def c2p(x, y):                    # slam_library.cpp:16-22
    return np.array([np.sqrt(x ** 2 + y ** 2), wrap(np.arctan2(y, x))])


class NpEKF:
    def __init__(self, robot, mp, Q, R):
        self.n = len(mp) // 2
        self.len = 3 + 2 * self.n
        self.seen = 0
        self.s = np.concatenate([robot, mp]).astype(float)
        self.Q = np.array(Q, float)
        self.R = np.array(R, float)
        self.P = np.zeros((self.len, self.len))
        for i in range(3, self.len):
            self.P[i, i] = 2147483647.0

    def predict(self, dth, dx, dy=0.0):
        th = self.s[0]
        if dth == 0.0:
            d = np.array([0.0, dx * np.cos(th), dx * np.sin(th)])
        else:
            r = dx / dth
            d = np.array([dth, -r * np.sin(th) + r * np.sin(th + dth), r * np.cos(th) - r * np.cos(th + dth)])
        self.s[:3] += d
        th = self.s[0]                                   # Jacobian at the advanced heading (:66-67,129)
        A = np.eye(self.len)
        if dth == 0:
            A[1, 0] = -dx * np.sin(th)
            A[2, 0] = dx * np.cos(th)
        else:
            r = dx / dth
            A[1, 0] = -r * np.cos(th) + r * np.cos(th + dth)
            A[2, 0] = -r * np.sin(th) + r * np.sin(th + dth)
        Qb = np.zeros_like(self.P)
        Qb[:3, :3] = self.Q
        self.P = A @ self.P @ A.T + Qb

    def h(self, j, s):
        z = c2p(s[3 + 2 * (j - 1)] - s[1], s[4 + 2 * (j - 1)] - s[2])
        z[1] = wrap(z[1] - s[0])
        return z

    def H(self, j, s):
        H = np.zeros((2, self.len))
        dx = s[3 + 2 * (j - 1)] - s[1]
        dy = s[4 + 2 * (j - 1)] - s[2]
        d = dx ** 2 + dy ** 2
        c = 3 + 2 * (j - 1)
        H[1, 0] = -1
        H[0, 1] = -dx / np.sqrt(d); H[1, 1] = dy / d
        H[0, 2] = -dy / np.sqrt(d); H[1, 2] = -dx / d
        H[0, c] = dx / np.sqrt(d); H[1, c] = -dy / d
        H[0, c + 1] = dy / np.sqrt(d); H[1, c + 1] = dx / d
        return H

    def init_landmark(self, z, j):
        self.s[3 + 2 * (j - 1)] = self.s[1] + z[0] * np.cos(z[1] + self.s[0])
        self.s[4 + 2 * (j - 1)] = self.s[2] + z[0] * np.sin(z[1] + self.s[0])

    def update(self, z, j):
        zh = self.h(j, self.s)
        H = self.H(j, self.s)
        K = self.P @ H.T @ np.linalg.inv(H @ self.P @ H.T + self.R)
        self.s = self.s + K @ (np.asarray(z) - zh)
        self.s[0] = wrap(self.s[0])
        self.P = (np.eye(self.len) - K @ H) @ self.P

    def associate(self, z):
        if self.seen == 0:
            self.seen = 1
            return 1
        if 4 + 2 * self.seen >= self.len:
            raise IndexError("full map")
        for k in range(1, self.seen + 1):
            H = self.H(k, self.s)
            psi = H @ self.P @ H.T + self.R
            dz = np.asarray(z) - self.h(k, self.s)
            d = dz @ np.linalg.inv(psi) @ dz
            if d < 0.01:
                return k
            if 0.01 < d < 60:
                return -1
        self.seen += 1
        return self.seen
