"""ctypes doorway onto oracle/libnuslam_oracle.so (the CPU checker) and, where it exists,
oracle/_ref/librigid2d_ref.so (the reference's own rigid2d + DiffDrive, compiled from /root/reference).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libnuslam_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "librigid2d_ref.so")

ORC_DENSE, ORC_STRUCTURED = 0, 1
ORC_OK, ORC_E_ARG, ORC_E_BOUNDS, ORC_E_SINGULAR = 0, 1, 2, 3

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _p(a):
    return a.ctypes.data_as(_dp)


def build_oracle(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present). Building the checker is not using it."""
    if force or not os.path.exists(ORACLE_SO) or (
            os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(os.path.join(ORACLE_DIR, f))
                                              for f in ("nuslam_oracle.c", "nuslam_oracle.h", "circle_fit_oracle.c",
                                                        "sim_oracle.c"))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "all"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/rigid2d/src") and (
            not os.path.exists(REF_SO)
            or os.path.getmtime(REF_SO) < os.path.getmtime(os.path.join(ORACLE_DIR, "ref_shim.cpp"))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)


class SimParams(C.Structure):
    """orc_sim_params (oracle/nuslam_oracle.h); include/nuslam_hip.h's nuslam_sim_params has the same layout.
    Defaults: nuturtlesim/config/tube_world_params.yaml + nuturtle_description/config/diff_params.yaml."""
    _fields_ = [(k, C.c_double) for k in ("wheel_base", "wheel_radius", "dt", "twist_noise", "slip_min", "slip_max",
                                          "tube_radius", "robot_radius", "tube_var", "marker_sigma", "max_range",
                                          "lidar", "lidar_min_range", "lidar_max_range", "fov", "min_range")]

    def __init__(self, **kw):
        d = dict(wheel_base=0.16, wheel_radius=0.033, dt=1.0 / 50, twist_noise=0.0, slip_min=0.9, slip_max=1.0,
                 tube_radius=0.0381, robot_radius=0.08, tube_var=0.001, marker_sigma=0.0, max_range=1.0,
                 lidar=0.0, lidar_min_range=0.05, lidar_max_range=1.0, fov=0.0, min_range=0.0)
        d.update(kw)
        super().__init__(**d)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(ORACLE_SO)
        L.orc_normalize_angle.restype = C.c_double
        L.orc_normalize_angle.argtypes = [C.c_double]
        L.orc_transform_twist.argtypes = [_dp, _dp, _dp]
        L.orc_integrate_twist.argtypes = [_dp, _dp]
        L.orc_dd_convert_twist.argtypes = [_dp, _dp, _dp]
        L.orc_dd_get_twist.argtypes = [_dp, C.c_double, C.c_double, _dp]
        L.orc_dd_step.argtypes = [_dp, C.c_double, C.c_double]
        L.orc_cartesian2polar.argtypes = [C.c_double, C.c_double, _dp]
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [_dp, _dp, C.c_int, _dp, _dp]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_mode.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_get_threads.restype = C.c_int
        L.orc_predict.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.orc_predict_dense.argtypes = [C.c_void_p, _dp]
        L.orc_measurement.argtypes = [_dp, C.c_int, _dp]
        L.orc_jacobian.argtypes = [_dp, C.c_int, C.c_int, _dp]
        L.orc_associate.argtypes = [C.c_void_p, C.c_double, C.c_double, _ip, _dp]
        L.orc_init_landmark.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int]
        L.orc_update.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int]
        L.orc_tick.argtypes = [C.c_void_p, _dp, C.c_double, C.c_double, _dp, C.c_int, _dp, _dp, _ip, C.c_int, _ip]
        for f in (L.orc_len, L.orc_n, L.orc_seen):
            f.argtypes = [C.c_void_p]
            f.restype = C.c_int
        L.orc_set_seen.argtypes = [C.c_void_p, C.c_int]
        L.orc_state.argtypes = [C.c_void_p]
        L.orc_state.restype = _dp
        L.orc_cov.argtypes = [C.c_void_p]
        L.orc_cov.restype = _dp
        L.orc_circle_fit.argtypes = [_dp, _dp, C.c_int, _dp, _dp]
        L.orc_classify_cluster.argtypes = [_dp, _dp, C.c_int, _dp]
        L.orc_cluster_points.argtypes = [C.POINTER(C.c_float), C.c_double, C.c_double, _dp, _dp, _ip, _ip]
        L.orc_tf_make.argtypes = [C.c_double, C.c_double, C.c_double, _dp]
        L.orc_tf_inv.argtypes = [_dp, _dp]
        L.orc_tf_mul.argtypes = [_dp, _dp, _dp]
        L.orc_tf_point.argtypes = [_dp, C.c_double, C.c_double, _dp]
        L.orc_map_to_odom.argtypes = [_dp, _dp, _dp]
        _up = C.POINTER(C.c_uint)
        L.orc_philox4x32_10.argtypes = [_up, _up, _up]
        L.orc_sim_normal_pair.argtypes = [C.c_ulonglong, C.c_uint, C.c_uint, C.c_uint, C.c_uint, _dp]
        L.orc_sim_scan.argtypes = [_dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                   C.POINTER(C.c_float)]
        L.orc_simulate.restype = C.c_longlong
        L.orc_simulate.argtypes = [C.POINTER(SimParams), _dp, C.c_int, _dp, C.c_int, C.c_int, C.c_ulonglong,
                                   C.c_uint, _dp, _dp, _dp, _ip, _dp, _dp]
        _lib = L
    return _lib


def circle_fit(xs, ys):
    """circle_fit::circleFit (circle_fit_library.cpp:15-134): (status, centre_x, centre_y, radius)."""
    xs = np.ascontiguousarray(xs, dtype=np.float64)
    ys = np.ascontiguousarray(ys, dtype=np.float64)
    work = np.zeros(4 * max(xs.size, 1))
    out = np.zeros(3)
    st = lib().orc_circle_fit(_p(xs), _p(ys), xs.size, _p(work), _p(out))
    return st, out[0], out[1], out[2]


def classify_cluster(xs, ys):
    xs = np.ascontiguousarray(xs, dtype=np.float64)
    ys = np.ascontiguousarray(ys, dtype=np.float64)
    sd = C.c_double()
    ok = lib().orc_classify_cluster(_p(xs), _p(ys), xs.size, C.byref(sd))
    return bool(ok), sd.value


def cluster_points(ranges, min_range, max_range):
    """circle_fit::clusterPoints (circle_fit_library.cpp:136-206) on a 360-ray scan: list of (xs, ys) clusters."""
    r = np.ascontiguousarray(ranges, dtype=np.float32)
    assert r.size == 360
    px = np.zeros(400); py = np.zeros(400); cl = np.zeros(400, dtype=np.int32)
    ncl = C.c_int()
    npts = lib().orc_cluster_points(r.ctypes.data_as(C.POINTER(C.c_float)), min_range, max_range, _p(px), _p(py),
                                    cl.ctypes.data_as(_ip), C.byref(ncl))
    return [(px[:npts][cl[:npts] == k].copy(), py[:npts][cl[:npts] == k].copy()) for k in range(ncl.value)]


def philox(ctr, key):
    c = (C.c_uint * 4)(*[int(x) for x in ctr]); k = (C.c_uint * 2)(*[int(x) for x in key]); o = (C.c_uint * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def sim_normal_pair(seed, filt, tick, stream, idx):
    z = np.zeros(2)
    lib().orc_sim_normal_pair(int(seed), int(filt), int(tick), int(stream), int(idx), _p(z))
    return z


def simulate(params, landmarks, cmd, m, seed, filt=0):
    """orc_simulate: dict(tw (T,2), mx, my (T,m), ids (T,m), truth (T,3), joints (T,2), empty)."""
    lm = np.ascontiguousarray(landmarks, dtype=np.float64).reshape(-1)
    cmd = np.ascontiguousarray(cmd, dtype=np.float64).reshape(-1, 2)
    T, n = cmd.shape[0], lm.size // 2
    mm = max(m, 1)
    tw = np.zeros((T, 2)); mx = np.zeros((T, mm)); my = np.zeros((T, mm)); ids = np.zeros((T, mm), dtype=np.int32)
    truth = np.zeros((T, 3)); joints = np.zeros((T, 2))
    empty = lib().orc_simulate(C.byref(params), _p(lm), n, _p(cmd), T, m, int(seed), int(filt), _p(tw), _p(mx),
                               _p(my), ids.ctypes.data_as(_ip), _p(truth), _p(joints))
    if empty < 0:
        raise OracleError(ORC_E_ARG)
    return dict(tw=tw, mx=mx[:, :m], my=my[:, :m], ids=ids[:, :m], truth=truth, joints=joints, empty=int(empty))


def sim_scan(landmarks, tube_radius, max_scan_range, pose_th_x_y):
    """simulate_lidar_scanner (tube_world.cpp:405-471): 360 float ranges for the robot at pose (th, x, y)."""
    lm = np.ascontiguousarray(landmarks, dtype=np.float64).reshape(-1)
    out = np.zeros(360, dtype=np.float32)
    th, x, y = pose_th_x_y
    lib().orc_sim_scan(_p(lm), lm.size // 2, tube_radius, max_scan_range, x, y, th, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def scan_markers(ranges, min_range, max_range):
    """The landmarks node's loop body (nuslam/src/landmarks.cpp:63, 82-108): clusterPoints -> classifyCluster ->
    circleFit -> keep fits with marker.id >= 0 and radius <= 1.  Returns the marker centres, in cluster order."""
    out = []
    for xs, ys in cluster_points(ranges, min_range, max_range):
        ok, _ = classify_cluster(xs, ys)
        if not ok:
            continue
        st, cx, cy, r = circle_fit(xs, ys)
        if st != 0 or r > 1:
            continue
        out.append((cx, cy))
    return out


def map_to_odom(odom, state3):
    out = np.zeros(3)
    lib().orc_map_to_odom(_p(np.ascontiguousarray(odom, dtype=np.float64)),
                          _p(np.ascontiguousarray(state3, dtype=np.float64)), _p(out))
    return out


def usable_cpus():
    """CPUs this process may actually use: affinity mask clipped by the cgroup CPU quota (a GPU box hands a
    16-CPU share of a 256-thread host to each GPU)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    env = os.environ.get("NUSLAM_ORACLE_THREADS")
    if env:
        n = int(env)
    elif n > 32:
        n = 16      # no quota visible on a big shared host: stay within the documented per-GPU share
    return n


def set_threads(n):
    lib().orc_set_threads(int(n))


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__("oracle status %d" % code)
        self.code = code


def _chk(rc):
    if rc != ORC_OK:
        raise OracleError(rc)


class OracleEKF:
    """slam_library::ExtendedKalman as restated by the oracle. Matrices are column-major (Fortran order)."""

    def __init__(self, robot, map_state, Q, R, mode=ORC_DENSE):
        L = lib()
        robot = np.ascontiguousarray(robot, dtype=np.float64)
        map_state = np.ascontiguousarray(map_state, dtype=np.float64)
        Qc = np.asfortranarray(np.asarray(Q, dtype=np.float64).reshape(3, 3))
        Rc = np.asfortranarray(np.asarray(R, dtype=np.float64).reshape(2, 2))
        n = map_state.size // 2
        self._h = L.orc_create(_p(robot), _p(map_state) if map_state.size else None, n,
                               Qc.ctypes.data_as(_dp), Rc.ctypes.data_as(_dp))
        if not self._h:
            raise MemoryError("orc_create")
        self.n = n
        self.len = 3 + 2 * n
        L.orc_set_mode(self._h, mode)

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_mode(self, mode):
        lib().orc_set_mode(self._h, mode)

    @property
    def state(self):
        return np.ctypeslib.as_array(lib().orc_state(self._h), shape=(self.len,))

    @property
    def cov(self):
        """Live view, shape (len, len), cov[i, j] = P(i, j)."""
        flat = np.ctypeslib.as_array(lib().orc_cov(self._h), shape=(self.len * self.len,))
        return flat.reshape((self.len, self.len), order="F")

    @property
    def seen(self):
        return lib().orc_seen(self._h)

    @seen.setter
    def seen(self, v):
        lib().orc_set_seen(self._h, int(v))

    def restore(self, state, cov, seen):
        self.state[:] = state
        self.cov[:, :] = cov
        self.seen = seen

    def predict(self, dth, dx, dy=0.0):
        lib().orc_predict(self._h, dth, dx, dy)

    def predict_dense(self, F):
        Fc = np.asfortranarray(F, dtype=np.float64)
        _chk(lib().orc_predict_dense(self._h, Fc.ctypes.data_as(_dp)))

    def update(self, r, phi, idx):
        _chk(lib().orc_update(self._h, r, phi, idx))

    def init_landmark(self, r, phi, idx):
        _chk(lib().orc_init_landmark(self._h, r, phi, idx))

    def associate(self, r, phi, want_d=False):
        out = C.c_int(0)
        d = np.full(max(self.seen, 1), np.nan)
        _chk(lib().orc_associate(self._h, r, phi, C.byref(out), _p(d)))
        return (out.value, d) if want_d else out.value

    def tick(self, tw=None, dd=None, thL=0.0, thR=0.0, mx=(), my=(), known_ids=None, total_landmarks=None):
        mx = np.ascontiguousarray(mx, dtype=np.float64)
        my = np.ascontiguousarray(my, dtype=np.float64)
        m = mx.size
        ids_out = np.zeros(max(m, 1), dtype=np.int32)
        kid = None
        if known_ids is not None:
            kid = np.ascontiguousarray(known_ids, dtype=np.int32)
        twp = None
        if tw is not None:
            twa = np.ascontiguousarray(tw, dtype=np.float64)
            twp = _p(twa)
        _chk(lib().orc_tick(self._h, _p(dd) if dd is not None else None, thL, thR, twp, m,
                            _p(mx), _p(my), kid.ctypes.data_as(_ip) if kid is not None else None,
                            self.n if total_landmarks is None else total_landmarks,
                            ids_out.ctypes.data_as(_ip)))
        return ids_out[:m].copy()


def measurement(state, j):
    s = np.ascontiguousarray(state, dtype=np.float64)
    out = np.zeros(2)
    lib().orc_measurement(_p(s), j, _p(out))
    return out


def jacobian(state, j):
    s = np.ascontiguousarray(state, dtype=np.float64)
    H = np.zeros((2, s.size), order="F")
    lib().orc_jacobian(_p(s), s.size, j, H.ctypes.data_as(_dp))
    return H


def cartesian2polar(x, y):
    out = np.zeros(2)
    lib().orc_cartesian2polar(x, y, _p(out))
    return out


def normalize_angle(a):
    return lib().orc_normalize_angle(a)


def integrate_twist(tw):
    tw = np.ascontiguousarray(tw, dtype=np.float64)
    T = np.zeros(4)
    lib().orc_integrate_twist(_p(tw), _p(T))
    return T


def transform_twist(T, tw):
    T = np.ascontiguousarray(T, dtype=np.float64)
    tw = np.ascontiguousarray(tw, dtype=np.float64)
    out = np.zeros(3)
    lib().orc_transform_twist(_p(T), _p(tw), _p(out))
    return out


def dd_new(base, rad, x=0.0, y=0.0, th=0.0, thL=0.0, thR=0.0):
    return np.array([base, rad, x, y, th, thL, thR], dtype=np.float64)


def dd_get_twist(dd, thL, thR):
    out = np.zeros(3)
    lib().orc_dd_get_twist(_p(dd), thL, thR, _p(out))
    return out


def dd_step(dd, thL, thR):
    lib().orc_dd_step(_p(dd), thL, thR)
    return dd


def dd_convert_twist(dd, tw):
    tw = np.ascontiguousarray(tw, dtype=np.float64)
    out = np.zeros(2)
    lib().orc_dd_convert_twist(_p(dd), _p(tw), _p(out))
    return out


# ---------------------------------------------------------------- the real reference (rigid2d only)
_ref = None


def ref_available():
    if not os.path.exists(REF_SO):
        try:
            build_oracle()
        except Exception:
            pass
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        R = C.CDLL(REF_SO)
        R.ref_normalize_angle.restype = C.c_double
        R.ref_normalize_angle.argtypes = [C.c_double]
        R.ref_integrate_twist.argtypes = [_dp, _dp]
        R.ref_transform_twist.argtypes = [C.c_double, C.c_double, C.c_double, _dp, _dp]
        R.ref_dd_convert_twist.argtypes = [_dp, _dp, _dp]
        R.ref_dd_get_twist.argtypes = [_dp, C.c_double, C.c_double, _dp]
        R.ref_dd_step.argtypes = [_dp, C.c_double, C.c_double]
        R.ref_tf_inv.argtypes = [C.c_double] * 3 + [_dp]
        R.ref_tf_mul.argtypes = [C.c_double] * 6 + [_dp]
        R.ref_tf_point.argtypes = [C.c_double] * 5 + [_dp]
        R.ref_map_to_odom.argtypes = [_dp, _dp, _dp]
        _ref = R
    return _ref
