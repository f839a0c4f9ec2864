"""ctypes binding of libnuslam_hip.so (the C ABI of include/nuslam_hip.h) for tests, smoke() and bench.py.

This is a thin harness binding, not a second implementation: every numerical entry point below is one call
into the HIP library, and loading fails loudly when the library has not been built (there is no CPU fallback).
The host-language mirror of the reference's C++ class lives in ../cpp/nuslam/slam_library.hpp.
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("NUSLAM_HIP_LIB") or os.path.join(PKG_DIR, "libnuslam_hip.so")   # override: A/B experiments only

OK, E_ARG, E_BOUNDS, E_SINGULAR, E_HIP, E_NODEV, E_NOMEM, E_CAPACITY, E_COMM, E_SYNC = range(10)
COMM_ID_BYTES = 128
F64, F32 = 0, 1
K_PREDICT, K_ASSOCIATE, K_UPDATE, K_DENSE_GEMM, K_UPDATE_DEFERRED, K_FLUSH, K_UPDATE2, K_TICK_CHAIN, K_TICK_PANELS, K_TICK_APPLY, K_TICK_NEXT, K_DA_BEGIN, K_DA_STEP, K_TICK_RANK = range(14)
# nuslam_batch_set_pass_variant: rank-2m pass on the matrix cores (default) / exact chain, plain kernel / exact chain, two-unit kernel
PASS_RANK, PASS_EXACT_PLAIN, PASS_EXACT = 0, 1, 2

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

# every symbol include/nuslam_hip.h declares: (name, restype, argtypes)
_vp = C.c_void_p
_vpp = C.POINTER(C.c_void_p)
SYMBOLS = [
    ("nuslam_strerror", C.c_char_p, [C.c_int]),
    ("nuslam_last_hip_error", C.c_char_p, []),
    ("nuslam_abi_version", C.c_int, []),
    ("nuslam_device_count", C.c_int, [_ip]),
    ("nuslam_cartesian2polar", C.c_int, [C.c_double, C.c_double, _dp]),
    ("nuslam_measurement", C.c_int, [_dp, C.c_int, C.c_int, _dp]),
    ("nuslam_jacobian", C.c_int, [_dp, C.c_int, C.c_int, _dp]),
    ("nuslam_ekf_create", C.c_int, [_dp, _dp, C.c_int, _dp, _dp, C.c_int, C.c_int, _vpp]),
    ("nuslam_ekf_destroy", C.c_int, [_vp]),
    ("nuslam_ekf_clone", C.c_int, [_vp, _vpp]),
    ("nuslam_ekf_predict", C.c_int, [_vp, C.c_double, C.c_double, C.c_double]),
    ("nuslam_ekf_predict_dense", C.c_int, [_vp, _dp, C.c_int]),
    ("nuslam_ekf_use_dense_predict", C.c_int, [_vp, C.c_int]),
    ("nuslam_ekf_update", C.c_int, [_vp, C.c_double, C.c_double, C.c_int]),
    ("nuslam_ekf_associate", C.c_int, [_vp, C.c_double, C.c_double, _ip]),
    ("nuslam_ekf_init_landmark", C.c_int, [_vp, C.c_double, C.c_double, C.c_int]),
    ("nuslam_ekf_tick", C.c_int, [_vp, C.c_double, C.c_double, C.c_double, C.c_int, _dp, _dp, _ip, C.c_int, _ip]),
    ("nuslam_ekf_tick_ex", C.c_int, [_vp, _dp, C.c_int, _dp, _dp, C.c_int, _ip, C.c_int, _ip]),
    ("nuslam_ekf_set_lazy", C.c_int, [_vp, C.c_int]),
    ("nuslam_batch_inject_fault", C.c_int, [_vp, C.c_int]),
    ("nuslam_ekf_len", C.c_int, [_vp, _ip]),
    ("nuslam_ekf_get_state", C.c_int, [_vp, _dp, C.c_int]),
    ("nuslam_ekf_get_cov", C.c_int, [_vp, _dp, C.c_int]),
    ("nuslam_ekf_get_seen", C.c_int, [_vp, _ip]),
    ("nuslam_ekf_restore", C.c_int, [_vp, _dp, _dp, C.c_int, C.c_int]),
    ("nuslam_ekf_sync", C.c_int, [_vp]),
    ("nuslam_ekf_status", C.c_int, [_vp, C.c_int, _ip]),
    ("nuslam_batch_create", C.c_int, [C.c_int, _dp, _dp, C.c_int, _dp, _dp, C.c_int, C.c_int, _vpp]),
    ("nuslam_batch_destroy", C.c_int, [_vp]),
    ("nuslam_batch_size", C.c_int, [_vp, _ip, _ip]),
    ("nuslam_batch_load_trace", C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp, _ip, C.c_int]),
    ("nuslam_batch_run", C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    ("nuslam_batch_get_state", C.c_int, [_vp, C.c_int, _dp, C.c_int]),
    ("nuslam_batch_get_cov", C.c_int, [_vp, C.c_int, _dp, C.c_int]),
    ("nuslam_batch_get_seen", C.c_int, [_vp, C.c_int, _ip]),
    ("nuslam_batch_restore", C.c_int, [_vp, C.c_int, _dp, _dp, C.c_int, C.c_int]),
    ("nuslam_batch_sync", C.c_int, [_vp]),
    ("nuslam_batch_status", C.c_int, [_vp, C.c_int, _ip, _ip]),
    ("nuslam_batch_stats", C.c_int, [_vp, _dp, C.c_int]),
    ("nuslam_ekf_as_batch", C.c_int, [_vp, _vpp]),
    ("nuslam_batch_set_deferred", C.c_int, [_vp, C.c_int]),
    ("nuslam_ekf_set_deferred", C.c_int, [_vp, C.c_int]),
    ("nuslam_batch_set_pairing", C.c_int, [_vp, C.c_int]),
    ("nuslam_batch_set_tick_mode", C.c_int, [_vp, C.c_int]),
    ("nuslam_batch_set_overlap", C.c_int, [_vp, C.c_int]),
    ("nuslam_batch_set_interleave", C.c_int, [_vp, C.c_int]),
    ("nuslam_batch_set_pass_variant", C.c_int, [_vp, C.c_int]),
    ("nuslam_circle_fit_batch", C.c_int, [C.c_int, _ip, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _dp, C.c_int, _dp]),
    ("nuslam_batch_profile", C.c_int, [_vp, C.c_int]),
    ("nuslam_batch_profile_read", C.c_int, [_vp, C.c_int, _dp, C.POINTER(C.c_longlong)]),
    ("nuslam_batch_timer_start", C.c_int, [_vp]),
    ("nuslam_batch_timer_stop", C.c_int, [_vp, _dp]),
    ("nuslam_map_to_odom", C.c_int, [_dp, _dp, _dp]),
    ("nuslam_batch_simulate", C.c_int, [_vp, C.c_void_p, _dp, C.c_int, _dp, C.c_int, C.c_int, C.c_ulonglong, C.c_uint,
                                        C.c_int, C.POINTER(C.c_longlong)]),
    ("nuslam_batch_get_trace", C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _ip, _dp]),
    ("nuslam_batch_get_scan", C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    ("nuslam_philox4x32_10", C.c_int, [C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_int]),
    ("nuslam_device_normalize_angle", C.c_int, [_dp, C.c_int, _dp, C.c_int]),
    ("nuslam_build_info", C.c_char_p, []),
    ("nuslam_ekf_snapshot", C.c_int, [_vp, _dp, C.c_int, _dp, C.c_int, _ip]),
    ("nuslam_comm_unique_id", C.c_int, [C.POINTER(C.c_ubyte)]),
    ("nuslam_comm_create", C.c_int, [C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int, _vpp]),
    ("nuslam_comm_destroy", C.c_int, [_vp]),
    ("nuslam_comm_size", C.c_int, [_vp, _ip, _ip]),
    ("nuslam_batch_reduce_stats", C.c_int, [_vp, _vp, _dp, C.c_int, _dp]),
]


class SimParams(C.Structure):
    """nuslam_sim_params (include/nuslam_hip.h).  Defaults: nuturtlesim/config/tube_world_params.yaml and
    nuturtle_description/config/diff_params.yaml of the reference."""
    _fields_ = [(k, C.c_double) for k in ("wheel_base", "wheel_radius", "dt", "twist_noise", "slip_min", "slip_max",
                                          "tube_radius", "robot_radius", "tube_var", "marker_sigma", "max_range",
                                          "lidar", "lidar_min_range", "lidar_max_range", "fov", "min_range")]

    def __init__(self, **kw):
        d = dict(wheel_base=0.16, wheel_radius=0.033, dt=1.0 / 50, twist_noise=0.0, slip_min=0.9, slip_max=1.0,
                 tube_radius=0.0381, robot_radius=0.08, tube_var=0.001, marker_sigma=0.0, max_range=1.0,
                 lidar=0.0, lidar_min_range=0.05, lidar_max_range=1.0, fov=0.0, min_range=0.0)
        d.update(kw)
        super().__init__(**d)


def map_to_odom(odom_xyth, state3):
    """EKFSlam::broadcast_map2odom_tf (slam.cpp:175-210): (x, y, yaw) of the map -> odom transform."""
    out = np.zeros(3)
    _chk(lib().nuslam_map_to_odom(_p(np.ascontiguousarray(odom_xyth, dtype=np.float64)),
                                  _p(np.ascontiguousarray(state3, dtype=np.float64)), _p(out)), "map_to_odom")
    return out


def device_normalize_angle(a, device=0):
    """rigid2d::normalize_angle as the device evaluates it (csrc/ekf_device.h), elementwise."""
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
    out = np.zeros_like(a)
    _chk(lib().nuslam_device_normalize_angle(_p(a), a.size, _p(out), device), "device_normalize_angle")
    return out


def philox(ctr, key, device=0):
    c = (C.c_uint * 4)(*[int(x) for x in ctr]); k = (C.c_uint * 2)(*[int(x) for x in key]); o = (C.c_uint * 4)()
    _chk(lib().nuslam_philox4x32_10(c, k, o, device), "philox")
    return [int(x) for x in o]


class NuslamError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        msg = "%s: status %d" % (where, code)
        try:
            L = lib()
            msg = "%s: %s" % (where, L.nuslam_strerror(code).decode())
            if code in (E_HIP, E_COMM):
                msg += " (" + L.nuslam_last_hip_error().decode() + ")"
        except Exception:
            pass
        super().__init__(msg)


def build(force=False):
    """Compile libnuslam_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", PKG_DIR, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", PKG_DIR, "-j4", "all"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    """Load the HIP library; raises if it is missing -- the product path never falls back to anything else."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libnuslam_hip.so is not built (%s); run __graft_entry__.build() or `make -C %s`"
                              % (LIB_PATH, PKG_DIR))
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            f = getattr(L, name)       # AttributeError if the library does not export a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _chk(rc, where):
    if rc != OK:
        raise NuslamError(rc, where)


def _p(a):
    return a.ctypes.data_as(_dp)


def build_info():
    """'csrc=<hash>' of the kernel sources the loaded library was built from."""
    return lib().nuslam_build_info().decode()


class Comm:
    """nuslam_comm_t: the RCCL communicator of the batch reduction (one rank per GPU)."""

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * COMM_ID_BYTES)()
        _chk(lib().nuslam_comm_unique_id(buf), "comm_unique_id")
        return bytes(buf)

    def __init__(self, uid, world, rank, device=0):
        buf = (C.c_ubyte * COMM_ID_BYTES)(*uid)
        h = C.c_void_p()
        _chk(lib().nuslam_comm_create(buf, world, rank, device, C.byref(h)), "comm_create")
        self._h = h
        self.world, self.rank = world, rank

    def close(self):
        if getattr(self, "_h", None):
            lib().nuslam_comm_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_count():
    n = C.c_int(0)
    lib().nuslam_device_count(C.byref(n))
    return n.value


def cartesian2polar(x, y):
    out = np.zeros(2)
    _chk(lib().nuslam_cartesian2polar(x, y, _p(out)), "cartesian2polar")
    return out


def measurement(state, j):
    s = np.ascontiguousarray(state, dtype=np.float64)
    out = np.zeros(2)
    _chk(lib().nuslam_measurement(_p(s), s.size, j, _p(out)), "measurement")
    return out


def jacobian(state, j):
    s = np.ascontiguousarray(state, dtype=np.float64)
    H = np.zeros((2, s.size), order="F")
    _chk(lib().nuslam_jacobian(_p(s), s.size, j, H.ctypes.data_as(_dp)), "jacobian")
    return H


def circle_fit_batch(clusters, device=0):
    """clusters: list of (xs, ys).  Returns dict of arrays: cx, cy, radius, status, is_circle, angle_std, kernel_ms."""
    n = len(clusters)
    sizes = [len(c[0]) for c in clusters]
    off = np.zeros(n + 1, dtype=np.int32)
    off[1:] = np.cumsum(sizes)
    xs = np.ascontiguousarray(np.concatenate([np.asarray(c[0], dtype=np.float64) for c in clusters]) if n else np.zeros(0))
    ys = np.ascontiguousarray(np.concatenate([np.asarray(c[1], dtype=np.float64) for c in clusters]) if n else np.zeros(0))
    cx, cy, rad, sd = (np.zeros(max(n, 1)) for _ in range(4))
    st = np.zeros(max(n, 1), dtype=np.int32)
    circ = np.zeros(max(n, 1), dtype=np.int32)
    ms = C.c_double()
    _chk(lib().nuslam_circle_fit_batch(n, off.ctypes.data_as(_ip), _p(xs) if xs.size else _p(np.zeros(1)),
                                       _p(ys) if ys.size else _p(np.zeros(1)), _p(cx), _p(cy), _p(rad),
                                       st.ctypes.data_as(_ip), circ.ctypes.data_as(_ip), _p(sd), device, C.byref(ms)),
         "circle_fit_batch")
    return {"cx": cx[:n], "cy": cy[:n], "radius": rad[:n], "status": st[:n], "is_circle": circ[:n].astype(bool),
            "angle_std": sd[:n], "kernel_ms": ms.value}


def _qr(Q, R):
    Qc = np.asfortranarray(np.asarray(Q, dtype=np.float64).reshape(3, 3))
    Rc = np.asfortranarray(np.asarray(R, dtype=np.float64).reshape(2, 2))
    return Qc, Rc


class Batch:
    """B independent filters resident on one GPU (nuslam_batch_t)."""

    def __init__(self, n_filters, n_landmarks, Q, R, robot=None, map_state=None, dtype=F64, device=0, _borrow=None):
        self._own = _borrow is None
        if _borrow is not None:
            self._h = _borrow
        else:
            Qc, Rc = _qr(Q, R)
            rb = None if robot is None else np.ascontiguousarray(robot, dtype=np.float64)
            mp = None if map_state is None else np.ascontiguousarray(map_state, dtype=np.float64)
            h = C.c_void_p()
            _chk(lib().nuslam_batch_create(n_filters, _p(rb) if rb is not None else None,
                                           _p(mp) if mp is not None else None, n_landmarks,
                                           Qc.ctypes.data_as(_dp), Rc.ctypes.data_as(_dp), dtype, device, C.byref(h)),
                 "batch_create")
            self._h = h
        b, l = C.c_int(), C.c_int()
        _chk(lib().nuslam_batch_size(self._h, C.byref(b), C.byref(l)), "batch_size")
        self.B, self.len = b.value, l.value
        self.n = (self.len - 3) // 2

    def close(self):
        if getattr(self, "_h", None) and self._own:
            lib().nuslam_batch_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_trace(self, tw, mx, my, ids=None, bcast=False):
        """tw: (B|1, T, 2); mx, my: (B|1, T, m); ids: same shape int32 or None (data association)."""
        tw = np.ascontiguousarray(tw, dtype=np.float64)
        mx = np.ascontiguousarray(mx, dtype=np.float64)
        my = np.ascontiguousarray(my, dtype=np.float64)
        if tw.ndim == 2:
            tw, mx, my = tw[None], mx[None], my[None]
            if ids is not None:
                ids = np.asarray(ids)[None]
        nb, T = tw.shape[0], tw.shape[1]
        m = mx.shape[2]
        assert nb == (1 if bcast else self.B), "trace batch dimension"
        idp = None
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype=np.int32)
            idp = ids.ctypes.data_as(_ip)
        _chk(lib().nuslam_batch_load_trace(self._h, T, m, _p(tw), _p(mx), _p(my), idp, 1 if bcast else 0),
             "batch_load_trace")
        self._trace_shape = (T, m, ids is not None)
        self._trace_generated = False

    def simulate(self, params, landmarks, cmd, m, seed, first_filter=0, known_ids=True):
        """Generate the resident trace on the device (tube_world.cpp:509-533 per filter): returns the number of
        unused marker slots."""
        lm = np.ascontiguousarray(landmarks, dtype=np.float64).reshape(-1)
        cmd = np.ascontiguousarray(cmd, dtype=np.float64).reshape(-1, 2)
        empty = C.c_longlong()
        _chk(lib().nuslam_batch_simulate(self._h, C.addressof(params), _p(lm), lm.size // 2, _p(cmd), cmd.shape[0],
                                         int(m), int(seed), int(first_filter), 1 if known_ids else 0,
                                         C.byref(empty)), "batch_simulate")
        self._trace_shape = (cmd.shape[0], int(m), True)      # generated traces always carry ids (identity or presence)
        self._trace_generated = True
        return empty.value

    def get_scan(self, b, tick):
        out = np.zeros(360, dtype=np.float32)
        _chk(lib().nuslam_batch_get_scan(self._h, b, tick, out.ctypes.data_as(C.POINTER(C.c_float))), "batch_get_scan")
        return out

    def get_trace(self, b=0):
        """Filter b's resident trace: dict(tw (T,2), mx, my, ids (T,m), truth (T,3)).  For a generated data-association
        trace `ids` only marks the filled slots (> 0) and the empty ones (-1)."""
        T, m, known = self._trace_shape
        generated = getattr(self, "_trace_generated", False)
        tw = np.zeros((T, 2)); mx = np.zeros((T, m)); my = np.zeros((T, m))
        truth = np.zeros((T, 3)) if generated else None
        ids = np.zeros((T, m), dtype=np.int32) if known else None
        _chk(lib().nuslam_batch_get_trace(self._h, b, _p(tw), _p(mx), _p(my),
                                          ids.ctypes.data_as(_ip) if known else None,
                                          _p(truth) if generated else None), "batch_get_trace")
        return dict(tw=tw, mx=mx, my=my, ids=ids, truth=truth)

    def run(self, t_begin, t_end, total_landmarks=None):
        _chk(lib().nuslam_batch_run(self._h, t_begin, t_end, self.n if total_landmarks is None else total_landmarks),
             "batch_run")

    def state(self, b=0):
        out = np.zeros(self.len)
        _chk(lib().nuslam_batch_get_state(self._h, b, _p(out), self.len), "batch_get_state")
        return out

    def cov(self, b=0):
        out = np.zeros((self.len, self.len), order="F")
        _chk(lib().nuslam_batch_get_cov(self._h, b, out.ctypes.data_as(_dp), self.len), "batch_get_cov")
        return out

    def seen(self, b=0):
        s = C.c_int()
        _chk(lib().nuslam_batch_get_seen(self._h, b, C.byref(s)), "batch_get_seen")
        return s.value

    def restore(self, b, state, cov, seen):
        st = np.ascontiguousarray(state, dtype=np.float64)
        cv = np.asfortranarray(cov, dtype=np.float64)
        _chk(lib().nuslam_batch_restore(self._h, b, _p(st), cv.ctypes.data_as(_dp), self.len, int(seen)), "batch_restore")

    def sync(self):
        _chk(lib().nuslam_batch_sync(self._h), "batch_sync")

    def status(self, clear=False):
        bad, st = C.c_int(), C.c_int()
        _chk(lib().nuslam_batch_status(self._h, 1 if clear else 0, C.byref(bad), C.byref(st)), "batch_status")
        return bad.value, st.value

    def stats(self):
        """{sum state (len), sum state^2 (len), sum pose error^2 (3), sum NEES, sum trace(P), count}: 2 len + 6."""
        out = np.zeros(2 * self.len + 6)
        _chk(lib().nuslam_batch_stats(self._h, _p(out), out.size), "batch_stats")
        return out

    def reduce_stats(self, comm):
        """All ranks: (total, per_rank) of the statistics vector, gathered over RCCL and added in rank order on the device."""
        n = 2 * self.len + 6
        total = np.zeros(n)
        per_rank = np.zeros((comm.world, n))
        _chk(lib().nuslam_batch_reduce_stats(self._h, comm._h, _p(total), n, _p(per_rank)), "batch_reduce_stats")
        return total, per_rank

    def set_deferred(self, enable=True):
        _chk(lib().nuslam_batch_set_deferred(self._h, 1 if enable else 0), "batch_set_deferred")

    def set_pairing(self, enable=True):
        """False / True: k_update2 for consecutive plain corrections (tick mode 0)."""
        _chk(lib().nuslam_batch_set_pairing(self._h, int(enable)), "batch_set_pairing")

    def set_tick_mode(self, mode):
        """1 (default): known-id ticks as chain + panels + ONE pass over P (one filter: one launch per tick, a nuslam_batch_run one launch);
        0: one pass per correction / pair; 3 / 4: the tick as three / two launches; 5: as 1 with a launch per tick in a run."""
        _chk(lib().nuslam_batch_set_tick_mode(self._h, int(mode)), "batch_set_tick_mode")

    def set_interleave(self, groups):
        """groups of filters on streams of their own in nuslam_batch_run (1..4; < 0: the library's default)."""
        _chk(lib().nuslam_batch_set_interleave(self._h, int(groups)), "batch_set_interleave")

    def set_overlap(self, enable=True):
        """True / False / None (library default); 2: the test hook (second stream = the handle's own)."""
        _chk(lib().nuslam_batch_set_overlap(self._h, -1 if enable is None else (2 if enable == 2 and enable is not True else (1 if enable else 0))),
             "batch_set_overlap")

    def set_pass_variant(self, variant):
        _chk(lib().nuslam_batch_set_pass_variant(self._h, int(variant)), "batch_set_pass_variant")

    def inject_fault(self, kind):
        """1: one expired device-side wait (-> NUSLAM_E_SYNC at the next status read); 2: NaN into the gain / factor strips."""
        _chk(lib().nuslam_batch_inject_fault(self._h, int(kind)), "batch_inject_fault")

    def profile(self, enable):
        _chk(lib().nuslam_batch_profile(self._h, 1 if enable else 0), "batch_profile")

    def profile_read(self, kernel):
        ms, n = C.c_double(), C.c_longlong()
        _chk(lib().nuslam_batch_profile_read(self._h, kernel, C.byref(ms), C.byref(n)), "batch_profile_read")
        return ms.value, n.value

    def timer_start(self):
        _chk(lib().nuslam_batch_timer_start(self._h), "timer_start")

    def timer_stop(self):
        ms = C.c_double()
        _chk(lib().nuslam_batch_timer_stop(self._h, C.byref(ms)), "timer_stop")
        return ms.value


class EKF:
    """One filter (nuslam_ekf_t) -- method for method slam_library::ExtendedKalman (slam_library.hpp:23-113)."""

    def __init__(self, robot, map_state, Q, R, dtype=F64, device=0, _handle=None):
        if _handle is not None:
            self._h = _handle
        else:
            Qc, Rc = _qr(Q, R)
            rb = np.ascontiguousarray(robot, dtype=np.float64)
            mp = np.ascontiguousarray(map_state, dtype=np.float64)
            h = C.c_void_p()
            _chk(lib().nuslam_ekf_create(_p(rb), _p(mp) if mp.size else None, mp.size // 2, Qc.ctypes.data_as(_dp),
                                         Rc.ctypes.data_as(_dp), dtype, device, C.byref(h)), "ekf_create")
            self._h = h
        l = C.c_int()
        _chk(lib().nuslam_ekf_len(self._h, C.byref(l)), "ekf_len")
        self.len = l.value
        self.n = (self.len - 3) // 2

    def close(self):
        if getattr(self, "_h", None):
            lib().nuslam_ekf_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clone(self):
        h = C.c_void_p()
        _chk(lib().nuslam_ekf_clone(self._h, C.byref(h)), "ekf_clone")
        return EKF(None, None, None, None, _handle=h)

    def as_batch(self):
        h = C.c_void_p()
        _chk(lib().nuslam_ekf_as_batch(self._h, C.byref(h)), "ekf_as_batch")
        b = Batch(1, self.n, None, None, _borrow=h)
        b._keep = self
        return b

    def predict(self, dth, dx, dy=0.0):
        _chk(lib().nuslam_ekf_predict(self._h, dth, dx, dy), "ekf_predict")

    def predict_dense(self, F=None):
        if F is None:
            _chk(lib().nuslam_ekf_predict_dense(self._h, None, 0), "ekf_predict_dense")
            return
        Fc = np.asfortranarray(F, dtype=np.float64)
        _chk(lib().nuslam_ekf_predict_dense(self._h, Fc.ctypes.data_as(_dp), Fc.shape[0]), "ekf_predict_dense")

    def set_deferred(self, enable=True):
        _chk(lib().nuslam_ekf_set_deferred(self._h, 1 if enable else 0), "ekf_set_deferred")

    def use_dense_predict(self, enable=True):
        """2: the reference's predict on the matrix cores (getA formed on the device every tick); True / 1: the Jacobian
        last staged by predict_dense, unchanged; False / 0: the O(len) shortcut."""
        _chk(lib().nuslam_ekf_use_dense_predict(self._h, int(enable)), "ekf_use_dense_predict")

    def update(self, r, phi, idx):
        _chk(lib().nuslam_ekf_update(self._h, r, phi, idx), "ekf_update")

    def associate(self, r, phi):
        out = C.c_int()
        _chk(lib().nuslam_ekf_associate(self._h, r, phi, C.byref(out)), "ekf_associate")
        return out.value

    def init_landmark(self, r, phi, idx):
        _chk(lib().nuslam_ekf_init_landmark(self._h, r, phi, idx), "ekf_init_landmark")

    def tick(self, tw, mx, my, known_ids=None, total_landmarks=None, want_ids=True):
        mx = np.ascontiguousarray(mx, dtype=np.float64)
        my = np.ascontiguousarray(my, dtype=np.float64)
        m = mx.size
        assert my.size == m, "marker x / y arrays differ in length"
        kid = None
        if known_ids is not None:
            kid = np.ascontiguousarray(known_ids, dtype=np.int32)
            assert kid.size == m, "known_ids must have one id per marker"
        ids_out = np.zeros(max(m, 1), dtype=np.int32)
        _chk(lib().nuslam_ekf_tick(self._h, tw[0], tw[1], tw[2] if len(tw) > 2 else 0.0, m, _p(mx), _p(my),
                                   kid.ctypes.data_as(_ip) if kid is not None else None,
                                   self.n if total_landmarks is None else total_landmarks,
                                   ids_out.ctypes.data_as(_ip) if want_ids else None), "ekf_tick")
        return ids_out[:m].copy() if want_ids else None

    def set_lazy(self, enable=True):
        """Lazy ticks (default on): predict / init_landmark / update are recorded and applied as one tick; False: every call
        launches its own kernels (one pass over the covariance per update -- the reference arithmetic of the bitwise tests)."""
        _chk(lib().nuslam_ekf_set_lazy(self._h, 1 if enable else 0), "ekf_set_lazy")

    def tick_ex(self, tw, a, b, polar=False, known_ids=None, total_landmarks=None, want_ids=True):
        """nuslam_ekf_tick_ex: tw None = no predict (the markers continue the current tick); polar: a / b are (range, bearing)."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        m = a.size
        assert b.size == m
        kid = None
        if known_ids is not None:
            kid = np.ascontiguousarray(known_ids, dtype=np.int32)
            assert kid.size == m
        twp = None
        if tw is not None:
            twa = np.ascontiguousarray([tw[0], tw[1]], dtype=np.float64)
            twp = _p(twa)
        ids_out = np.zeros(max(m, 1), dtype=np.int32)
        _chk(lib().nuslam_ekf_tick_ex(self._h, twp, m, _p(a), _p(b), 1 if polar else 0,
                                      kid.ctypes.data_as(_ip) if kid is not None else None,
                                      self.n if total_landmarks is None else total_landmarks,
                                      ids_out.ctypes.data_as(_ip) if want_ids else None), "ekf_tick_ex")
        return ids_out[:m].copy() if want_ids else None

    @property
    def state(self):
        out = np.zeros(self.len)
        _chk(lib().nuslam_ekf_get_state(self._h, _p(out), self.len), "ekf_get_state")
        return out

    @property
    def cov(self):
        out = np.zeros((self.len, self.len), order="F")
        _chk(lib().nuslam_ekf_get_cov(self._h, out.ctypes.data_as(_dp), self.len), "ekf_get_cov")
        return out

    @property
    def seen(self):
        s = C.c_int()
        _chk(lib().nuslam_ekf_get_seen(self._h, C.byref(s)), "ekf_get_seen")
        return s.value

    def restore(self, state, cov, seen):
        st = np.ascontiguousarray(state, dtype=np.float64)
        cv = np.asfortranarray(cov, dtype=np.float64)
        _chk(lib().nuslam_ekf_restore(self._h, _p(st), cv.ctypes.data_as(_dp), self.len, int(seen)), "ekf_restore")

    def snapshot(self):
        """(state, cov, seen) in one call -- what restore() takes back."""
        st = np.zeros(self.len)
        cv = np.zeros((self.len, self.len), order="F")
        seen = C.c_int()
        _chk(lib().nuslam_ekf_snapshot(self._h, _p(st), self.len, cv.ctypes.data_as(_dp), self.len, C.byref(seen)),
             "ekf_snapshot")
        return st, cv, seen.value

    def sync(self):
        _chk(lib().nuslam_ekf_sync(self._h), "ekf_sync")

    def status(self, clear=False):
        st = C.c_int()
        _chk(lib().nuslam_ekf_status(self._h, 1 if clear else 0, C.byref(st)), "ekf_status")
        return st.value
