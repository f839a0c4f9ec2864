"""The device's rigid2d::normalize_angle (csrc/ekf_device.h) is a two-constant range reduction, not atan2(sin, cos)
(rigid2d/src/rigid2d.cpp:9-13).  This is the ALGORITHM restated in Python with exact fused multiply-adds (fractions), held on
the CPU to the compiled reference's own vectors (tests/golden/rigid2d_ref.npz) and to glibc's atan2(sin, cos): <= 1 ulp, edges
included.  The device implementation itself is held to the same vectors by tests/test_gpu_more.py (-m gpu)."""
import math
import os
from fractions import Fraction

import numpy as np

PI = 3.141592653589793
TWO_PI_HI = 6.283185307179586
TWO_PI_LO = 2.4492935982947064e-16
INV_TWO_PI = 0.15915494309189535


def fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))          # one rounding, like v_fma_f64


def normalize_angle(rad):
    if abs(rad) <= PI:
        return rad
    k = float(np.rint(rad * INV_TWO_PI))
    r = fma(-k, TWO_PI_HI, rad)
    r = fma(-k, TWO_PI_LO, r)
    if r > PI:
        r = fma(-1.0, TWO_PI_LO, r - TWO_PI_HI)
    elif r < -PI:
        r = fma(1.0, TWO_PI_LO, r + TWO_PI_HI)
    return r


def ulps(x, ref):
    return 0.0 if x == ref else abs(x - ref) / np.spacing(max(abs(x), abs(ref)))


def test_constants_are_two_pi_split_in_two():
    from mpmath import mp, mpf
    mp.dps = 60
    two_pi = 2 * mp.pi
    assert TWO_PI_HI == float(two_pi) and TWO_PI_LO == float(two_pi - mpf(TWO_PI_HI))
    assert INV_TWO_PI == float(1 / two_pi) and PI == math.pi


def test_range_reduction_gives_the_references_values():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rigid2d_ref.npz"))
    worst = max(ulps(normalize_angle(float(a)), float(n)) for a, n in zip(g["ang"], g["norm"]))
    assert worst <= 1.0
    rng = np.random.default_rng(5)
    for a in np.concatenate([rng.uniform(-2 * PI, 2 * PI, 4000), rng.uniform(-40, 40, 1000)]):
        a = float(a)
        assert ulps(normalize_angle(a), math.atan2(math.sin(a), math.cos(a))) <= 1.0, a


def test_edges():
    up, down = float(np.nextafter(PI, 4.0)), float(np.nextafter(-PI, -4.0))
    assert normalize_angle(PI) == PI and normalize_angle(-PI) == -PI            # the fp64 values lie inside (-pi, pi]
    assert normalize_angle(up) == math.atan2(math.sin(up), math.cos(up)) == -PI
    assert normalize_angle(down) == math.atan2(math.sin(down), math.cos(down)) == PI
    for a in (3 * PI, -3 * PI, -7.5, 7.5, 2 * PI, -2 * PI, 100.0, 0.0, 1e-300):
        assert ulps(normalize_angle(a), math.atan2(math.sin(a), math.cos(a))) <= 1.0, a
    assert normalize_angle(normalize_angle(5.0)) == normalize_angle(5.0)        # idempotent (the chain relies on it)
