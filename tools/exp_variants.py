#!/usr/bin/env python3
"""A/B harness for sweep-kernel variants (NUSLAM_HIP_LIB = a -DNUSLAM_PHASE_CLOCK build).  Prints, for N = 1000 fp64
pair launches: the per-launch distribution of workgroup exit times, the event-timed launch and the back-to-back cadence."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "shermbot-navigation_amd"))
import nuslam_hip as nh  # noqa: E402
from nuslam_hip import synth  # noqa: E402


def main():
    n, m = int(os.environ.get("N", 1000)), 16
    L = nh.lib()
    has_phase = hasattr(L, "nuslam_debug_wg")
    tr = synth.make_trace(n, 300, m)
    bx, by, wid = synth.warmup_observations(tr.landmarks)
    ekf = nh.EKF(np.zeros(3), np.zeros(2 * n), synth.Q_DEFAULT, synth.R_DEFAULT)
    ekf.tick(np.zeros(3), bx, by, known_ids=wid, want_ids=False)
    ekf.sync()
    bt = ekf.as_batch()
    bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, tr.ids, bcast=True)
    bt.run(0, 20)
    bt.sync()
    t0 = time.perf_counter()
    bt.run(20, 220)
    bt.sync()
    dt = time.perf_counter() - t0
    bt.profile(True)
    bt.run(220, 260)
    bt.sync()
    ms2, n2 = bt.profile_read(nh.K_UPDATE2)
    ms1, n1 = bt.profile_read(nh.K_UPDATE)
    msp, npd = bt.profile_read(nh.K_PREDICT)
    bt.profile(False)
    line = "tick %.2f us; pair launch (events) %.2f us x%d; single %.2f us x%d; predict %.2f us" % (
        1e6 * dt / 200, 1e3 * ms2 / max(n2, 1), n2, 1e3 * ms1 / max(n1, 1), n1, 1e3 * msp / max(npd, 1))
    if has_phase:
        L.nuslam_debug_wg.argtypes = [C.POINTER(C.c_longlong), C.c_int]
        W = int(os.environ.get("NUSLAM_FORCE_WAVES", 8))
        ld = (3 + 2 * n + 31) // 32 * 32
        gx, gy = (ld + 127) // 128, ((3 + 2 * n + 15) // 16 + W - 1) // W
        stats = []
        for t in range(260, 300):
            for i in range(0, m, 2):
                ekf.tick(np.zeros(3) if i else tr.tw[t], tr.mx[t, i:i + 2], tr.my[t, i:i + 2], known_ids=tr.ids[t, i:i + 2],
                         want_ids=False)
                ekf.sync()
                wg = (C.c_longlong * (2 * gx * gy))()
                L.nuslam_debug_wg(wg, gx * gy)
                a = np.array(wg[:], dtype=np.int64).reshape(-1, 2)
                ex = (a[:, 1] - a[:, 0].min()) * 0.01
                stats.append([np.percentile(ex, q) for q in (0, 50, 90, 99, 100)] + [(a[:, 0] - a[:, 0].min()).max() * 0.01])
        s = np.median(np.array(stats), axis=0)
        line += "; exits min %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f (last entry %.2f)" % tuple(s)
    if os.environ.get("SINGLE"):
        bt.set_pairing(False)
        bt.run(0, 10); bt.sync()
        t0 = time.perf_counter(); bt.run(10, 110); bt.sync(); dts = time.perf_counter() - t0
        bt.profile(True); bt.run(110, 150); bt.sync()
        ms1, n1 = bt.profile_read(nh.K_UPDATE); bt.profile(False)
        bt.set_pairing(True)
        line += "; SINGLE k_update tick %.2f us, launch (events) %.2f us x%d" % (1e6 * dts / 100, 1e3 * ms1 / max(n1, 1), n1)
    if os.environ.get("SKIP"):
        # k_update on its skip path: the pure ping-pong copy of P
        bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, -np.ones_like(tr.ids), bcast=True)
        bt.run(0, 10); bt.sync()
        t0 = time.perf_counter(); bt.run(10, 110); bt.sync(); dts = time.perf_counter() - t0
        bt.profile(True); bt.run(110, 150); bt.sync()
        ms1, n1 = bt.profile_read(nh.K_UPDATE); bt.profile(False)
        line += "; SKIP-COPY k_update tick %.2f us, launch (events) %.2f us x%d" % (1e6 * dts / 100, 1e3 * ms1 / max(n1, 1), n1)
    print(os.path.basename(os.environ.get("NUSLAM_HIP_LIB", "product")), os.environ.get("NUSLAM_FORCE_WAVES", ""), line)


if __name__ == "__main__":
    main()
