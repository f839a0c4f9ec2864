#!/usr/bin/env python3
"""Per-correction time of k_tick_panels' two roles (wave 0 of workgroup 1).  NUSLAM_HIP_LIB = a -DNUSLAM_CHAIN_CLOCK build
(make -C shermbot-navigation_amd chainclock)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "shermbot-navigation_amd"))
import nuslam_hip as nh
from nuslam_hip import synth
n, m = 1000, 16
tr = synth.make_trace(n, 40, m)
bx, by, wid = synth.warmup_observations(tr.landmarks)
ekf = nh.EKF(np.zeros(3), np.zeros(2 * n), synth.Q_DEFAULT, synth.R_DEFAULT)
ekf.tick(np.zeros(3), bx, by, known_ids=wid, want_ids=False)
bt = ekf.as_batch(); bt.set_tick_mode(1); bt.set_overlap(False)
bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, tr.ids, bcast=True)
bt.run(0, 30); bt.sync()
L = nh.lib(); L.nuslam_debug_panels_clock.argtypes = [C.POINTER(C.c_longlong)]
acc = []
for t in range(30, 40):
    bt.run(t, t + 1); bt.sync()
    out = (C.c_longlong * 40)(); L.nuslam_debug_panels_clock(out); acc.append(list(out))
a = np.median(np.array(acc, dtype=np.float64), axis=0).reshape(2, 20) * 0.01
for r in range(2):
    if r == 0:
        print("role 0 prologue split: kernel entry -> before the plan loads %.2f us, plan staging issued %.2f us, gathers issued %.2f us, barrier passed %.2f us" % (a[0, 17], a[0, 18], a[0, 19], a[0, 0]))
    print("role %d: prologue %.2f us; per correction " % (r, a[r, 0]) + " ".join("%.2f" % x for x in a[r, 2:17]) + " | first stamp %.2f" % a[r, 1])
