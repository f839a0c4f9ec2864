#!/usr/bin/env python3
"""Where does one correction of the resident unknown-association round (k_da_round) spend its time?
NUSLAM_HIP_LIB = a -DNUSLAM_DA_CLOCK build (make -C shermbot-navigation_amd daclock)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "shermbot-navigation_amd"))
import nuslam_hip as nh
from nuslam_hip import synth
n, n_world, m = 1000, 998, 16
Qs = np.diag([1e-4, 1e-4, 1e-4])
lm = synth.make_landmarks(n_world)
tr = synth.make_trace(n_world, 40, m, landmarks=lm, noise_sigma=1e-4)
bx, by, wid = synth.warmup_observations(lm, noise_sigma=1e-4)
ekf = nh.EKF(np.zeros(3), np.zeros(2 * n), Qs, synth.R_DEFAULT)
ekf.tick(np.zeros(3), bx, by, known_ids=wid, want_ids=False)
bt = ekf.as_batch(); bt.set_tick_mode(1)
bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, None, bcast=True)
bt.run(0, 30); bt.sync()
L = nh.lib(); L.nuslam_debug_da_clock.argtypes = [C.POINTER(C.c_longlong)]
acc = []
for t in range(30, 40):
    bt.run(t, t + 1); bt.sync()
    out = (C.c_longlong * 64)(); L.nuslam_debug_da_clock(out); acc.append(list(out))
a = np.median(np.array(acc, dtype=np.float64), axis=0).reshape(4, 16) * 0.01 / m   # us per correction
names = ["loop top", "keys+decision", "loads+head", "barrier 1", "replay/gain/state", "barrier 2", "TC/TD", "barrier 3",
         "candidates", "meet"]
for w in range(4):
    print("wave %d: " % w + ", ".join("%s %.2f" % (names[k], a[w, k]) for k in range(10)) + "  | sum %.2f us/correction" % a[w].sum())
