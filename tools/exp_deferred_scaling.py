#!/usr/bin/env python3
"""GPU experiment: k_update_deferred / k_flush time versus the number of pending factors (corrections per tick)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "shermbot-navigation_amd"))
import numpy as np
import nuslam_hip as nh
from nuslam_hip import synth

n = 1000
lm = synth.make_landmarks(n)
bx, by, wid = synth.warmup_observations(lm)
for m in (1, 2, 4, 8, 16):
    T = 40
    tr = synth.make_trace(n, T, m, landmarks=lm)
    ekf = nh.EKF(np.zeros(3), np.zeros(2 * n), synth.Q_DEFAULT, synth.R_DEFAULT)
    ekf.tick(np.zeros(3), bx, by, known_ids=wid, want_ids=False)
    bt = ekf.as_batch()
    bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, tr.ids, bcast=True)
    bt.set_deferred(True)
    bt.run(0, 5); bt.sync()
    bt.profile(True)
    bt.run(5, T); bt.sync()
    u, un = bt.profile_read(nh.K_UPDATE_DEFERRED)
    f, fn = bt.profile_read(nh.K_FLUSH)
    p, pn = bt.profile_read(nh.K_PREDICT)
    bt.profile(False)
    print("m=%2d: update_deferred %.2f us (avg over J=0..%d), flush %.2f us (J=%d), predict %.2f us"
          % (m, 1e3 * u / un, m - 1, 1e3 * f / fn, m, 1e3 * p / pn))
