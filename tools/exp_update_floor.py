#!/usr/bin/env python3
"""GPU experiment: duration of k_update on its normal path vs its skip path (pure ping-pong copy of P)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "shermbot-navigation_amd"))
import numpy as np
import nuslam_hip as nh
from nuslam_hip import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
m, T = 16, 60
tr = synth.make_trace(n, T, m)
bx, by, wid = synth.warmup_observations(tr.landmarks)
ekf = nh.EKF(np.zeros(3), np.zeros(2 * n), synth.Q_DEFAULT, synth.R_DEFAULT)
ekf.tick(np.zeros(3), bx, by, known_ids=wid, want_ids=False)
bt = ekf.as_batch()
for name, ids in (("update", tr.ids), ("skip-copy", -np.ones_like(tr.ids))):
    bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, ids, bcast=True)
    bt.run(0, 10); bt.sync()
    bt.profile(True)
    bt.run(10, T); bt.sync()
    ms, cnt = bt.profile_read(nh.K_UPDATE)
    pms, pc = bt.profile_read(nh.K_PREDICT)
    bt.profile(False)
    L = 3 + 2 * n
    print("%-10s k_update avg %.2f us (%d launches) -> %.0f GB/s algorithmic; predict %.2f us" %
          (name, 1e3 * ms / cnt, cnt, 2.0 * L * L * 8 / (1e-3 * ms / cnt) / 1e9, 1e3 * pms / max(pc, 1)))
