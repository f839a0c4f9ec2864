#!/usr/bin/env python3
"""Period of one tick inside k_run_fused (nuslam_batch_run on one filter as one launch): wall time of a 200-tick run / 200.
For measurement builds whose results are wrong by construction (NUSLAM_RUN_EXP): the status the run leaves is ignored."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "shermbot-navigation_amd"))
import nuslam_hip as hip
from nuslam_hip import synth

n, m, T = 1000, 16, 420
lm = synth.make_landmarks(n)
tr = synth.make_wellposed_trace(n, T, m, landmarks=lm)
bx, by, wid = synth.warmup_observations(lm)
for mode in (1, 5):
    f = hip.EKF(np.zeros(3), np.zeros(2 * n), synth.Q_DEFAULT, synth.R_DEFAULT)
    bt = f.as_batch()
    bt.set_tick_mode(mode)
    for i0 in range(0, len(wid), 16):
        f.tick(np.zeros(3), bx[i0:i0 + 16], by[i0:i0 + 16], known_ids=wid[i0:i0 + 16], want_ids=False)
    bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, tr.ids, bcast=True)
    bt.run(0, 20)
    hip.lib().nuslam_batch_sync(bt._h)
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        bt.run(20 + 100 * rep, 120 + 100 * rep)
        hip.lib().nuslam_batch_sync(bt._h)
        best = min(best, (time.perf_counter() - t0) / 100)
    print("tick mode %d: %.2f us per tick (best of 4 runs of 100 ticks)" % (mode, 1e6 * best))
