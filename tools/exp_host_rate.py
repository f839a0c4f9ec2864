import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shermbot-navigation_amd"))
import nuslam_hip as nh
from nuslam_hip import synth
n, m, T = 1000, 16, 440
lm = synth.make_landmarks(n)
tr = synth.make_wellposed_trace(n, T, m, landmarks=lm)
bx, by, wid = synth.warmup_observations(lm)
for overlap in (False, True):
    ekf = nh.EKF(np.zeros(3), np.zeros(2 * n), synth.Q_DEFAULT, synth.R_DEFAULT)
    ekf.tick(np.zeros(3), bx, by, known_ids=wid, want_ids=False)
    bt = ekf.as_batch(); bt.set_overlap(overlap)
    bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, tr.ids, bcast=True)
    bt.run(0, 20); bt.sync()
    for rep in range(2):
        t0 = time.perf_counter(); bt.run(20 + 200 * rep, 220 + 200 * rep); t1 = time.perf_counter(); bt.sync(); t2 = time.perf_counter()
        print("overlap=%s: enqueue of 200 ticks %.2f ms (%.1f us/tick), until done %.2f ms (%.1f us/tick)" % (overlap, (t1 - t0) * 1e3, (t1 - t0) * 5e3, (t2 - t0) * 1e3, (t2 - t0) * 5e3))
