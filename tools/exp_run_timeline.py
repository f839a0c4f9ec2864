#!/usr/bin/env python3
"""The hand-off between two ticks inside k_run_fused (NUSLAM_HIP_LIB = a chainclock build): absolute 100 MHz stamps of the launch's last
chain tick and of a middle pass workgroup in the tick before it."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "shermbot-navigation_amd"))
import nuslam_hip as nh
from nuslam_hip import synth
n, m, T = 1000, 16, 120
lm = synth.make_landmarks(n)
tr = synth.make_wellposed_trace(n, T, m, landmarks=lm)
bx, by, wid = synth.warmup_observations(lm)
ekf = nh.EKF(np.zeros(3), np.zeros(2 * n), synth.Q_DEFAULT, synth.R_DEFAULT)
for i0 in range(0, len(wid), 16):
    ekf.tick(np.zeros(3), bx[i0:i0 + 16], by[i0:i0 + 16], known_ids=wid[i0:i0 + 16], want_ids=False)
bt = ekf.as_batch()
bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, tr.ids, bcast=True)
bt.run(0, 20); bt.sync()
L = nh.lib(); L.nuslam_debug_front_timeline.argtypes = [C.POINTER(C.c_longlong)]
rows = []
for r in range(8):
    L.nuslam_debug_front_timeline((C.c_longlong * 32)())
    bt.run(20 + 10 * r, 30 + 10 * r); bt.sync()
    tl = (C.c_longlong * 32)(); L.nuslam_debug_front_timeline(tl); rows.append(list(tl))
t = np.array(rows, dtype=np.float64) * 0.01
# reference: the chain's entry into the launch's LAST tick (= its exit from the tick before)
rel = t - t[:, 0:1]
med = np.median(rel, axis=0)
print("us relative to the chain's entry into the last tick (= the end of the tick before):")
print("  pass workgroup 100, tick before: last unit announced %.2f, k-steps done %.2f, exports issued %.2f, acknowledged %.2f, counted %.2f" % tuple(med[17:22]))
print("  the last to say so, tick before: the chain's entries stored -- interior pass workgroup %.2f, edge pass workgroup %.2f; state stored -- strip workgroup %.2f; every export stored -- pass workgroup %.2f" % (med[22], med[24], med[23], med[25]))
print("  the last through its k-steps, tick before: interior pass workgroup %.2f, edge pass workgroup %.2f" % (med[27], med[26]))
print("  chain, last tick: previous exports complete %.2f, block gathered %.2f, loop start %.2f, loop end %.2f, exit %.2f" % (med[16], med[10], med[1], med[2], med[3]))
