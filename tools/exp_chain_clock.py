#!/usr/bin/env python3
"""Where does a k_tick_chain step spend its time?  NUSLAM_HIP_LIB = a -DNUSLAM_CHAIN_CLOCK build."""
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
bt = ekf.as_batch()
if os.environ.get("OVERLAP"):
    bt.set_overlap(True)
if os.environ.get("CHAIN_ONLY"):
    bt.set_tick_mode(3)        # k_tick_chain as a launch of its own (plain plan stores) instead of k_tick_front's workgroup 0
bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, tr.ids, bcast=True)
bt.run(0, 30); bt.sync()
L = nh.lib(); L.nuslam_debug_chain_clock.argtypes = [C.POINTER(C.c_longlong)]
acc = []
tls = []
L.nuslam_debug_front_timeline.argtypes = [C.POINTER(C.c_longlong)]
for t in range(30, 40, 2 if os.environ.get("OVERLAP") else 1):
    bt.run(t, t + (2 if os.environ.get("OVERLAP") else 1)); bt.sync()
    out = (C.c_longlong * 32)(); L.nuslam_debug_chain_clock(out); acc.append(list(out))
    tl = (C.c_longlong * 16)(); L.nuslam_debug_front_timeline(tl); tls.append(list(tl))
raw = np.median(np.array(acc, dtype=np.float64), axis=0).reshape(4, 8)
print("shader clock over the loop: %.0f MHz (%d cycles in %.2f us)" % (100.0 * raw[3, 2] / raw[3, 3], raw[3, 2], raw[3, 3] * 0.01))
raw[3, 2] = raw[3, 3] = 0
a = raw * 0.01 / m   # us per step
# stamps inside the one phase of a correction (ekf_tick.h, CK(k)): waves 1, 2: [2] plan stores (wave 1), [3] gain rows at set_s,
# [4] wave 2: the 25 entries / wave 1: landmark offset in polar form, [5] broadcasts + Jacobian, [6] S, [0] rest up to the barrier
names = ["rest", "barrier", "stores", "rows", "entries|polar", "bcast+H", "S", "loop top"]
for w in range(4):
    print("wave %d: " % w + ", ".join("%s %.2f" % (names[k], a[w, k]) for k in (7, 2, 3, 4, 5, 6, 0, 1)) + "  | sum %.2f us/step" % a[w].sum())

if not os.environ.get("CHAIN_ONLY"):
    # the launch's timeline, us from the chain's entry (median over the ticks)
    t = np.array(tls, dtype=np.float64)
    t = (t - t[:, :1]) * 0.01
    med = np.median(t, axis=0)
    print("k_tick_front timeline (us after the chain workgroup's entry): chain loop start %.2f, loop end %.2f, exit %.2f | predict workgroup %.2f .. %.2f"
          " | strips: middle workgroup %.2f .. %.2f, last workgroup %.2f .. %.2f" % tuple(med[1:10]))
    print("   chain prologue: block gathered and predict applied %.2f | overlapped runs: previous strips complete %.2f, replayed on the block %.2f (plan scalars %.2f, strips loaded %.2f, replay %.2f)" % (med[10], med[13], med[14], med[10], med[11], med[12]))
    if med[11] != 0:
        print("   fused pass (a middle workgroup): entry %.2f, first k-step may go %.2f, last k-step may go %.2f, stores issued %.2f" % tuple(med[11:15]))
