#!/usr/bin/env python3
"""Would the chain / strips of one half of a batch beside the pass of the other half pay?  Two handles of 512 filters (N = 200), each
on its own stream, ticks enqueued alternately without waiting, against one handle of 1024 -- wall time per tick of the whole job."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "shermbot-navigation_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import nuslam_hip as nh
from nuslam_hip import synth
import _oracle as O

n, m, T = 200, 16, 140
tr = synth.make_wellposed_trace(n, 4, m, seed=12345)
bx, by, wid = synth.warmup_observations(tr.landmarks, seed=12345)
sim = nh.SimParams(marker_sigma=float(np.sqrt(1e-3)), max_range=0.0, fov=synth.FOV_DEFAULT, min_range=synth.MIN_RANGE_DEFAULT)
rng = np.random.default_rng(17)
uL, uR = 0.30 * 50, 0.36 * 50
cmd = np.zeros((T, 2)); cmd[:, 0] = (synth.WHEEL_RADIUS / synth.WHEEL_BASE) * (uR - uL); cmd[:, 1] = (synth.WHEEL_RADIUS / 2) * (uL + uR)


def make(B, first):
    bt = nh.Batch(B, n, synth.Q_DEFAULT, synth.R_DEFAULT)
    for i0 in range(0, len(wid), m):
        bt.load_trace(np.zeros((1, 2)), bx[None, i0:i0 + m], by[None, i0:i0 + m], wid[None, i0:i0 + m], bcast=True)
        bt.run(0, 1)
    bt.simulate(sim, synth.make_landmarks(n, 12345), cmd, m, 12345, first_filter=first, known_ids=True)
    bt.run(0, 10); bt.sync()
    return bt


def timed(handles, t0, t1, step):
    for h in handles:
        h.sync()
    s = time.perf_counter()
    for t in range(t0, t1, step):
        for h in handles:
            h.run(t, t + step)
    for h in handles:
        nh.lib().nuslam_batch_sync(h._h)
    return (time.perf_counter() - s) / (t1 - t0)


one = make(1024, 0)
groups = {g: [make(1024 // g, k * (1024 // g)) for k in range(g)] for g in (2, 4, 8)}
for step in (1, 2):
    r1 = min(timed([one], 10 + 40 * k, 50 + 40 * k, step) for k in range(3))
    line = "ticks enqueued %d at a time: one handle of 1024: %.1f us per tick (%.2f M updates/s)" % (step, 1e6 * r1, 1024 * m / r1 / 1e6)
    for g, hs in groups.items():
        r2 = min(timed(hs, 10 + 40 * k, 50 + 40 * k, step) for k in range(3))
        line += "; %d handles of %d on %d streams: %.1f us (%.2f M)" % (g, 1024 // g, g, 1e6 * r2, 1024 * m / r2 / 1e6)
    print(line)
