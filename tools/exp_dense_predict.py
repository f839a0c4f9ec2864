#!/usr/bin/env python3
"""GPU experiment: time of nuslam_ekf_predict_dense (two MFMA GEMMs) and its flop rate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "shermbot-navigation_amd"))
import numpy as np
import nuslam_hip as nh
from nuslam_hip import synth

for n, dtype, name, peak in ((1000, nh.F64, "f64", 78.6), (5000, nh.F32, "f32", 157.3)):
    L = 3 + 2 * n
    rng = np.random.default_rng(1)
    F = np.eye(L) + 0.01 * rng.standard_normal((L, L)) / np.sqrt(L)
    ekf = nh.EKF(np.zeros(3), np.zeros(2 * n), synth.Q_DEFAULT, synth.R_DEFAULT, dtype=dtype)
    P0 = np.eye(L) * 0.1
    ekf.restore(np.zeros(L), P0, n)
    bt = ekf.as_batch()
    ekf.predict_dense(F); ekf.sync()
    bt.profile(True)
    reps = 3
    for _ in range(reps):
        ekf.predict_dense(F)
    ekf.sync()
    ms, cnt = bt.profile_read(nh.K_DENSE_GEMM)
    bt.profile(False)
    per_predict = ms / reps
    print("dense predict N=%d %s: %.3f ms per predict (2 GEMMs, %d launches), %.1f TFLOP/s = %.1f%% of %.1f"
          % (n, name, per_predict, cnt, 4.0 * L ** 3 / (per_predict * 1e-3) / 1e12,
             100 * 4.0 * L ** 3 / (per_predict * 1e-3) / 1e12 / peak, peak))
