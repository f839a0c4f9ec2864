#!/usr/bin/env python3
"""Where does a k_update2 launch spend its time?  Needs the debug variant of the library:
  make -C shermbot-navigation_amd phase          (compiles csrc/nuslam_hip.hip with -DNUSLAM_PHASE_CLOCK)
  NUSLAM_HIP_LIB=shermbot-navigation_amd/build/variants/phase.so python tools/exp_phase_clock.py
Wave 0 of one mid-grid workgroup stamps the 100 MHz wall clock at every phase boundary; printed are the medians over
many launches of the time since kernel entry (of that workgroup), in microseconds."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "shermbot-navigation_amd"))
import nuslam_hip as nh  # noqa: E402
from nuslam_hip import synth  # noqa: E402

NAMES = ["entry", "block+state arrived (wave 0)", "head 1 done", "phase A done", "barrier 1 passed", "phase B done (head 2)",
         "barrier 2 passed", "rows done (M1, P1 cols, M2)", "strips done", "sweep issued", "stores acknowledged", "before the first vector load", "load burst issued"]


def nh_ld(n):
    L = 3 + 2 * n
    return (L + 31) // 32 * 32          # leading dimension (see nuslam_hip.hip: alloc_batch)


def main():
    n, m = int(os.environ.get("N", 1000)), 16
    L = nh.lib()
    L.nuslam_debug_phase.argtypes = [C.POINTER(C.c_longlong)]
    tr = synth.make_trace(n, 60, m)
    bx, by, wid = synth.warmup_observations(tr.landmarks)
    ekf = nh.EKF(np.zeros(3), np.zeros(2 * n), synth.Q_DEFAULT, synth.R_DEFAULT)
    ekf.tick(np.zeros(3), bx, by, known_ids=wid, want_ids=False)
    ekf.sync()
    rows = []
    L.nuslam_debug_wg.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    W = int(os.environ.get('K2_WAVES', 8))
    gx, gy = (nh_ld(n) + 127) // 128, ((3 + 2 * n + 15) // 16 + W - 1) // W
    spans, resid, exits = [], [], []
    for t in range(60):
        # one pair at a time so that the stamps belong to a known launch
        for i in range(0, m, 2):
            ekf.tick(np.zeros(3) if i else tr.tw[t], tr.mx[t, i:i + 2], tr.my[t, i:i + 2], known_ids=tr.ids[t, i:i + 2],
                     want_ids=False)
            ekf.sync()
            out = (C.c_longlong * 32)()
            L.nuslam_debug_phase(out)
            rows.append([out[k] - out[0] for k in range(13)] + [out[16 + k] - out[0] for k in range(13)])
            wg = (C.c_longlong * (2 * gx * gy))()
            L.nuslam_debug_wg(wg, gx * gy)
            a = np.array(wg[:], dtype=np.int64).reshape(-1, 2)
            t0 = a[:, 0].min()
            resid.append(a[:, 1] - a[:, 0]); exits.append(a[:, 1] - t0)
            spans.append([(a[:, 0] - t0).max(), np.median(a[:, 0] - t0), (a[:, 1] - t0).min(), np.median(a[:, 1] - t0),
                          (a[:, 1] - t0).max(), np.median(a[:, 1] - a[:, 0]), (a[:, 1] - a[:, 0]).max()])
    med = np.median(np.array(rows[40:], dtype=np.float64), axis=0) * 0.01      # 100 MHz ticks -> us
    print("workgroup (3,10) | workgroup (3,26), both relative to the entry of (3,10)")
    for k, name in enumerate(NAMES):
        print("%6.2f us | %6.2f us  %s" % (med[k], med[13 + k], name))
    res = np.median(np.array(resid[40:], dtype=np.float64), axis=0).reshape(gy, gx) * 0.01
    ext = np.median(np.array(exits[40:], dtype=np.float64), axis=0).reshape(gy, gx) * 0.01
    np.set_printoptions(linewidth=250, precision=1, suppress=True)
    print("exit time by workgroup (rows: blockIdx.y = column-strip group, cols: blockIdx.x = row block), us:")
    print(ext)
    lin = np.arange(gx * gy).reshape(gy, gx)
    print("median exit by XCD (linear id % 8):", [round(float(np.median(ext[lin % 8 == k])), 2) for k in range(8)])
    L.nuslam_debug_hwid.argtypes = [C.POINTER(C.c_uint), C.c_int]
    hw = (C.c_uint * (16 * gx * gy))()
    L.nuslam_debug_hwid(hw, gx * gy)
    hw = np.array(hw[:], dtype=np.uint32).reshape(gy * gx, 8, 2)[:, :W]
    print("placement of the last launch: workgroup -> (xcc, se, cu) and the SIMD of each of its four waves")
    place = {}
    for wg in list(range(0, 4)) + list(range(gx * gy // 2, gx * gy // 2 + 4)):
        ids = hw[wg, :, 0]
        print("  wg %3d: xcc %s se %s cu %s simd %s wave-slot %s" % (wg, hw[wg, :, 1] & 0xf, (ids >> 13) & 7, (ids >> 8) & 15,
                                                                   (ids >> 4) & 3, ids & 15))
    for wg in range(gx * gy):
        key = (int(hw[wg, 0, 1] & 0xf), int((hw[wg, 0, 0] >> 13) & 7), int((hw[wg, 0, 0] >> 8) & 15))
        place.setdefault(key, []).append(wg)
    sizes = np.bincount([len(v) for v in place.values()])
    print("  CUs in use: %d; workgroups per CU histogram: %s" % (len(place), list(enumerate(sizes))))
    ex = ext.ravel()
    both = [v for v in place.values() if len(v) == 2]
    if both:
        print("  CUs with two workgroups: median exit of the earlier-id one %.2f, of the later-id one %.2f us" % (
            np.median([ex[min(v)] for v in both]), np.median([ex[max(v)] for v in both])))
    sp = np.median(np.array(spans[40:], dtype=np.float64), axis=0) * 0.01
    print("all %d workgroups, relative to the first entry: last entry %.2f us (median %.2f); exits first %.2f / median "
          "%.2f / last %.2f us; residence median %.2f, max %.2f us" % ((gx * gy,) + tuple(sp)))


if __name__ == "__main__":
    main()
