"""How deep can a 1e-6 parity contract hold at all?  A property of the reference's ALGORITHM on a given input trace,
measured with the oracle against itself: the same trace with one tick's markers moved by one ulp.

  * the all-around trace (every tick sees the m nearest landmarks, behind the robot too): the reference subtracts the two
    bearings without wrapping (slam_library.cpp:272), a landmark behind the robot gives a ~2 pi innovation, the pose
    estimate jumps by a radian per tick and never tracks the truth, and the two runs lose all common digits;
  * a 'straight' tick whose dth comes out as 4.6e-17 instead of 0 takes the arc branch (slam_library.cpp:77: an exact
    compare) and amplifies a 1e-13 difference in the heading to 1e-4 in one predict;
  * the well-posed trace (synth.make_wellposed_trace: limited field of view, exactly representable wheel increments):
    the filter tracks the truth and the two runs stay within 1e-8 -- the regime in which tests/test_gpu_depth.py holds
    the GPU paths to 1e-6 over 200 ticks.
CPU only (the oracle is the subject here, not the checker of anything)."""
import numpy as np

import _oracle as O
from nuslam_hip import synth

Q, R = synth.Q_DEFAULT, synth.R_DEFAULT


def rel(a, ref):
    return float((np.abs(a - ref) / np.maximum(np.abs(ref), 1e-12 * np.abs(ref).max())).max())


def twin_run(tr, lm, T, perturb_tick=0):
    n = lm.shape[0]
    bx, by, wid = synth.warmup_observations(lm)
    o = O.OracleEKF(np.zeros(3), np.zeros(2 * n), Q, R, O.ORC_STRUCTURED)
    o.tick(tw=np.zeros(3), mx=bx, my=by, known_ids=wid)
    p = O.OracleEKF(np.zeros(3), np.zeros(2 * n), Q, R, O.ORC_STRUCTURED)
    p.restore(o.state.copy(), o.cov.copy(), o.seen)
    mx = tr.mx.copy()
    mx[perturb_tick] = np.nextafter(mx[perturb_tick], np.inf)          # one ulp on every marker of one tick
    hist, pose_err = [], []
    for t in range(T):
        o.tick(tw=tr.tw[t], mx=tr.mx[t], my=tr.my[t], known_ids=tr.ids[t])
        p.tick(tw=tr.tw[t], mx=mx[t], my=tr.my[t], known_ids=tr.ids[t])
        hist.append((rel(p.state, o.state), rel(p.cov, o.cov)))
        e = o.state[:3] - tr.truth[t]
        pose_err.append(float(np.abs(np.array([np.arctan2(np.sin(e[0]), np.cos(e[0])), e[1], e[2]])).max()))
    return np.array(hist), np.array(pose_err)


def test_all_around_trace_loses_parity_with_itself():
    n, m, T = 60, 16, 120
    lm = synth.make_landmarks(n)
    tr = synth.make_trace(n, T, m, landmarks=lm)
    hist, pose_err = twin_run(tr, lm, T)
    r, b = tr.polar()
    print("all-around trace: oracle vs oracle + 1 ulp: state %.1e / %.1e / %.1e at ticks 10 / 50 / %d; pose error up to %.2f"
          % (hist[9, 0], hist[49, 0], hist[-1, 0], T, pose_err.max()))
    assert np.abs(b).max() > 3.0                      # landmarks behind the robot are in the trace
    assert pose_err.max() > 0.5                       # the reference filter itself does not track the truth there
    assert hist[:, 0].max() > 1e-6                    # and two runs one ulp apart leave the 1e-6 contract


def test_near_zero_dth_is_an_amplifier():
    """The plain trace's 'straight' tick 50 has dth = -3.7e-16 (accumulated wheel angles round), not 0: predict() takes the
    arc branch (slam_library.cpp:77-86) with r = dx / dth ~ 3e13 and forms -r sin(th) + r sin(th + dth), two products of
    ~1e13 whose ulp is 2^-9 m: the displacement is a multiple of that, and which one is decided by the last bits of the
    heading.  Headings ONE ulp apart must be found whose predicted positions differ by millimetres."""
    tr = synth.make_trace(60, 50, 16)
    dth, dx = float(tr.tw[49, 0]), float(tr.tw[49, 1])
    assert dth != 0.0 and abs(dth) < 1e-15
    pos, th = [], 0.5
    for _ in range(64):
        o = O.OracleEKF(np.array([th, 0.0, 0.0]), np.zeros(2), Q, R, O.ORC_STRUCTURED)
        o.predict(dth, dx)
        pos.append(o.state[1:3].copy())
        th = float(np.nextafter(th, 1.0))
    jumps = [float(np.abs(pos[k + 1] - pos[k]).max()) for k in range(63)]
    print("dth = %.2e: one ulp of heading moves the predicted position by up to %.3e m (true step %.4f m)" % (dth, max(jumps), dx))
    assert max(jumps) > 1e-3


def test_wellposed_trace_keeps_two_runs_together():
    n, m, T = 60, 16, 120
    lm = synth.make_landmarks(n)
    tr = synth.make_wellposed_trace(n, T, m, landmarks=lm)
    assert (tr.tw[24::25, 0] == 0.0).all() and (tr.tw[:24, 0] != 0.0).all()       # straight ticks are EXACTLY straight
    r, b = tr.polar()
    assert np.abs(b).max() < synth.FOV_DEFAULT + 0.5                               # clear of the +-pi cut, noise included
    hist, pose_err = twin_run(tr, lm, T)
    print("well-posed trace: oracle vs oracle + 1 ulp: worst state %.1e cov %.1e over %d ticks; pose error up to %.3f"
          % (hist[:, 0].max(), hist[:, 1].max(), T, pose_err.max()))
    assert pose_err.max() < 0.2                       # the filter tracks the truth
    assert hist[:, 0].max() < 1e-8 and hist[:, 1].max() < 1e-7
