#!/usr/bin/env python3
"""bench.py -- EKF-SLAM updates/s on MI355X (BASELINE.json metric), one JSON line on rank 0.

A "step" is one tick of the slam node's loop (nuslam/src/slam.cpp:250-319): 1 predict + m sequential
corrections on one batch of synthetic odometry + range-bearing input that is already resident in HBM.

Workloads
  ekf1000   (default; BASELINE configs[1]) one EKF per GPU, N = 1000 landmarks, fp64 covariance, m = 16,
            known association; the map is initialised (one update per landmark) before the timed region.
            With --gpus N every rank runs its own independent replica (Monte-Carlo trials): weak scaling.
  batch     (BASELINE configs[3]) 1024 independent EKFs of N = 200 landmarks split over the ranks in contiguous
            blocks (nuslam_hip.dist.shard; --filters F: F filters per GPU instead), one launch per kernel for a
            rank's whole block; no collective in the data path, one RCCL all-gather of the statistics vector.
  da1000    (BASELINE configs[4]) as ekf1000 with unknown data association (associateLandmark per marker).
  ekf5000   (BASELINE configs[2]) one EKF, N = 5000, fp32 covariance, every predict propagates P with a resident
            dense Jacobian on the matrix cores (F P F^T, 4 L^3 flop) -- the MFMA-bound configuration.
`value` = corrections (EKF updates) per second over all ranks.

Timing: W untimed warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + synchronize on both
sides and maxed over ranks.  One block at N = 1000 is only K x 0.1 ms, so the block is repeated (on fresh ticks of the
same resident trace) until the timed blocks add up to >= 100 ms; `ms_per_step` is the MEDIAN block / K and the spread is
reported beside it.

Launch: `python bench.py --gpus N` starts N ranks itself (fresh child processes, started before this process touches
the GPU); under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks already exist
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) and are used as they are.

Extra objects: "roofline" for the dominant kernel (duration from per-dispatch HIP events on the handle's stream) and
"cpu_baseline" (the reference's dense algebra timed on this host's cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "shermbot-navigation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
# the timed blocks must add up to at least this (ms), per workload: a 0.1 s region is thin beside a 20 s command (the driver's
# utilisation samples never saw it); the single-filter workloads time 2 s, the batch 1 s, the 33 ms steps of ekf5000 0.5 s
MIN_TIMED_MS = {"ekf1000": 2000.0, "da1000": 2000.0, "batch": 1000.0, "ekf5000": 500.0}
MIN_BLOCKS = 5          # ... and never fewer than this many K-step blocks (the reported figure is the median block)
MAX_BLOCKS = 5000
# rough step times (ms) only to size the resident trace (taken a little LOW: the trace must hold enough ticks even on a fast box);
# the number of blocks actually run is decided by the clock
STEP_MS_GUESS = {"ekf1000": 0.040, "da1000": 0.085, "batch": 0.75, "ekf5000": 30.0}


# what the input is (synth.make_wellposed_trace / the simulator's fov, min_range): NOT SURVEY 8(d)'s all-around "m nearest" trace, on
# which the reference algorithm itself is chaotic at the level of one ulp (tests/test_trace_conditioning.py)
TRACE_KIND = "wellposed: fov 2.0 rad, min_range 0.2 m, the m nearest landmarks inside it, wheel increments dL 5/16, dR 3/8 rad per tick"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="ekf1000", choices=["ekf1000", "batch", "da1000", "ekf5000"])
    ap.add_argument("--landmarks", type=int, default=None)
    ap.add_argument("--filters", type=int, default=None, help="batch workload: filters PER GPU (weak scaling); default: "
                                                              "--filters-total split over the ranks (strong scaling)")
    ap.add_argument("--filters-total", type=int, default=1024, help="batch workload: filters over all ranks")
    ap.add_argument("--m", type=int, default=16, help="corrections per tick")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--deferred", action="store_true",
                    help="opt-in deferred application: a tick's corrections are kept as rank-2 factors and applied to P "
                         "once per tick (csrc/ekf_deferred.h); results agree with the default to rounding, not bitwise")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for barriers / max-over-ranks (nccl = RCCL over xGMI; gloo only to "
                         "rehearse the multi-rank path with several ranks on one GPU)")
    ap.add_argument("--trace", default=None, choices=["host", "device"],
                    help="batch workload: per-filter Monte-Carlo traces generated on the device by the simulator kernels "
                         "(default), or one host-made trace per rank replayed by every filter")
    ap.add_argument("--no-pairing", action="store_true", help="one k_update launch per correction (disable k_update2)")
    ap.add_argument("--tick-pipeline", action="store_true",
                    help="force the tick pipeline (chain + strips + ONE pass over P per tick) where the library's default "
                         "would pick the per-pair kernels (a single filter)")
    ap.add_argument("--tick-mode", type=int, default=None, choices=[0, 1, 2, 3, 4, 5],
                    help="nuslam_batch_set_tick_mode: 0 one pass over P per correction, 1 tick pipelines, 2 as 1 but unknown "
                         "association as one launch per marker instead of the resident round kernel")
    ap.add_argument("--pass-variant", type=int, default=None,
                    help="nuslam_batch_set_pass_variant: 0 (default) the rank-2m pass on the matrix cores, 1 / 2 the exact chain "
                         "(plain / two-unit kernel), 10 + k the rank-2m pass with tile shape k")
    ap.add_argument("--interleave", type=int, default=None, help="nuslam_batch_set_interleave: groups of filters on streams of their own (1..4; 10 + G: without the passes taking turns)")
    ap.add_argument("--plain-pass", action="store_true", help="nuslam_batch_set_pass_variant(1): the exact chain, plain kernel")
    ap.add_argument("--dense-random-f", action="store_true",
                    help="ekf5000: propagate with a fixed dense random Jacobian instead of the reference's A = I + B formed on the "
                         "device every tick -- a GEMM micro-measurement on fully dense operands (MFMA loops on mostly-zero "
                         "operands hold a higher clock), not the reference's predict")
    ap.add_argument("--parity-ticks", type=int, default=40, help="ticks of the same-run parity leg (through nuslam_batch_run)")
    ap.add_argument("--no-api", action="store_true", help="skip the api_driven leg (the C++ class driven call by call)")
    ap.add_argument("--no-overlap", action="store_true", help="tick pipeline on ONE stream (no chain running ahead)")
    ap.add_argument("--overlap", action="store_true", help="force the chain of tick t+1 onto a second stream (nuslam_batch_set_overlap; default: on for one filter, off for batches)")
    ap.add_argument("--per-correction", action="store_true",
                    help="round-1 path: one pass over P per correction / pair instead of the tick pipeline (same bits)")
    ap.add_argument("--min-timed-ms", type=float, default=None, help="the timed blocks add up to at least this (default per workload: MIN_TIMED_MS)")
    ap.add_argument("--blocks", type=int, default=0, help="run exactly this many timed K-step blocks (0: until --min-timed-ms)")
    ap.add_argument("--events-in-timed-region", action="store_true",
                    help="attach the per-dispatch HIP events inside the timed region itself (costs throughput: every "
                         "dispatch then carries a completion signal); default: one more block of K steps right after")
    ap.add_argument("--dump", default=None, help="batch workload: write every local filter's final state / seen and the "
                                                 "reduced statistics to this .npz (rank suffix added) -- for the sharding test")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` with no rank environment: start N fresh rank processes (this process has not touched
    the GPU and never will), wait, relay rank 0's line.  Children are started, never exec'd into."""
    import socket
    import torch
    ndev = torch.cuda.device_count()            # counts devices without initialising the GPU (safe before a spawn)
    if args.backend == "nccl" and ndev < args.gpus:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible (RCCL needs one GPU per rank; use --backend gloo "
                         "to rehearse several ranks on one GPU)\n" % (args.gpus, ndev))
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = rc or p.wait()
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return rc


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(n, m, tr, budget_s, warm_state):
    """The reference CPU path (dense algebra: two L^3 GEMMs per predict, one per update -- slam_library.cpp:104,279) on
    this host's usable cores, on the first ticks of the same trace from the same post-initialisation snapshot.  Two
    restatements are timed and the FASTER one is reported as the baseline:
      oracle-c    oracle/nuslam_oracle.c dense mode, OpenMP-blocked loops
      numpy-blas  tests/_np_ekf.py, the same chain through numpy's BLAS dgemm (OpenBLAS) -- what Armadillo itself
                  would dispatch to."""
    import numpy as np
    import _oracle as O
    import _np_ekf
    from nuslam_hip import synth
    cores = O.usable_cpus()
    cands = {}

    O.set_threads(cores)
    o = O.OracleEKF(np.zeros(3), np.zeros(2 * n), synth.Q_DEFAULT, synth.R_DEFAULT, O.ORC_DENSE)
    o.restore(*warm_state)
    done, t0 = 0, time.perf_counter()
    while done < tr.ticks and (done == 0 or time.perf_counter() - t0 < budget_s / 2):
        o.tick(tw=tr.tw[done], mx=tr.mx[done], my=tr.my[done], known_ids=tr.ids[done])
        done += 1
    dt = time.perf_counter() - t0
    cands["oracle-c"] = (done * m / dt, done, dt)
    O.set_threads(1)

    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=cores)
    except Exception:
        limiter = None
    e = _np_ekf.NpEKF(np.zeros(3), np.zeros(2 * n), synth.Q_DEFAULT, synth.R_DEFAULT)
    e.s[:] = warm_state[0]
    e.P[:, :] = warm_state[1]
    r, b = tr.polar()
    done, t0 = 0, time.perf_counter()
    while done < tr.ticks and (done == 0 or time.perf_counter() - t0 < budget_s / 2):
        e.predict(tr.tw[done][0], tr.tw[done][1])
        for i in range(m):
            e.update([r[done, i], b[done, i]], int(tr.ids[done, i]))
        done += 1
    dt = time.perf_counter() - t0
    cands["numpy-blas"] = (done * m / dt, done, dt)
    if limiter is not None and hasattr(limiter, "unregister"):
        limiter.unregister()

    # ---- SURVEY 8(d)'s other two CPU rows.  structured: the same O(L^2) sums the GPU computes (exact-zero terms of the dense
    # products skipped), all cores -- the fair fight.  dense_1thread: the reference algebra on ONE core through BLAS dgemm
    # (how Armadillo behaves on an install with reference BLAS); one tick is ~5 s there, so the sample is one tick.
    O.set_threads(cores)
    osx = O.OracleEKF(np.zeros(3), np.zeros(2 * n), synth.Q_DEFAULT, synth.R_DEFAULT, O.ORC_STRUCTURED)
    osx.restore(*warm_state)
    sd, t0 = 0, time.perf_counter()
    while sd < tr.ticks and (sd == 0 or time.perf_counter() - t0 < min(3.0, budget_s / 4)):
        osx.tick(tw=tr.tw[sd], mx=tr.mx[sd], my=tr.my[sd], known_ids=tr.ids[sd])
        sd += 1
    sdt = time.perf_counter() - t0
    O.set_threads(1)
    structured = {"value": sd * m / sdt, "unit": "updates/s", "cores": cores, "impl": "oracle-c structured mode, OpenMP",
                  "sample": "%d ticks, %.1f s" % (sd, sdt), "ms_per_step": 1e3 * sdt / sd}
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            e1 = _np_ekf.NpEKF(np.zeros(3), np.zeros(2 * n), synth.Q_DEFAULT, synth.R_DEFAULT)
            e1.s[:] = warm_state[0]
            e1.P[:, :] = warm_state[1]
            t0 = time.perf_counter()
            e1.predict(tr.tw[0][0], tr.tw[0][1])
            for i in range(m):
                e1.update([r[0, i], b[0, i]], int(tr.ids[0, i]))
            d1 = time.perf_counter() - t0
        dense_1thread = {"value": m / d1, "unit": "updates/s", "cores": 1, "impl": "numpy-blas, one thread",
                         "sample": "1 tick, %.1f s" % d1, "ms_per_step": 1e3 * d1}
    except Exception as ex:
        dense_1thread = {"value": None, "error": str(ex)[:200]}

    best = max(cands, key=lambda k: cands[k][0])
    v, done, dt = cands[best]
    return {"structured": structured, "dense_1thread": dense_1thread,"value": v, "unit": "updates/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port", "impl": best,
            "sample": "%d tick(s) (1 predict + %d updates each) of the same N=%d trace from the same post-initialisation "
                      "snapshot, reference algebra (two L^3 GEMMs per predict, one per update), %.1f s; faster of "
                      "{oracle-c: %.2f, numpy-blas: %.2f} updates/s" % (done, m, n, dt, cands["oracle-c"][0], cands["numpy-blas"][0]),
            "ms_per_step": 1e3 * dt / done}, o, cands["oracle-c"][1]


def api_driven(n, m, tr, bx, by, wid, Q, R, exe_name="api_rate"):
    """The rate the UNCHANGED slam node would see: slam_library::ExtendedKalman (the C++ host mirror,
    shermbot-navigation_amd/cpp/nuslam/slam_library.hpp) driven call by call as slam.cpp:250-319 drives it -- predict, then per
    marker update() (known ids) or associateLandmark() + update() (unknown ids: a synchronising call per marker) -- every
    call crossing the C ABI on its own.  Run as a child process (cpp/tests/api_rate) after the timed region."""
    import numpy as np
    exe = os.path.join(ROOT, "shermbot-navigation_amd", "cpp", "tests", exe_name)
    if not os.path.exists(exe):
        return {"error": "cpp/tests/%s is not built" % exe_name}
    res = {}
    ticks = min(tr.ticks, 256)
    for name, known in (("known_ids", 1), ("unknown_ids", 0)):
        if known:
            q, wx, wy, ww, t2 = float(Q[0, 0]), bx, by, wid, tr
        else:
            # association only matches when the innovation is far below sqrt(R): 1e-4 m marker noise, Q = 1e-4 (as da1000),
            # and one free slot in the map (associateLandmark indexes out of bounds on a full one, slam_library.cpp:206-207)
            from nuslam_hip import synth
            q = 1e-4
            t2 = synth.make_wellposed_trace(n - 1, ticks, m, seed=12345, noise_sigma=1e-4)
            wx, wy, ww = synth.warmup_observations(t2.landmarks, seed=12345, noise_sigma=1e-4)
        lines = ["%d %d %d %d %.17g %.17g" % (n, ticks, m, known, q, float(R[0, 0]))]
        pad = n - len(wx)
        for i in range(len(wx)):
            lines.append("%.17g %.17g %d" % (wx[i], wy[i], ww[i]))
        for i in range(pad):                      # (unknown ids: the map holds n - 1 landmarks; the last line repeats landmark 1)
            lines.append("%.17g %.17g %d" % (wx[0], wy[0], ww[0]))
        for t in range(ticks):
            lines.append("%.17g %.17g" % (t2.tw[t][0], t2.tw[t][1]))
            for i in range(m):
                lines.append("%.17g %.17g %d" % (t2.mx[t, i], t2.my[t, i], t2.ids[t, i]))
        try:
            p = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=300)
            res[name] = json.loads(p.stdout.strip().splitlines()[-1]) if p.stdout.strip() else {"error": p.stderr[-300:]}
        except Exception as ex:
            res[name] = {"error": str(ex)[:300]}
    res["note"] = ("slam_library::ExtendedKalman per call through the C ABI, the unchanged loop of slam.cpp:250-319 (predict + per marker "
                   "update; unknown ids: associateLandmark, which returns the id to the host, in front of each) -- the drop-in rate.  The "
                   "library records the calls and applies them tick by tick (lazy ticks: k_tick_fused; a round served to the host through "
                   "a mailbox in mapped memory for associateLandmark); `value` is the resident-trace entry point nuslam_batch_run")
    return res


def da_parity(nh, args, n, m, Q, R, dtype, dev, seed):
    """The da1000 line's same-run parity leg: associateLandmark + update per marker (slam_library.cpp:188-282 in the loop
    slam.cpp:279-318) for `--parity-ticks` ticks of the line's own kind of input, oracle (structured mode, marker by marker) against
    the GPU through nuslam_batch_run with the timed handle's settings; the ids come from a twin filter stepped with nuslam_ekf_tick
    (same kernels), which must equal the nuslam_batch_run filter bit for bit."""
    import numpy as np
    from nuslam_hip import synth
    O_ = __import__("_oracle")
    T = int(args.parity_ticks)
    n_world = n - 1
    tr = synth.make_wellposed_trace(n_world, T, m, seed=seed, noise_sigma=1e-4)
    bx, by, wid = synth.warmup_observations(tr.landmarks, seed=seed, noise_sigma=1e-4)
    O_.set_threads(O_.usable_cpus())
    o = O_.OracleEKF(np.zeros(3), np.zeros(2 * n), Q, R, O_.ORC_STRUCTURED)
    o.tick(tw=np.zeros(3), mx=bx, my=by, known_ids=wid)
    snap = (o.state.copy(), o.cov.copy(), o.seen)
    ids_o = np.zeros((T, m), dtype=np.int64)
    seen_o = np.zeros(T, dtype=np.int64)
    margin = np.inf
    for t in range(T):
        o.predict(tr.tw[t][0], tr.tw[t][1])
        cached = o.seen
        for i in range(m):
            z = O_.cartesian2polar(tr.mx[t, i], tr.my[t, i])
            k, dk = o.associate(z[0], z[1], want_d=True)
            dk = dk[~np.isnan(dk)]
            if dk.size:
                margin = min(margin, float(np.min(np.abs(dk - 0.01) / 0.01)), float(np.min(np.abs(dk - 60.0) / 60.0)))
            ids_o[t, i] = k
            if k > cached:
                o.init_landmark(z[0], z[1], k)
            elif k < 0:
                continue
            o.update(z[0], z[1], k)
        seen_o[t] = o.seen
    O_.set_threads(1)
    g = nh.EKF(np.zeros(3), np.zeros(2 * n), Q, R, dtype=dtype, device=dev)
    h = nh.EKF(np.zeros(3), np.zeros(2 * n), Q, R, dtype=dtype, device=dev)
    ids_equal = seen_equal = True
    for f in (g, h):
        f.restore(*snap)
        fb = f.as_batch()
        if args.tick_mode is not None:
            fb.set_tick_mode(args.tick_mode)
        if args.pass_variant is not None:
            fb.set_pass_variant(args.pass_variant)
    for t in range(T):
        idg = g.tick(tr.tw[t], tr.mx[t], tr.my[t])
        ids_equal = ids_equal and bool(np.array_equal(idg, ids_o[t]))
        seen_equal = seen_equal and g.seen == seen_o[t]
    hb = h.as_batch()
    hb.load_trace(tr.tw[:, :2], tr.mx, tr.my, None, bcast=True)
    hb.run(0, T)
    gs, gP, os_, oP = h.state, h.cov, o.state, o.cov
    floor = 1e-12 * np.abs(oP).max()
    return {"against": "oracle/nuslam_oracle.c, structured mode, driven marker by marker as slam.cpp:269-318 (EKF parity unpinned, see "
                       "DESIGN.md); input trace " + TRACE_KIND + "; marker noise 1e-4 m",
            "path": "nuslam_batch_run on a resident trace without ids (tick mode %s, pass variant %s); ids from a twin filter stepped with "
                    "nuslam_ekf_tick" % ("default" if args.tick_mode is None else args.tick_mode,
                                         "default" if args.pass_variant is None else args.pass_variant),
            "ticks": T, "associations": int(T * m), "ids_equal_every_tick": bool(ids_equal), "seen_equal_every_tick": bool(seen_equal),
            "matches": int((ids_o > 0).sum()), "gray_zone": int((ids_o < 0).sum()),
            "min_relative_margin_of_any_decision_distance_from_a_threshold": float(margin),
            "batch_run_equals_tick_by_tick_bitwise": bool(np.array_equal(gs, g.state) and np.array_equal(gP, g.cov)),
            "device_status": int(h.status()),
            "max_rel_err_state": float((np.abs(gs - os_) / np.maximum(np.abs(os_), 1e-12)).max()),
            "max_rel_err_cov": float((np.abs(gP - oP) / np.maximum(np.abs(oP), floor)).max()),
            "tolerance": 1e-6}


def pmc_traffic(nh, sweep_kernel_sig, min_bytes):
    """HBM bytes per launch from a committed rocprofv3 --pmc measurement (profiles/r*/**pmc_hbm*.json made by
    tools/summarize_pmc.py), quoted ONLY when that record was taken from the very kernel sources the loaded library
    was built from (nuslam_build_info), names the kernel this run's launches used and was taken at this run's size (same
    minimum bytes per launch); otherwise null."""
    import glob
    have = nh.build_info()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*pmc_hbm*.json")), reverse=True):
        try:
            rec = json.load(open(f))
        except Exception:
            continue
        if (rec.get("build_info") == have and sweep_kernel_sig in rec.get("kernel_name", "")
                and abs(rec.get("per_launch_bytes", {}).get("min_bytes", -1) - min_bytes) < 1):
            return rec["per_launch_bytes"]["hbm_traffic"], os.path.relpath(f, ROOT)
    return None, None


def main():
    args = parse()
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import numpy as np
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    ndev = torch.cuda.device_count()
    dev = (local_rank % max(ndev, 1)) if world > 1 else 0
    coll_dev = "cuda" if args.backend == "nccl" else None
    ranks_seen = dist.get_world_size() if world > 1 else 1

    import nuslam_hip as nh
    from nuslam_hip import synth
    from nuslam_hip import dist as nd

    if args.workload == "ekf5000":
        args.dtype = "f32"
        if args.steps == 200 and args.warmup == 20:
            args.steps, args.warmup = 5, 1          # a tick is ~35 ms here
    dtype = nh.F64 if args.dtype == "f64" else nh.F32
    w = 8 if dtype == nh.F64 else 4
    first_filter = 0
    if args.workload == "batch":
        n = args.landmarks or 200
        if args.filters:
            B, first_filter, filters_total = args.filters, rank * args.filters, world * args.filters
        else:
            first_filter, B = nd.shard(args.filters_total, world, rank)      # contiguous blocks, remainders included
            filters_total = args.filters_total
        if B < 1:
            raise SystemExit("bench.py: rank %d would own no filter (%d filters over %d ranks)" % (rank, filters_total, world))
    else:
        n = args.landmarks or (5000 if args.workload == "ekf5000" else 1000)
        B, filters_total = 1, world
    m = min(args.m, n)
    L = 3 + 2 * n
    K, W = args.steps, args.warmup
    known = args.workload != "da1000"

    # how many K-step blocks the resident trace must hold: enough for >= the timed target even if the code were several
    # times faster than today, plus one block for the kernel-event pass
    if args.min_timed_ms is None:
        args.min_timed_ms = MIN_TIMED_MS[args.workload]
    blocks_cap = int(min(MAX_BLOCKS, max(MIN_BLOCKS, np.ceil(args.min_timed_ms / (K * STEP_MS_GUESS[args.workload] * (B / 1024.0 if args.workload == "batch" else 1.0))))))
    if args.blocks > 0:
        blocks_cap = args.blocks
    ticks_total = W + K * (blocks_cap + 2)       # (+ the kernel-event pass, + the pass-on-its-own pass of a fused single-filter tick)

    # ---- synthetic input (seeded; Monte-Carlo replica r uses seed 12345 + r), made resident in HBM
    seed = 12345 if args.workload == "batch" else 12345 + rank      # the batch is ONE world split over the ranks
    # data association needs one free slot: associateLandmark writes a hypothetical landmark at index seen+1 and
    # indexes out of bounds on a full map (slam_library.cpp:206-207), so the world holds n-1 landmarks there
    n_world = n if known else n - 1
    trace_kind = args.trace or ("device" if args.workload == "batch" else "host")
    host_ticks = ticks_total if trace_kind == "host" else 4
    # well-posed input (synth.make_wellposed_trace): a +-2 rad field of view and exactly representable wheel increments keep
    # the REFERENCE algorithm in its working regime -- its un-wrapped bearing innovation (slam_library.cpp:272) and a
    # near-zero dth in the arc branch (:77) otherwise make the trajectory chaotic at the level of one ulp
    tr = synth.make_wellposed_trace(n_world, host_ticks, m, seed=seed, noise_sigma=None if known else 1e-4)
    # association only matches when the innovation is ~100x below sqrt(R) (threshold 0.01, slam_library.cpp:193,238),
    # so that workload measures with 1e-4 m marker noise, in the map-initialising pass too
    bx, by, wid = synth.warmup_observations(tr.landmarks, seed=seed, noise_sigma=None if known else 1e-4)
    Q, R = synth.Q_DEFAULT, synth.R_DEFAULT
    if not known:
        # With the node's Q = diag(0.1) every prediction inflates the pose covariance so much that an earlier-indexed
        # neighbour lands in the (0.01, 60) gray zone and associateLandmark returns -1 before reaching the true
        # landmark (slam_library.cpp:243-246): almost nothing is ever corrected.  The association workload uses the
        # well-conditioned Q = diag(1e-4) of SURVEY section 8d so that matches (and corrections) actually happen.
        Q = np.diag([1e-4, 1e-4, 1e-4])

    if B == 1 and args.workload != "batch":
        ekf = nh.EKF(np.zeros(3), np.zeros(2 * n), Q, R, dtype=dtype, device=dev)
        bt = ekf.as_batch()
        if args.per_correction or args.no_pairing or args.tick_mode == 0:
            bt.set_tick_mode(0)        # (before the map is initialised: pairing needs the host's mirror of `seen`, which
                                       # only the per-correction path maintains)
        ekf.tick(np.zeros(3), bx, by, known_ids=wid, want_ids=False)     # initialise the whole map (untimed)
        ekf.sync()
        warm_state = (ekf.state, ekf.cov, ekf.seen) if (rank == 0 and world == 1 and args.cpu_seconds > 0 and args.workload == "ekf1000") else None   # cpu_baseline: N=1 only
        if args.workload == "ekf5000" and args.dense_random_f:
            # a DENSE Jacobian kept resident in HBM: the reference's A = I + B for the first twist plus a small dense
            # random perturbation -- every operand non-zero, because MFMA loops on mostly-zero operands hold a higher
            # clock and would flatter the number (cdna_hip_programming.md section 5.4 rule 25)
            dth, dx = tr.tw[0][0], tr.tw[0][1]
            F = np.eye(L) + (1e-3 / np.sqrt(L)) * np.random.default_rng(seed).standard_normal((L, L))
            F[1, 0] += -(dx / dth) * np.cos(dth) + (dx / dth) * np.cos(2 * dth)
            F[2, 0] += -(dx / dth) * np.sin(dth) + (dx / dth) * np.sin(2 * dth)
            s0, P0, sn = ekf.state, ekf.cov, ekf.seen
            ekf.predict_dense(F)
            ekf.restore(s0, P0, sn)
            ekf.use_dense_predict(True)
            del F, P0
        elif args.workload == "ekf5000":
            # configs[2] as the reference runs it: every predict forms A = I + B(theta', twist) on the device
            # (slam_library.cpp:127-148) and propagates the covariance with it through the two dense MFMA products (:104)
            ekf.use_dense_predict(2)
    else:
        bt = nh.Batch(B, n, Q, R, dtype=dtype, device=dev)
        if args.interleave is not None:
            bt.set_interleave(args.interleave)
        # initialise every filter's map with one resident warm-up tick of n observations
        bt.load_trace(np.zeros((1, 2)), bx[None, :], by[None, :], wid[None, :], bcast=True)
        bt.run(0, 1)
        bt.sync()
        warm_state = None
    ids = tr.ids if known else None
    if trace_kind == "device":
        # every filter is its own Monte-Carlo trial -- its trace is generated in HBM by the simulator
        # kernels (tube_world.cpp:509-533 per filter) from the random streams of its GLOBAL filter index, so the
        # sharded run reproduces the unsharded one and nothing but the six-number parameter block crosses PCIe
        uL, uR = 0.30 * 50, 0.36 * 50                      # the wheel increments of synth.make_trace, per second
        cmd = np.zeros((ticks_total, 2))
        cmd[:, 0] = (synth.WHEEL_RADIUS / synth.WHEEL_BASE) * (uR - uL)
        cmd[:, 1] = (synth.WHEEL_RADIUS / 2) * (uL + uR)
        # (no straight ticks here: the joint angles accumulate, a commanded dth of 0 comes back from getTwist as ~1e-17 and
        # takes the arc branch, where one ulp of heading decides about centimetres -- tests/test_trace_conditioning.py)
        sim = nh.SimParams(marker_sigma=float(np.sqrt(1e-3)) if known else 1e-4, max_range=0.0, fov=synth.FOV_DEFAULT,
                           min_range=synth.MIN_RANGE_DEFAULT)
        world_lm = synth.make_landmarks(n_world, 12345)    # ONE world for all ranks: only the noise streams differ
        bt.simulate(sim, world_lm, cmd, m, 12345, first_filter=first_filter, known_ids=known)
    else:
        bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, ids, bcast=True)
    if args.deferred:
        bt.set_deferred(True)
    if args.per_correction or args.no_pairing:
        bt.set_tick_mode(0)
    elif args.tick_pipeline:
        bt.set_tick_mode(1)
    if args.tick_mode is not None:
        bt.set_tick_mode(args.tick_mode)
    if args.plain_pass:
        bt.set_pass_variant(1)
    if args.pass_variant is not None:
        bt.set_pass_variant(args.pass_variant)
    if args.interleave is not None:
        bt.set_interleave(args.interleave)
    if args.no_overlap:
        bt.set_overlap(False)
    if args.overlap:
        bt.set_overlap(True)
    if args.no_pairing:
        bt.set_pairing(False)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        bt.sync()

    bt.run(0, W)                       # W untimed warm-up steps
    barrier()
    in_region = args.events_in_timed_region
    bt.profile(in_region)
    block_s = []
    t_at = W
    while len(block_s) < blocks_cap:
        barrier()
        t0 = time.perf_counter()
        bt.run(t_at, t_at + K)         # EXACTLY K timed steps
        torch.cuda.synchronize()       # (every stream of the device: the handle's own included)
        if world > 1:
            dist.barrier()
        block_s.append(time.perf_counter() - t0)
        bt.sync()                      # the handle's status words (two small device-to-host copies): checked, not timed
        t_at += K
        if world == 1 and args.blocks == 0 and len(block_s) >= MIN_BLOCKS and sum(block_s) >= 1e-3 * args.min_timed_ms:
            break                      # (with several ranks every rank runs the same, precomputed number of blocks)
    if not in_region:
        # kernel durations: the next K steps of the same trace, every dispatch bracketed by its own HIP events
        # on the handle's stream (hipExtLaunchKernelGGL start/stop events)
        # (a large batch's timed ticks run as groups of filters on streams of their own, one group's pass beside the other's chain and
        # strips: the kernels' own durations are taken with the batch as ONE group, every kernel alone on the chip)
        grouped_batch = args.workload == "batch" and B >= 512 and args.interleave in (None, -1)
        if grouped_batch:
            bt.set_interleave(1)
        bt.profile(True)
        bt.run(t_at, t_at + K)
        bt.sync()
        t_at += K
        if grouped_batch:
            bt.set_interleave(-1)
    fused_launch = None
    if known and B == 1 and args.workload == "ekf1000" and args.tick_mode in (None, 1, 5) and not (args.per_correction or args.no_pairing
                                                                                                 or args.deferred or args.overlap):
        # The default single-filter tick is ONE launch (k_tick_fused: predict || chain || strips || the rank-2m pass as workgroups of
        # one grid), whose duration is the serial chain's.  The HBM-bound kernel's own figures come from K more steps of the same
        # trace with the pass as a launch of its own (tick mode 4: k_tick_front + k_tick_rank -- the same arithmetic, the same bits).
        fz_ms, fz_n = bt.profile_read(nh.K_TICK_CHAIN)
        rk_ms, rk_n = bt.profile_read(nh.K_TICK_RANK)
        if fz_n and not rk_n:
            per_launch = K // fz_n if fz_n < K else 1              # (a nuslam_batch_run of one filter is ONE launch: k_run_fused)
            fused_launch = {"kernel": ("k_run_fused (the run's ticks in one launch: predict || chain || strips || pass, P resident in the pass "
                                       "workgroups' registers between ticks, csrc/ekf_fused.h)" if per_launch > 1 else
                                       "k_tick_fused (predict || chain || strips || pass, csrc/ekf_fused.h)"),
                            "avg_launch_us": 1e3 * fz_ms / fz_n, "launches": fz_n, "ticks_per_launch": per_launch,
                            "avg_tick_us": 1e3 * fz_ms / fz_n / per_launch}
            bt.set_tick_mode(4)
            bt.profile(True)
            bt.run(t_at, t_at + K)
            bt.sync()
            t_at += K
            bt.set_tick_mode(1 if args.tick_mode is None else args.tick_mode)
    sweep_ms, sweep_n = bt.profile_read(nh.K_UPDATE)
    pair_ms, pair_n = bt.profile_read(nh.K_UPDATE2)
    pred_ms, pred_n = bt.profile_read(nh.K_PREDICT)
    asso_ms, asso_n = bt.profile_read(nh.K_ASSOCIATE)
    gemm_ms, gemm_n = bt.profile_read(nh.K_DENSE_GEMM)
    dupd_ms, dupd_n = bt.profile_read(nh.K_UPDATE_DEFERRED)
    flush_ms, flush_n = bt.profile_read(nh.K_FLUSH)
    chain_ms, chain_n = bt.profile_read(nh.K_TICK_CHAIN)
    panel_ms, panel_n = bt.profile_read(nh.K_TICK_PANELS)
    apply_ms, apply_n = bt.profile_read(nh.K_TICK_APPLY)
    rank_ms, rank_n = bt.profile_read(nh.K_TICK_RANK)
    next_ms, next_n = bt.profile_read(nh.K_TICK_NEXT)
    dab_ms, dab_n = bt.profile_read(nh.K_DA_BEGIN)
    das_ms, das_n = bt.profile_read(nh.K_DA_STEP)
    bt.profile(False)
    bad, st = bt.status()
    if st != 0:
        raise RuntimeError("device status %d on filter %d" % (st, bad))

    block_s = np.array(block_s)
    rccl = None
    if world > 1:
        # per-block max over ranks
        t = torch.tensor(block_s, dtype=torch.float64)
        if coll_dev:
            t = t.to(coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        block_s = t.cpu().numpy()
        # the batch reduction: Monte-Carlo statistics of all trials, gathered and summed in rank order -- once through
        # torch.distributed (backend as given), and, with one GPU per rank, through the library's own C-ABI RCCL path
        local_stats = bt.stats()
        total, per_rank = nd.reduce_stats(local_stats, device=coll_dev)
        n_filters_seen = int(total[-1])
        if args.backend == "nccl":
            try:
                uid = [nh.Comm.unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                comm = nh.Comm(uid[0], world, rank, dev)
                tot2, rows2 = bt.reduce_stats(comm)
                rccl = {"c_abi_allgather": "ok", "ranks": comm.world,
                        "equals_torch_distributed_bitwise": bool(np.array_equal(tot2, total) and np.array_equal(rows2, per_rank))}
                comm.close()
            except Exception as e:      # reported in the line, never hidden: the torch.distributed result above stands
                rccl = {"c_abi_allgather": "failed", "error": str(e)[:300]}
    else:
        local_stats = bt.stats()
        total, per_rank = local_stats, local_stats[None, :]
        n_filters_seen = int(total[-1])

    if args.dump:
        states = np.stack([bt.state(b) for b in range(B)])
        seens = np.array([bt.seen(b) for b in range(B)])
        np.savez(args.dump + ".rank%d.npz" % rank, first_filter=first_filter, states=states, seens=seens,
                 local_stats=local_stats, total=total, per_rank=per_rank)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    dt_med = float(np.median(block_s))
    updates_per_block = float(filters_total if args.workload == "batch" else world) * m * K
    out = {
        "metric": "EKF updates/sec (predict+correct, N landmarks)",
        "value": updates_per_block / dt_med,
        "unit": "updates/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": 1e3 * dt_med / K,
        "higher_is_better": True,
        # single filter: one replica per GPU (per-GPU work fixed -> weak); batch without --filters: 1024 filters split
        # over the ranks (total work fixed -> strong); batch with --filters F: F filters per GPU (weak)
        "scaling": "strong" if (args.workload == "batch" and not args.filters) else "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": {"ekf1000": "single EKF per GPU, known association (BASELINE configs[1])",
                                "batch": "batch of independent EKFs sharded over the GPUs (BASELINE configs[3])",
                                "da1000": "single EKF per GPU, unknown data association (BASELINE configs[4])",
                                "ekf5000": "single EKF per GPU, fp32, dense MFMA F P F^T predict (BASELINE configs[2]); F = "
                                           + ("a fixed dense random Jacobian (GEMM micro-measurement)" if args.dense_random_f else
                                              "the reference's A = I + B(theta', twist), formed on the device every tick")}[args.workload],
                   "landmarks": n, "state_len": L, "filters_rank0": B, "filters_total": filters_total,
                   "filters_counted_by_reduction": n_filters_seen,
                   "updates_per_step": m, "parallelism": "replicas x%d" % world if args.workload != "batch" else "filters sharded x%d" % world,
                   "ranks_seen_by_backend": ranks_seen, "backend": (args.backend if world > 1 else None),
                   "trace": "per-filter, generated on the device (k_sim_path / k_sim_markers)" if trace_kind == "device"
                            else "one host-made trace per rank, resident in HBM",
                   "trace_kind": TRACE_KIND + ("" if known else "; marker noise 1e-4 m, %d of %d map slots filled" % (n_world, n)),
                   "kernel_events_in_timed_region": in_region, "Q_diag": float(Q[0, 0]), "R_diag": float(R[0, 0])},
        "timing": {"blocks": int(block_s.size), "steps_per_block": K, "timed_region_ms": 1e3 * float(block_s.sum()),
                   "ms_per_step_median": 1e3 * dt_med / K, "ms_per_step_min": 1e3 * float(block_s.min()) / K,
                   "ms_per_step_max": 1e3 * float(block_s.max()) / K,
                   "ms_per_step_stdev": 1e3 * (statistics.pstdev(block_s.tolist()) if block_s.size > 1 else 0.0) / K,
                   "note": "each block = exactly K steps bracketed by barrier + synchronize, max over ranks; value and "
                           "ms_per_step are the median block"},
        "ticks_per_s": float(filters_total if args.workload == "batch" else world) * K / dt_med,
        "build_info": nh.build_info(),
    }
    if rccl is not None:
        out["rccl"] = rccl
    if args.workload == "batch":
        out["batch_stats"] = {"mean_sq_pose_error_th_x_y": (total[2 * L:2 * L + 3] / max(n_filters_seen, 1)).tolist(),
                              "mean_nees": float(total[2 * L + 3] / max(n_filters_seen, 1)),
                              "mean_trace_P": float(total[2 * L + 4] / max(n_filters_seen, 1))}
    sweep_kernel, units = "k_update", 1
    exact_pass = None
    if rank_n:
        # the tick pipeline's ONE pass over P as a rank-2m update on the matrix cores (csrc/ekf_rank.h): all m corrections
        sweep_ms, sweep_n, sweep_kernel, units = rank_ms, rank_n, "k_tick_rank", m
        if apply_n:
            exact_pass = {"kernel": "k_tick_apply (exact chain, launched behind k_tick_rank for rounds that may hold a first "
                                    "sighting; exits at once for the others)", "avg_launch_us": 1e3 * apply_ms / apply_n, "launches": apply_n}
    elif apply_n:
        # the exact chain: ONE pass over P applies all m corrections of the tick, 7 FMAs per element and correction
        sweep_ms, sweep_n, sweep_kernel, units = apply_ms, apply_n, "k_tick_apply", m
        # which instantiation of the pass ran (csrc/nuslam_hip.hip::launch_pass): one big fp64 filter with the chip to
        # itself -> the two-unit kernel; overlapped with the next tick's chain, batches, fp32 -> k_tick_apply
        overlapped = known and B == 1 and not args.no_overlap and args.workload == "ekf1000" and args.tick_mode != 0
        if dtype == nh.F64 and B == 1 and L >= 1000 and not overlapped and not args.plain_pass and args.pass_variant != 1:
            sweep_kernel = "k_tick_apply_units"
    elif pair_n > sweep_n:
        # most corrections went through k_update2: TWO corrections per pass over P (bit-identical to two k_update)
        sweep_ms, sweep_n, sweep_kernel, units = pair_ms, pair_n, "k_update2", 2
    if sweep_n:
        # roofline.frac is PHYSICAL: the bytes one launch cannot avoid moving -- every element of P of every filter of
        # the launch read once and written once, 2*L^2*w*B, however many corrections the launch fuses -- over the
        # launch's measured duration, over the 8 TB/s HBM peak.  SURVEY 8(d)'s per-correction figure (2*L^2*w per
        # correction x corrections per launch) is kept beside it as effective_*: it can exceed the peak for a launch
        # that applies several corrections in one pass and is a throughput figure, not a roofline fraction.
        min_bytes = 2.0 * L * L * w * B
        avg_s = 1e-3 * sweep_ms / sweep_n
        ach = min_bytes / avg_s / 1e9
        tname = "double" if dtype == nh.F64 else "float"
        traffic, traffic_src = pmc_traffic(nh, "k_tick_apply_units<" if sweep_kernel == "k_tick_apply_units" else "%s<%s" % (sweep_kernel, tname), min_bytes)
        # the arithmetic roof beside the byte roof: fp64 FMAs per element and correction -- 2 on the matrix cores for the
        # rank-2m pass (v_mfma_f64_16x16x4_f64: 78.6 TFLOP/s), 7 on the vector pipe for the exact chain (78.6 nominal, 60.7
        # measured by tools/micro/fp64_fma_peak.hip), 10 for the per-correction sweeps
        fma_per = {"k_tick_rank": 2.0, "k_tick_apply": 7.0, "k_tick_apply_units": 7.0}.get(sweep_kernel, 10.0)
        fl = 2.0 * fma_per * L * L * units * B
        pipe = "mfma_f64" if sweep_kernel == "k_tick_rank" else "valu_f64"
        arith_frac = fl / avg_s / 1e12 / 78.6
        out["roofline"] = {"bound": "hbm" if ach / HBM_PEAK_GBS >= arith_frac else pipe, "kernel": sweep_kernel, "achieved": ach,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                           "traffic_source": traffic_src, "avg_launch_us": 1e6 * avg_s, "launches": sweep_n,
                           "corrections_per_launch": units, "min_bytes_per_launch": min_bytes,
                           "effective_GBps": units * min_bytes / avg_s / 1e9,
                           "effective_frac": units * min_bytes / avg_s / 1e9 / HBM_PEAK_GBS,
                           pipe: {"achieved_TFLOPs": fl / avg_s / 1e12, "peak_TFLOPs": 78.6, "frac": arith_frac,
                                  "flop_per_launch": fl, "fma_per_element_and_correction": fma_per},
                           "note": "achieved/frac: minimum bytes one launch must move (2*L^2*w*B: P read once, written once) / "
                                   "launch duration (HIP events) / 8 TB/s; bound = the nearer of the byte roof and the arithmetic "
                                   "roof of the pipe the kernel computes on.  effective_*: SURVEY 8(d)'s 2*L^2*w per CORRECTION x "
                                   "corrections per launch -- a throughput figure that exceeds the physical one when a launch "
                                   "fuses corrections.  traffic: FETCH_SIZE x2 + WRITE_SIZE from separate rocprofv3 --pmc passes "
                                   "of this build (null when no record matches nuslam_build_info())"}
        if exact_pass:
            out["roofline"]["exact_pass_behind_it"] = exact_pass
        if fused_launch:
            if fused_launch["ticks_per_launch"] > 1:
                fused_launch["note"] = ("the timed region runs each block of %d ticks as this ONE launch: the covariance is read once when the "
                                        "launch begins and written once when it ends, in between it stays in the pass workgroups' accumulators "
                                        "and a tick stores only the rows / columns of the next tick's index set; roofline.achieved / frac above "
                                        "are k_tick_rank's, measured in %d more steps with the pass as a launch of its own (tick mode 4: same "
                                        "arithmetic, same bits)" % (K, K))
            else:
                fused_launch["whole_launch_GBps"] = min_bytes / (1e-6 * fused_launch["avg_launch_us"]) / 1e9
                fused_launch["note"] = ("the timed region runs the tick as this ONE launch: the pass's tile loads run under the serial chain, its "
                                        "k-steps follow the strips, its stores close the launch; roofline.achieved / frac above are k_tick_rank's, "
                                        "measured in %d more steps with the pass as a launch of its own (tick mode 4: same arithmetic, same bits)" % K)
            out["roofline"]["fused_launch"] = fused_launch
        # ... and the same bytes over the WHOLE tick (every kernel of it, launch gaps included): what the chip's HBM sees of a tick
        tick_s = dt_med / K
        out["roofline"]["whole_tick"] = {"bytes_per_step": min_bytes, "ms_per_step": 1e3 * tick_s, "GBps": min_bytes / tick_s / 1e9,
                                         "frac": min_bytes / tick_s / 1e9 / HBM_PEAK_GBS,
                                         "note": "2*L^2*w*B / ms_per_step: the bytes a pass over P moves, over the wall time of a whole tick"
                                                 + (" (algorithmic: in a one-launch run P stays on the chip between the ticks)"
                                                    if fused_launch and fused_launch.get("ticks_per_launch", 1) > 1 else "")}
        kernel_us_fused = fused_launch
        if args.workload == "batch" and B >= 512 and args.interleave in (None, -1):
            out["roofline"]["note_groups"] = ("the timed region runs the batch as 2 groups of filters on 2 streams (nuslam_batch_set_interleave default: one "
                                              "group's pass beside the other's chain and strips); avg_launch_us / frac and kernel_us are each kernel ALONE, "
                                              "from %d more steps with the batch as one group" % K)
        out["kernel_us"] = {"update": 1e3 * sweep_ms / sweep_n,
                            "predict": 1e3 * pred_ms / max(pred_n, 1),
                            "associate": 1e3 * asso_ms / max(asso_n, 1) if asso_n else None}
        if kernel_us_fused:
            out["kernel_us"]["tick_fused"] = kernel_us_fused["avg_tick_us"]
            out["kernel_us"]["note"] = "tick_fused: the one launch of the timed region; tick_chain (k_tick_front) / tick_rank: the two launches of tick mode 4"
        if apply_n or rank_n:
            out["kernel_us"].update({"tick_chain": 1e3 * chain_ms / chain_n if chain_n else None,
                                     "tick_panels": 1e3 * panel_ms / panel_n if panel_n else None,
                                     "tick_rank": 1e3 * rank_ms / rank_n if rank_n else None,
                                     "tick_apply": 1e3 * apply_ms / apply_n if apply_n else None})
        if das_n:
            # unknown association: one resident launch per round of <= 16 markers (k_da_round), or k_da_begin + one
            # k_da_step per marker when the handle's workgroups do not fit the chip
            out["kernel_us"].update({"da_round_or_step": 1e3 * das_ms / das_n, "da_launches_per_tick": das_n / max(pred_n, 1),
                                     "da_begin": 1e3 * dab_ms / dab_n if dab_n else None})
    if args.deferred and flush_n:
        # the covariance pass of this mode is k_flush: once per tick, 2*L^2*w bytes per filter (actual bytes moved)
        per_launch_bytes = 2.0 * L * L * w * B
        avg_s = 1e-3 * flush_ms / flush_n
        ach = per_launch_bytes / avg_s / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "k_flush", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": 1e6 * avg_s, "launches": flush_n,
                           "min_bytes_per_launch": per_launch_bytes,
                           "effective_GBps": per_launch_bytes * out["value"] / (world * B) / 1e9}
        out["kernel_us"] = {"update_deferred": 1e3 * dupd_ms / max(dupd_n, 1), "flush": 1e3 * flush_ms / flush_n,
                            "predict": 1e3 * pred_ms / max(pred_n, 1)}
        out["config"]["deferred"] = True
    if gemm_n:
        flop = 2.0 * L ** 3                                    # one of the two products of F P F^T
        avg_s = 1e-3 * gemm_ms / gemm_n
        peak = 157.3 if dtype == nh.F32 else 78.6              # MI355X_MICROARCH.md: dense f32 / f64 matrix peak, TFLOP/s
        out["roofline_hbm_kernel"] = out.get("roofline")
        out["roofline"] = {"bound": "mfma", "kernel": "k_gemm (F P, then T F^T + Qbar)", "achieved": flop / avg_s / 1e12,
                           "peak": peak, "unit": "TFLOP/s", "frac": flop / avg_s / 1e12 / peak, "traffic": None,
                           "avg_launch_us": 1e6 * avg_s, "launches": gemm_n, "algorithmic_flop_per_launch": flop}
        out.setdefault("kernel_us", {})["dense_gemm"] = 1e6 * avg_s
    if warm_state is not None and args.cpu_seconds > 0 and args.workload == "ekf1000":
        ptr = synth.make_wellposed_trace(n, max(64, args.parity_ticks), m, seed=12345)
        cb, orc, orc_ticks = cpu_baseline(n, m, ptr, args.cpu_seconds, warm_state)
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
        out["speedup_note"] = ("GPU O(L^2) correction vs the reference's dense O(L^3) algebra on the host cores; the fair fight is "
                               "cpu_baseline.structured (the same O(L^2) sums on the host): speedup_vs_structured_cpu")
        out["speedup_vs_structured_cpu"] = out["value"] / cb["structured"]["value"]
        # parity in the same run, THROUGH THE TIMED PATH: the same resident-trace entry point (nuslam_batch_run) with the
        # timed handle's settings (overlap, tick mode, pass variant) on a fresh filter restored from the same
        # post-initialisation snapshot, against the oracle's structured mode (bit-equal to the dense mode the baseline timed:
        # checked here on the ticks the dense run covered)
        pt = max(args.parity_ticks, orc_ticks)
        O_ = __import__("_oracle")
        O_.set_threads(O_.usable_cpus())
        os2 = O_.OracleEKF(np.zeros(3), np.zeros(2 * n), Q, R, O_.ORC_STRUCTURED)
        os2.restore(*warm_state)
        dense_equal = None
        for t in range(pt):
            os2.tick(tw=ptr.tw[t], mx=ptr.mx[t], my=ptr.my[t], known_ids=ptr.ids[t])
            if t + 1 == orc_ticks:
                dense_equal = bool(np.array_equal(os2.state, orc.state) and np.array_equal(os2.cov, orc.cov))
        O_.set_threads(1)
        g2 = nh.EKF(np.zeros(3), np.zeros(2 * n), Q, R, dtype=dtype, device=dev)
        g2.restore(*warm_state)
        b2 = g2.as_batch()
        if args.per_correction or args.no_pairing:
            b2.set_tick_mode(0)
        elif args.tick_pipeline:
            b2.set_tick_mode(1)
        if args.tick_mode is not None:
            b2.set_tick_mode(args.tick_mode)
        if args.plain_pass:
            b2.set_pass_variant(1)
        if args.pass_variant is not None:
            b2.set_pass_variant(args.pass_variant)
        if args.no_overlap:
            b2.set_overlap(False)
        if args.overlap:
            b2.set_overlap(True)
        if args.no_pairing:
            b2.set_pairing(False)
        b2.load_trace(ptr.tw[:pt, :2], ptr.mx[:pt], ptr.my[:pt], ptr.ids[:pt], bcast=True)
        b2.run(0, pt)
        bad2, st2 = b2.status()
        gs, gP, os_, oP = g2.state, g2.cov, os2.state.copy(), os2.cov.copy()
        floor = 1e-12 * np.abs(oP).max()
        out["parity"] = {"against": "oracle/nuslam_oracle.c, structured mode (the reference's algebra with exact-zero terms skipped; "
                                    "EKF parity unpinned, see DESIGN.md); input trace " + TRACE_KIND,
                         "path": "nuslam_batch_run on a resident trace with the timed handle's settings (overlap %s, pass variant %s)"
                                 % ("off" if args.no_overlap else "default", "default" if args.pass_variant is None else args.pass_variant),
                         "structured_oracle_equals_dense_oracle_bitwise_on_first_%d_ticks" % orc_ticks: dense_equal,
                         "ticks": int(pt), "corrections": int(pt * m), "device_status": int(st2),
                         "max_rel_err_state": float((np.abs(gs - os_) / np.maximum(np.abs(os_), 1e-12)).max()),
                         "max_rel_err_cov": float((np.abs(gP - oP) / np.maximum(np.abs(oP), floor)).max()),
                         "rel_frobenius_cov": float(np.linalg.norm(gP - oP) / np.linalg.norm(oP)),
                         "seen_equal": bool(g2.seen == os2.seen), "tolerance": 1e-6}
        if not args.no_api:
            atr = synth.make_wellposed_trace(n, 256, m, seed=12345)
            out["api_driven"] = api_driven(n, m, atr, bx, by, wid, Q, R)
            # ... and with what the node does at the top of every iteration (slam.cpp:184,250-251: getStateVector twice, getSeenLandmarks)
            nl = api_driven(n, m, atr, bx, by, wid, Q, R, exe_name="node_loop_rate")
            out["api_driven"]["with_the_nodes_loop_top"] = {k: v for k, v in nl.items() if k != "note"}
            out["api_driven"]["ticks"] = 256
    if args.workload == "da1000" and world == 1 and args.cpu_seconds > 0:
        out["parity"] = da_parity(nh, args, n, m, Q, R, dtype, dev, seed)
    print(json.dumps(out))
    sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
