#!/usr/bin/env python3
"""bench.py -- EKF-SLAM updates/s on MI355X (BASELINE.json metric), one JSON line on rank 0.

A "step" is one tick of the slam node's loop (nuslam/src/slam.cpp:250-319): 1 predict + m sequential
corrections on one batch of synthetic odometry + range-bearing input that is already resident in HBM.

Workloads
  ekf1000   (default; BASELINE configs[1]) one EKF per GPU, N = 1000 landmarks, fp64 covariance, m = 16,
            known association; the map is initialised (one update per landmark) before the timed region.
            With --gpus N every rank runs its own independent replica (Monte-Carlo trials): weak scaling.
  batch     (BASELINE configs[3]) --filters independent EKFs of N = 200 landmarks per GPU, one launch per
            kernel for the whole batch.
  da1000    (BASELINE configs[4]) as ekf1000 with unknown data association (associateLandmark per marker).
  ekf5000   (BASELINE configs[2]) one EKF, N = 5000, fp32 covariance, every predict propagates P with a resident
            dense Jacobian on the matrix cores (F P F^T, 4 L^3 flop) -- the MFMA-bound configuration.
`value` = corrections (EKF updates) per second over all ranks = ranks * filters * m * steps / seconds.

Extra objects: "roofline" for the dominant kernel (k_update; algorithmic bytes 2*L^2*w per launch per
filter, duration from per-dispatch HIP events on the handle's stream) and "cpu_baseline" (the oracle's dense
mode = the reference's algebra, timed on this host's cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "shermbot-navigation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="ekf1000", choices=["ekf1000", "batch", "da1000", "ekf5000"])
    ap.add_argument("--landmarks", type=int, default=None)
    ap.add_argument("--filters", type=int, default=None, help="filters per GPU (batch workload; default 1024/gpus)")
    ap.add_argument("--m", type=int, default=16, help="corrections per tick")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--deferred", action="store_true",
                    help="opt-in deferred application: a tick's corrections are kept as rank-2 factors and applied to P "
                         "once per tick (csrc/ekf_deferred.h); results agree with the default to rounding, not bitwise")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the batch reduction (nccl = RCCL over xGMI; gloo only to rehearse "
                         "the multi-rank path on one GPU)")
    ap.add_argument("--trace", default=None, choices=["host", "device"],
                    help="batch workload: per-filter Monte-Carlo traces generated on the device by the simulator kernels "
                         "(default), or one host-made trace per rank replayed by every filter")
    ap.add_argument("--no-pairing", action="store_true", help="one k_update launch per correction (disable k_update2)")
    ap.add_argument("--group", type=int, default=0, help="corrections per pass over P: 2 or 4 (0 = library default)")
    ap.add_argument("--events-in-timed-region", action="store_true",
                    help="attach the per-dispatch HIP events inside the timed region itself (costs ~25%% throughput: "
                         "every dispatch then carries a completion signal); default: a second pass of K identical steps "
                         "right after the timed one")
    return ap.parse_args()


def cpu_baseline(n, m, tr, budget_s, warm_state):
    """The reference CPU path (dense algebra: two L^3 GEMMs per predict, one per update -- slam_library.cpp:104,279) on
    this host's usable cores, on the first ticks of the same trace from the same post-initialisation snapshot.  Two
    restatements are timed and the FASTER one is reported as the baseline:
      oracle-c    oracle/nuslam_oracle.c dense mode, OpenMP-blocked loops
      numpy-blas  tests/_np_ekf.py, the same chain through numpy's BLAS dgemm (OpenBLAS) -- what Armadillo itself
                  would dispatch to."""
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
    if limiter is not None:
        limiter.unregister() if hasattr(limiter, "unregister") else None

    best = max(cands, key=lambda k: cands[k][0])
    v, done, dt = cands[best]
    return {"value": v, "unit": "updates/s", "cores": cores, "kind": "port", "impl": best,
            "sample": "%d tick(s) (1 predict + %d updates each) of the same N=%d trace from the same post-initialisation "
                      "snapshot, reference algebra (two L^3 GEMMs per predict, one per update), %.1f s; faster of "
                      "{oracle-c: %.2f, numpy-blas: %.2f} updates/s" % (done, m, n, dt, cands["oracle-c"][0], cands["numpy-blas"][0]),
            "ms_per_step": 1e3 * dt / done}, o, cands["oracle-c"][1]


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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

    import nuslam_hip as nh
    from nuslam_hip import synth

    if args.workload == "ekf5000":
        args.dtype = "f32"
        if args.steps == 200 and args.warmup == 20:
            args.steps, args.warmup = 5, 1          # a tick is ~45 ms here
    dtype = nh.F64 if args.dtype == "f64" else nh.F32
    w = 8 if dtype == nh.F64 else 4
    if args.workload == "batch":
        n = args.landmarks or 200
        B = args.filters or max(1, 1024 // world)
    else:
        n = args.landmarks or (5000 if args.workload == "ekf5000" else 1000)
        B = 1
    m = min(args.m, n)
    L = 3 + 2 * n
    K, W = args.steps, args.warmup
    known = args.workload != "da1000"

    # ---- synthetic input (seeded; Monte-Carlo replica r uses seed 12345 + r), made resident in HBM
    seed = 12345 + rank
    # data association needs one free slot: associateLandmark writes a hypothetical landmark at index seen+1 and
    # indexes out of bounds on a full map (slam_library.cpp:206-207), so the world holds n-1 landmarks there
    n_world = n if known else n - 1
    tr = synth.make_trace(n_world, W + 3 * K, m, seed=seed, noise_sigma=None if known else 1e-4)
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

    if B == 1:
        ekf = nh.EKF(np.zeros(3), np.zeros(2 * n), Q, R, dtype=dtype, device=dev)
        bt = ekf.as_batch()
        ekf.tick(np.zeros(3), bx, by, known_ids=wid, want_ids=False)     # initialise the whole map (untimed)
        ekf.sync()
        warm_state = (ekf.state, ekf.cov, ekf.seen) if (rank == 0 and world == 1 and args.cpu_seconds > 0 and args.workload == "ekf1000") else None   # cpu_baseline: N=1 only
        if args.workload == "ekf5000":
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
    else:
        bt = nh.Batch(B, n, Q, R, dtype=dtype, device=dev)
        # initialise every filter's map with one resident warm-up tick of n observations
        bt.load_trace(np.zeros((1, 2)), bx[None, :], by[None, :], wid[None, :], bcast=True)
        bt.run(0, 1)
        bt.sync()
        warm_state = None
    ids = tr.ids if known else None
    trace_kind = args.trace or ("device" if args.workload == "batch" else "host")
    if trace_kind == "device":
        # every filter is its own Monte-Carlo trial -- its trace is generated in HBM by the simulator
        # kernels (tube_world.cpp:509-533 per filter) from the random streams of its GLOBAL filter index, so the
        # sharded run reproduces the unsharded one and nothing but the six-number parameter block crosses PCIe
        ticks_total = W + 3 * K
        uL, uR = 0.30 * 50, 0.36 * 50                      # the wheel increments of synth.make_trace, per second
        cmd = np.zeros((ticks_total, 2))
        cmd[:, 0] = (synth.WHEEL_RADIUS / synth.WHEEL_BASE) * (uR - uL)
        cmd[:, 1] = (synth.WHEEL_RADIUS / 2) * (uL + uR)
        cmd[24::25, 0] = 0.0                               # every 25th tick straight: the dth == 0 branch
        sim = nh.SimParams(marker_sigma=float(np.sqrt(1e-3)) if known else 1e-4, max_range=0.0)
        bt.simulate(sim, tr.landmarks, cmd, m, 12345, first_filter=rank * B, known_ids=known)
    else:
        bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, ids, bcast=True)
    if args.deferred:
        bt.set_deferred(True)
    if args.no_pairing:
        bt.set_pairing(False)
    elif args.group:
        bt.set_pairing(args.group)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        bt.sync()

    bt.run(0, W)                       # W untimed warm-up steps
    barrier()
    in_region = args.events_in_timed_region
    bt.profile(in_region)
    t0 = time.perf_counter()
    bt.run(W, W + K)                   # EXACTLY K timed steps
    bt.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if not in_region:
        # kernel durations: the next K steps of the same trace, every dispatch bracketed by its own HIP events
        # on the handle's stream (hipExtLaunchKernelGGL start/stop events)
        bt.profile(True)
        bt.run(W + K, W + 2 * K)
        bt.sync()
    use_events = True
    sweep_ms, sweep_n = bt.profile_read(nh.K_UPDATE)
    pair_ms, pair_n = bt.profile_read(nh.K_UPDATE2)
    pred_ms, pred_n = bt.profile_read(nh.K_PREDICT)
    asso_ms, asso_n = bt.profile_read(nh.K_ASSOCIATE)
    gemm_ms, gemm_n = bt.profile_read(nh.K_DENSE_GEMM)
    dupd_ms, dupd_n = bt.profile_read(nh.K_UPDATE_DEFERRED)
    flush_ms, flush_n = bt.profile_read(nh.K_FLUSH)
    bt.profile(False)
    deferred_extra = None
    if args.workload == "ekf1000" and not args.deferred and not in_region and K >= 4:
        # the same K steps once more with the opt-in deferred application (reported beside the headline, never as it)
        bt.set_deferred(True)
        bt.run(W + 2 * K, W + 2 * K + 2)
        barrier()
        td = time.perf_counter()
        bt.run(W + 2 * K + 2, W + 3 * K)
        bt.sync()
        td = time.perf_counter() - td
        bt.set_deferred(False)
        deferred_extra = {"value": float(world) * B * m * (K - 2) / td, "unit": "updates/s", "ms_per_step": 1e3 * td / (K - 2),
                          "note": "opt-in mode (nuslam_ekf_set_deferred): a tick's corrections kept as rank-2 factors, P rewritten "
                                  "once per tick; agrees with the default path to rounding (tests/test_gpu_deferred.py), not bitwise"}
    bad, st = bt.status()
    if st != 0:
        raise RuntimeError("device status %d on filter %d" % (st, bad))

    from nuslam_hip import dist as nd
    if world > 1:
        dt = nd.max_over_ranks(dt, device=coll_dev)
        # the batch reduction over xGMI (RCCL): Monte-Carlo statistics of all trials, gathered and summed in rank order
        total, _ = nd.reduce_stats(bt.stats(), device=coll_dev)
        n_filters_total = int(total[-1])
    else:
        n_filters_total = B

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    updates = float(world) * B * m * K
    out = {
        "metric": "EKF updates/sec (predict+correct, N landmarks)",
        "value": updates / dt,
        "unit": "updates/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": 1e3 * dt / K,
        "higher_is_better": True,
        # single filter: one replica per GPU (per-GPU work fixed -> weak); batch without --filters: 1024 filters split
        # over the ranks (total work fixed -> strong); batch with --filters F: F filters per GPU (weak)
        "scaling": "strong" if (args.workload == "batch" and not args.filters) else "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": {"ekf1000": "single EKF per GPU, known association (BASELINE configs[1])",
                                "batch": "batch of independent EKFs per GPU (BASELINE configs[3])",
                                "da1000": "single EKF per GPU, unknown data association (BASELINE configs[4])",
                                "ekf5000": "single EKF per GPU, fp32, dense MFMA F P F^T predict (BASELINE configs[2])"}[args.workload],
                   "landmarks": n, "state_len": L, "filters_per_gpu": B, "filters_total": n_filters_total,
                   "updates_per_step": m, "parallelism": "replicas x%d" % world if B == 1 else "filters sharded x%d" % world,
                   "trace": "per-filter, generated on the device (k_sim_path / k_sim_markers)" if trace_kind == "device"
                            else "one host-made trace per rank, resident in HBM",
                   "kernel_events_in_timed_region": in_region, "Q_diag": float(Q[0, 0]), "R_diag": float(R[0, 0])},
        "ticks_per_s": float(world) * B * K / dt,
    }
    sweep_kernel, units = "k_update", 1
    if use_events and pair_n > sweep_n:
        # most corrections went through k_update2: TWO corrections per pass over P (bit-identical to two k_update)
        sweep_ms, sweep_n, sweep_kernel, units = pair_ms, pair_n, "k_update2", 2
    if use_events and sweep_n:
        # the algorithmic figure is 2*L^2*w bytes per correction per filter (read + write every P entry
        # once); `achieved` = that figure x the corrections one launch processes / the launch duration.  k_update2
        # really moves half of it per correction (temporal blocking), which `actual_bytes_per_launch` and `traffic` show.
        per_unit_bytes = 2.0 * L * L * w * B
        per_launch_bytes = per_unit_bytes * units
        avg_s = 1e-3 * sweep_ms / sweep_n
        ach = per_launch_bytes / avg_s / 1e9
        # HBM bytes per launch from the PMC counters: a committed rocprofv3 --pmc measurement of this very workload
        # (separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per the gfx950 correction); null for others
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01", "ekf1000_pmc_hbm_traffic.json")
        if args.workload == "ekf1000" and n == 1000 and dtype == nh.F64 and os.path.exists(pmc):
            rec = json.load(open(pmc))
            if rec.get("kernel", "k_update") == sweep_kernel:
                traffic = rec["per_launch_bytes"]["hbm_traffic"]
        out["roofline"] = {"bound": "hbm", "kernel": sweep_kernel, "achieved": ach, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                           "avg_launch_us": 1e6 * avg_s, "launches": sweep_n, "corrections_per_launch": units,
                           "algorithmic_bytes_per_launch": per_launch_bytes,
                           "actual_bytes_per_launch": per_unit_bytes,
                           "achieved_actual_bytes": per_unit_bytes / avg_s / 1e9,
                           "frac_actual_bytes": per_unit_bytes / avg_s / 1e9 / HBM_PEAK_GBS,
                           "note": "achieved/frac follow the algorithmic definition (2*L^2*w bytes per correction x "
                                   "corrections per launch); a launch that applies two corrections in one pass moves each "
                                   "byte once, so frac can exceed 1 -- achieved_actual_bytes / frac_actual_bytes / traffic "
                                   "are what crosses the memory interface"}
        out["kernel_us"] = {"update": 1e3 * sweep_ms / sweep_n,
                            "predict": 1e3 * pred_ms / max(pred_n, 1),
                            "associate": 1e3 * asso_ms / max(asso_n, 1) if asso_n else None}
    if args.deferred and flush_n:
        # the covariance pass of this mode is k_flush: once per tick, 2*L^2*w bytes per filter (actual bytes moved);
        # "effective" = the eager formula (2*L^2*w per correction) over the time actually spent per correction
        per_launch_bytes = 2.0 * L * L * w * B
        avg_s = 1e-3 * flush_ms / flush_n
        ach = per_launch_bytes / avg_s / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "k_flush", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": 1e6 * avg_s, "launches": flush_n,
                           "algorithmic_bytes_per_launch": per_launch_bytes,
                           "effective_GBps_eager_formula": per_launch_bytes * out["value"] / (world * B) / 1e9}
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
        out["kernel_us"]["dense_gemm"] = 1e6 * avg_s
    if deferred_extra is not None:
        out["deferred_mode"] = deferred_extra
    if warm_state is not None and args.cpu_seconds > 0 and args.workload == "ekf1000":
        ptr = synth.make_trace(n, W + 3 * K, m, seed=12345)
        cb, orc, orc_ticks = cpu_baseline(n, m, ptr, args.cpu_seconds, warm_state)
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
        # parity in the same run: the ticks the CPU oracle (dense reference algebra) just ran, replayed on the GPU from
        # the same post-initialisation snapshot through the same kernels the timed region used
        g2 = nh.EKF(np.zeros(3), np.zeros(2 * n), Q, R, dtype=dtype, device=dev)
        g2.restore(*warm_state)
        for t in range(orc_ticks):
            g2.tick(ptr.tw[t], ptr.mx[t], ptr.my[t], known_ids=ptr.ids[t], want_ids=False)
        gs, gP, os_, oP = g2.state, g2.cov, orc.state.copy(), orc.cov.copy()
        floor = 1e-12 * np.abs(oP).max()
        out["parity"] = {"against": "oracle/nuslam_oracle.c, dense mode (the reference's algebra; EKF parity unpinned, see DESIGN.md)",
                         "ticks": int(orc_ticks), "corrections": int(orc_ticks * m),
                         "max_rel_err_state": float((np.abs(gs - os_) / np.maximum(np.abs(os_), 1e-12)).max()),
                         "max_rel_err_cov": float((np.abs(gP - oP) / np.maximum(np.abs(oP), floor)).max()),
                         "rel_frobenius_cov": float(np.linalg.norm(gP - oP) / np.linalg.norm(oP)),
                         "seen_equal": bool(g2.seen == orc.seen), "tolerance": 1e-6}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
