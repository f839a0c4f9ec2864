"""The N > 1 path on the GPU: `bench.py --gpus 2` starts two rank processes itself, the batch's filters
are split with nuslam_hip.dist.shard (remainders included), every rank generates its filters' traces from their
GLOBAL indices, and the statistics are gathered and added in rank order.  Two ranks share GPU 0 here, so the
rendezvous backend is gloo (RCCL refuses two ranks on one device); the library's own RCCL reduction is exercised with a
one-rank communicator.  No scaling number is taken from this."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from nuslam_hip import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env["NUSLAM_SKIP_BUILD"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True,
                         timeout=timeout, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line: " + out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("total", [8, 7])
def test_two_ranks_reproduce_one_rank_bitwise(hip, tmp_path, total):
    common = ["--workload", "batch", "--filters-total", str(total), "--landmarks", "40", "--steps", "3", "--warmup", "1",
              "--blocks", "2", "--cpu-seconds", "0"]
    one = run_bench(["--gpus", "1"] + common + ["--dump", str(tmp_path / "one")])
    two = run_bench(["--gpus", "2", "--backend", "gloo"] + common + ["--dump", str(tmp_path / "two")])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["ranks_seen_by_backend"] == 2 and two["config"]["backend"] == "gloo"
    assert two["config"]["filters_total"] == total == two["config"]["filters_counted_by_reduction"]
    assert two["config"]["filters_rank0"] == (total + 1) // 2          # contiguous blocks, the remainder to the low ranks
    assert two["scaling"] == "strong" and two["value"] > 0
    a = np.load(str(tmp_path / "one") + ".rank0.npz")
    parts = [np.load(str(tmp_path / "two") + ".rank%d.npz" % r) for r in range(2)]
    assert [int(p["first_filter"]) for p in parts] == [0, (total + 1) // 2]
    states = np.concatenate([p["states"] for p in parts])
    seens = np.concatenate([p["seens"] for p in parts])
    assert states.shape == a["states"].shape
    assert np.array_equal(states, a["states"]) and np.array_equal(seens, a["seens"])      # per filter: bit for bit
    assert not np.array_equal(states[0], states[-1])                                       # and they are different trials
    # the reduced statistics: rows added in rank order == the one-rank, filter-ordered sums to 1e-12
    t2, t1 = parts[0]["total"], a["total"]
    assert np.array_equal(parts[0]["total"], parts[1]["total"])                            # same bits on every rank
    assert np.array_equal(parts[0]["per_rank"][0] + parts[0]["per_rank"][1], t2)
    assert np.allclose(t2, t1, rtol=1e-12, atol=1e-300) and t2[-1] == total


def test_c_abi_rccl_reduction_one_rank(hip):
    """nuslam_comm_* + nuslam_batch_reduce_stats (ncclAllGather on the batch's stream + rank-ordered device sum) with a
    one-rank communicator: must return exactly nuslam_batch_stats."""
    n, B = 12, 5
    Q, R = synth.Q_DEFAULT, synth.R_DEFAULT
    tr = synth.make_trace(n, 3, 4)
    bt = hip.Batch(B, n, Q, R)
    bt.load_trace(tr.tw[:, :2], tr.mx, tr.my, tr.ids, bcast=True)
    bt.run(0, 3)
    comm = hip.Comm(hip.Comm.unique_id(), 1, 0, 0)
    total, per_rank = bt.reduce_stats(comm)
    st = bt.stats()
    assert np.array_equal(total, st) and np.array_equal(per_rank[0], st) and total[-1] == B
    comm.close()


def test_snapshot_restore_round_trip(hip):
    n = 9
    Q, R = synth.Q_DEFAULT, synth.R_DEFAULT
    tr = synth.make_trace(n, 4, 5)
    g = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    for t in range(2):
        g.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t], want_ids=False)
    s, P, seen = g.snapshot()
    assert np.array_equal(s, g.state) and np.array_equal(P, g.cov) and seen == g.seen
    h = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
    h.restore(s, P, seen)
    for t in range(2, 4):
        g.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t], want_ids=False)
        h.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t], want_ids=False)
    assert np.array_equal(g.state, h.state) and np.array_equal(g.cov, h.cov) and g.seen == h.seen
