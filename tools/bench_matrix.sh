#!/bin/bash
# Runs the bench workloads over the pass variants (builder-run measurements; the summaries go to profiles/rNN/).
# usage: tools/bench_matrix.sh OUTDIR
out=${1:-gpurun_out/matrix}
mkdir -p $out
run() { name=$1; shift; echo "== $name: $*"; timeout -k 10 400 python bench.py "$@" > $out/$name.json 2> $out/$name.err || echo "FAILED $name"; python - <<PY
import json
try:
    d = json.loads(open("$out/$name.json").read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print("   value %.4g %s  ms/step %.4f  kernel %s %.1f us frac %.3f  kernel_us %s" % (d["value"], d["unit"], d["ms_per_step"], r.get("kernel"), r.get("avg_launch_us", 0), r.get("frac", 0), {k: (round(v, 1) if v else v) for k, v in d.get("kernel_us", {}).items()}))
    if "parity" in d: print("   parity", {k: v for k, v in d["parity"].items() if k.startswith("max") or k in ("ticks", "device_status")})
    if "api_driven" in d: print("   api", {k: (v.get("updates_per_s") if isinstance(v, dict) else None) for k, v in d["api_driven"].items()})
    if "cpu_baseline" in d: print("   cpu dense %.1f structured %.1f dense_1thread %s" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["structured"]["value"], d["cpu_baseline"]["dense_1thread"].get("value")))
except Exception as e:
    print("   (no line)", e)
PY
}
run ekf1000_default --workload ekf1000
for v in 10 11 12 13 1 2; do run ekf1000_onestream_v$v --workload ekf1000 --no-overlap --pass-variant $v --cpu-seconds 0; done
run ekf1000_overlap_v1 --workload ekf1000 --pass-variant 1 --cpu-seconds 0
for v in 10 11 12 13 1; do run batch_v$v --workload batch --pass-variant $v --steps 50 --warmup 5; done
for v in 10 1; do run da1000_v$v --workload da1000 --pass-variant $v; done
for v in 10 11 12 13 1; do run ekf5000_v$v --workload ekf5000 --pass-variant $v --steps 3 --warmup 1; done
