#!/bin/bash
# Regenerates the measurements kept under profiles/rNN/ on the GPU box.  usage (from the repo root, through gpurun):
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r02'
# Everything lands in gpurun_out/<round>/; copy what is to be judged into profiles/<round>/ afterwards.
# rocprofv3: the program itself after `--`, counters in their own passes with --kernel-trace only.
set -u
ROUND=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$ROUND
mkdir -p "$OUT"
export NUSLAM_SKIP_BUILD=1
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"

step() { echo "== $*" | tee -a "$OUT/progress.log"; }

if [ "${SKIP_BENCH:-0}" != "1" ]; then
step bench lines
timeout -k 10 400 $B > "$OUT/bench_ekf1000.json" 2> "$OUT/bench_ekf1000.err" || exit 1
timeout -k 10 200 $B --tick-mode 5 --cpu-seconds 0 > "$OUT/bench_ekf1000_launch_per_tick.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 200 $B --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_ekf1000_driver_command.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 200 $B --tick-mode 4 --cpu-seconds 0 > "$OUT/bench_ekf1000_pass_as_second_launch.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 200 $B --tick-mode 3 --cpu-seconds 0 > "$OUT/bench_ekf1000_three_launches.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 200 $B --pass-variant 2 --tick-mode 3 --no-overlap --cpu-seconds 0 > "$OUT/bench_ekf1000_exact_chain_one_stream.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 200 $B --tick-mode 4 --overlap --cpu-seconds 0 > "$OUT/bench_ekf1000_streamed_overlap.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 200 $B --per-correction --cpu-seconds 0 > "$OUT/bench_ekf1000_pairs.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 200 $B --no-pairing --cpu-seconds 0 > "$OUT/bench_ekf1000_per_correction.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 300 $B --workload batch --steps 50 --warmup 10 > "$OUT/bench_batch.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 300 $B --workload batch --steps 50 --warmup 10 --pass-variant 1 > "$OUT/bench_batch_exact_chain.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 300 $B --workload batch --steps 50 --warmup 10 --interleave 1 --cpu-seconds 0 > "$OUT/bench_batch_one_group.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 400 $B --workload da1000 --steps 100 --warmup 10 > "$OUT/bench_da1000.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 300 $B --workload da1000 --steps 100 --warmup 10 --tick-mode 2 --cpu-seconds 0 > "$OUT/bench_da1000_launch_per_marker.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 300 $B --workload ekf5000 > "$OUT/bench_ekf5000.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 300 python3 $ROOT/tools/api_rate_run.py 256 > "$OUT/api_rates.json" 2>> "$OUT/bench.err" || exit 1
for v in 10 11 12 13 14 15 16 18; do
    timeout -k 10 200 $B --tick-mode 4 --pass-variant $v --cpu-seconds 0 --min-timed-ms 300 > "$OUT/tile_ekf1000_v$v.json" 2>> "$OUT/bench.err"
    timeout -k 10 300 $B --workload batch --steps 20 --warmup 5 --pass-variant $v --min-timed-ms 300 > "$OUT/tile_batch_v$v.json" 2>> "$OUT/bench.err"
    [ $v -le 13 ] && timeout -k 10 300 $B --workload ekf5000 --pass-variant $v --steps 3 --warmup 1 > "$OUT/tile_ekf5000_v$v.json" 2>> "$OUT/bench.err"
done
python3 - "$OUT" > "$OUT/pass_tiles.txt" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
print("the rank-2m pass (k_tick_rank), tile shape <RB, CB, WR, WC> per nuslam_batch_set_pass_variant(10 + k): launch duration (HIP events), fraction of 8 TB/s")
names = {"f64": {10: "<2,2,2,2> 128x64 (N=1000 default)", 11: "<4,1,1,4>", 12: "<1,4,4,1>", 13: "<2,2,1,4>", 14: "<1,2,2,2> 64x64 (default for L <= 600)", 15: "<1,1,2,2> 64x32", 16: "<2,1,2,2> 128x32", 18: "<2,3,2,2> 128x96"},
         "f32": {10: "<2,1,1,4>", 11: "<2,2,1,4>", 12: "<1,4,4,1>", 13: "<1,4,2,2>"}}
for wl, dt in (("ekf1000", "f64"), ("batch", "f64"), ("ekf5000", "f32")):
    for k in [v - 10 for v in sorted(names[dt])]:
        f = os.path.join(out, "tile_%s_v%d.json" % (wl, 10 + k))
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
            r = d.get("roofline_hbm_kernel") or d["roofline"]
            print("%-8s %s tile %d %-36s %9.1f us  %.3f of HBM peak   (%.4g updates/s)" % (wl, dt, k, names[dt][10 + k], r["avg_launch_us"], r["frac"], d["value"]))
        except Exception as e:
            print("%-8s tile %d: no line (%s)" % (wl, k, e))
PY

fi
# second pass (ONLY_BENCH=1): the bench lines again, now that profiles/<round>/ holds counter records of this build, so that
# their `roofline.traffic` is filled in
[ "${ONLY_BENCH:-0}" = "1" ] && { step "done (bench lines only)"; exit 0; }

stats() {   # name, bench args...
    local name=$1; shift
    step "rocprofv3 --kernel-trace --stats: $name"
    rm -rf "$OUT/prof_$name"
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$name" -- python3 "$ROOT/bench.py" "$@" \
        > "$OUT/prof_$name.json" 2> "$OUT/prof_$name.err" || return 1
    local f
    f=$(find "$OUT/prof_$name" -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && cp "$f" "$OUT/${name}_kernel_stats.csv"
    rm -rf "$OUT/prof_$name"
}
stats ekf1000 --steps 200 --warmup 20 --blocks 3 --cpu-seconds 0 || exit 1
stats ekf1000_launch_per_tick --steps 200 --warmup 20 --blocks 3 --cpu-seconds 0 --tick-mode 5 || exit 1
stats ekf1000_pass_as_second_launch --steps 200 --warmup 20 --blocks 3 --cpu-seconds 0 --tick-mode 4 || exit 1
# (the batch as ONE group: every kernel alone on the chip, as in the bench line's own event pass; the default runs it as two groups)
stats batch --workload batch --steps 20 --warmup 3 --blocks 2 --cpu-seconds 0 --interleave 1 || exit 1
stats da1000 --workload da1000 --steps 50 --warmup 5 --blocks 2 --cpu-seconds 0 || exit 1
stats ekf5000 --workload ekf5000 --cpu-seconds 0 || exit 1

pmc() {     # name, counter, bench args...
    local name=$1 ctr=$2; shift 2
    step "rocprofv3 --pmc $ctr: $name"
    rm -rf "$OUT/pmc_${name}_$ctr"
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$ctr" --output-format csv -d "$OUT/pmc_${name}_$ctr" -- python3 "$ROOT/bench.py" "$@" \
        > /dev/null 2> "$OUT/pmc_${name}_$ctr.err" || return 1
}
# (the default run has no dependency between streams: one stream, the chain / strips hand-offs inside one launch)
PM="--steps 40 --warmup 10 --blocks 1 --cpu-seconds 0"
pmc ekf1000 FETCH_SIZE $PM || exit 1
pmc ekf1000 WRITE_SIZE $PM || exit 1
# (the default run launches k_run_fused in its timed blocks and k_tick_rank in the extra tick-mode-4 block of the event pass)
python3 "$ROOT/tools/summarize_pmc.py" "$OUT/pmc_ekf1000_FETCH_SIZE" "$OUT/pmc_ekf1000_WRITE_SIZE" "$OUT/ekf1000_pmc_hbm_traffic.json" \
    "k_tick_rank<double" 40 2003 8 1 >> "$OUT/progress.log" 2>&1
# (k_run_fused: one launch per bench block -- warm-up 10 ticks, the timed block 40, the event pass 40: the last one is summarised)
python3 "$ROOT/tools/summarize_pmc.py" "$OUT/pmc_ekf1000_FETCH_SIZE" "$OUT/pmc_ekf1000_WRITE_SIZE" "$OUT/ekf1000_run_fused_pmc_hbm_traffic.json" \
    "k_run_fused<double" 1 2003 8 1 40 >> "$OUT/progress.log" 2>&1
pmc ekf1000_tick FETCH_SIZE $PM --tick-mode 5 || exit 1
pmc ekf1000_tick WRITE_SIZE $PM --tick-mode 5 || exit 1
python3 "$ROOT/tools/summarize_pmc.py" "$OUT/pmc_ekf1000_tick_FETCH_SIZE" "$OUT/pmc_ekf1000_tick_WRITE_SIZE" "$OUT/ekf1000_fused_pmc_hbm_traffic.json" \
    "k_tick_fused<double" 80 2003 8 1 >> "$OUT/progress.log" 2>&1
PB="--workload batch --steps 6 --warmup 2 --blocks 1 --cpu-seconds 0 --interleave 1"
pmc batch FETCH_SIZE $PB || exit 1
pmc batch WRITE_SIZE $PB || exit 1
python3 "$ROOT/tools/summarize_pmc.py" "$OUT/pmc_batch_FETCH_SIZE" "$OUT/pmc_batch_WRITE_SIZE" "$OUT/batch_pmc_hbm_traffic.json" \
    "k_tick_rank<double" 12 403 8 1024 >> "$OUT/progress.log" 2>&1
PE="--workload ekf5000 --steps 3 --warmup 1 --blocks 1 --cpu-seconds 0"
pmc ekf5000 FETCH_SIZE $PE && pmc ekf5000 WRITE_SIZE $PE && \
python3 "$ROOT/tools/summarize_pmc.py" "$OUT/pmc_ekf5000_FETCH_SIZE" "$OUT/pmc_ekf5000_WRITE_SIZE" "$OUT/ekf5000_pmc_hbm_traffic.json" \
    "k_tick_rank<float" 6 10003 4 1 >> "$OUT/progress.log" 2>&1
rm -rf "$OUT"/pmc_*_SIZE
step done
