#!/bin/bash
# Regenerates the measurements kept under profiles/rNN/ on the GPU box.  usage (from the repo root, through gpurun):
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r02'
# Everything lands in gpurun_out/<round>/; copy what is to be judged into profiles/<round>/ afterwards.
# rocprofv3: the program itself after `--`, counters in their own passes with --kernel-trace only.
set -u
ROUND=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$ROUND
mkdir -p "$OUT"
export NUSLAM_SKIP_BUILD=1
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"

step() { echo "== $*" | tee -a "$OUT/progress.log"; }

step bench lines
timeout -k 10 300 $B > "$OUT/bench_ekf1000.json" 2> "$OUT/bench_ekf1000.err" || exit 1
timeout -k 10 200 $B --no-overlap --cpu-seconds 0 > "$OUT/bench_ekf1000_one_stream.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 200 $B --per-correction --cpu-seconds 0 > "$OUT/bench_ekf1000_pairs.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 200 $B --no-pairing --cpu-seconds 0 > "$OUT/bench_ekf1000_per_correction.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 300 $B --workload batch --steps 50 --warmup 10 > "$OUT/bench_batch.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 300 $B --workload batch --steps 50 --warmup 10 --trace device > "$OUT/bench_batch_device_trace.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 300 $B --workload da1000 --steps 100 --warmup 10 > "$OUT/bench_da1000.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 300 $B --workload da1000 --steps 100 --warmup 10 --tick-mode 2 > "$OUT/bench_da1000_launch_per_marker.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 300 $B --workload da1000 --steps 100 --warmup 10 --tick-mode 0 > "$OUT/bench_da1000_per_correction.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 300 $B --workload ekf5000 > "$OUT/bench_ekf5000.json" 2>> "$OUT/bench.err" || exit 1

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
stats ekf1000_one_stream --steps 200 --warmup 20 --blocks 3 --cpu-seconds 0 --no-overlap || exit 1
stats batch --workload batch --steps 20 --warmup 3 --blocks 2 --cpu-seconds 0 || exit 1
stats da1000 --workload da1000 --steps 50 --warmup 5 --blocks 2 --cpu-seconds 0 || exit 1
stats ekf5000 --workload ekf5000 --cpu-seconds 0 || exit 1

pmc() {     # name, counter, bench args...
    local name=$1 ctr=$2; shift 2
    step "rocprofv3 --pmc $ctr: $name"
    rm -rf "$OUT/pmc_${name}_$ctr"
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$ctr" --output-format csv -d "$OUT/pmc_${name}_$ctr" -- python3 "$ROOT/bench.py" "$@" \
        > /dev/null 2> "$OUT/pmc_${name}_$ctr.err" || return 1
}
PM="--steps 40 --warmup 10 --blocks 1 --cpu-seconds 0"
# (counter passes serialise the dispatches, so they run the one-stream order: the kernel and its bytes are the same)
pmc ekf1000 FETCH_SIZE $PM --no-overlap || exit 1
pmc ekf1000 WRITE_SIZE $PM --no-overlap || exit 1
python3 "$ROOT/tools/summarize_pmc.py" "$OUT/pmc_ekf1000_FETCH_SIZE" "$OUT/pmc_ekf1000_WRITE_SIZE" "$OUT/ekf1000_one_stream_pmc_hbm_traffic.json" \
    "k_tick_apply_units<" 80 2003 8 1 >> "$OUT/progress.log" 2>&1
# the default (overlapped) run's pass, k_tick_apply<double, 8, 2, true>: counter passes serialise the dispatches, under which
# the chain of an overlapped run never meets its strips (the waits expire, NUSLAM_E_SYNC) -- so the same kernel is counted
# in a one-stream run that selects it (--plain-pass)
pmc ekf1000ov FETCH_SIZE $PM --no-overlap --plain-pass && pmc ekf1000ov WRITE_SIZE $PM --no-overlap --plain-pass && \
python3 "$ROOT/tools/summarize_pmc.py" "$OUT/pmc_ekf1000ov_FETCH_SIZE" "$OUT/pmc_ekf1000ov_WRITE_SIZE" "$OUT/ekf1000_pmc_hbm_traffic.json" \
    "k_tick_apply<double" 80 2003 8 1 >> "$OUT/progress.log" 2>&1
PB="--workload batch --steps 6 --warmup 2 --blocks 1 --cpu-seconds 0"
pmc batch FETCH_SIZE $PB || exit 1
pmc batch WRITE_SIZE $PB || exit 1
python3 "$ROOT/tools/summarize_pmc.py" "$OUT/pmc_batch_FETCH_SIZE" "$OUT/pmc_batch_WRITE_SIZE" "$OUT/batch_pmc_hbm_traffic.json" \
    "k_tick_apply<double" 12 403 8 1024 >> "$OUT/progress.log" 2>&1
rm -rf "$OUT"/pmc_*_SIZE
step done
