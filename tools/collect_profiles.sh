#!/bin/bash
# Copies what tools/profile_round.sh left in gpurun_out/<round>/ into profiles/<round>/ (the judged, committed copy).
set -eu
ROUND=${1:-r03}
SRC=gpurun_out/$ROUND
DST=profiles/$ROUND
mkdir -p "$DST"
cp "$SRC"/bench_*.json "$SRC"/*_kernel_stats.csv "$SRC"/*_pmc_hbm_traffic.json "$SRC"/pass_tiles.txt "$DST"/
[ -f gpurun_out/$ROUND/pytest_gpu.log ] && cp gpurun_out/$ROUND/pytest_gpu.log "$DST"/pytest_gpu.log
ls -la "$DST"
