#!/bin/bash
# A/B of library builds on ONE box: tools/ab_bench.sh <runs> <bench args...> -- lib1.so lib2.so ...   (alternating, median-free: prints every run)
runs=$1; shift
args=()
while [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
for r in $(seq $runs); do
  for lib in "$@"; do
    NUSLAM_HIP_LIB=$PWD/$lib python bench.py --cpu-seconds 0 --no-api --parity-ticks 0 "${args[@]}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_us']
print('%-60s %9.0f /s  %.2f us/step  chain %.2f rank %.2f' % ('$lib', d['value'], d['ms_per_step']*1e3, k.get('tick_chain') or 0, k.get('tick_rank') or 0))"
  done
done
