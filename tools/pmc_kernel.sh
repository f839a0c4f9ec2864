#!/bin/bash
# SQ / cache counters of ONE kernel (name substring $1) under a bench.py command line ($2...), one rocprofv3 --pmc pass per counter
# group; prints the mean per launch over the second half of the launches.  Run on the GPU box:
#   gpurun -- 'bash tools/pmc_kernel.sh k_tick_strips_lane --workload batch --steps 10 --warmup 3'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=$1; shift
O=$R/gpurun_out/pmc_$K
mkdir -p $O
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM" "SQ_WAIT_ANY SQ_WAVES" "SQC_DCACHE_REQ SQC_DCACHE_HITS" "SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM"; do
  d=$O/$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/bench.py "$@" --cpu-seconds 0 > $d.log 2>&1 || echo "failed $c"
done
KERNEL=$K OUT=$O python3 - <<'PY'
import csv, glob, os, collections
for f in sorted(glob.glob(os.environ["OUT"]+"/*/*/*counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if os.environ["KERNEL"] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        v=v[len(v)//2:]
        print(k, "n=%d"%len(v), "mean %.0f"%(sum(v)/len(v)))
PY
