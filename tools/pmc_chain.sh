#!/bin/bash
# SQ instruction / cycle counters of k_tick_chain as a launch of its own (bench.py --tick-mode 3), one rocprofv3 --pmc pass per
# counter pair; prints the mean per launch.  Run on the GPU box: gpurun -- 'bash tools/pmc_chain.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_chain
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
  d=$R/gpurun_out/pmc_chain/$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/bench.py --tick-mode 3 --steps 20 --warmup 5 --cpu-seconds 0 --no-api --parity-ticks 0 > $d.log 2>&1 || echo "failed $c"
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R+"/gpurun_out/pmc_chain/*/*/*counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_tick_chain" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        v=v[len(v)//2:]
        print(k, "n=%d"%len(v), "mean %.0f"%(sum(v)/len(v)))
PY
