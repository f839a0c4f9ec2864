#!/bin/bash
# kernel timeline of a pipelined run (start / end of every dispatch): gpurun -- 'bash tools/trace_pipeline.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/trace_pipe
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_pipe -- python3 $R/tools/exp_host_rate.py > $R/gpurun_out/trace_pipe.log 2>&1
python3 - <<'PY'
import csv, glob, os
R=os.environ["GRAFT_REPO_ROOT"]
f=sorted(glob.glob(R+"/gpurun_out/trace_pipe/*/*kernel_trace.csv"))[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-70:-14]
t0=int(rows[0]["Start_Timestamp"])
for r in rows:
    n=r["Kernel_Name"]; n=n[n.find("k_"):][:22] if "k_" in n else n[:22]
    print("%-22s q%-3s %9.2f .. %9.2f  (%.2f us)" % (n, r.get("Queue_Id","?"), (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
