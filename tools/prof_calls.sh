#!/bin/bash
# Per-launch durations of kernels matching a pattern in the fused top-k (developer tool; run through gpurun):
#   gpurun -- 'bash tools/prof_calls.sh to_bf16'
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pc
rm -rf $out
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/tune.py --only "score_topk k=20" --iters 10 > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 - "$1" <<'PY'
import csv, glob, sys
pat = sys.argv[1]
f = glob.glob("gpurun_out/pc/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
for r in rows[-12:]:
    print(r["Kernel_Name"][:40], "grid", r.get("Grid_Size_X", r.get("Grid_Size", "?")), "dur us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
rm -rf $out $out.log
