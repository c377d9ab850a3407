#!/bin/bash
# rocprof per-kernel breakdown of one tools/tune.py case (developer tool; run through gpurun): bash tools/prof_case.sh "<case>"
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pcase
rm -rf $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/tune.py --only "$1" --iters 5 > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
grep -o '"us_per_launch": [0-9.]*' $out.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/pcase/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.5:
        print("  %-60s calls %5s avg %10.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
rm -rf $out $out.log
