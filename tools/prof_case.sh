#!/bin/bash
# rocprof per-kernel breakdown of one tools/tune.py case (developer tool; run through gpurun):
#   bash tools/prof_case.sh "<case substring>" [ENV=VALUE ...]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
c=$1; shift
for kv in "$@"; do export "$kv"; done
out=gpurun_out/pcase
rm -rf $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/tune.py --only "$c" --iters 20 > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
grep -o '"case": "[^"]*", "us_per_launch": [0-9.]*' $out.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/pcase/*/*kernel_stats.csv")[0]
for r in sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"])):
    if int(r["Calls"]) >= 20:
        print("  %-90s calls %5s avg %8.2f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $out $out.log
