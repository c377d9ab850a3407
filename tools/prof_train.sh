#!/bin/bash
# rocprofv3 per-kernel breakdown of the graphed BPR + lsh training step (developer tool; run through gpurun)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/ptrain
rm -rf $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/train_step_time.py --graph > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
tail -2 $out.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ptrain/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("  %-70s calls %6s avg %8.1f us  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
rm -rf $out $out.log
