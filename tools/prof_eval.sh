#!/bin/bash
# Where the wall time of driver.evaluate goes (run THROUGH gpurun from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/prof_eval.sh r04a'
# 1. plain run of tools/eval_time.py 40000 250 (10.04 M rows)        -> wall seconds
# 2. the same under rocprofv3 --kernel-trace --stats                 -> GPU time by kernel (all four evaluations of the tool)
# 3. the same under cProfile                                         -> host time by function
# Summary: gpurun_out/<tag>_eval_split.txt
set -o pipefail
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
rm -rf $out/${tag}_eval_trace
python3 tools/eval_time.py 40000 250 > $out/${tag}_eval_plain.log 2>&1 || { tail -5 $out/${tag}_eval_plain.log; exit 2; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_eval_trace -- python3 tools/eval_time.py 40000 250 > $out/${tag}_eval_trace.log 2>&1 || { tail -5 $out/${tag}_eval_trace.log; exit 3; }
python3 -c "
import cProfile, pstats, sys, io
sys.argv = ['tools/eval_time.py', '40000', '250']
sys.path.insert(0, 'tools')
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path('tools/eval_time.py', run_name='__main__')
finally:
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue())
" > $out/${tag}_eval_cprofile.log 2>&1 || { tail -5 $out/${tag}_eval_cprofile.log; exit 4; }
{
  echo "# tools/eval_time.py 40000 250  (driver.evaluate: 40 000 positives x 251 candidates = 10.04 M rows, BPR + lsh, 2 M-item vocabulary)"
  echo "## wall time (plain run)"; grep '^{' $out/${tag}_eval_plain.log
  echo "## GPU time by kernel, all four evaluations of the tool together (rocprofv3 --kernel-trace --stats)"
  python3 - <<PY
import csv, glob
f = glob.glob("$out/${tag}_eval_trace/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms in {sum(int(r['Calls']) for r in rows)} launches")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print("  %-90s calls %6s total %9.2f ms" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6))
PY
  echo "## host time by function (cProfile, tottime)"; grep -A 40 'Ordered by' $out/${tag}_eval_cprofile.log | cut -c1-200
} > $out/${tag}_eval_split.txt
rm -rf $out/${tag}_eval_trace
cat $out/${tag}_eval_split.txt | cut -c1-180
