#!/bin/bash
# Per-kernel breakdown of the fused top-k for several builds of libmi_oov.so (developer tool; run through gpurun):
#   gpurun -- 'bash tools/prof_topk_libs.sh improving-inductive-oov-recsys_amd/lib/ab/*.so'
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
L=improving-inductive-oov-recsys_amd/lib/libmi_oov.so
cp $L /tmp/libmi_oov_keep.so
for v in "$@"; do
  cp "$v" $L
  tag=$(basename "$v" .so)
  rm -rf $out/pt_$tag
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/pt_$tag -- python3 tools/tune.py --only score_topk --iters 20 > $out/pt_$tag.log 2>&1 || { tail -5 $out/pt_$tag.log; cp /tmp/libmi_oov_keep.so $L; exit 1; }
  echo "== $tag"
  python3 - <<PY
import csv, glob
f = glob.glob("$out/pt_$tag/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if float(r["TotalDurationNs"]) > 0 and int(r["Calls"]) >= 20:
        print("  %-60s calls %5s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $out/pt_$tag
done
cp /tmp/libmi_oov_keep.so $L
