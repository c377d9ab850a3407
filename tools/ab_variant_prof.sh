#!/bin/bash
# Per-kernel rocprof breakdown of the fused top-k for each variant library (developer tool; run through gpurun):
#   gpurun -- 'bash tools/ab_variant_prof.sh improving-inductive-oov-recsys_amd/lib/ab/*.so'
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
L=improving-inductive-oov-recsys_amd/lib/libmi_oov.so
cp $L /tmp/libmi_oov_keep.so
out=gpurun_out
for v in "$@"; do
  cp "$v" $L
  tag=$(basename $v .so)
  rm -rf $out/pv_$tag
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/pv_$tag -- python3 tools/tune.py --only score_topk --iters 20 > $out/pv_$tag.log 2>&1 || { tail -5 $out/pv_$tag.log; cp /tmp/libmi_oov_keep.so $L; exit 1; }
  echo "== $tag  $(grep -o '"us_per_launch": [0-9.]*' $out/pv_$tag.log)"
  python3 - <<PY
import csv, glob
f = glob.glob("$out/pv_$tag/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if int(r["Calls"]) >= 20 and "bf16_tile" in r["Name"]:
        print("  %-50s avg %9.1f us" % (r["Name"][:50], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $out/pv_$tag $out/pv_$tag.log
done
cp /tmp/libmi_oov_keep.so $L
