#!/bin/bash
# Per-kernel breakdown of the fused top-k under its knobs (developer tool; run through gpurun).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
run() {  # tag, env...
  tag=$1; shift
  rm -rf $out/pt_$tag
  env "$@" true
  ( export "$@"; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/pt_$tag -- python3 tools/tune.py --only score_topk --iters 20 > $out/pt_$tag.log 2>&1 ) || { tail -5 $out/pt_$tag.log; return 1; }
  echo "== $tag: $*"
  python3 - <<PY
import csv, glob
f = glob.glob("$out/pt_$tag/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if float(r["TotalDurationNs"]) > 0 and int(r["Calls"]) >= 20:
        print("  %-60s calls %5s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $out/pt_$tag
}
run direct MI_OOV_FILTER_DIRECT=1
run staged MI_OOV_FILTER_DIRECT=0
for w in ${PT_WGS2:-}; do run direct_wgs$w MI_OOV_FILTER_DIRECT=1 MI_OOV_STRIP_WGS2=$w; done
for t in ${PT_TILES:-}; do for w in ${PT_TILES_WGS:-512}; do run direct_tiles${t}_wgs$w MI_OOV_FILTER_DIRECT=1 MI_OOV_FILTER_TILES=$t MI_OOV_STRIP_WGS2=$w; done; done
