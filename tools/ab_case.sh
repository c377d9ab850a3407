#!/bin/bash
# time one tools/tune.py case with each variant library:  gpurun -- 'bash tools/ab_case.sh "score_topk degenerate" lib/ab/*.so'
L=improving-inductive-oov-recsys_amd/lib/libmi_oov.so
cp $L /tmp/libmi_oov_keep.so
pat="$1"; shift
for v in "$@"; do
  cp "$v" $L
  echo "== $(basename $v)"; timeout -k 10 300 python tools/tune.py --only "$pat" --iters 5 | sed 's/"units_per_launch.*//'
done
cp /tmp/libmi_oov_keep.so $L
