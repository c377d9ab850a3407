#!/bin/bash
# A/B builds of libmi_oov.so with tools/graph_time.py:  gpurun -- 'bash tools/ab_graph.sh "lsh_embed (rows" lib/ab/*.so'
L=improving-inductive-oov-recsys_amd/lib/libmi_oov.so
cp $L /tmp/libmi_oov_keep.so
pat="$1"; shift
for rep in 1 2; do
  for v in "$@"; do
    cp "$v" $L
    echo "== $(basename $v)"; timeout -k 10 200 python tools/graph_time.py 1000 "$pat" || exit 1
  done
done
cp /tmp/libmi_oov_keep.so $L
