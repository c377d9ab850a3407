#!/bin/bash
# tools/topk_chain.sh for several builds of libmi_oov.so (developer tool; run through gpurun):
#   bash tools/topk_chain_libs.sh <D> <prepared> lib/ab/a.so lib/ab/b.so ...
cd "$GRAFT_REPO_ROOT" || exit 1
D=$1; prep=$2; shift; shift
L=improving-inductive-oov-recsys_amd/lib/libmi_oov.so
cp $L /tmp/libmi_oov_keep.so
for v in "$@"; do
  cp "$v" $L
  echo "== $(basename "$v" .so)  (D = $D)"
  bash tools/topk_chain.sh $D $prep | grep -E "calls;|filter|finalize|tile_kernel"
done
cp /tmp/libmi_oov_keep.so $L
