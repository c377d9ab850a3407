#!/bin/bash
# A/B of the persistent launch's modes (tools/multi_bench.cpp), one GPU session:
#   gpurun -- 'bash tools/multi_modes.sh "20 64" "score rows lookup_score lookup_rows" "0 1" improving-inductive-oov-recsys_amd/lib/ab/new.so'
# arguments: list of K, list of MB_MODE, list of MB_PREP, libraries
KS=$1; MODES=$2; PREPS=$3; shift 3
for v in "$@"; do
  for m in $MODES; do
    for p in $PREPS; do
      for K in $KS; do
        echo "== $(basename $v) mode=$m prep=$p K=$K"
        MB_MODE=$m MB_PREP=$p timeout -k 10 120 tools/multi_bench "$v" $K $((1280 / K)) 1024 | grep -v "^single" || { echo "FAILED: $v $m $p $K"; exit 1; }
      done
    done
  done
done
