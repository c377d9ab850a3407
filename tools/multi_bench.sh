#!/bin/bash
# A/B of libmi_oov.so variants on the multi-batch persistent launch (tools/multi_bench.cpp), one GPU session:
#   gpurun -- 'bash tools/multi_bench.sh "64 20" improving-inductive-oov-recsys_amd/lib/ab/*.so'
# first argument: space-separated list of K (batches per launch); launches per timing = 1280 / K
KS=$1; shift
for v in "$@"; do
  for K in $KS; do
    echo "== $v K=$K"
    timeout -k 10 120 tools/multi_bench "$v" $K $((1280 / K)) 1024 | grep -v "^single" || { echo "FAILED: $v"; exit 1; }
  done
done
