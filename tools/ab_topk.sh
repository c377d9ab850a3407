#!/bin/bash
# Fused top-k: parity tests, then the k = 20 case under each strip-count setting (developer tool, GPU box only).
#   gpurun -- 'AB_CFGS="1024 2048" bash tools/ab_topk.sh'
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -q -x -m gpu -k "topk or knn or full_sort" > gpurun_out/ab_topk_tests.log 2>&1 || { tail -30 gpurun_out/ab_topk_tests.log; exit 1; }
tail -2 gpurun_out/ab_topk_tests.log
for w in ${AB_CFGS:-1024 2048 4096 512}; do
  echo "== MI_OOV_STRIP_WGS=$w"
  MI_OOV_STRIP_WGS=$w timeout -k 10 200 python tools/tune.py --only "score_topk k=20" 2>&1 | grep -E "score_topk|rror" || true
done
