#!/bin/bash
# A/B of the fused top-k under its tuning knobs (developer tool, GPU box only).
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -q -x -m gpu -k "topk or knn or full_sort" > gpurun_out/ab_topk_tests.log 2>&1 || { tail -30 gpurun_out/ab_topk_tests.log; exit 1; }
tail -2 gpurun_out/ab_topk_tests.log
for cfg in ${AB_CFGS:-1024 2048 4096 512}; do
  set -- $cfg 0
  echo "== STRIP_WGS=$1 STRIP_BUF=$2"
  MI_OOV_STRIP_WGS=$1 MI_OOV_STRIP_BUF=$2 timeout -k 10 200 python tools/tune.py --only score_topk 2>&1 | grep -E "score_topk|rror" || true
done
