for st in ${STRIDES:-8 4}; do echo "== stride $st"; MI_OOV_TOPK_STRIDE=$st timeout -k 10 200 python tools/tune.py --only score_topk 2>&1 | grep -E "score_topk|rror"; done
