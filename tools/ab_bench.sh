#!/bin/bash
# A/B two builds of libmi_oov.so with bench.py, alternating, on one box:
#   gpurun -- 'bash tools/ab_bench.sh improving-inductive-oov-recsys_amd/lib/ab/libmi_oov_old.so improving-inductive-oov-recsys_amd/lib/ab/libmi_oov_new.so 3'
L=improving-inductive-oov-recsys_amd/lib/libmi_oov.so
for rep in $(seq 1 ${3:-3}); do
  for v in "$1" "$2"; do
    cp "$v" $L
    timeout -k 10 200 python bench.py --no-cpu-baseline | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v'.split('/')[-1], round(d['roofline']['avg_launch_us'],3), round(d['value']/1e9,3))" || exit 1
  done
done
