#!/bin/bash
# A/B builds of libmi_oov.so with bench.py (graph replay: +-0.01 us), alternating, on one box:
#   gpurun -- 'bash tools/ab_bench.sh 2 improving-inductive-oov-recsys_amd/lib/ab/*.so'
L=improving-inductive-oov-recsys_amd/lib/libmi_oov.so
cp $L /tmp/libmi_oov_keep.so
reps=$1; shift
for rep in $(seq 1 $reps); do
  for v in "$@"; do
    cp "$v" $L
    timeout -k 10 200 python bench.py --no-cpu-baseline 2>/tmp/ab_err.log | tail -1 > /tmp/ab_line.json || { tail -5 /tmp/ab_err.log; exit 1; }
    [ -s /tmp/ab_line.json ] || { echo "no output from $v"; tail -8 /tmp/ab_err.log; exit 1; }
    cat /tmp/ab_line.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-24s %.3f us  %.3f G/s' % ('$v'.split('/')[-1], d['roofline']['avg_launch_us'], d['value']/1e9))" || exit 1
  done
done
cp /tmp/libmi_oov_keep.so $L
