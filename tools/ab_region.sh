#!/bin/bash
# A/B builds of libmi_oov.so on the driver's 20-step region (tools/region_cost.py: host segments + HIP events), one box:
#   gpurun -- 'bash tools/ab_region.sh improving-inductive-oov-recsys_amd/lib/ab/*.so'
L=improving-inductive-oov-recsys_amd/lib/libmi_oov.so
cp $L /tmp/libmi_oov_keep.so
for v in "$@"; do
  cp "$v" $L
  echo "== $v"
  timeout -k 10 200 python tools/region_cost.py 2>/dev/null | tail -13 || { cp /tmp/libmi_oov_keep.so $L; exit 1; }
done
cp /tmp/libmi_oov_keep.so $L
