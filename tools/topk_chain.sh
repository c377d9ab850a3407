#!/bin/bash
# The fused top-k's launch CHAIN as the GPU sees it (developer tool; run through gpurun):
#   bash tools/topk_chain.sh [D=64] [prepared=0|1] [ENV=VALUE ...]
# eager calls of ops.score_topk (no HIP graph) under rocprofv3 --kernel-trace; per call: every kernel's duration, the gap
# in front of it (previous kernel's end -> this kernel's start) and the span first start -> last end.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
D=${1:-64}; prep=${2:-0}; shift; shift
for kv in "$@"; do export "$kv"; done
out=gpurun_out/tchain
rm -rf $out
cat > /tmp/topk_chain_run.py <<PY
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, mi_oov
from mi_oov import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
U = torch.randn((4096, $D), generator=g, device=dev)
E = torch.randn((50000, $D), generator=g, device=dev)
cat = ops.TopkCatalogue(E) if $prep else E
for i in range(30):
    ops.score_topk(U, cat, 20, 1)
torch.cuda.synchronize()
PY
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 /tmp/topk_chain_run.py > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob("gpurun_out/tchain/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "mi_oov" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void mi_oov::", "")[:44]))
rows.sort()
# a call = the kernels from one to_bf16_norm / first kernel to the finalize kernel
calls, cur = [], []
for r in rows:
    cur.append(r)
    if "finalize" in r[2]:
        calls.append(cur); cur = []
calls = calls[10:]  # warm
agg = collections.OrderedDict()
spans, sums = [], []
for c in calls:
    prev_end = None
    for s, e, n in c:
        d = agg.setdefault(n, [[], []])
        d[0].append((e - s) / 1e3)
        d[1].append(0.0 if prev_end is None else (s - prev_end) / 1e3)
        prev_end = e
    spans.append((c[-1][1] - c[0][0]) / 1e3)
    sums.append(sum((e - s) / 1e3 for s, e, _ in c))
print(f"{len(calls)} calls; span first start -> last end {sum(spans)/len(spans):.1f} us; sum of kernel durations {sum(sums)/len(sums):.1f} us")
for n, (du, ga) in agg.items():
    print(f"  {n:46s} duration {sum(du)/len(du):6.2f} us   gap in front {sum(ga)/len(ga):5.2f} us")
if len(calls) > 1:
    between = [(calls[i + 1][0][0] - calls[i][-1][1]) / 1e3 for i in range(len(calls) - 1)]
    print(f"  between calls (host-side: allocation, 5 launches): {sum(between)/len(between):.1f} us")
PY
rm -rf $out $out.log
