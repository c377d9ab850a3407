#!/bin/bash
# The launch chain of any ops call as the GPU sees it (developer tool; run through gpurun):
#   bash tools/chain.sh '<python statements; SETUP then a line starting with "CALL:">' [ENV=VALUE ...]
# e.g. bash tools/chain.sh 'ids = torch.randint(0, 10**7, (65536,), device=dev); over = torch.zeros(1, dtype=torch.int32, device=dev)
#      CALL: ops.bucket_by_owner(ids, 10**7, 1250000, 8, 10240, over)'
# 40 eager calls (no HIP graph) under rocprofv3 --kernel-trace; per call: the span first start -> last end and every
# kernel's completion-to-completion duration.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
code=$1; shift
for kv in "$@"; do export "$kv"; done
out=gpurun_out/chain
rm -rf $out
python3 - "$code" > /tmp/chain_run.py <<'PY'
import sys
setup, call = [], None
for ln in sys.argv[1].splitlines():
    ln = ln.strip()
    if ln.startswith("CALL:"):
        call = ln[5:].strip()
    elif ln:
        setup.append(ln)
print("import os, sys\nsys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])\nimport torch, mi_oov\nfrom mi_oov import ops\ndev = torch.device('cuda', 0)")
print("\n".join(setup))
print("marker = torch.zeros(1, device=dev)")
print(f"with torch.no_grad():\n    for _i in range(40):\n        marker.add_(1.0)\n        {call}\ntorch.cuda.synchronize()")
PY
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 /tmp/chain_run.py > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob("gpurun_out/chain/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mi_oov::", "")[:110]))
rows.sort()
calls, cur = [], None
for r in rows:  # a call = the kernels between two marker.add_ launches (an elementwise kernel on one element)
    if "CUDAFunctorOnSelf_add" in r[2] or "AddFunctor" in r[2] or ("elementwise" in r[2] and r[1] - r[0] < 4000 and cur is not None and len(cur) == 0):
        if cur:
            calls.append(cur)
        cur = []
    elif cur is not None:
        cur.append(r)
if cur:
    calls.append(cur)
if not calls:
    print("no call boundaries found; kernels seen:", sorted({r[2] for r in rows})[:12])
calls = [c for c in calls[10:] if c]
agg = collections.OrderedDict()
spans = []
for c in calls:
    prev = None
    for s, e, n in c:
        agg.setdefault(n, []).append(((e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3))
        prev = e
    spans.append((c[-1][1] - c[0][0]) / 1e3)
print(f"{len(calls)} calls; span first start -> last end {sum(spans)/max(1,len(spans)):.2f} us")
for n, v in agg.items():
    print(f"  {n[:62]:62s} x{len(v)/max(1,len(calls)):.1f}  duration {sum(d for d, _ in v)/len(v):6.2f} us  gap in front {sum(g for _, g in v)/len(v):5.2f} us")
PY
rm -rf $out $out.log
