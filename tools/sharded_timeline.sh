#!/bin/bash
# What a sustained sharded run's GPU time is made of (developer tool; run through gpurun):
#   bash tools/sharded_timeline.sh [bench flags]
# rocprofv3 --kernel-trace of `bench.py --force-sharded --steps 20 --sustained-steps 640 ...`; the LAST 10 exchanges'
# kernels (the sustained region): per kernel total time, and how much of the region's span has 0 / 1 / 2 / 3+ kernels running.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/shtl
rm -rf $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --force-sharded --steps 20 --warmup 5 --no-cpu-baseline --no-also --sustained-steps 640 "$@" > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections, json
rows = []
for f in glob.glob("gpurun_out/shtl/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mi_oov::", "")[:70]))
rows.sort()
# the sustained region = the last 640 requester launches' neighbourhood: take the window that holds the last 10 exchanges
# (10 bucket launches of 64 batches) up to the last requester kernel, before the replicated run starts
req = [i for i, r in enumerate(rows) if "lsh64_persistent_kernel<8, 2" in r[2]]
buck = [i for i, r in enumerate(rows) if "bucket_by_owner" in r[2]]
line = [l for l in open("gpurun_out/shtl.log") if l.startswith("{")]
if line:
    s = json.loads(line[-1])["sharded"]
    print("bench line: 20 steps %.2f us/step; sustained %.2f us/step (%d exchanges)" % (s["us_per_step_hip_events"], s["sustained"]["us_per_step_hip_events"], s["sustained"]["exchanges"]))
if len(req) >= 10 and len(buck) >= 10:
    # the sustained run is the last group of 10 bucket launches that precede the final requester launch of the sharded phase
    last_req = req[-1]
    bs = [i for i in buck if i < last_req][-10:]
    lo, hi = rows[bs[0]][0], rows[last_req][1]
    win = [r for r in rows if r[0] >= lo and r[1] <= hi]
    span = (hi - lo) / 1e3
    agg = collections.Counter()
    for s_, e_, n in win:
        agg[n] += (e_ - s_) / 1e3
    print(f"sustained window: {span:.1f} us for 640 steps = {span / 640:.2f} us per step; {len(win)} kernels; sum of kernel time {sum(agg.values()):.1f} us")
    for n, t in agg.most_common(10):
        print(f"  {n:72s} {t:9.1f} us  ({t / 640:.2f} per step)")
    ev = sorted([(s_, 1) for s_, _, _ in win] + [(e_, -1) for _, e_, _ in win])
    depth, prev, occ = 0, lo, collections.Counter()
    for t, d in ev:
        occ[min(depth, 3)] += (t - prev) / 1e3
        depth, prev = depth + d, t
    print("  time with 0 / 1 / 2 / 3+ kernels running: " + " / ".join(f"{occ[i]:.1f}" for i in range(4)) + " us")
PY
rm -rf $out $out.log
