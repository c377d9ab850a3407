#!/bin/bash
# Per-kernel durations of every kernel on the path (run THROUGH gpurun from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/profile_all.sh r01'
# rocprofv3 --kernel-trace --stats over tools/tune.py; the summary (kernel stats + the tool's own per-case lines)
# lands in gpurun_out/<tag>_all_kernels.json -- copy it to profiles/.
set -o pipefail
tag=${1:-r03}
commit=${2:-}   # `git rev-parse --short HEAD` of the build container (the GPU box has no .git)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
rm -rf $out/${tag}_all_trace
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_all_trace -- python3 tools/tune.py > $out/${tag}_all_cases.log 2>&1 || { tail -5 $out/${tag}_all_cases.log; exit 2; }
python3 tools/summarize_profile.py --trace $out/${tag}_all_trace --out $out/${tag}_all_kernels.json \
  --command "rocprofv3 --kernel-trace --stats -- python3 tools/tune.py" --commit "$commit" > /dev/null
rm -f $out/${tag}_all_trace/*/*kernel_trace.csv
python3 - <<PY
import json
p = "$out/${tag}_all_kernels.json"
d = json.load(open(p))
d["note"] = ("kernel durations from rocprofv3 --kernel-trace --stats; cases = tools/tune.py's own HIP-event timing of a graph "
             "replay with the algorithmic bytes/flops of SURVEY section 8(d); MI355X, N=10M, B=65536 unless the case says otherwise")
d["cases"] = [json.loads(l) for l in open("$out/${tag}_all_cases.log") if l.startswith("{")]
json.dump(d, open(p, "w"), indent=1)
for c in d["cases"]:
    print(c)
PY
