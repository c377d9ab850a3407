#!/bin/bash
# Collect the round's judged profile of bench.py on the GPU box (run THROUGH gpurun from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 900 -- "bash tools/profile_bench.sh r03 '--steps 20 --warmup 5' $(git rev-parse --short HEAD)"
# 1. rocprofv3 --kernel-trace --stats of the bench command            -> per-kernel durations
# 2. separate --pmc passes for FETCH_SIZE and WRITE_SIZE              -> HBM traffic per launch
# Summaries land in gpurun_out/<tag>_*; copy <tag>_bench_summary.json / _kernel_stats.csv to profiles/.
set -o pipefail
tag=${1:-r03}
flags=${2:-}
commit=${3:-}   # the build container passes `git rev-parse --short HEAD` (the GPU box has no .git)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
rm -rf $out/${tag}_trace $out/${tag}_pmc_fetch $out/${tag}_pmc_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -- python3 bench.py $flags --no-cpu-baseline > $out/${tag}_trace.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -- python3 bench.py $flags --no-cpu-baseline > $out/${tag}_pmc_fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -- python3 bench.py $flags --no-cpu-baseline > $out/${tag}_pmc_write.log 2>&1 || exit 4
python3 tools/summarize_profile.py --trace $out/${tag}_trace --trace-line $out/${tag}_trace.log \
  --pmc $out/${tag}_pmc_fetch $out/${tag}_pmc_write --pmc-line $out/${tag}_pmc_fetch.log \
  --out $out/${tag}_bench_summary.json --commit "$commit" \
  --command "rocprofv3 --kernel-trace --stats -- python3 bench.py $flags --no-cpu-baseline ; rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py $flags --no-cpu-baseline" > /dev/null || exit 5
(head -1 $out/${tag}_trace/*/*kernel_stats.csv; grep mi_oov $out/${tag}_trace/*/*kernel_stats.csv) | cut -c1-400 > $out/${tag}_bench_kernel_stats.csv
grep '"metric"' $out/${tag}_trace.log | tail -1 > $out/${tag}_bench_line_under_rocprof.json
rm -f $out/${tag}_trace/*/*kernel_trace.csv $out/${tag}_pmc_fetch/*/*counter_collection.csv $out/${tag}_pmc_write/*/*counter_collection.csv  # raw per-launch rows
cat $out/${tag}_bench_kernel_stats.csv
python3 -c "import json; d=json.load(open('$out/${tag}_bench_summary.json')); print(json.dumps(d.get('timed'), indent=1)); print({k: {c: v[c] for c in v if c in ('hbm_bytes_per_batch','algorithmic_bytes_per_batch','traffic_over_algorithmic')} for k, v in d['pmc_per_launch'].items()})"
