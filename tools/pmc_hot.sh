#!/bin/bash
# Hardware-counter passes over the hot kernel (run THROUGH gpurun from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/pmc_hot.sh'
# Each pass is its own rocprofv3 --pmc run of tools/microbench (MB_LIB_ONLY: the product kernels through
# the C ABI at the BASELINE shape); per-kernel means land in gpurun_out/pmc_hot.txt.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmc_hot
rm -rf $out && mkdir -p $out
export MB_LIB_ONLY=1
i=0
for set in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
  "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $out/pass$i -- ./tools/microbench > $out/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/pass$i.log; }
done
python3 - <<'PY' | tee gpurun_out/pmc_hot.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_hot/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "lsh64" not in k and "rowdot" not in k:
            continue
        acc[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:44s} mean {sum(v)/len(v):16.1f}   n={len(v)}")
PY
