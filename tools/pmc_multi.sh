#!/bin/bash
# Hardware-counter passes over the persistent multi-batch kernel (run THROUGH gpurun from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/pmc_multi.sh'
# Each pass is its own rocprofv3 --pmc run of tools/multi_bench (the product library through the C ABI: 64 batches of
# 65536 lookups per launch, N = 10 M, 512 distinct id batches, 1 GiB user-row ring); per-kernel means land in
# gpurun_out/pmc_multi.txt.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmc_multi
rm -rf $out && mkdir -p $out
L=improving-inductive-oov-recsys_amd/lib/libmi_oov.so
i=0
for set in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
  "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU" \
  "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
  "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $out/pass$i -- ./tools/multi_bench $L 64 6 1024 > $out/pass$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $out/pass$i.log; }
done
python3 - <<'PY' | tee gpurun_out/pmc_multi.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_multi/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "lsh64" not in k:
            continue
        acc[k[:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:44s} mean {sum(v)/len(v):18.1f}   n={len(v)}")
PY
rm -f $out/pass*/*/*counter_collection.csv
