#!/bin/bash
# Matrix-core counters of the MFMA kernels on the path (run THROUGH gpurun from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/pmc_mfma.sh'
# one rocprofv3 --pmc pass over tools/tune.py's full-sort / fused top-k / dhe-MLP cases -> gpurun_out/pmc_mfma.txt
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmc_mfma
rm -rf $out && mkdir -p $out
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU"; do
  i=$((i+1))
  for c in full_sort score_topk "dhe MLP"; do
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$out/pass${i}_${c// /_}" -- python3 tools/tune.py --only "$c" --iters 10 > "$out/pass${i}_${c// /_}.log" 2>&1 || { echo "pass $i $c failed"; tail -3 "$out/pass${i}_${c// /_}.log"; }
  done
done
python3 - <<'PY' | tee gpurun_out/pmc_mfma.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_mfma/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "full_sort_kernel" not in k and "bf16_" not in k:
            continue
        acc[k.split("(")[0].replace("void mi_oov::", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("# full_sort_kernel<VEC, EPI>: EPI 0 = scores stored, 1-3 = Linear (+GELU / +sigmoid), 4 = tile maxima, 5 = filter (f32 MFMA);")
print("# bf16_tile_kernel<4, MASKED> = first top-k pass (tile maxima), bf16_filter_direct_kernel<MASKED> = second pass (bf16 MFMA)")
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} mean {sum(v)/len(v):16.1f}   n={len(v)}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d:
        mf, bz = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(d["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(d["SQ_BUSY_CYCLES"]) / len(d["SQ_BUSY_CYCLES"])
        print(f"   -> MFMA busy / SQ busy cycles = {mf / bz:.3f}")
PY
rm -f gpurun_out/pmc_mfma/*/*/*counter_collection.csv
