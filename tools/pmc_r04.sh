#!/bin/bash
# Round-4 counter evidence (run THROUGH gpurun from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/pmc_r04.sh [topk|x3|all]'
# 1. fused top-k: matrix-core busy / wait counters of bf16_filter_direct_kernel<...,1> (D = 64) and <...,2> (D = 128), of the
#    first pass and of the finalize kernel                                    -> gpurun_out/r04_pmc_mfma.txt
# 2. linear_x3_fast_kernel<1,true> at 65536 x 1024 -> 512 on random and on zero operands (matrix busy, VALU, LDS conflicts,
#    GRBM_GUI_ACTIVE / kernel time = effective clock)                          -> gpurun_out/r04_pmc_x3.txt
# Every pass is its own rocprofv3 --pmc run (no trace domains beside it).
set -o pipefail
what=${1:-all}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
L=improving-inductive-oov-recsys_amd/lib/libmi_oov.so

summarize() {  # dir, filter substring list (comma), title
python3 - "$1" "$2" <<'PY'
import csv, glob, collections, sys
root, want = sys.argv[1], sys.argv[2].split(",")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    tag = f[len(root):].strip("/").split("/")[0]
    tag = tag.split("_", 1)[1] if "_" in tag else ""
    seen = set()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if not any(w in k for w in want):
            continue
        name = (tag + " :: " if tag else "") + k.split("(")[0].replace("void mi_oov::", "")
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
        if "Start_Timestamp" in row and "End_Timestamp" in row and (row["Dispatch_Id"], name) not in seen:
            seen.add((row["Dispatch_Id"], name))
            dur[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} mean {sum(v)/len(v):18.1f}   n={len(v)}")
    m = lambda c: sum(d[c]) / len(d[c]) if c in d else None
    if m("SQ_VALU_MFMA_BUSY_CYCLES") and m("SQ_BUSY_CYCLES"):
        print(f"   -> MFMA busy / SQ busy cycles = {m('SQ_VALU_MFMA_BUSY_CYCLES') / m('SQ_BUSY_CYCLES'):.3f}")
    if m("SQ_WAIT_INST_ANY") and m("SQ_WAVE_CYCLES"):
        print(f"   -> waves waiting on any instruction / wave cycles = {m('SQ_WAIT_INST_ANY') / m('SQ_WAVE_CYCLES'):.3f}")
    if k in dur and dur[k]:
        us = sum(dur[k]) / len(dur[k])
        print(f"   kernel duration under the counters: {us:.1f} us (n={len(dur[k])})")
        if m("GRBM_GUI_ACTIVE"):
            print(f"   -> GRBM_GUI_ACTIVE / duration = {m('GRBM_GUI_ACTIVE') / us / 1e3:.3f} GHz effective clock")
PY
}

if [ "$what" = all ] || [ "$what" = topk ]; then
  out=gpurun_out/r04_pmc_mfma_raw
  rm -rf $out && mkdir -p $out
  i=0
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$out/pass$i" -- python3 tools/tune.py --only "score_topk k=20" --iters 10 > "$out/pass$i.log" 2>&1 || { echo "topk pass $i failed"; tail -3 "$out/pass$i.log"; }
  done
  { echo "# rocprofv3 --pmc <set> -- python3 tools/tune.py --only 'score_topk k=20' --iters 10  (4096 x 50 000, k = 20; D = 64 and D = 128)";
    echo "# bf16_filter_direct_kernel<MASKED, TILES, KH>: KH = 1 -> D <= 64, KH = 2 -> 64 < D <= 128";
    summarize $out "bf16_,topk_finalize,tile_kth,to_bf16"; } | tee gpurun_out/r04_pmc_mfma.txt
  rm -rf $out
fi

if [ "$what" = all ] || [ "$what" = x3 ]; then
  out=gpurun_out/r04_pmc_x3_raw
  rm -rf $out && mkdir -p $out
  export XB_SKIP_F32=1
  for z in random zero; do
    if [ $z = zero ]; then export XB_ZERO=1; else unset XB_ZERO; fi
    i=0
    for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16" \
               "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
      i=$((i+1))
      timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d "$out/pass${i}_$z" -- ./tools/x3_bench $L 65536 1024 512 1 10 > "$out/pass${i}_$z.log" 2>&1 || { echo "x3 pass $i $z failed"; tail -3 "$out/pass${i}_$z.log"; }
    done
    # the same command without counters: the launch time the counters are to be read against
    ./tools/x3_bench $L 65536 1024 512 1 20 > "$out/plain_$z.log" 2>&1
  done
  unset XB_ZERO
  { echo "# rocprofv3 --pmc <set> -- ./tools/x3_bench libmi_oov.so 65536 1024 512 1 10   (XB_ZERO=1 for the 'zero' rows: all operands 0)";
    echo "# plain runs (no counters):"; grep -h linear_x3 $out/plain_random.log | sed 's/^/#   random: /'; grep -h linear_x3 $out/plain_zero.log | sed 's/^/#   zero:   /';
    summarize $out "linear_x3"; } | tee gpurun_out/r04_pmc_x3.txt
  rm -rf $out
fi

# 3. the gather kernels on ONE large call (2 M lookups, tools/large_calls.py): what bounds slsh64_kernel at 24 planes, lsh64g at
#    16 / 32 planes and the persistent kernel -- vector-ALU activity against wave cycles and waiting
#                                                                              -> gpurun_out/r04_pmc_lsh.txt
if [ "$what" = all ] || [ "$what" = lsh ]; then
  out=gpurun_out/r04_pmc_lsh_raw
  rm -rf $out && mkdir -p $out
  i=0
  for set in "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$out/pass$i" -- python3 tools/large_calls.py > "$out/pass$i.log" 2>&1 || { echo "lsh pass $i failed"; tail -3 "$out/pass$i.log"; }
  done
  { echo "# rocprofv3 --pmc <set> -- python3 tools/large_calls.py   (one call of 2 097 152 lookups per op, 10 M x 64 table)";
    echo "# SQ_* are summed over the chip's 1024 SIMDs; SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES x (1024 / 4 per CU ...) is read as a RATIO between kernels";
    summarize $out "slsh64_kernel,lsh64g_kernel,lsh64_persistent,lsh_kernel,row_copy,gather_mean,rowdot"; } | tee gpurun_out/r04_pmc_lsh.txt
  rm -rf $out
fi
