#!/usr/bin/env python3
"""Compiler-reported resources of every kernel in libmi_oov.so (no GPU needed):

    python3 tools/kernel_resources.py profiles/r03_kernel_resources.json

Compiles each csrc/*.hip with the library's flags + -Rpass-analysis=kernel-resource-usage and records, per kernel, the
architected VGPRs, AGPRs, SGPRs, static LDS, scratch and the occupancy the compiler derives.  rocprofv3's per-dispatch
columns (VGPR_Count, LDS_Block_Size in profiles/*_bench_summary.json) are the DISPATCH PACKET's view: the register
count in allocation granules of the architected file only, and the STATIC LDS of the code object -- a kernel that takes
its LDS dynamically (`extern __shared__`: the persistent kernel's 64 KiB table) shows 0 there; this file is the other
half of the picture."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "improving-inductive-oov-recsys_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-mllvm", "-amdgpu-kernarg-preload-count=16", f"-I{os.path.join(ROOT, 'include')}", "-Rpass-analysis=kernel-resource-usage",
         "--cuda-device-only", "-c", "-o", "/dev/null"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [o.replace("void mi_oov::", "").split("(")[0] for o in out]


def main():
    res = {}
    for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
        err = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(CSRC, src)], capture_output=True, text=True).stderr
        cur, names = None, []
        for ln in err.splitlines():
            m = re.search(r"remark: Function Name: (\S+)", ln)
            if m:
                cur = {"file": src}
                names.append(m.group(1))
                res[m.group(1)] = cur
                continue
            m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", ln)
            if m and cur is not None:
                cur[{"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
                     "Occupancy [waves/SIMD]": "occupancy_waves_per_simd", "LDS Size [bytes/block]": "static_lds_bytes"}[m.group(1)]] = int(m.group(2))
    keys = list(res)
    nice = dict(zip(keys, demangle(keys)))
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    out = {"commit": commit, "flags": " ".join(FLAGS[:-4]),
           "note": "compiler view (hipcc -Rpass-analysis=kernel-resource-usage); dynamic LDS is a launch argument and not listed: "
                   "lsh64_persistent_kernel takes (2^H + 2 H) x 256 B (69632 B at H = 8) in the score / rows modes, H x 256 B in the codes mode, "
                   "none in the mean mode; full_sort_kernel 36864 B; bf16_tile_kernel 36864 B (+ 4096 B in the filter form); "
                   "bf16_filter_direct_kernel 16960 B",
           "kernels": {nice[k]: res[k] for k in keys}}
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "kernel_resources.json")
    json.dump(out, open(path, "w"), indent=1)
    print(f"{len(keys)} kernels -> {path}")


if __name__ == "__main__":
    main()
