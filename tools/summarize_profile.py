#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/...) into the small per-round summaries kept under
profiles/: per-kernel time statistics of the mi_oov kernels and per-launch PMC means.

    python tools/summarize_profile.py --trace gpurun_out/r01_bench_trace --pmc gpurun_out/r01_pmc_fetch \
        gpurun_out/r01_pmc_write --out profiles/r01_bench_summary.json
"""
import argparse
import collections
import csv
import glob
import json
import os


def short(name):
    name = name.replace("void mi_oov::", "")
    return name.split("(")[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace")
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--out", required=True)
    ap.add_argument("--command", default="")
    a = ap.parse_args()
    out = {"command": a.command, "kernels": [], "pmc_per_launch": {}}
    if a.trace:
        for f in glob.glob(os.path.join(a.trace, "**", "*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "mi_oov" in r["Name"]:
                    out["kernels"].append({"kernel": short(r["Name"]), "calls": int(r["Calls"]),
                                           "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                                           "max_us": float(r["MaxNs"]) / 1e3, "stddev_us": float(r["StdDev"]) / 1e3,
                                           "pct_of_gpu_time": float(r["Percentage"])})
    for d in a.pmc:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(list)
            meta = {}
            for r in csv.DictReader(open(f)):
                if "mi_oov" in r["Kernel_Name"]:
                    k = (short(r["Kernel_Name"]), r["Counter_Name"])
                    agg[k].append(float(r["Counter_Value"]))
                    meta[short(r["Kernel_Name"])] = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                                                     "lds_bytes": int(r["LDS_Block_Size"]),
                                                     "grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"])}
            for (k, c), v in agg.items():
                e = out["pmc_per_launch"].setdefault(k, dict(meta[k]))
                e[c] = {"mean": sum(v) / len(v), "min": min(v), "max": max(v), "launches": len(v)}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
