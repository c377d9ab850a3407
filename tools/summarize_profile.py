#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/...) into the small per-round summaries kept under
profiles/: per-kernel time statistics of the mi_oov kernels and per-launch PMC means.

    python tools/summarize_profile.py --trace gpurun_out/r02_trace --trace-line gpurun_out/r02_trace.json \
        --pmc gpurun_out/r02_pmc_fetch gpurun_out/r02_pmc_write --pmc-line gpurun_out/r02_pmc_fetch.json \
        --out profiles/r02_bench_summary.json

bench.py's timed region is the LAST `roofline.launches` launches of the dominant kernel in the process (the clock
ramp and the warm-up launch the same kernel with other batch counts before it), so the per-launch rows of the
dominant kernel are cut to those: `timed` holds their mean duration / counter values, next to the all-launch
statistics rocprofv3 --stats prints.  `--trace-line` / `--pmc-line` are the JSON lines bench.py printed in those runs.
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


def load_line(path):
    if not path or not os.path.exists(path):
        return None
    for ln in reversed(open(path).read().splitlines()):
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            return json.loads(ln)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace")
    ap.add_argument("--trace-line")
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--pmc-line")
    ap.add_argument("--out", required=True)
    ap.add_argument("--command", default="")
    ap.add_argument("--commit", default="")
    a = ap.parse_args()
    out = {"command": a.command, "commit": a.commit, "kernels": [], "pmc_per_launch": {},
           "resource_columns": "vgpr / sgpr / lds_bytes below are the dispatch packet's: architected VGPRs in allocation granules, STATIC "
                               "LDS only (dynamic LDS -- the persistent kernel's table -- shows 0); the compiler's figures per kernel are "
                               "in profiles/<round>_kernel_resources.json (tools/kernel_resources.py)"}
    tline, pline = load_line(a.trace_line), load_line(a.pmc_line)
    for ln in (tline, pline):
        if ln:
            c = ln["config"]
            out["shape"] = {"items": c["items"], "feat": c["feat"], "dim": c["dim"], "hashes": c["hashes"],
                            "batch": c["batch_per_gpu"]}
    if a.trace:
        for f in glob.glob(os.path.join(a.trace, "**", "*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "mi_oov" in r["Name"]:
                    out["kernels"].append({"kernel": short(r["Name"]), "calls": int(r["Calls"]),
                                           "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                                           "max_us": float(r["MaxNs"]) / 1e3, "stddev_us": float(r["StdDev"]) / 1e3,
                                           "pct_of_gpu_time": float(r["Percentage"])})
        if tline:
            roof = tline["roofline"]
            rows = []
            for f in glob.glob(os.path.join(a.trace, "**", "*kernel_trace.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "mi_oov" in r["Kernel_Name"] and short(r["Kernel_Name"]) == roof["kernel"]:
                        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
            rows.sort()
            timed = rows[-int(roof["launches"]):] if rows else []
            if timed:
                durs = [(e - s) / 1e3 for s, e in timed]
                out["timed"] = {"kernel": roof["kernel"], "launches": len(durs), "batches_per_launch": roof["batches_per_launch"],
                                "avg_us": sum(durs) / len(durs), "min_us": min(durs), "max_us": max(durs),
                                "us_per_batch": sum(durs) / len(durs) / roof["batches_per_launch"],
                                "first_start_to_last_end_us": (timed[-1][1] - timed[0][0]) / 1e3,
                                "bench_line_avg_launch_us": roof["avg_launch_us"], "bench_line_frac": roof["frac"],
                                "all_launches_of_kernel_in_process": len(rows)}
    for d in a.pmc:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(list)
            meta = {}
            for r in csv.DictReader(open(f)):
                if "mi_oov" in r["Kernel_Name"]:
                    k = (short(r["Kernel_Name"]), r["Counter_Name"])
                    agg[k].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
                    meta[short(r["Kernel_Name"])] = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                                                     "lds_bytes": int(r["LDS_Block_Size"]),
                                                     "grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"])}
            for (k, c), v in agg.items():
                v.sort()
                e = out["pmc_per_launch"].setdefault(k, dict(meta[k]))
                vals = [x for _, x in v]
                if pline and k == pline["roofline"]["kernel"]:  # the timed launches only
                    vals = vals[-int(pline["roofline"]["launches"]):]
                    e["batches_per_launch"] = pline["roofline"]["batches_per_launch"]
                    e["launch_selection"] = "timed region (last launches of the process)"
                e[c] = {"mean": sum(vals) / len(vals), "min": min(vals), "max": max(vals), "launches": len(vals)}
    for k, e in out["pmc_per_launch"].items():
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e and "shape" in out and e.get("batches_per_launch"):
            per_batch = (2.0 * e["FETCH_SIZE"]["mean"] + e["WRITE_SIZE"]["mean"]) * 1024.0 / e["batches_per_launch"]
            s = out["shape"]
            alg = (16 + 4 * s["feat"] + 4 * s["dim"] + 4) * s["batch"]
            e["hbm_bytes_per_batch"] = per_batch  # FETCH_SIZE x 2: gfx950 correction for 16-B-per-lane reads
            e["algorithmic_bytes_per_batch"] = alg
            e["traffic_over_algorithmic"] = per_batch / alg
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
