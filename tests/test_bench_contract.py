"""CPU: the bench line's schema (driver contract) on the committed round-1 line, the PMC traffic lookup against
the committed rocprof summary, and bench.py's defaults.  bench.py itself needs an MI355X and is run by the driver."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_bench_line_has_the_contract_fields():
    line = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_line.json")))
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(line[key], typ), key
    assert line["vs_baseline"] is None and line["scaling"] == "weak" and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    # achieved = algorithmic bytes per launch / average launch duration
    want = roof["bytes_per_lookup"] * roof["lookups_per_launch"] / (roof["avg_launch_us"] * 1e-6) / 1e9
    assert abs(roof["achieved"] - want) < 1e-6 * want
    assert roof["traffic"] is None or 0.9 < roof["traffic"] / (roof["bytes_per_lookup"] * roof["lookups_per_launch"]) < 1.2
    cpu = line["cpu_baseline"]
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["unit"] == line["unit"] and cpu["sample"]
    # value is the whole job: lookups of all steps over the timed region
    assert abs(line["value"] - line["n_gpus"] * roof["lookups_per_launch"] / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]


def test_pmc_traffic_and_rocprof_duration_agree_with_the_line():
    bench = _bench()
    line = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_line.json")))
    summary = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_summary.json")))
    kernel = line["roofline"]["kernel"]
    assert bench.pmc_traffic(kernel) is not None
    rocprof = [k for k in summary["kernels"] if k["kernel"] == kernel]
    assert rocprof and abs(rocprof[0]["avg_us"] - line["roofline"]["avg_launch_us"]) < 0.05 * rocprof[0]["avg_us"]


def test_defaults_finish_in_minutes(monkeypatch):
    bench = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.gpus == 1 and a.steps * 10e-6 < 1.0 and a.steps >= 2000 and a.batch == 65536 and a.items == 10_000_000 and a.hashes == 8
