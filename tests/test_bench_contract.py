"""CPU: the bench line's schema (driver contract) on the committed round-4 line, the PMC traffic lookup against
the committed rocprof summary, and bench.py's defaults.  bench.py itself needs an MI355X and is run by the driver."""
import importlib.util
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _line(name="r04_bench_line.json"):
    return json.load(open(os.path.join(ROOT, "profiles", name)))


def test_committed_bench_line_has_the_contract_fields():
    line = _line()
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(line[key], typ), key
    assert line["vs_baseline"] is None and line["scaling"] == "weak" and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert "pipelined" not in line  # round 1's host-clock multi-stream figure is gone: the persistent launch replaced it
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    # achieved = algorithmic bytes per launch / average launch duration
    want = roof["bytes_per_lookup"] * roof["lookups_per_launch"] / (roof["avg_launch_us"] * 1e-6) / 1e9
    assert abs(roof["achieved"] - want) < 1e-6 * want
    assert roof["bytes_per_lookup"] == 532 and roof["bytes_moved_per_lookup"] == 524  # SURVEY 8d's formula / what moves
    assert roof["lookups_per_launch"] == line["config"]["batch_per_gpu"] * roof["batches_per_launch"]
    assert roof["launches"] * roof["batches_per_launch"] == line["steps"]
    assert roof["traffic"] is None or 0.9 < roof["traffic"] / (roof["bytes_per_lookup"] * roof["lookups_per_launch"]) < 1.2
    # the user rows come from a ring far larger than the 256 MiB Infinity Cache
    assert line["config"]["user_row_ring_bytes"] >= 4 * (256 << 20)
    cpu = line["cpu_baseline"]
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["unit"] == line["unit"] and cpu["sample"]
    # value is the whole job: lookups of all steps over the timed region (host clock)
    total = line["n_gpus"] * line["config"]["batch_per_gpu"] * line["steps"]
    assert abs(line["value"] - total / (line["ms_per_step"] * 1e-3 * line["steps"])) < 1e-6 * line["value"]
    # the kernel cannot be slower than the region it was timed in
    assert roof["us_per_batch"] <= line["ms_per_step"] * 1e3


def test_pmc_traffic_and_rocprof_duration_agree_with_the_line():
    bench = _bench()
    line, under = _line(), _line("r04_bench_line_under_rocprof.json")
    summary = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_summary.json")))
    kernel = line["roofline"]["kernel"]
    c = line["config"]
    args = types.SimpleNamespace(items=c["items"], feat=c["feat"], dim=c["dim"], hashes=c["hashes"], batch=c["batch_per_gpu"])
    per_batch, src = bench.pmc_traffic(kernel, args)
    assert src == "profiles/r04_bench_summary.json"
    alg = line["roofline"]["bytes_per_lookup"] * c["batch_per_gpu"]
    assert 0.95 < per_batch / alg < 1.05  # FETCH x 2 + WRITE ~ algorithmic bytes: no wasted re-reads
    other = types.SimpleNamespace(items=c["items"] // 2, feat=c["feat"], dim=c["dim"], hashes=c["hashes"], batch=c["batch_per_gpu"])
    assert bench.pmc_traffic(kernel, other) == (None, None)  # another shape: no figure rather than a stale one
    timed = summary["timed"]
    assert timed["kernel"] == kernel and timed["launches"] == under["roofline"]["launches"]
    assert timed["batches_per_launch"] == line["roofline"]["batches_per_launch"]
    # rocprof's kernel duration of the timed launch against the bench's HIP events: the plain run within 7 % (the event
    # pair around ONE launch of ~106 us also holds the dispatch and the two event packets, 3-6 us, and two runs on
    # different boxes differ by ~3 %), the run under the profiler (whose event records carry the tool's own overhead
    # around a single launch) within 15 %
    assert abs(timed["avg_us"] - line["roofline"]["avg_launch_us"]) < 0.07 * timed["avg_us"]
    assert 0 <= under["roofline"]["avg_launch_us"] - timed["avg_us"] < 0.15 * timed["avg_us"]
    assert timed["avg_us"] * 1e-6 > 0  # and the roofline fraction it implies meets north_star's 0.70
    frac = line["roofline"]["bytes_per_lookup"] * line["roofline"]["lookups_per_launch"] / (timed["avg_us"] * 1e-6) / 8e12
    assert frac >= 0.70


def test_defaults_finish_in_minutes(monkeypatch):
    bench = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.gpus == 1 and a.steps * 10e-6 < 1.0 and a.steps >= 2000 and a.batch == 65536 and a.items == 10_000_000 and a.hashes == 8
    assert a.ring_mib >= 1024 and a.table == "sharded"
