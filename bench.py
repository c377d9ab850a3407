#!/usr/bin/env python3
"""Headline benchmark: OOV embed lookups+scores per second on the synthetic north-star workload.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json north_star / SURVEY.md section 8d "Config S-lsh"): item features
f32[N=10M, F=64] row-normalised, planes f32[8,64], OOV bucket table f32[8,64], batch B=65536
random ids, user rows f32[B,64].  One STEP = one pass of the hot path over one batch = ONE launch of
    mi_oov_lsh_embed_score  (gather feat[id] -> 8 sign projections -> masked mean of bucket rows ->
                             dot with the user row -> score[b]; what BPR.predict returns for OOV
                             items behind LSHInductiveEmbedder.embed_item_ids, embedding never
                             materialised: 16 + 4F + 4D + 4 = 532 algorithmic bytes per lookup)
`--unfused` runs the same step as the two launches the plugin boundary implies (mi_oov_lsh_embed
writing [B,64] rows, then mi_oov_rowdot).  Every step uses a fresh id batch (no cache reuse across
steps); all inputs are resident in HBM before the timed region.  value = lookups (each embedded
and scored) per second, whole job.

N > 1: one process per GPU; the 2.56 GB table is REPLICATED (it fits 288 GB HBM 100x over), ranks
process independent batches and there is no data-path collective -> weak
scaling.  `--sharded` instead row-shards the table and exchanges ids/rows with RCCL all-to-all.

The JSON line also carries `roofline` (dominant kernel = the fused lsh kernel, HBM-bound,
algorithmic bytes 8+4F+4D = 520 B per lookup, timed with HIP events on its own stream inside the
timed region) and, at N=1, `cpu_baseline` (the reference's torch-CPU op sequence on the host).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 20000 timed steps = 0.17 s on the GPU, so that the two barrier + synchronize fences around the timed
    # region (~0.9 ms together) stay below 1 % of it; 22000 distinct id batches = 11.5 GB of the 288 GB
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--ramp-seconds", type=float, default=1.0,
                    help="untimed pre-warm-up that lets the GPU leave its idle clock (sclk idles at ~500 MHz and "
                         "needs tens of ms of load to ramp; 200 x 11 us steps alone are over before it does)")
    ap.add_argument("--items", type=int, default=10_000_000)
    ap.add_argument("--feat", type=int, default=64)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--hashes", type=int, default=8)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--sharded", action="store_true", help="row-shard the table + RCCL all-to-all exchange")
    ap.add_argument("--unfused", action="store_true", help="two launches (lsh_embed + rowdot) instead of the fused kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every step from Python instead of replaying one HIP graph of the K launches")
    ap.add_argument("--in-flight", type=int, default=4,
                    help="N=1 only: after the timed region, also measure the same K steps with this many batches "
                         "allowed to overlap (one graph chain per stream) and report it as `pipelined` (0 = skip)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal on a 1-GPU box)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    return ap.parse_args()


def make_inputs(args, dev, rank, n_rows, row_offset=0):
    """Deterministic synthetic tables, generated on the device in chunks."""
    g = torch.Generator(device=dev)
    feat = torch.empty((n_rows, args.feat), dtype=torch.float32, device=dev)
    chunk = 1 << 20
    for lo in range(0, n_rows, chunk):
        hi = min(n_rows, lo + chunk)
        g.manual_seed(1_000_003 * ((row_offset + lo) // chunk) + 0)
        x = torch.randn((hi - lo, args.feat), generator=g, device=dev)
        feat[lo:hi] = torch.nn.functional.normalize(x, dim=-1)
    g.manual_seed(1)
    planes = torch.randn((args.hashes, args.feat), generator=g, device=dev)
    g.manual_seed(2)
    buckets = torch.randn((args.hashes, args.dim), generator=g, device=dev)
    return feat, planes, buckets


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup quota (the
    GPU box exposes 256 logical CPUs but grants 16 per GPU)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(args, feat, planes, buckets, seconds):
    """The reference's CPU path (torch ops, oracle/ref_torch.py) on a bounded sample of batches."""
    from oracle import ref_torch
    cores = host_cores()
    torch.set_num_threads(cores)
    f, p, w = feat.cpu(), planes.cpu(), buckets.cpu()
    g = torch.Generator().manual_seed(7)
    users = torch.randn((args.batch, args.dim), generator=g)
    n_done, t_total = 0, 0.0
    for it in range(100000):
        ids = torch.randint(0, f.shape[0], (args.batch,), generator=g)
        t0 = time.perf_counter()
        e = ref_torch.lsh_embed(ids, f, p, w)
        s = ref_torch.rowdot(users, e)
        t1 = time.perf_counter()
        if it >= 2:  # two untimed warm-up batches
            n_done += 1
            t_total += t1 - t0
        if t_total > seconds:
            break
    del s
    return {"value": n_done * args.batch / t_total, "unit": "lookups/s", "cores": cores, "kind": "port",
            "sample": f"{n_done} batches of {args.batch} lookups+scores through the reference's torch-CPU op "
                      f"sequence (oracle/ref_torch.py), {t_total:.1f} s, torch.set_num_threads({cores})"}


def pipelined(args, step, per_lookup):
    """The same K steps with `--in-flight` batches allowed to overlap: one HIP-graph chain per stream, replayed
    together.  NOT the headline (`value` times the K steps serialised, as the contract's single-stream HIP-event
    timing implies); it shows what the same kernel delivers to a serving loop that keeps several batches in
    flight: the launch -> ids -> rows -> store chain of one batch hides under the row traffic of the others."""
    C, K, B = min(args.in_flight, args.steps), args.steps, args.batch
    if C < 2:
        return None
    with torch.no_grad():
        streams = [torch.cuda.Stream() for _ in range(C)]
        graphs = []
        for c, s in enumerate(streams):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for k in range(c, K, C):
                    step(args.warmup + k)
            graphs.append(gr)
        times = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for gr, s in zip(graphs, streams):
                with torch.cuda.stream(s):
                    gr.replay()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    return {"batches_in_flight": C, "value": K * B / dt, "unit": "lookups/s", "us_per_step": dt / K * 1e6,
            "achieved": K * B * per_lookup / dt / 1e9, "achieved_unit": "GB/s (algorithmic)", "timing": "host clock around "
            "one concurrent replay of the C graph chains (median of 5), K steps in total"}


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_bench_summary.json, written by tools/summarize_profile.py from separate --pmc
    FETCH_SIZE / --pmc WRITE_SIZE runs of this same command).  FETCH_SIZE/WRITE_SIZE are in KiB; on
    gfx950 FETCH_SIZE reports half of the bytes of 16-B-per-lane reads (MI355X_MICROARCH.md, HBM
    section), hence the factor 2.  None when no profile is committed for this kernel."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_summary.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        for k, v in d.get("pmc_per_launch", {}).items():
            if k.startswith(kernel_prefix.rstrip(">")) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                best = (2.0 * v["FETCH_SIZE"]["mean"] + v["WRITE_SIZE"]["mean"]) * 1024.0
    return best


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    dev_index = 0 if args.single_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.dist_backend)
    ctl_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")  # where control tensors live

    import mi_oov  # noqa: F401
    from mi_oov import ops, sharded

    B, D, F, H, N = args.batch, args.dim, args.feat, args.hashes, args.items
    if args.sharded and world > 1:
        lo, hi, per = sharded.shard_bounds(N, world, rank)
        feat, planes, buckets = make_inputs(args, dev, rank, hi - lo, lo)
        table = sharded.ShardedLSHTable(feat, N)
        embed = lambda ids: table.embed(ids, planes, buckets)  # noqa: E731
    else:
        feat, planes, buckets = make_inputs(args, dev, rank, N)
        embed = lambda ids: ops.lsh_embed(ids, feat, planes, buckets)  # noqa: E731

    total = args.warmup + args.steps
    g = torch.Generator(device=dev)
    g.manual_seed(3 + 1000 * rank)
    all_ids = torch.randint(0, N, (total, B), generator=g, device=dev)
    g.manual_seed(4 + 1000 * rank)
    n_user_bufs = 8
    users = torch.randn((n_user_bufs, B, D), generator=g, device=dev)

    fused = not args.unfused and not (args.sharded and world > 1)
    # The K timed launches are captured once into a HIP graph (stream capture sees the C-ABI launches, which
    # go to torch's current stream) and the timed region is ONE replay: the host contributes nothing per
    # step.  Launched from Python, a step costs ~8.6 us of host time against ~9.5 us on the GPU, so any
    # host jitter shows up in the number (tools/stability.py).  Same kernels, same K distinct id batches.
    use_graph = fused and not args.no_graph
    scores = torch.empty((n_user_bufs, B), dtype=torch.float32, device=dev)

    def step(i, ev=None):
        ids = all_ids[i]
        if fused:
            return ops.lsh_embed_score(ids, feat, planes, buckets, users[i % n_user_bufs], score_out=scores[i % n_user_bufs])
        if ev:
            ev[0].record()
        e = embed(ids)
        if ev:
            ev[1].record()
        return ops.rowdot(users[i % n_user_bufs], e)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        graph = None
        if use_graph:
            for i in range(3):  # first-use initialisation outside the capture
                step(i % total)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for k in range(args.steps):
                    step(args.warmup + k)
        # Clock ramp, before the W warm-up steps: keep launching until the time of a burst (256 steps, or one
        # replay of the graph) has settled (two consecutive bursts within 2 %) and at least --ramp-seconds
        # have passed; give up after 5 x that.  Untimed; uses the same id batches as the run.
        t_ramp = time.perf_counter()
        i, prev = 0, None
        while True:
            t_b = time.perf_counter()
            if graph is not None:
                graph.replay()
            else:
                for _ in range(256):
                    step(i % total)
                    i += 1
            torch.cuda.synchronize()
            now = time.perf_counter()
            burst = now - t_b
            settled = prev is not None and abs(burst - prev) <= 0.02 * burst
            prev = burst
            if (settled and now - t_ramp >= args.ramp_seconds) or now - t_ramp >= 5 * args.ramp_seconds:
                break
        for i in range(args.warmup):
            step(i)
        # HIP events on the launch stream (torch's current stream is the one handed to the C ABI).
        # Fused: the timed region holds nothing but K launches of the dominant kernel, so one event
        # pair around the region gives its average launch duration (inter-launch gaps included).
        # Unfused: one pair around every lsh_embed launch.
        region = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        events = None if fused else [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                                     for _ in range(args.steps)]
        fence()
        t0 = time.perf_counter()
        region[0].record()
        if graph is not None:
            graph.replay()  # the K captured steps
        else:
            for k in range(args.steps):
                step(args.warmup + k, events[k] if events else None)
        region[1].record()
        fence()
        t1 = time.perf_counter()

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=ctl_dev)
    if fused:
        kern_ms = region[0].elapsed_time(region[1]) / max(1, args.steps)
    else:
        kern_ms = sum(a.elapsed_time(b) for a, b in events) / max(1, args.steps)
    kern = torch.tensor([kern_ms], dtype=torch.float64, device=ctl_dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(kern, op=dist.ReduceOp.MAX)
    elapsed_s, kern_ms = float(elapsed.item()), float(kern.item())

    if rank == 0:
        per_lookup = (16 + 4 * F + 4 * D + 4) if fused else (8 + 4 * F + 4 * D)
        achieved = (B * per_lookup / (kern_ms * 1e-3)) / 1e9 if kern_ms > 0 else 0.0
        out = {
            "metric": "OOV embed lookups+scores/sec",
            "value": world * B * args.steps / elapsed_s,
            "unit": "lookups/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_s / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "synthetic lsh hash-gather-aggregate + pairwise score "
                                   f"({N}-item x {F}-feature table, {H} hashes/buckets, {D}-d, batch {B} per GPU)",
                       "items": N, "feat": F, "dim": D, "hashes": H, "batch_per_gpu": B,
                       "table": "row-sharded + all-to-all" if (args.sharded and world > 1) else "replicated per GPU",
                       "launches_per_step": "lsh_embed_score" if fused else "lsh_embed + rowdot",
                       "launch_mode": "one HIP graph of the K launches, replayed once" if use_graph else "one launch per step from the host"},
            "roofline": {"bound": "hbm", "kernel": "lsh64_kernel<8, true, false, false, false>" if fused else "lsh64_kernel<8, false, true, false, false>",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "bytes_per_lookup": per_lookup, "lookups_per_launch": B, "avg_launch_us": kern_ms * 1e3,
                         "traffic": None},
        }
        out["roofline"]["traffic"] = pmc_traffic(out["roofline"]["kernel"])
        if world == 1 and use_graph and args.in_flight > 1:
            try:  # an extra, never allowed to cost the headline line
                pl = pipelined(args, step, per_lookup)
            except Exception as e:  # noqa: BLE001
                pl = {"error": f"{type(e).__name__}: {e}"}
            if pl is not None:
                out["pipelined"] = pl
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, feat, planes, buckets, args.cpu_seconds)
            except Exception as e:  # noqa: BLE001 -- the GPU line is still worth printing
                out["cpu_baseline"] = {"value": None, "unit": "lookups/s", "cores": host_cores(), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
