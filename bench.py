#!/usr/bin/env python3
"""Headline benchmark: OOV embed lookups+scores per second on the synthetic north-star workload.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json north_star / SURVEY.md section 8d "Config S-lsh"): item features
f32[N=10M, F=64] row-normalised, planes f32[8,64], OOV bucket table f32[8,64], batch B=65536
random ids, user rows f32[B,64].  One STEP = one pass of the hot path over one batch:
    gather feat[id] -> 8 sign projections -> masked mean of bucket rows -> dot with the user row -> score[b]
(what BPR.predict returns for OOV items behind LSHInductiveEmbedder.embed_item_ids, embedding never
materialised: 16 + 4F + 4D + 4 = 532 algorithmic bytes per lookup by SURVEY 8d's formula, 524 of which this
path moves -- it is handed user ROWS, not user ids).

The K timed steps are K distinct id batches queued to mi_oov_lsh_embed_score_multi (csrc/lsh64p.hip): ONE
persistent launch per `--batches-per-launch` steps, whose waves walk the tiles of all queued batches with the ids
/ rows of the next tiles already requested (what a serving loop with K batches in its queue calls).
`--per-batch` launches every step separately (mi_oov_lsh_embed_score, K launches in one HIP graph: the round-1
headline); `--unfused` runs the two launches the plugin boundary implies (mi_oov_lsh_embed writing [B,64] rows,
then mi_oov_rowdot).  Every step has its own id batch, the user rows come from a ring of >= 1 GiB (4x the
256 MiB Infinity Cache) and the clock ramp runs on batches of its own, so nothing the timed region reads has been
touched recently; all inputs are resident in HBM before the timed region.  value = lookups (each embedded and
scored) per second, whole job.

N > 1: `python bench.py --gpus N` starts the N ranks itself (one process per GPU, RCCL) when no launcher did.
Default N > 1 table mode is `--table sharded` (feature table row-sharded, ids out / codes back
by RCCL all-to-all); the replicated-table figure (no data-path collective) is reported beside it as `replicated`.

The JSON line also carries `roofline` (dominant kernel, HBM-bound, timed with HIP events on its own stream inside
the timed region) and, at N=1, `cpu_baseline` (the reference's torch-CPU op sequence on the host).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 20000 timed steps = 0.11 s on the GPU, so that the two barrier + synchronize fences around the timed
    # region stay below 1 % of it; 22000 distinct id batches = 11.5 GB of the 288 GB
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--batches-per-launch", type=int, default=1000,
                    help="steps queued to one persistent launch (the whole run when --steps is smaller)")
    ap.add_argument("--ramp-seconds", type=float, default=5.0,
                    help="untimed pre-warm-up under load, on id batches and user rows of its own.  sclk leaves its ~500 MHz "
                         "idle state within tens of ms, but a box that has been idle reaches its steady state only after "
                         "~4 s of load: ten back-to-back `--steps 20` runs on a fresh box with a 1 s ramp gave 0.757, 0.766, "
                         "then 0.777-0.784 of the HBM peak; with a 6 s ramp the FIRST run gives 0.785 (round 4)")
    ap.add_argument("--items", type=int, default=10_000_000)
    ap.add_argument("--feat", type=int, default=64)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--hashes", type=int, default=8)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--ring-mib", type=int, default=1024, help="size of the ring of user-row buffers")
    ap.add_argument("--batches-per-exchange", type=int, default=64, help="--table sharded: steps carried by one all-to-all")
    ap.add_argument("--force-sharded", action="store_true",
                    help="developer: run the sharded path at N = 1 too (one-rank RCCL group, self-exchange) to time its kernels")
    ap.add_argument("--local-fast", action="store_true",
                    help="--table sharded: lookups a rank owns itself skip the exchange (fused kernel on the local block)")
    ap.add_argument("--sustained-steps", type=int, default=640,
                    help="--table sharded: also time this many steps in one go (exchanges overlapping) and report them under "
                         "sharded.sustained; 0 = skip")
    ap.add_argument("--cap-factor", type=float, default=1.0,
                    help="--table sharded: segment capacity = this multiple of the expected share of an exchange + 8 "
                         "standard deviations (ids are uniform; the run fails loudly if a segment ever overflows)")
    ap.add_argument("--table", choices=["sharded", "replicated"], default="sharded",
                    help="N > 1: row-shard the feature table + RCCL all-to-all (default), or replicate it per GPU")
    ap.add_argument("--sharded", action="store_true", help="same as --table sharded (kept for round-1 command lines)")
    ap.add_argument("--per-batch", action="store_true", help="one launch per step (K launches in one HIP graph)")
    ap.add_argument("--unfused", action="store_true", help="two launches per step (lsh_embed + rowdot)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the untimed rows-stored measurement reported under 'also'")
    ap.add_argument("--no-graph", action="store_true", help="with --per-batch: launch every step from Python")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal on a 1-GPU box)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-spin", action="store_true", help="do not poll the end event before the closing fence")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` from a bare shell: start the N ranks as CHILD processes -- before this process has
    made a single GPU call -- with the rendezvous environment torch.distributed.run would have set, wait for them, and
    exit with the first failure.  Rank 0 prints the JSON line to our stdout."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):  # a rank that died leaves the others in a collective: end them
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


def make_inputs(args, dev, n_rows, row_offset=0):
    """Deterministic synthetic tables, generated on the device in chunks (row r gets the same values whichever rank
    and shard holds it)."""
    g = torch.Generator(device=dev)
    feat = torch.empty((n_rows, args.feat), dtype=torch.float32, device=dev)
    chunk = 1 << 20
    lo = 0
    while lo < n_rows:
        glo = row_offset + lo
        c0 = glo // chunk
        hi = min(n_rows, lo + (c0 + 1) * chunk - glo)
        g.manual_seed(1_000_003 * c0)
        x = torch.randn((chunk, args.feat), generator=g, device=dev)
        x = x[glo - c0 * chunk: glo - c0 * chunk + (hi - lo)]
        feat[lo:hi] = torch.nn.functional.normalize(x, dim=-1)
        lo = hi
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


def also_rows_stored(args, ops, feat, planes, buckets, all_ids, n_ramp, dev):
    """Outside the timed region, for the record: what the PLUGIN returns -- LSHInductiveEmbedder.embed_item_ids' [B, 64]
    rows (lsh_embedder.py:161-179) -- for 20 queued batches per persistent launch (mi_oov_lsh_multi, rows mode, prepared
    table), and ONE such call per launch (mi_oov_lsh_embed).  HIP events over 10 launches that alternate between two
    queues of distinct id batches and two sets of output buffers; 8 + 4F + 4D = 520 algorithmic bytes per lookup."""
    B, K = args.batch, 20
    qs = [ops.LshBatchQueue([all_ids[j * K + i] for i in range(K)], rows=True) for j in range(2)]  # (40 of the 64 ramp batches)
    sc = ops.LshMultiScorer(feat, planes, buckets)
    calls = [sc.bind(q) for q in qs]
    res = {}
    with torch.no_grad():
        for name, fn, lookups in (("lsh_embed rows, 20 queued batches per launch (mi_oov_lsh_multi)", lambda i: calls[i % 2](), K * B),
                                  ("lsh_embed rows, one batch per launch (mi_oov_lsh_embed)",
                                   lambda i: ops.lsh_embed(all_ids[i % n_ramp], feat, planes, buckets), B)):
            for i in range(4):
                fn(i)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10 if lookups > B else 40
            a.record()
            for i in range(n):
                fn(i)
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) * 1e3 / n
            gbs = lookups * 520 / (us * 1e-6) / 1e9
            res[name] = {"us_per_launch": us, "us_per_65536_lookups": us * 65536 / lookups, "GB_per_s": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS}
    return res


def pmc_traffic(kernel, args):
    """HBM bytes per BATCH of the dominant kernel from the committed rocprofv3 PMC passes (profiles/*_bench_summary.json,
    written by tools/summarize_profile.py from separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this command).
    FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of 16-B-per-lane reads
    (MI355X_MICROARCH.md, HBM section), hence the factor 2.  Only a profile of the SAME kernel on the SAME shape is
    used (the newest one); returns (bytes_per_batch, source file) or (None, None)."""
    import glob
    shape = {"items": args.items, "feat": args.feat, "dim": args.dim, "hashes": args.hashes, "batch": args.batch}
    best = (None, None)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_summary.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("shape") != shape:
            continue
        for k, v in d.get("pmc_per_launch", {}).items():
            if k == kernel and "FETCH_SIZE" in v and "WRITE_SIZE" in v:  # (the exact instantiation: the rows / lookup modes are other kernels)
                per_launch = (2.0 * v["FETCH_SIZE"]["mean"] + v["WRITE_SIZE"]["mean"]) * 1024.0
                best = (per_launch / max(1, int(v.get("batches_per_launch", 1))), os.path.relpath(f, ROOT))
    return best


def main():
    args = parse()
    if args.sharded:
        args.table = "sharded"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    # stdout carries ONE JSON line and nothing else: RCCL prints a version banner to stdout when its first communicator is
    # made (seen with --force-sharded: "RCCL version : 2.26.6 ..." in front of the line), so file descriptor 1 points at
    # stderr for the whole run and the line is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    dev_index = 0 if args.single_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1 or args.force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.dist_backend)
    ctl_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")  # where control tensors live

    import mi_oov  # noqa: F401
    from mi_oov import ops

    B, D, F, H, N = args.batch, args.dim, args.feat, args.hashes, args.items
    K, W = args.steps, args.warmup
    mode = "unfused" if args.unfused else ("per_batch" if args.per_batch else "multi")
    bpl = max(1, min(args.batches_per_launch, K)) if mode == "multi" else 1

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def reduce_max(x):
        t = torch.tensor([x], dtype=torch.float64, device=ctl_dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- inputs ---------------------------------------------------------------------------------------------------
    n_ramp = 64  # id batches (and as many user buffers) that only the clock ramp touches
    total = n_ramp + W + K
    g = torch.Generator(device=dev)
    g.manual_seed(3 + 1000 * rank)
    all_ids = torch.randint(0, N, (total, B), generator=g, device=dev)
    g.manual_seed(4 + 1000 * rank)
    ring = max(2, -(-(args.ring_mib << 20) // (B * D * 4)))
    users = torch.randn((ring, B, D), generator=g, device=dev)
    ramp_users = torch.randn((n_ramp, B, D), generator=g, device=dev)
    scores = torch.empty((ring, B), dtype=torch.float32, device=dev)
    # batch i (0 <= i < total): ramp batches first, then the W warm-up steps, then the K timed steps
    user_of = lambda i: ramp_users[i] if i < n_ramp else users[(i - n_ramp) % ring]  # noqa: E731

    def time_region(run_steps):
        """ramp -> W warm-up steps -> fence -> K timed steps -> fence.  `run_steps(i0, n)` enqueues steps [i0, i0+n)
        on the current stream and returns the number of launches of the dominant kernel it made."""
        with torch.no_grad():
            t_ramp = time.perf_counter()
            prev, j = None, 0
            while True:  # clock ramp on the ramp batches only; settled = two consecutive bursts within 2 %
                t_b = time.perf_counter()
                run_steps(0, n_ramp)
                torch.cuda.synchronize()
                now = time.perf_counter()
                burst = now - t_b
                settled = prev is not None and abs(burst - prev) <= 0.02 * burst
                prev = burst
                j += 1
                done = (settled and now - t_ramp >= args.ramp_seconds) or now - t_ramp >= 2 * args.ramp_seconds
                # the ranks must leave the ramp TOGETHER: a step of the sharded path is a collective, and a rank that
                # has settled earlier would otherwise go on to the warm-up exchange while another repeats the ramp's
                if reduce_max(0.0 if done else 1.0) == 0.0:
                    break
            if W:
                run_steps(n_ramp, W)
            return timed_steps(run_steps, n_ramp + W, K)

    def timed_steps(run_steps, i0, n):
        """fence -> n steps between two HIP events on the launch stream -> fence; (host seconds, event ms, launches), each
        the max over the ranks."""
        with torch.no_grad():
            region = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            # first use of an event creates it (~40 us) and the first query / synchronize after one is slower too
            # (tools/region_cost.py): done here, untimed, so that the n steps are what the region holds
            region[0].record()
            region[1].record()
            while not region[1].query():
                pass
            fence()
            t0 = time.perf_counter()
            region[0].record()
            launches = run_steps(i0, n)
            region[1].record()
            if not args.no_spin:
                # poll the end event instead of sleeping in hipDeviceSynchronize: a blocking wait wakes up tens of us
                # after the GPU is done, which at --steps 20 (a 0.11 ms region) is a third of the measurement.  The
                # closing fence below then finds an idle device.
                while not region[1].query():
                    pass
            fence()
            t1 = time.perf_counter()
        return reduce_max(t1 - t0), reduce_max(region[0].elapsed_time(region[1])), launches

    def replicated_runner(feat, planes, buckets):
        """Local table, no collective: the N = 1 path, and the `replicated` figure at N > 1."""
        if mode == "multi":
            scorer = ops.LshMultiScorer(feat, planes, buckets)
            q = ops.LshBatchQueue([all_ids[i] for i in range(total)], [user_of(i) for i in range(total)],
                                  [scores[i % ring] for i in range(total)])

            bound = {}  # (first batch, batches) -> prevalidated launch: what a serving loop over preallocated buffers keeps

            def run_steps(i0, n):
                launches = 0
                for k0 in range(i0, i0 + n, bpl):
                    nb = min(bpl, i0 + n - k0)
                    call = bound.get((k0, nb))
                    if call is None:
                        call = bound[(k0, nb)] = scorer.bind(q, k0, nb) if scorer.persistent and B <= (1 << 23) else \
                            (lambda k0=k0, nb=nb: scorer.run(q, k0, nb))
                    call()
                    launches += 1
                return launches
            for i0_, n_ in ((0, n_ramp), (n_ramp, W), (n_ramp + W, K)):  # bind outside the timed region
                for k0 in range(i0_, i0_ + n_, bpl):
                    nb = min(bpl, i0_ + n_ - k0)
                    if scorer.persistent and B <= (1 << 23):
                        bound[(k0, nb)] = scorer.bind(q, k0, nb)
            return run_steps, "lsh64_persistent_kernel<8, 0, true, false>", "mi_oov_lsh_multi (score mode, prepared 2^H table; = mi_oov_lsh_embed_score_multi)"
        if mode == "per_batch":
            scorer = ops.LshScorer(feat, planes, buckets)
            graphs = {}

            def run_steps(i0, n):
                if args.no_graph:
                    for i in range(i0, i0 + n):
                        scorer(all_ids[i], user_of(i), score_out=scores[i % ring])
                    return n
                if (i0, n) not in graphs:  # the n launches captured once, replayed as one graph
                    scorer(all_ids[i0], user_of(i0), score_out=scores[i0 % ring])
                    torch.cuda.synchronize()
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr):
                        for i in range(i0, i0 + n):
                            scorer(all_ids[i], user_of(i), score_out=scores[i % ring])
                    graphs[(i0, n)] = gr
                graphs[(i0, n)].replay()
                return n
            for span in ((0, n_ramp), (n_ramp, W), (n_ramp + W, K)):  # capture outside the timed region
                if span[1] and not args.no_graph:
                    run_steps(*span)
            torch.cuda.synchronize()
            return run_steps, "lsh64_kernel<8, true, false, false, false>", "mi_oov_lsh_embed_score"

        def run_steps(i0, n):
            for i in range(i0, i0 + n):
                e = ops.lsh_embed(all_ids[i], feat, planes, buckets)
                ops.rowdot(user_of(i), e)
            return n
        return run_steps, "lsh64_kernel<8, false, true, false, false>", "mi_oov_lsh_embed + mi_oov_rowdot"

    per_lookup = (16 + 4 * F + 4 * D + 4) if mode != "unfused" else (8 + 4 * F + 4 * D)
    moved = per_lookup - 8 if mode != "unfused" else per_lookup

    def sharded_runner():
        """N > 1, `--table sharded`: the feature table row-sharded over the ranks, every rank's batches looked up through
        the owner-computes exchange (mi_oov.sharded: ids out, 8-byte codes back, both by RCCL all_to_all_single), steps
        software-pipelined three deep.  One exchange carries `--batches-per-exchange` steps."""
        from mi_oov import sharded
        S = max(1, min(args.batches_per_exchange, ring, n_ramp))
        lo, hi, _ = sharded.shard_bounds(N, world, rank)
        feat_l, planes_l, buckets_l = make_inputs(args, dev, hi - lo, lo)
        table = sharded.ShardedLSHTable(feat_l, N, cap_factor=args.cap_factor, uniform_batches=True)
        pipe = sharded.LshPipeline(table, planes_l, buckets_l, local_fast=args.local_fast)
        flat_ids = all_ids.view(-1)
        ramp_rows, ring_rows, ring_scores = ramp_users.view(-1, D), users.view(-1, D), scores.view(-1)

        def blocks(i0, n):
            """[i0, i0+n) in exchanges of up to S steps whose user rows are contiguous (cut where the ring wraps)."""
            out = []
            while n > 0:
                if i0 < n_ramp:  # the clock ramp's own batches and user rows
                    take = min(n, S, n_ramp - i0)
                    rows, sc0 = ramp_rows[i0 * B:(i0 + take) * B], 0
                else:
                    sc0 = (i0 - n_ramp) % ring
                    take = min(n, S, ring - sc0)
                    rows = ring_rows[sc0 * B:(sc0 + take) * B]
                out.append((flat_ids[i0 * B:(i0 + take) * B], rows, ring_scores[sc0 * B:(sc0 + take) * B]))
                i0, n = i0 + take, n - take
            return out

        def run_steps(i0, n):
            bl = blocks(i0, n)
            pipe.run([b[0] for b in bl], [b[1] for b in bl], [b[2] for b in bl])
            return len(bl)

        elapsed_s, region_ms, n_blocks = time_region(run_steps)
        torch.cuda.synchronize()
        last = n_ramp + W + K - 1
        kept = scores[(last - n_ramp) % ring].clone()  # the last timed batch's scores: compared with the unsharded kernel below
        # A 20-step run is ONE exchange (a serial chain, nothing overlaps); the sustained figure is what the pipeline
        # delivers when exchanges overlap: `--sustained-steps` (>= 640) steps on id batches of their own, same user-row ring.
        sustained = None
        n_sus = int(args.sustained_steps)
        if n_sus > 0 and n_sus != K:
            g2 = torch.Generator(device=dev)
            g2.manual_seed(5 + 1000 * rank)
            sus_ids = torch.randint(0, N, (n_sus * B,), generator=g2, device=dev)

            def run_sus(i0, n):
                bl, done = [], 0
                while done < n:
                    sc0 = done % ring
                    take = min(n - done, S, ring - sc0)
                    bl.append((sus_ids[done * B:(done + take) * B], ring_rows[sc0 * B:(sc0 + take) * B], ring_scores[sc0 * B:(sc0 + take) * B]))
                    done += take
                pipe.run([b[0] for b in bl], [b[1] for b in bl], [b[2] for b in bl])
                return len(bl)
            run_sus(0, min(n_sus, 2 * S))  # untimed: first use of these buffers
            s_el, s_ms, s_blocks = timed_steps(run_sus, 0, n_sus)
            sustained = {"steps": n_sus, "value": world * B * n_sus / s_el, "ms_per_step": 1e3 * s_el / n_sus,
                         "us_per_step_hip_events": s_ms * 1e3 / n_sus, "exchanges": s_blocks}
            del sus_ids
        torch.cuda.synchronize()
        dropped = reduce_max(float(table.overflow.item()))  # every rank learns of an overflow on any rank
        if dropped:
            raise RuntimeError(f"--cap-factor {args.cap_factor} too tight: a segment overflowed by {int(dropped)} lookups")
        if rank != 0:
            return {"kept_scores": kept}
        # The three kernels of an exchange (bucketing of step t+2, owner of step t, requester of step t-1) run on three
        # streams at once and share HBM, so no kernel has a duration of its own: the roofline object is the whole step --
        # algorithmic bytes of all three per lookup over the HIP-event time of the region.
        step_bytes = (8 + 8 + 4) + (8 + 4 * F + H) + (4 + H + 4 * D + 4)
        step_us = region_ms * 1e3 / K
        achieved = step_bytes * B / (step_us * 1e-6) / 1e9 if step_us > 0 else 0.0
        return {"value": world * B * K / elapsed_s, "ms_per_step": 1e3 * elapsed_s / K,
                "table": f"feature table row-sharded over {world} ranks ({hi - lo} rows here), planes + bucket table replicated",
                "entry_point": "mi_oov_bucket_by_owner + mi_oov_lsh_embed(bits) + mi_oov_lsh_codes_embed",
                "launch_mode": f"owner-computes exchange, up to {S} steps per exchange, three exchanges in flight on three streams",
                "roofline": {"bound": "hbm", "kernel": "bucket_by_owner_small_kernel + lsh64_persistent_kernel<8, 1, false, false> (owner) + "
                                                       "lsh64_persistent_kernel<8, 2, false, false> (requester), concurrent",
                             "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                             "bytes_per_lookup": step_bytes, "lookups_per_launch": B * K / n_blocks,
                             "launches": n_blocks, "avg_launch_us": region_ms * 1e3 / n_blocks, "traffic": None,
                             "note": "one 'launch' = one exchange (three kernels + two all-to-alls); HIP events over the timed region"},
                "detail": {"steps_per_exchange": S, "exchanges": n_blocks, "segment_capacity": table.capacity(min(S, K) * B),
                           "cap_factor": args.cap_factor, "bytes_on_wire_per_lookup": 8 + H, "local_fast": bool(args.local_fast),
                           "region_ms_hip_events": region_ms, "us_per_step_hip_events": region_ms * 1e3 / K,
                           "overflowed_lookups": 0, "sustained": sustained},
                "kept_scores": kept}

    sharded_line = None
    if (world > 1 or args.force_sharded) and args.table == "sharded":
        # A failure of the sharded runner is FATAL (non-zero exit on this rank; the launcher ends the others): the line's
        # `value` at N > 1 is the sharded figure, and printing the replicated one in its place would report the wrong mode.
        sharded_line = sharded_runner()
        torch.cuda.empty_cache()

    feat, planes, buckets = make_inputs(args, dev, N)
    run_steps, kernel, entry = replicated_runner(feat, planes, buckets)
    elapsed_s, region_ms, launches = time_region(run_steps)

    # ---- after the timed region: the LAST timed batch's scores against ONE single-batch launch of mi_oov_lsh_embed_score
    # (csrc/lsh64.hip: another kernel, the same arithmetic) on the same ids and user rows -- bit for bit (NaN = NaN).  A run
    # whose timed output is wrong exits non-zero instead of printing a line.
    torch.cuda.synchronize()
    last = n_ramp + W + K - 1
    got = scores[(last - n_ramp) % ring]
    want = ops.lsh_embed_score(all_ids[last], feat, planes, buckets, user_of(last))
    torch.cuda.synchronize()

    def same_bits(a, b):
        return bool((((a == b) | (torch.isnan(a) & torch.isnan(b))).all()).item())
    if mode == "unfused":  # the two-launch mode keeps no scores (a developer comparison, never the headline): re-run its last step
        got = ops.rowdot(user_of(last), ops.lsh_embed(all_ids[last], feat, planes, buckets))
    checked = {"batch": last - n_ramp - W, "lookups": B, "against": "mi_oov_lsh_embed_score (single-batch kernel), bit for bit",
               "ok": same_bits(got, want), "nan_scores": int(torch.isnan(want).sum().item())}
    if sharded_line is not None:
        checked["sharded_ok"] = same_bits(sharded_line.pop("kept_scores"), want)
    bad = 0.0 if (checked["ok"] and checked.get("sharded_ok", True)) else 1.0
    if reduce_max(bad) != 0.0:
        raise SystemExit(f"bench.py rank {rank}: the timed region's scores differ from the single-batch kernel's: {checked}")

    if rank == 0:
        launch_us = region_ms * 1e3 / max(1, launches)
        lookups_per_launch = B * K / max(1, launches)
        achieved = lookups_per_launch * per_lookup / (launch_us * 1e-6) / 1e9 if launch_us > 0 else 0.0
        rep_value = world * B * K / elapsed_s
        traffic_batch, traffic_src = pmc_traffic(kernel, args)
        roofline = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "bytes_per_lookup": per_lookup,
                    "bytes_moved_per_lookup": moved, "frac_moved": achieved / HBM_PEAK_GBS * moved / per_lookup,
                    "lookups_per_launch": lookups_per_launch, "batches_per_launch": K / max(1, launches),
                    "launches": launches, "avg_launch_us": launch_us, "us_per_batch": region_ms * 1e3 / K,
                    "traffic": None if traffic_batch is None else traffic_batch * K / max(1, launches),
                    "traffic_source": traffic_src}
        out = {
            "metric": "OOV embed lookups+scores/sec",
            "value": rep_value,
            "unit": "lookups/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": 1e3 * elapsed_s / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "synthetic lsh hash-gather-aggregate + pairwise score "
                                   f"({N}-item x {F}-feature table, {H} hashes/buckets, {D}-d, batch {B} per GPU)",
                       "items": N, "feat": F, "dim": D, "hashes": H, "batch_per_gpu": B,
                       "table": "replicated per GPU", "entry_point": entry,
                       "launch_mode": {"multi": f"persistent launch, {bpl} queued batches per launch",
                                       "per_batch": "one launch per step" + ("" if args.no_graph else ", one HIP graph of the K launches"),
                                       "unfused": "two launches per step from the host"}[mode],
                       "user_row_ring_bytes": ring * B * D * 4, "ranks": world,
                       "backend": None if world == 1 else ("rccl" if args.dist_backend == "nccl" else args.dist_backend)},
            "roofline": roofline,
            "checked": True,
            "check": checked,
        }
        if sharded_line is not None:
            # N > 1 headline = the row-sharded table north_star specifies; the replicated figure stays beside it
            out["replicated"] = {"value": rep_value, "ms_per_step": out["ms_per_step"], "roofline": roofline,
                                 "table": "replicated per GPU, no data-path collective"}
            out["value"] = sharded_line["value"]
            out["ms_per_step"] = sharded_line["ms_per_step"]
            out["config"]["table"] = sharded_line["table"]
            out["config"]["entry_point"] = sharded_line["entry_point"]
            out["config"]["launch_mode"] = sharded_line["launch_mode"]
            out["roofline"] = sharded_line["roofline"]
            out["sharded"] = sharded_line.get("detail")
        if world == 1 and mode == "multi" and F == 64 and D == 64 and H <= 8 and not args.no_also:
            out["also"] = also_rows_stored(args, ops, feat, planes, buckets, all_ids, n_ramp, dev)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, feat, planes, buckets, args.cpu_seconds)
                out["vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            except Exception as e:  # noqa: BLE001 -- the GPU line is still worth printing
                out["cpu_baseline"] = {"value": None, "unit": "lookups/s", "cores": host_cores(), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or args.force_sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
