"""-m gpu: the row-sharded exchange on real kernels.

* the two device ends (mi_oov_bucket_by_owner, mi_oov_lsh_codes_embed) against the oracle;
* a world-size-1 RCCL group in this process: ShardedLSHTable / LshPipeline through the real collectives are
  bit-identical to the unsharded kernels, and a step runs under torch.cuda.set_sync_debug_mode("error");
* world size 2 on ONE device (gloo rendezvous, exchanges staged through the host -- the 1-GPU box has no second
  GPU for RCCL): every rank's local compute is the HIP kernels on its row block, results bit-identical to the
  unsharded kernels for lsh (BASELINE headline shape) and slsh with 128-d rows (BASELINE config 4 shape).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import bits_equal

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def ops():
    import mi_oov
    from mi_oov import ops as _ops
    assert mi_oov.available()
    return _ops


@pytest.mark.parametrize("B,N,world,cap", [(1, 10, 2, 1), (1000, 777, 8, 1000), (65536, 1_000_003, 8, 9000), (5000, 64, 3, 100),
                                           (70000, 500, 7, 70000), (300, 1000, 64, 300)])
def test_bucket_by_owner_properties(B, N, world, cap, oracle, ops, dev):
    rng = np.random.default_rng(B + world)
    ids = rng.integers(-3, N + 3, size=B).astype(np.int64)
    per = -(-N // world)
    over = torch.zeros((1,), dtype=torch.int32, device=dev)
    send, slot, counts = (t.cpu().numpy() for t in ops.bucket_by_owner(T(ids, dev), N, per, world, cap, over))
    o_send, o_slot, o_counts = oracle.bucket_by_owner(ids, N, per, world, cap)
    assert np.array_equal(counts, o_counts)  # lookups per owner (may exceed cap)
    assert int(over.item()) == max(0, int(o_counts.max()) - cap)
    valid = (ids >= 0) & (ids < N)
    owner = np.minimum(ids // per, world - 1)
    assert (slot[~valid] == -2).all()
    placed = valid & (slot >= 0)
    # as many lookups placed per owner as the oracle places (the rest dropped: slot -1), each in its owner's segment
    assert np.array_equal(np.bincount(owner[placed], minlength=world), np.minimum(o_counts, cap))
    assert (slot[valid & ~placed] == -1).all()
    assert np.array_equal(slot[placed] // cap, owner[placed])
    assert len(np.unique(slot[placed])) == placed.sum()  # nobody shares a slot
    flat = send.reshape(-1)
    assert np.array_equal(flat[slot[placed]], ids[placed] - owner[placed] * per)  # owner-local row numbers
    assert (flat == -1).sum() == world * cap - placed.sum()  # the rest of every segment is padding
    for w in range(world):  # a segment is filled from its start
        n = min(int(o_counts[w]), cap)
        assert (send[w, :n] >= 0).all() and (send[w, n:] == -1).all()
        assert np.array_equal(np.sort(send[w, :n]), np.sort(o_send[w, :n])) or o_counts[w] > cap


@pytest.mark.parametrize("B,N,world,cap,me", [(1, 10, 2, 1, 1), (1000, 777, 8, 1000, 0), (65536, 1_000_003, 8, 9000, 5), (5000, 64, 3, 100, 2),
                                              (70000, 500, 7, 70000, 3), (300000, 40_000_000, 16, 20000, 15), (4097, 9001, 1, 4097, 0)])
def test_bucket_by_owner_one_launch_and_local_compaction(B, N, world, cap, me, oracle, ops, dev, monkeypatch):
    """mi_oov_bucket_by_owner_fused (round 4): ONE launch -- reservations in caller-owned scratch words that every launch leaves
    at zero, the last workgroup publishes the counts and fills the tails -- gives what the three-operation form gives
    (same counts, same segments up to the order inside a segment, same -1 tails), launch after launch on the same scratch;
    with my_rank the lookups this rank owns are compacted into local_rows with slots behind the exchanged ones."""
    rng = np.random.default_rng(B + world)
    per = -(-N // world)
    for rep in range(3):  # the same scratch words serve every launch
        ids = rng.integers(-3, N + 3, size=B).astype(np.int64)
        o_send, o_slot, o_counts = oracle.bucket_by_owner(ids, N, per, world, cap)
        valid = (ids >= 0) & (ids < N)
        owner = np.minimum(ids // per, world - 1)
        over = torch.zeros((1,), dtype=torch.int32, device=dev)
        send, slot, counts = (t.cpu().numpy() for t in ops.bucket_by_owner(T(ids, dev), N, per, world, cap, over))
        monkeypatch.setenv("MI_OOV_BUCKET_FUSED", "0")
        send3, slot3, counts3 = (t.cpu().numpy() for t in ops.bucket_by_owner(T(ids, dev), N, per, world, cap))
        monkeypatch.delenv("MI_OOV_BUCKET_FUSED")
        assert np.array_equal(counts, o_counts) and np.array_equal(counts3, o_counts)
        assert int(over.item()) == max(0, int(o_counts.max()) - cap)
        for w in range(world):  # the same rows in every segment, the same padding (a segment that overflows keeps WHICHEVER cap
            if o_counts[w] <= cap:  # lookups reserved first: only their number is fixed)
                assert np.array_equal(np.sort(send[w]), np.sort(send3[w])) and np.array_equal(np.sort(send[w]), np.sort(o_send[w]))
            else:
                assert (send[w] >= 0).all() and (send3[w] >= 0).all()
        if (o_counts <= cap).all():
            assert np.array_equal(slot < 0, slot3 < 0) and np.array_equal(slot[slot < 0], slot3[slot3 < 0])
        placed = slot >= 0
        assert np.array_equal(np.bincount(owner[valid & placed], minlength=world), np.minimum(o_counts, cap))
        assert (slot[~valid] == -2).all() and (slot[valid & ~placed] == -1).all() and len(np.unique(slot[placed])) == placed.sum()
        assert np.array_equal(send.reshape(-1)[slot[placed]], ids[placed] - owner[placed] * per)
        # ... and with the local share compacted
        send_l, slot_l, counts_l, local_rows = (t.cpu().numpy() for t in ops.bucket_by_owner(T(ids, dev), N, per, world, cap, None, my_rank=me))
        assert np.array_equal(counts_l, o_counts)
        mine = valid & (owner == me)
        n_me = min(int(o_counts[me]), cap)
        assert (send_l[me] == -1).all()                                         # nothing of mine is sent
        assert (local_rows[:n_me] >= 0).all() and (local_rows[n_me:] == -1).all()
        kept = mine & (slot_l >= 0)
        assert kept.sum() == n_me and (slot_l[mine & ~kept] == -1).all()
        assert (slot_l[kept] >= world * cap).all() and (slot_l[kept] < (world + 1) * cap).all()
        assert len(np.unique(slot_l[kept])) == n_me
        assert np.array_equal(local_rows[slot_l[kept] - world * cap], ids[kept] - me * per)
        others = valid & (owner != me)
        if (o_counts <= cap).all():
            assert np.array_equal(slot_l[others] < 0, slot[others] < 0)
        po = others & (slot_l >= 0)
        assert np.array_equal(send_l.reshape(-1)[slot_l[po]], ids[po] - owner[po] * per) and (slot_l[po] // cap == owner[po]).all()
        assert (slot_l[~valid] == -2).all()
        for w in range(world):
            if w != me and o_counts[w] <= cap:
                assert np.array_equal(np.sort(send_l[w]), np.sort(send[w]))


@pytest.mark.parametrize("B,H,D", [(1, 8, 64), (777, 8, 64), (4099, 3, 64), (500, 16, 36), (333, 27, 128), (129, 40, 50),
                                   (64, 12, 256), (257, 5, 1),
                                   (900, 300, 64), (333, 2000, 64), (257, 700, 130), (130, 257, 256)])  # bucket rows a chunk at a time
def test_lsh_codes_embed_vs_oracle(B, H, D, oracle, ops, dev):
    rng = np.random.default_rng(B + H)
    M = B + 50
    codes = rng.integers(0, 2, size=(M, H)).astype(np.uint8)
    codes[3 % M] = 0      # an all-zero code: 0/0 -> NaN row
    codes[7 % M] = 0xFF   # the owner's answer for an id outside its shard
    slot = rng.permutation(M)[:B].astype(np.int32)
    if B > 10:
        slot[5], slot[6], slot[8] = -1, -2, 3 % M
    buckets = rng.standard_normal((H, D), dtype=np.float32)
    other = rng.standard_normal((B, D), dtype=np.float32)
    o_score, o_emb = oracle.lsh_codes_embed(codes, slot, buckets, other)
    score, emb = ops.lsh_codes_embed(T(codes, dev), T(slot, dev), T(buckets, dev), T(other, dev))
    assert bits_equal(emb.cpu().numpy(), o_emb) and bits_equal(score.cpu().numpy(), o_score)
    _, emb2 = ops.lsh_codes_embed(T(codes, dev), T(slot, dev), T(buckets, dev))
    assert bits_equal(emb2.cpu().numpy(), o_emb)
    buf = torch.full((B,), -3.0, device=dev)
    score3, none = ops.lsh_codes_embed(T(codes, dev), T(slot, dev), T(buckets, dev), T(other, dev), want_emb=False, score_out=buf)
    assert none is None and score3.data_ptr() == buf.data_ptr() and bits_equal(buf.cpu().numpy(), o_score)


@pytest.fixture(scope="module")
def rccl_world1(dev):
    """A one-rank RCCL process group in the test process: the collectives are real RCCL calls (self-exchange)."""
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=dev)
    yield
    dist.destroy_process_group()


def test_sharded_world1_rccl_matches_unsharded_and_never_syncs(rccl_world1, ops, dev):
    from mi_oov import sharded
    rng = np.random.default_rng(1)
    N, B = 50_000, 4097
    feat = T(rng.standard_normal((N, 64), dtype=np.float32), dev)
    planes = T(rng.standard_normal((8, 64), dtype=np.float32), dev)
    buckets = T(rng.standard_normal((8, 64), dtype=np.float32), dev)
    ids_np = rng.integers(0, N, size=B).astype(np.int64)
    ids_np[5], ids_np[6] = -1, N + 9
    ids = T(ids_np, dev)
    other = T(rng.standard_normal((B, 64), dtype=np.float32), dev)
    table = sharded.ShardedLSHTable(feat, N, max_batch=B)
    want_emb = ops.lsh_embed(ids, feat, planes, buckets)
    want_score = ops.lsh_embed_score(ids, feat, planes, buckets, other)
    assert bits_equal(table.embed(ids, planes, buckets).cpu().numpy(), want_emb.cpu().numpy())  # also warms RCCL up
    out = torch.empty((B,), device=dev)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")  # any host synchronisation inside the step raises
    try:
        table.embed_score(ids, planes, buckets, other, score_out=out)
        steps = [ids, ids.flip(0).contiguous(), ids.roll(7).contiguous(), ids]
        oth = [other, other.flip(0).contiguous(), other.roll(7, 0).contiguous(), other]
        sc = [torch.empty((B,), device=dev) for _ in steps]
        sharded.LshPipeline(table, planes, buckets).run(steps, oth, sc)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert bits_equal(out.cpu().numpy(), want_score.cpu().numpy())
    for i, o, s in zip(steps, oth, sc):
        assert bits_equal(s.cpu().numpy(), ops.lsh_embed_score(i, feat, planes, buckets, o).cpu().numpy())
    table.check_overflow()


def test_sharded_embedding_table_world1_rccl(rccl_world1, ops, dev):
    """ShardedEmbeddingTable through real RCCL collectives (one rank: everything is local, the exchange carries padding
    only): gather, the knn aggregate, the full-catalogue top-k and the lsh local fast path equal the unsharded kernels."""
    from mi_oov import sharded
    rng = np.random.default_rng(2)
    N, B, D = 30_001, 3000, 64
    W = T(rng.standard_normal((N, D), dtype=np.float32), dev)
    ids_np = rng.integers(0, N, size=B).astype(np.int64)
    ids_np[1], ids_np[2] = -4, N
    ids = T(ids_np, dev)
    tab = sharded.ShardedEmbeddingTable(W, N, max_batch=2 * B)
    assert bits_equal(tab.gather(ids).cpu().numpy(), ops.gather_rows(ids, W).cpu().numpy())
    idx2 = T(rng.integers(0, N, size=(B, 2)).astype(np.int64), dev)
    assert bits_equal(tab.gather_mean(idx2, 2).cpu().numpy(), ops.gather_mean(idx2, W, 2).cpu().numpy())
    U = T(rng.standard_normal((200, D), dtype=np.float32), dev)
    gv, gi = tab.topk(U, 20, 1)
    wv, wi = ops.score_topk(U, W, 20, 1)
    assert torch.equal(gi, wi) and torch.equal(gv, wv)
    feat = T(rng.standard_normal((N, 64), dtype=np.float32), dev)
    planes = T(rng.standard_normal((8, 64), dtype=np.float32), dev)
    buckets = T(rng.standard_normal((8, 64), dtype=np.float32), dev)
    other = T(rng.standard_normal((B, 64), dtype=np.float32), dev)
    lt = sharded.ShardedLSHTable(feat, N, max_batch=B)
    want = ops.lsh_embed_score(ids, feat, planes, buckets, other)
    assert bits_equal(lt.embed_score(ids, planes, buckets, other, local_fast=True).cpu().numpy(), want.cpu().numpy())
    tab.check_overflow()


def _rank_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fails = []
    try:
        import mi_oov  # noqa: F401
        from mi_oov import ops, sharded
        dev = torch.device("cuda:0")  # every rank on the one device of the box
        g = torch.Generator(device=dev).manual_seed(11)  # same tables on every rank
        N, B, F = 2_000_003, 65536, 64
        feat = torch.nn.functional.normalize(torch.randn((N, F), generator=g, device=dev), dim=-1)
        planes = torch.randn((8, F), generator=g, device=dev)
        buckets = torch.randn((8, 64), generator=g, device=dev)
        lo, hi, per = sharded.shard_bounds(N, world, rank)
        gr = torch.Generator(device=dev).manual_seed(100 + rank)  # every rank its own batch
        ids = torch.randint(0, N, (B,), generator=gr, device=dev)
        ids[3], ids[4] = -7, N
        other = torch.randn((B, 64), generator=gr, device=dev)
        want_score = ops.lsh_embed_score(ids, feat, planes, buckets, other)
        want_emb = ops.lsh_embed(ids, feat, planes, buckets)
        for cap_factor in (None, 1.0):
            table = sharded.ShardedLSHTable(feat[lo:hi].contiguous(), N, cap_factor=cap_factor, max_batch=B)
            if not torch.equal(torch.nan_to_num(table.embed(ids, planes, buckets), 7.0), torch.nan_to_num(want_emb, 7.0)):
                fails.append(f"lsh embed cap_factor={cap_factor}")
            sc = [torch.empty((B,), device=dev) for _ in range(4)]
            sharded.LshPipeline(table, planes, buckets).run([ids] * 4, [other] * 4, sc)
            for s in sc:
                if not torch.equal(torch.nan_to_num(s, 7.0), torch.nan_to_num(want_score, 7.0)):
                    fails.append(f"lsh pipeline cap_factor={cap_factor}")
            table.check_overflow()
        # BASELINE config 4 shape: slsh, 128-d rows, feature table AND bucket table row-sharded
        NB, D = N, 128
        n_pl = int(np.ceil(np.log2(NB)))
        planes_s = torch.randn((n_pl, F), generator=g, device=dev)
        big = torch.randn((NB, D), generator=g, device=dev)
        blo, bhi, _ = sharded.shard_bounds(NB, world, rank)
        window, win_lo = sharded.ShardedSLSHTable.gather_window(big[blo:bhi].contiguous(), NB, n_pl)
        st = sharded.ShardedSLSHTable(feat[lo:hi].contiguous(), N, window, win_lo, NB)
        got, gidx = st.embed(ids, planes_s)
        want = ops.slsh_embed(ids, feat, planes_s, big)
        widx = ops.slsh_index(ids, feat, planes_s, NB)
        if not torch.equal(gidx, widx):
            fails.append("slsh idx")
        if not torch.equal(torch.nan_to_num(got, 7.0), torch.nan_to_num(want, 7.0)):
            fails.append("slsh rows")
        # row e': the D-wide tables sharded -- BPR's in-vocabulary gather, the knn aggregate, the sharded full-catalogue
        # top-k on the HIP kernels per shard, and the lsh exchange with the local share served by the fused kernel
        et = sharded.ShardedEmbeddingTable(big[blo:bhi].contiguous(), NB)
        if not torch.equal(torch.nan_to_num(et.gather(ids), 7.0), torch.nan_to_num(ops.gather_rows(ids, big), 7.0)):
            fails.append("embedding gather")
        idx2 = torch.randint(0, NB, (B // 4, 2), generator=gr, device=dev)
        if not torch.equal(et.gather_mean(idx2, 2), ops.gather_mean(idx2, big, 2)):
            fails.append("embedding gather_mean")
        items = torch.randn((120_000, 64), generator=g, device=dev)
        items[100_000] = items[5]  # a tie across shards
        ilo, ihi, _ = sharded.shard_bounds(items.shape[0], world, rank)
        it = sharded.ShardedEmbeddingTable(items[ilo:ihi].contiguous(), items.shape[0])
        U = torch.randn((512, 64), generator=g, device=dev)  # users replicated
        gv, gi = it.topk(U, 20, 1)
        wv, wi = ops.score_topk(U, items, 20, 1)
        if not (torch.equal(gi, wi) and torch.equal(gv, wv)):
            fails.append("sharded top-k")
        lt = sharded.ShardedLSHTable(feat[lo:hi].contiguous(), N, max_batch=B)
        got = lt.embed_score(ids, planes, buckets, other, local_fast=True)
        if not torch.equal(torch.nan_to_num(got, 7.0), torch.nan_to_num(want_score, 7.0)):
            fails.append("lsh local fast path")
        sc = [torch.empty((B,), device=dev) for _ in range(3)]
        sharded.LshPipeline(lt, planes, buckets, local_fast=True).run([ids] * 3, [other] * 3, sc)
        for s_ in sc:
            if not torch.equal(torch.nan_to_num(s_, 7.0), torch.nan_to_num(want_score, 7.0)):
                fails.append("lsh pipeline local fast path")
        torch.cuda.synchronize()
        ret[rank] = fails
    finally:
        dist.destroy_process_group()


def test_world2_single_device_hip_local_compute(dev):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rank_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: [], 1: []}


def test_world3_single_device_hip_local_compute(dev):
    """Three ranks (shards of 666 668 / 666 668 / 666 667 rows): the exchange with the HIP kernels as local compute on a
    world size that is not a power of two."""
    world = 3
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rank_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: [], 1: [], 2: []}


def test_bench_self_spawned_two_ranks_sharded_line(dev):
    """`python bench.py --gpus 2` from a bare shell: the parent starts the ranks itself, the N > 1 headline is the
    row-sharded mode with the replicated figure beside it.  Two ranks on the one device, gloo rendezvous (rehearsal
    backend).  Run twice with a short clock ramp: ranks whose ramps settle at different times must still leave the
    ramp together (they once did not, and the exchanges of the two ranks then mismatched)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for rep in range(2):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--single-device", "--dist-backend",
                            "gloo", "--steps", "9", "--warmup", "3", "--items", "600001", "--batch", "8192", "--ring-mib", "64",
                            "--ramp-seconds", "0.05", "--batches-per-exchange", "4"],
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, p.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["steps"] == 9 and d["scaling"] == "weak" and d["config"]["backend"] == "gloo"
        assert "row-sharded over 2 ranks" in d["config"]["table"] and d["sharded"]["overflowed_lookups"] == 0
        assert d["sharded"]["bytes_on_wire_per_lookup"] == 16 and d["sharded"]["exchanges"] >= 3
        assert d["replicated"]["value"] > 0 and d["value"] > 0 and d["roofline"]["bound"] == "hbm"
