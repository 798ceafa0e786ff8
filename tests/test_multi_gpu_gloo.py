"""N > 1 path on CPU: two processes over gloo run the fused MoCo step (emulator kernels) on
different seed-batch shards; checks the RCCL-side contract of SURVEY.md §8(e):
  * keys are all-gathered in rank order before the enqueue -> identical queues,
  * gradients are averaged -> identical weights / EMA weights on every rank, equal to the
    single-process result on the mean of the per-rank gradients,
  * the per-rank sample ids tile the global batch exactly (bit-exact sharding of the sampler)."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

GOLD_PATH = os.path.join(os.path.dirname(__file__), "golden", "encoder_golden.pt")


def _build(rank_views, world, rank):
    from gcc_amd.contrast import MemoryMoCo
    from gcc_amd.train_step import MoCoTrainStep
    from tests.hipemu.emu_encoder import CpuBatch, emu_engine, reference_encoder
    from tests.test_nce_emu import emu_nce

    gold = torch.load(GOLD_PATH, weights_only=False)
    g = gold["moco"]
    model, ema = reference_encoder(), reference_encoder()
    model.load_state_dict(g["init"]["model"])
    ema.load_state_dict(g["init"]["model_ema"])
    model._engine = ema._engine = emu_engine()
    contrast = MemoryMoCo(64, None, g["K"], g["T"], use_softmax=True)
    contrast._engine = emu_nce()
    contrast.memory.copy_(g["init"]["memory"])

    class Stub:
        batch_size = 6

        def sample(self, first_id, prof=None):
            return CpuBatch(gold["views"][rank_views[0]]), CpuBatch(gold["views"][rank_views[1]])

    step = MoCoTrainStep(model, ema, contrast, Stub(), posemb=lambda gr, prof=None: gr, prefetch=False,
                         world_size=world, rank=rank, clip_norm=0.0)   # raw gradients: clipping is nonlinear
    return gold, step, model, ema, contrast


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        views = (0, 1) if rank == 0 else (1, 0)          # the two ranks see different shards
        gold, step, model, ema, contrast = _build(views, world, rank)
        masks = gold["moco"]["masks"].contiguous()
        step.mask_fn = lambda: masks
        out = step.step(0, gold["moco"]["lr"])
        torch.save(dict(model=model.state_dict(), ema=ema.state_dict(), memory=contrast.memory.clone(),
                        index=contrast.index, grad=step.flat_grad.clone(), loss=out["loss"].clone(),
                        feat_k=step.gin._bufs and None), os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_step_over_gloo(tmp_path):
    port = 29500 + os.getpid() % 1000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    # identical replicas after the step
    for k in r0["model"]:
        if "running_" in k or "num_batches" in k:
            continue                                     # BatchNorm statistics are per rank
        torch.testing.assert_close(r0["model"][k], r1["model"][k], rtol=0, atol=0, msg=k)
        torch.testing.assert_close(r0["ema"][k], r1["ema"][k], rtol=0, atol=0, msg=k)
    torch.testing.assert_close(r0["memory"], r1["memory"], rtol=0, atol=0)
    torch.testing.assert_close(r0["grad"], r1["grad"], rtol=0, atol=0)
    assert r0["index"] == r1["index"] == 12              # 2 ranks x 6 keys enqueued
    # single-process references for each shard: averaged gradient and rank-ordered keys
    grads, keys = [], []
    for views in ((0, 1), (1, 0)):
        gold, step, model, ema, contrast = _build(views, 1, 0)
        masks = gold["moco"]["masks"].contiguous()
        step.mask_fn = lambda: masks
        step.step(0, gold["moco"]["lr"])
        grads.append(step.flat_grad.clone())
        keys.append(contrast.memory[:6].clone())         # this shard's keys were enqueued at rows 0..5
    torch.testing.assert_close(r0["grad"], (grads[0] + grads[1]) / 2, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(r0["memory"][:6], keys[0], rtol=1e-6, atol=1e-7)      # rank 0's keys first
    torch.testing.assert_close(r0["memory"][6:12], keys[1], rtol=1e-6, atol=1e-7)    # then rank 1's


def test_rank_shards_tile_the_global_batch_bit_exactly():
    from gcc_amd.graphgen import powerlaw_graph
    from tests.hipemu.emu_driver import EmuGraph, emu_sample_batch

    rp, ci = powerlaw_graph(3000, 30000, 3)
    g = EmuGraph(rp, ci, rw_hops=32)
    world, B, step = 4, 3, 5
    whole, _, seeds_whole = emu_sample_batch(g, world * B, 17, step * world * B)
    for view in range(2):
        parts = []
        for rank in range(world):
            res, status, seeds = emu_sample_batch(g, B, 17, (step * world + rank) * B)   # MoCoTrainStep._first_id
            assert status == 0
            assert seeds.tolist() == seeds_whole[rank * B:(rank + 1) * B].tolist()
            parts.append(res[view])
        assert np.array_equal(np.concatenate([p["parent_nid"] for p in parts]), whole[view]["parent_nid"])
        sizes = np.concatenate([np.diff(p["node_off"]) for p in parts])
        assert np.array_equal(sizes, np.diff(whole[view]["node_off"]))
        nnz = np.concatenate([np.diff(p["edge_off"]) for p in parts])
        assert np.array_equal(nnz, np.diff(whole[view]["edge_off"]))


# --------------------------------------------------------------------------------------------------------------------
# 8 ranks (the size the driver's scaling run uses): the real host code of MoCoTrainStep over gloo on the emulator kernels
def _worker8(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gcc_amd.contrast import MemoryMoCo
        from gcc_amd.train_step import MoCoTrainStep
        from oracle import encoder as E
        from tests.hipemu.emu_encoder import CpuBatch, emu_engine, reference_encoder
        from tests.test_nce_emu import emu_nce

        gold = torch.load(GOLD_PATH, weights_only=False)
        g = gold["moco"]
        B, K = 6, 2 * world * 6                                   # two steps fill the queue exactly once

        class Stub:
            batch_size = B
            ids = []

            def sample(self, first_id, prof=None):
                Stub.ids.append(first_id)
                v = (first_id // B) % 2                           # the shard decides which golden views this rank sees
                return CpuBatch(gold["views"][v]), CpuBatch(gold["views"][1 - v])

            def check_status(self):
                if rank == 5 and getattr(Stub, "fail", False):
                    raise RuntimeError("gcc_sample_batch overflow: induction scratch (injected on rank 5)")

        class Pos:
            def __call__(self, gr, prof=None):
                return gr

            def check_status(self, strict=False):
                return 8 if rank == 3 else (256 if rank == 6 else 0)     # flag words differ by rank: a bit mask, not a maximum

        def build(K_):
            model, ema = reference_encoder(), reference_encoder()
            model.load_state_dict(g["init"]["model"])
            ema.load_state_dict(g["init"]["model_ema"])
            model._engine = ema._engine = emu_engine()
            contrast = MemoryMoCo(64, None, K_, g["T"], use_softmax=True)
            contrast._engine = emu_nce()
            contrast.memory.copy_(E.memory_init(K_, 64, generator=torch.Generator().manual_seed(0)))
            return model, ema, contrast

        # (a) the queue must hold one step's keys of ALL ranks (memory_moco.py:55-61 enqueues every key of a step)
        try:
            MoCoTrainStep(*build(world * B - 1), Stub(), Pos(), prefetch=False, world_size=world, rank=rank)
            refused = False
        except ValueError as e:
            refused = "nce_k" in str(e) or "--nce-k" in str(e)
        model, ema, contrast = build(K)
        tr = MoCoTrainStep(model, ema, contrast, Stub(), Pos(), prefetch=False, world_size=world, rank=rank, clip_norm=1.0)
        masks = g["masks"].contiguous()
        tr.mask_fn = lambda: masks
        losses = [tr.step(s, g["lr"])["loss"].clone() for s in range(2)]
        # (b) status words: OR over ranks on every rank; an error on one rank raises on all
        flags = tr.check_status()
        Stub.fail = True
        try:
            tr.check_status()
            raised = None
        except RuntimeError as e:
            raised = str(e)
        torch.save(dict(model=model.state_dict(), ema=ema.state_dict(), memory=contrast.memory.clone(), index=contrast.index,
                        grad=tr.flat_grad.clone(), ids=list(Stub.ids), flags=flags, raised=raised, refused=refused,
                        loss=torch.stack(losses)), os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_eight_rank_step_over_gloo(tmp_path):
    """world_size 8 (the driver's SCALE run): queue divisibility, rank-rotated sample ids, rank-ordered key all-gather over
    two steps, one averaged gradient, agreement on status flags (bitwise OR) and on errors.  CPU / gloo / emulator kernels:
    correctness of the host code only -- no N > 1 throughput has been measured anywhere in this repository."""
    world, B = 8, 6
    port = 29700 + os.getpid() % 200
    mp.spawn(_worker8, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(tmp_path / f"rank{r}.pt", weights_only=False) for r in range(world)]
    for r, d in enumerate(rs):
        assert d["refused"], r                                        # (a)
        assert d["ids"] == [(s * world + r) * B for s in range(2)]    # rank-rotated shards: MoCoTrainStep._first_id
        assert d["flags"] == (8 | 256), (r, d["flags"])               # (b) OR of the ranks' words, the same everywhere
        assert d["raised"] is not None and ("rank 5" in d["raised"] or "another rank" in d["raised"]), (r, d["raised"])
        assert d["index"] == 0                                        # 2 steps x 8 ranks x 6 keys = K: wrapped exactly once
    for d in rs[1:]:                                                  # identical replicas
        torch.testing.assert_close(d["memory"], rs[0]["memory"], rtol=0, atol=0)
        torch.testing.assert_close(d["grad"], rs[0]["grad"], rtol=0, atol=0)
        for k in d["model"]:
            if "running_" in k or "num_batches" in k:
                continue
            torch.testing.assert_close(d["model"][k], rs[0]["model"][k], rtol=0, atol=0, msg=k)
            torch.testing.assert_close(d["ema"][k], rs[0]["ema"][k], rtol=0, atol=0, msg=k)
    # keys in rank order: ranks with the same shard parity enqueue the same keys, step 0 rows [0, 48), step 1 rows [48, 96)
    mem = rs[0]["memory"]
    for s in range(2):
        blocks = [mem[(s * world + r) * B:(s * world + r + 1) * B] for r in range(world)]
        for r in range(2, world):
            same = blocks[r - 2]                                      # shard parity repeats every 2 ranks
            torch.testing.assert_close(blocks[r], same, rtol=0, atol=0)
        assert not torch.equal(blocks[0], blocks[1])
    assert torch.isfinite(torch.stack([d["loss"] for d in rs])).all()


# --------------------------------------------------------------------------------------------------------------------
# --hidden-size above 64 (train.py:93) data parallel: the wide branch of MoCoTrainStep over gloo, two ranks, width 128
def _build_wide(rank_views, world, rank, hidden=128):
    from gcc_amd.contrast import MemoryMoCo
    from gcc_amd.train_step import MoCoTrainStep
    from tests.golden import wide_init
    from tests.hipemu.emu_encoder import CpuBatch
    from tests.test_nce_emu import emu_nce
    from tests.test_wide_encoder_emu import emu_wide_engine, emu_wide_nce, wide_encoder
    from tests.wide_golden_check import gold as wide_gold

    G = wide_gold()
    c = G["cases"][hidden]
    model, ema = wide_init.fill_(wide_encoder(hidden, hidden), 0), wide_init.fill_(wide_encoder(hidden, hidden), 1)
    model._wide_engine = ema._wide_engine = emu_wide_engine()
    contrast = MemoryMoCo(hidden, None, c["K"], c["T"], use_softmax=True)
    with torch.no_grad():
        contrast.memory.copy_(wide_init.tensor_for("contrast.memory", contrast.memory) * c["memory0_scale"])
    contrast._engine = emu_wide_nce()
    views = [CpuBatch(G["views"][v]) for v in rank_views]

    class Stub:
        batch_size = views[0].batch_size

        def sample(self, first_id, prof=None):
            return views[0], views[1]

    step = MoCoTrainStep(model, ema, contrast, Stub(), posemb=lambda gr, prof=None: gr, prefetch=False, world_size=world, rank=rank,
                         clip_norm=0.0, flat_engine=emu_nce())
    return c, step, model, ema, contrast


def _worker_wide(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c, step, model, ema, contrast = _build_wide((0, 1) if rank == 0 else (1, 0), world, rank)
        assert step.wide and step.collectives
        masks = c["masks"].contiguous()
        step.mask_fn = lambda: masks
        out = step.step(0, c["lr"])
        torch.save(dict(model=model.state_dict(), ema=ema.state_dict(), memory=contrast.memory.clone(), index=contrast.index,
                        grad=step.flat_grad.clone(), loss=out["loss"].clone()), os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_wide_step_over_gloo(tmp_path):
    port = 29500 + (os.getpid() + 7) % 1000
    mp.spawn(_worker_wide, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    for k in r0["model"]:
        if "running_" in k or "num_batches" in k:
            continue                                     # BatchNorm statistics are per rank
        torch.testing.assert_close(r0["model"][k], r1["model"][k], rtol=0, atol=0, msg=k)
        torch.testing.assert_close(r0["ema"][k], r1["ema"][k], rtol=0, atol=0, msg=k)
    torch.testing.assert_close(r0["memory"], r1["memory"], rtol=0, atol=0)
    torch.testing.assert_close(r0["grad"], r1["grad"], rtol=0, atol=0)
    grads, keys = [], []
    for views in ((0, 1), (1, 0)):
        c, step, model, ema, contrast = _build_wide(views, 1, 0)
        masks = c["masks"].contiguous()
        step.mask_fn = lambda: masks
        step.step(0, c["lr"])
        grads.append(step.flat_grad.clone())
        B = step.B
        keys.append(contrast.memory[:B].clone())
    assert r0["index"] == r1["index"] == 2 * B
    torch.testing.assert_close(r0["grad"], (grads[0] + grads[1]) / 2, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(r0["memory"][:B], keys[0], rtol=1e-6, atol=1e-7)          # rank 0's keys first
    torch.testing.assert_close(r0["memory"][B:2 * B], keys[1], rtol=1e-6, atol=1e-7)     # then rank 1's
