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
