#!/usr/bin/env python
"""bench.py -- GCC pre-training hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--mode train|sampler]

--mode train (default; BASELINE configs[1], configs[2] at N = 8): one "step" = one pass of the hot path over one
batch of bsz samples per GPU (2*bsz sampled subgraphs): draw seeds -> RWR walks -> induced subgraphs -> batched CSR
-> positional embedding -> GIN encoder q/k -> MoCo/InfoNCE -> backward -> clip + Adam -> EMA.  Default workload:
MoCo K=16384, bsz 256, rw_hops 256, restart 0.8 on the synthetic 1M-node/10M-edge power-law graph G1 (SURVEY.md
§8d), everything resident in HBM before the timed region.
--mode sampler (BASELINE configs[3]): the sampler alone (seed draw, walks, induction, batch packing) on the
10M-node/200M-edge graph G2; no collective.
--mode sample-ready (SURVEY.md 8d, C2 "sample-ready subgraphs/s"): sampler + device positional embedding on the producer
lanes, no training step -- what the reference's DataLoader workers deliver (graph_dataset.py:94-179 incl. the ARPACK
call of data_util.py:242-281), and the figure the >= 10x target is quoted on.
--mode e2e (BASELINE configs[0], train.py:396-401): in-batch negatives (K = bsz - 1, NCESoftmaxLossNS), both views
through `model`, fused clip + Adam, on G1 (small.bin is not obtainable offline); --batch-size 32 is the reference's
README setting, 256 the default here.  Single GPU (the reference has no data-parallel E2E mode).

Prints ONE JSON line on rank 0.  Multi-GPU: one process per GPU; when WORLD_SIZE is not set and --gpus N > 1 this
script re-launches itself under torch.distributed.run (rendezvous on 127.0.0.1).  The seed batch is sharded by rank
(rank r owns samples [step*N*bsz + r*bsz, +bsz)), the graph is replicated in every GPU's HBM, weak scaling.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")      # the command processor serves few queues well (tools/contention_probe.py)
os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")   # RCCL's stream must not share a hardware queue with a producer lane
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_sampler.json")
SAMPLER_SRC = os.path.join(ROOT, "gcc_amd", "csrc", "sampler.hip")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=192)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--mode", choices=["train", "sampler", "sample-ready", "e2e"], default="train")
    ap.add_argument("--batch-size", type=int, default=256)
    ap.add_argument("--nce-k", type=int, default=16384)
    ap.add_argument("--hidden-size", type=int, default=64,
                    help="train.py:93 (GraphEncoder output / hidden width, MemoryMoCo feature size).  Above 64 the step runs on the "
                         "any-width kernels (csrc/ginx.hip) launch by launch; the headline config is 64")
    ap.add_argument("--rw-hops", type=int, default=256)
    ap.add_argument("--restart-prob", type=float, default=0.8)
    ap.add_argument("--nodes", type=int, default=None, help="default 1,000,000 (train) / 10,000,000 (sampler)")
    ap.add_argument("--edges", type=int, default=None, help="default 10,000,000 (train) / 200,000,000 (sampler)")
    ap.add_argument("--run-seed", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the post-clock checker legs (parity_step / parity_batch / parity_posemb vs oracle/); they run by default, "
                         "with or without the CPU baseline")
    ap.add_argument("--posemb", choices=["device", "placeholder"], default="device")
    ap.add_argument("--lanes", type=int, default=2, help="producer streams (sampler + positional embedding)")
    ap.add_argument("--reserved-cus", type=int, default=0, help="compute units the producer streams are masked off (kept for the training step)")
    ap.add_argument("--cu-layout", default="interleaved", choices=["interleaved", "block"])
    ap.add_argument("--chunk", type=int, default=16, help="most steps a producer lane prepares per turn (one multi-view eigensolver call); "
                                                          "the largest divisor of --steps not above it is used, so that the timed region "
                                                          "launches exactly the chunks it consumes")
    ap.add_argument("--depth", type=int, default=2, help="chunks in flight per producer lane")
    ap.add_argument("--steady-steps", type=int, default=0, help="extra: run at least this many untimed steps before the clock (the first ~100 steps "
                                                                "of a process run 2-3 %% slower than the rest); 0 (default) = exactly --warmup "
                                                                "untimed steps.  The line's `warmup` field reports what was run")
    ap.add_argument("--collectives", action="store_true", help="--gpus 1: run the multi-GPU step (key all-gather + gradient all-reduce over "
                                                               "RCCL, segmented graph replay) on a 1-rank process group")
    ap.add_argument("--sampler-steps", type=int, default=16,
                    help="--mode sampler: steps (DataLoader batches) per sampler call (gcc_sample_multi); the training modes "
                         "sample a producer chunk per call")
    ap.add_argument("--ahead", type=int, default=None, help="chunks launched beyond the one being consumed (default lanes * (depth - 1))")
    ap.add_argument("--strict-streams", action="store_true", help="bracket every step with caller <-> step stream hand-offs (the API default; two event hops on the step's chain)")
    ap.add_argument("--e2e-graph", action="store_true", help="--mode e2e: replay the step as a captured hipGraph (off by default: no gain measured)")
    ap.add_argument("--no-graph", action="store_true", help="issue every step launch by launch instead of replaying the captured hipGraph of its ring slot")
    ap.add_argument("--hub-degree", type=int, default=0, help="member rows of at least this degree are not scanned by the induction "
                                                              "(0 = the library's default 512, -1 = scan every row: rounds 1-3)")
    ap.add_argument("--scratch-entries", type=int, default=0, help="induction scratch of the sampler (int32 slots); 0 = default")
    ap.add_argument("--edge-cap", type=int, default=0, help="edge capacity of a batch view; 0 = default")
    ap.add_argument("--pmc-traffic", type=float, default=None,
                    help="HBM bytes per launch of the roofline kernel from a separate rocprofv3 --pmc pass "
                         "(default: profiles/pmc_sampler.json if it was collected from this build of sampler.hip)")
    ap.add_argument("--posemb-fork", type=int, default=None, choices=[0, 1, 2],
                    help="gcc_posemb_set_fork: solver classes of a call on side streams (default: 1 in --mode sample-ready, where nothing "
                         "else needs the GPU; 0 in the training modes, where the in-order stream is the throttle the step needs)")
    ap.add_argument("--allow-posemb-flags", action="store_true",
                    help="do not fail when an eigen-iteration hit its restart cap (status bit 8); the count is reported either way")
    args = ap.parse_args(argv)
    if args.nodes is None:
        args.nodes = 10_000_000 if args.mode == "sampler" else 1_000_000
    if args.edges is None:
        args.edges = 200_000_000 if args.mode == "sampler" else 10_000_000
    return args


# ------------------------------------------------------------------ launcher ----
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(n, argv, port=None, script=None):
    """`python bench.py --gpus N` without a launcher: the command that runs N ranks of this script on this node."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()),
            script or os.path.abspath(__file__)] + list(argv)


def workload_name(args, world, v, e):
    """config.workload from what actually runs (not a fixed string)."""
    graph = f"synthetic power-law graph {args.nodes:,}-node/{args.edges:,}-edge requested ({v:,}/{e:,} after dedup and isolated-node removal)"
    if args.mode == "sampler":
        tag = "BASELINE configs[3]: " if (args.nodes, args.edges, args.rw_hops) == (10_000_000, 200_000_000, 256) \
            and abs(args.restart_prob - 0.8) < 1e-12 else ""
        return f"{tag}sampler-only rw_hops={args.rw_hops} restart={args.restart_prob} bsz={args.batch_size}/GPU, {graph}, {world}xMI355X"
    if args.mode == "sample-ready":
        tag = "BASELINE configs[1] data pipeline (SURVEY 8d C2 sample-ready): " \
            if (args.nodes, args.edges, args.batch_size, args.rw_hops) == (1_000_000, 10_000_000, 256, 256) else ""
        return (f"{tag}sampler + device positional embedding, no training step, bsz={args.batch_size}/GPU rw_hops={args.rw_hops} "
                f"restart={args.restart_prob}, {graph}, {world}xMI355X")
    if args.mode == "e2e":
        return (f"BASELINE configs[0] on G1: E2E K={args.batch_size - 1} (in-batch negatives, NCESoftmaxLossNS) bsz={args.batch_size} "
                f"hid=64 rw_hops={args.rw_hops} restart={args.restart_prob}, {graph}, {world}xMI355X")
    tag = ""
    hs = getattr(args, "hidden_size", 64)
    if (args.nodes, args.edges, args.batch_size, args.nce_k, args.rw_hops, hs) == (1_000_000, 10_000_000, 256, 16384, 256, 64):
        tag = "BASELINE configs[1]: " if world == 1 else ("BASELINE configs[2]: " if world == 8 else "BASELINE configs[1] per GPU: ")
    elif hs != 64:
        tag = f"--hidden-size {hs} (train.py:93; not a BASELINE config): "
    return (f"{tag}MoCo K={args.nce_k} m=0.999 bsz={args.batch_size}/GPU (global {args.batch_size * world}) rw_hops={args.rw_hops} "
            f"restart={args.restart_prob}, {graph}, {world}xMI355X")


def sampler_algorithmic_bytes(rp, views):
    """per subgraph 12*steps + sum_{v in S}(8 + 4 deg v) + 4n + 4(n+1) + 4 nnz,
    split by the kernel that moves them.  `views` = list of (csr dict, L array)."""
    deg = np.diff(rp).astype(np.int64)
    walk = induce = pack = 0
    for c, L in views:
        n = len(c["parent_nid"])
        nnz = len(c["col_idx"])
        walk += 12 * int(L.sum()) + 8 * n + 4 * 4 * n          # steps + row extents + node/extents written
        induce += int(4 * deg[c["parent_nid"]].sum()) + 4 * nnz + 4 * n   # row scans + hits + induced degrees
        pack += 4 * n + 4 * (n + 1) + 4 * nnz + 4 * nnz + 4 * n            # outputs (+ scratch re-read)
    return dict(walk=walk, induce=induce, pack=pack, total=walk + induce + pack)


def _posemb_init():
    os.environ["OMP_NUM_THREADS"] = "1"          # one ARPACK process per core, no BLAS oversubscription
    os.environ["MKL_NUM_THREADS"] = "1"
    os.environ["OPENBLAS_NUM_THREADS"] = "1"


def _posemb_chunk(job):
    from oracle import posemb as P

    node_off, row_ptr, col_idx, seed = job
    return P.batched_positional_embedding(node_off, row_ptr, col_idx, 32, seed=seed)


def cpu_baseline_sampler(rp, ci, args):
    """BASELINE configs[3] on the host cores ("port"): the C oracle of the sampler (seed draw, RWR walks, node set,
    induced subgraph, batching), OpenMP over subgraphs."""
    from oracle import sampler as O

    c = O.COracle()
    cores = os.cpu_count() or 1
    threads = min(c.max_threads(), cores)
    cdf = O.seed_cdf(rp)
    lt = O.max_nodes_table(int(np.diff(rp).max()), args.rw_hops, args.restart_prob)
    thr = O.restart_threshold(args.restart_prob)
    B = args.batch_size
    deg = np.diff(rp)
    done, first = 0, 10_000_000
    t0 = time.perf_counter()
    while True:
        seeds = c.draw_seeds(cdf, args.run_seed, first, B)
        L = lt[deg[seeds]]
        for view in range(2):
            c.sample_batch(rp, ci, seeds, L, view, args.run_seed, first, thr, threads=threads)
        done += 2 * B
        first += B
        dt = time.perf_counter() - t0
        if dt >= args.cpu_seconds:
            break
    return dict(value=done / dt, unit="subgraphs/s", cores=threads, kind="port",
                sample=f"{done} subgraphs ({done // (2 * B)} batches of bsz {B}, both views) in {dt:.1f}s: C sampler oracle, OpenMP x{threads}")


def cpu_baseline(rp, ci, args):
    """The CPU path on this box's host cores, two ways.

    ``value`` (kind "port", the STRONGER one): C oracle of the sampler (OpenMP over subgraphs) + SciPy ARPACK positional
    embedding (data_util.py:242-281, one process per core like the reference's DataLoader workers) + the torch-CPU
    oracle of encoder/head/Adam/EMA (train.py:378-431), the three stages PIPELINED one step deep the way the reference
    overlaps its workers with the trainer (the eigen-solves of step i + 1 run while step i trains): full steps.
    ``reference_shaped``: BASELINE.md's B-ref-N -- the reference's per-sample Python data pipeline in ``nproc`` worker
    processes and in train.py:49's default 12: sample-ready subgraphs/s, no training step."""
    import multiprocessing as mp

    import torch

    from oracle import encoder as E
    from oracle import sampler as O

    c = O.COracle()
    cores = os.cpu_count() or 1
    threads = min(c.max_threads(), cores)
    workers = cores
    torch.set_num_threads(min(cores, 32))
    cdf = O.seed_cdf(rp)
    lt = O.max_nodes_table(int(np.diff(rp).max()), args.rw_hops, args.restart_prob)
    thr = O.restart_threshold(args.restart_prob)
    B = args.batch_size
    deg = np.diff(rp)
    res = None
    if args.mode != "sample-ready":
        model, ema = E.OracleGraphEncoder(), E.OracleGraphEncoder()
        ema.load_state_dict(model.state_dict())
        memory = E.memory_init(args.nce_k, 64)
        opt = torch.optim.Adam(model.parameters(), lr=0.005, betas=(0.9, 0.999), weight_decay=1e-5)
        model.train()
        ema.train()
        index, done = 0, 0
        pool = mp.get_context("spawn").Pool(workers, initializer=_posemb_init)

        def submit(first):                       # sampler (C, OpenMP) now, the eigen-solves as asynchronous pool jobs
            seeds = c.draw_seeds(cdf, args.run_seed, first, B)
            L = lt[deg[seeds]]
            out = []
            for view in range(2):
                r = c.sample_batch(rp, ci, seeds, L, view, args.run_seed, first, thr, threads=threads)
                nb = len(r["node_off"]) - 1
                cuts = np.linspace(0, nb, workers + 1).astype(int)
                jobs = []
                for w in range(workers):
                    lo, hi = cuts[w], cuts[w + 1]
                    if hi > lo:
                        n0, n1 = r["node_off"][lo], r["node_off"][hi]
                        e0, e1 = r["row_ptr"][n0], r["row_ptr"][n1]
                        jobs.append((r["node_off"][lo:hi + 1] - n0, r["row_ptr"][n0:n1 + 1] - e0,
                                     r["col_idx"][e0:e1] - n0, first + w))
                out.append((r, pool.map_async(_posemb_chunk, jobs)))
            return out
        try:
            pool.map(_posemb_chunk, [(np.array([0, 3]), np.array([0, 2, 4, 6]), np.array([1, 2, 0, 2, 0, 1]), 0)] * workers)   # import scipy in every worker before the clock
            first = 10_000_000
            t0 = time.perf_counter()
            pending = submit(first)
            while True:
                nxt = submit(first + B)          # step i + 1's data pipeline runs under step i's training
                (rq, pq), (rk, pk) = [(r, torch.from_numpy(np.concatenate(a.get(timeout=300)))) for r, a in pending]
                keep = (torch.rand(5, B, 64) >= 0.5).float()
                fq = model(rq["node_off"].astype(np.int64), rq["row_ptr"].astype(np.int64),
                           rq["col_idx"].astype(np.int64), pq, dropout_masks=keep)
                if args.mode == "e2e":                           # train.py:396-401
                    keep_k = (torch.rand(5, B, 64) >= 0.5).float()
                    fk = model(rk["node_off"].astype(np.int64), rk["row_ptr"].astype(np.int64),
                               rk["col_idx"].astype(np.int64), pk, dropout_masks=keep_k)
                    loss = E.nce_softmax_loss_ns(fk @ fq.t() / 0.07)
                else:
                    with torch.no_grad():
                        fk = ema(rk["node_off"].astype(np.int64), rk["row_ptr"].astype(np.int64),
                                 rk["col_idx"].astype(np.int64), pk)
                    out, index = E.moco_forward(memory, index, fq, fk, 0.07)
                    loss = E.nce_softmax_loss(out)
                opt.zero_grad()
                loss.backward()
                torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
                opt.step()
                if args.mode != "e2e":
                    E.moment_update(model, ema, 0.999)
                done += 2 * B
                first += B
                pending = nxt
                dt = time.perf_counter() - t0
                if dt >= args.cpu_seconds:
                    break
        finally:
            pool.terminate()
        head = "encoder/E2E-NS/Adam" if args.mode == "e2e" else "encoder/MoCo/Adam/EMA"
        res = dict(value=done / dt, unit="subgraphs/s", cores=cores, kind="port",
                   sample=f"{done // (2 * B)} full steps of bsz {B} ({done} subgraphs) in {dt:.1f}s: C sampler oracle "
                          f"(OpenMP x{threads}) + SciPy eigsh pos-emb ({workers} processes) + torch-CPU "
                          f"{head} oracle ({torch.get_num_threads()} threads), stages pipelined one step deep")
    # BASELINE.md B-ref-N: the reference-shaped data pipeline on every host core, and on train.py:49's default 12 workers
    # (every core, a quarter of them capped at 64, and the reference's default 12: the per-sample Python pipeline does not scale
    #  to 256 processes on the GPU box's host -- 12 workers deliver 2.8 k subgraphs/s, 256 deliver 0.5-0.9 k: page faults of the
    #  per-seed O(|V|) clear and the interpreter's allocator across 256 address spaces -- so the line carries the curve and
    #  `best`, the figure the >= 10x statement is made against)
    shaped = {}
    for procs in sorted({cores, min(64, max(12, cores // 4)), min(12, cores)}, reverse=True):
        if procs <= cores:
            shaped["nproc" if procs == cores else "workers_%d" % procs] = cpu_baseline_reference_shaped(rp, ci, args, procs)
    if cores <= 12:
        shaped["workers_12"] = dict(shaped["nproc"], note="this host has no more than 12 cores: the same run")
    best = max(shaped, key=lambda k: shaped[k].get("value") or 0.0)
    shaped["best"] = dict(config=best, value=shaped[best].get("value"), cores=shaped[best].get("cores"))
    if res is None:                              # --mode sample-ready: the data pipeline IS the workload
        top = shaped[best]
        res = dict(value=top.get("value"), unit="subgraphs/s", cores=top.get("cores"), kind="port", sample=top.get("sample"))
    elif (shaped[best].get("value") or 0.0) > (res.get("value") or 0.0):
        # `value` is the STRONGEST CPU leg measured here (round 5's line carried the pipelined full-step port, 1.5 k/s, next to a
        # reference-shaped data pipeline that delivered 4.3 k/s with 64 workers): the reference-shaped pipeline alone -- no training
        # step -- bounds what the reference's CPU path can deliver from above; the full-step port stays in `full_step_port`
        top = shaped[best]
        res = dict(value=top.get("value"), unit="subgraphs/s", cores=top.get("cores"), kind="port",
                   sample=(top.get("sample") or "") + " -- the strongest CPU leg of this run (data pipeline only, an upper bound of the "
                                                       "reference-shaped CPU path; the pipelined full-step port: `full_step_port`)",
                   value_is="reference_shaped." + best, full_step_port=res)
    res["reference_shaped"] = shaped
    return res


def cpu_baseline_reference_shaped(rp, ci, args, procs):
    """SURVEY.md 8(d) baseline (i) / BASELINE.md B-ref-N: the reference's data pipeline the way the reference runs it -- a
    Python loop per sample mirroring graph_dataset.py:94-179 call for call (walker in C standing in for DGL's C++
    one, torch.unique, SciPy slicing for g.subgraph, SciPy ARPACK exactly as data_util.py:242-281), in ``procs`` worker
    processes -- sample-ready subgraphs/s, no encoder step.  Spawned processes (a forked child of a process holding a HIP
    context is unsafe); the graph reaches them as mapped files."""
    from tests.tools.cpu_baseline_reference_shaped import run_timed

    cores = os.cpu_count() or 1
    try:
        v, dt, n = run_timed(rp, ci, procs, max(4.0, 0.5 * args.cpu_seconds), clear=True, rw_hops=args.rw_hops,
                             restart=args.restart_prob, start="spawn")
    except Exception as e:                                   # the other figures stand on their own
        return dict(value=None, error=repr(e)[:200])
    return dict(value=v, unit="subgraphs/s", cores=procs, host_cores=cores, kind="port",
                sample=f"{n} samples (2 views each) in {dt:.1f}s: per-sample Python loop of graph_dataset.py:94-179 in {procs} "
                       f"worker processes (BASELINE.md B-ref-N; train.py:49 default num_workers=12), C walker + per-seed O(|V|) "
                       f"visit-count clear (DGL-recalled) + torch.unique + SciPy subgraph slicing + SciPy ARPACK pos-emb; data pipeline only")


def parity_batch(sampler, rp, ci, args, first_sample_id):
    """After the clock, in the checker leg: the batch with sample ids ``first_sample_id`` + [0, bsz) -- the first one the timed
    region produced -- sampled again by the device sampler and compared with oracle/sampler_oracle.c bit for bit: seeds,
    node ids, batched CSR of both views (graph_dataset.py:94-130, data_util.py:218-239, 26-32).  Raises on a mismatch."""
    from oracle import sampler as O

    c = O.COracle()
    B = args.batch_size
    q, k = sampler.sample(first_sample_id)
    sampler.check_status()
    seeds = c.draw_seeds(O.seed_cdf(rp), args.run_seed, first_sample_id, B)
    if sampler.last_seeds().cpu().numpy().tolist() != seeds.tolist():
        raise AssertionError("parity_batch: the device's seed draw differs from the oracle's")
    L = O.max_nodes_table(int(np.diff(rp).max()), args.rw_hops, args.restart_prob)[np.diff(rp)[seeds]]
    nodes = edges = 0
    for view, gb in enumerate((q, k)):
        ref = c.sample_batch(rp, ci, seeds, L, view, args.run_seed, first_sample_id, O.restart_threshold(args.restart_prob))
        got = gb.csr_numpy()
        for key in ("node_off", "edge_off", "parent_nid", "row_ptr", "col_idx"):
            if not np.array_equal(got[key], ref[key]):
                raise AssertionError(f"parity_batch: view {view} {key} differs from oracle/sampler_oracle.c")
        nodes += len(ref["parent_nid"])
        edges += len(ref["col_idx"])
    return dict(first_sample_id=int(first_sample_id), subgraphs=2 * B, nodes=int(nodes), edges=int(edges),
                checked="seeds, node_off, edge_off, parent_nid, row_ptr, col_idx of both views of the first timed batch, "
                        "re-sampled after the clock: bit-equal to oracle/sampler_oracle.c")


def parity_posemb(view, pos, evals, picks=6, tol=2e-4):
    """Checker leg of --mode sample-ready: the device positional embedding of a few subgraphs of one view against a dense
    float64 eigendecomposition of D^-1/2 A D^-1/2 (data_util.py:242-281) on the invariants tests/test_posemb_gpu.py uses:
    eigenvalues, residual, orthonormality of the raw vectors is not available here, so: eigenvalues and unit rows."""
    from oracle import posemb as P

    c = view.csr_numpy()
    no, rpv, civ = c["node_off"], c["row_ptr"], c["col_idx"]
    sizes = np.diff(no)
    order = np.argsort(sizes)
    chosen = sorted({int(order[int(f * (len(order) - 1))]) for f in np.linspace(0.05, 1.0, picks)})
    worst = 0.0
    for b in chosen:
        lo, hi = int(no[b]), int(no[b + 1])
        n = hi - lo
        k = min(n - 2, 32)
        if k <= 0:
            continue
        M = P.normalized_adjacency(rpv[lo:hi + 1] - rpv[lo], civ[rpv[lo]:rpv[hi]] - lo).toarray()
        s = np.linalg.eigvalsh(M)
        err = float(np.abs(evals[b, :k] - s[-k:]).max())
        worst = max(worst, err)
        if err > tol:
            raise AssertionError(f"parity_posemb: subgraph {b} (n = {n}): eigenvalues off by {err:.2e}")
        norms = np.linalg.norm(pos[lo:hi, :k], axis=1)
        if not np.all((np.abs(norms - 1) < 1e-4) | (norms == 0)):
            raise AssertionError(f"parity_posemb: subgraph {b}: rows are not unit vectors")
    return dict(subgraphs_checked=len(chosen), sizes=[int(sizes[b]) for b in chosen], worst_eigenvalue_error=worst, tol=tol,
                checked="top-k eigenvalues against numpy eigvalsh (float64) of D^-1/2 A D^-1/2, unit rows (data_util.py:242-281)")


def parity_step(args, graph, dev):
    """After the clock, in the checker leg: ONE fused step of the benchmarked trainer class at the benchmarked config
    (fresh weights, device sampler + device eigensolvers, host-drawn dropout masks) against oracle/encoder.py on the same
    batch -- the assertion tests/test_headline_parity_gpu.py makes, repeated on the box the number comes from.  Raises
    when loss / embeddings / gradients / post-step state leave north_star's 1e-3; returns the report for the line."""
    import torch

    from gcc_amd.contrast import MemoryMoCo
    from gcc_amd.encoder import GraphEncoder
    from gcc_amd.posemb import DevicePosEmb
    from gcc_amd.sampler import DeviceRWRSampler
    from gcc_amd.train_step import E2ETrainStep, MoCoTrainStep
    from tests.headline_step_check import check_e2e_step, check_moco_step

    B = args.batch_size
    HS = getattr(args, "hidden_size", 64)
    torch.manual_seed(12345)
    enc_kw = dict(positional_embedding_size=32, max_node_freq=16, max_edge_freq=16, max_degree=512,
                  freq_embedding_size=16, degree_embedding_size=16, output_dim=HS, node_hidden_dim=HS,
                  edge_hidden_dim=HS, num_layers=5, num_step_set2set=6, num_layer_set2set=3, norm=True,
                  gnn_model="gin", degree_input=True)
    smp = DeviceRWRSampler(graph, B, run_seed=args.run_seed, num_buffers=2)
    pe = DevicePosEmb(B, smp.node_cap, 32, device=dev, seed=args.run_seed, num_buffers=2, max_views=2)
    model = GraphEncoder(**enc_kw).to(dev)
    masks = [(torch.rand(5, B, max(HS, 64)) >= 0.5).float().to(dev).contiguous() for _ in range(2)]
    if args.mode == "e2e":
        tr = E2ETrainStep(model, smp, pe, prefetch=False)
        rep = check_e2e_step(tr, model, 0.005, masks[0], masks[1], sync=torch.cuda.synchronize, step_id=7)
    else:
        ema = GraphEncoder(**enc_kw).to(dev)
        ema.load_state_dict(model.state_dict())
        contrast = MemoryMoCo(HS, None, args.nce_k, 0.07, use_softmax=True).to(dev)
        tr = MoCoTrainStep(model, ema, contrast, smp, pe, prefetch=False)
        rep = check_moco_step(tr, model, ema, contrast, 0.005, masks[0], sync=torch.cuda.synchronize, step_id=7)
    rep.pop("_graphs", None)
    tr.check_status(strict_posemb=True)
    rep["checked"] = "one fused step vs oracle/encoder.py (fp32 and float64) on the same sampled batch: embeddings, loss, prob, grad norm, every gradient, Adam update, EMA, running statistics, queue -- all inside 1e-3 (tests/headline_step_check.py)"
    return rep


def wide_step_roofline(trainer, last, args, torch, reps=5):
    """--hidden-size above 64: the step's training-stream launches (encoder forward of both views, head, encoder backward, clip +
    Adam + EMA, enqueue) replayed on the last timed batch with the producers idle, timed with events on the step's stream; against
    the exact-f32 MFMA peak the FLOPs of every Linear (forward, data gradient, weight gradient: the strided 64 x 64 GEMM of
    csrc/ginx.hip, v_mfma_f32_16x16x4_f32).  The fraction is a LOWER bound of the GEMM kernel's own rate: the interval also holds
    the step's ~230 other launches (gathers, BatchNorm passes, column sums)."""
    q, k = last["graph_q"], last["graph_k"]
    B, W, K = args.batch_size, args.hidden_size, args.nce_k
    nq, nk = int(q.node_off[B].item()), int(k.node_off[B].item())
    L = trainer.L
    d_in = 32 + 16 + 1
    lin = lambda n: 2.0 * n * (d_in * W + W * W + (L - 1) * 2 * W * W)        # linears.0 / .1 of every layer, one pass over n rows
    pred = lambda: 2.0 * B * (d_in * W + L * W * W)                            # linears_prediction on the pooled rows
    fwd_q, fwd_k = lin(nq) + pred(), lin(nk) + pred()
    head = 2.0 * B * K * W * 2                                                  # logits + d loss / d q
    flops = fwd_q + fwd_k + 2.0 * fwd_q + head                                  # backward of q: data + weight gradient of every Linear
    st = trainer.main if trainer.main is not None else torch.cuda.current_stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        raw = st.cuda_stream
        for _ in range(2):
            segs, _res = trainer._body_wide(q, k, None, {}, raw)
            for _c, fn in segs:
                fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            segs, _res = trainer._body_wide(q, k, None, {}, raw)
            for _c, fn in segs:
                fn()
        e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ach = flops / (ms * 1e-3) / 1e12
    return dict(bound="mfma", kernel="ginx_gemm_kernel (every Linear of the any-width encoder and head: forward, data gradient, weight gradient)",
                achieved=ach, peak=F32_MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / F32_MFMA_PEAK_TFLOPS, traffic=None,
                gemm_gflop_per_step=flops / 1e9, training_stream_ms_isolated=ms, nodes_q=nq, nodes_k=nk, hidden=W,
                dominant_by="FLOPs: at width 256 the Linears are 94 GFLOP per step against < 1 GFLOP of everything else",
                note="achieved = GEMM FLOPs of a step / the step's whole training-stream interval (isolated, producers idle, HIP events on "
                     "the step's stream): a lower bound of the GEMM kernel's own rate.  The kernel is a plain 64 x 64 tile with scalar LDS "
                     "operand loads, one launch per operator (DESIGN 4f): the any-width path is a correctness path first")


def sampler_source_hash():
    h = hashlib.sha256()
    for name in ("sampler.hip", "device_compat.h", "host_common.h"):
        with open(os.path.join(ROOT, "gcc_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def committed_pmc_traffic(args, v, e, steps_per_call=1):
    """PMC counters cannot be collected from inside the benchmarked process; the figure comes from the separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes summarised by tools/pmc_sampler.py into
    profiles/pmc_sampler.json.  That file records the hash of the kernel source and the workload it was collected
    on; a file from another build or workload is refused (traffic = null) instead of quoted."""
    if args.pmc_traffic is not None:
        return args.pmc_traffic, "--pmc-traffic"
    if getattr(args, "hub_degree", 0) != 0:
        return None, "profiles/pmc_sampler.json was collected with the default hub-row settings, this run uses --hub-degree %d: refused" % args.hub_degree
    if not os.path.exists(PMC_FILE):
        return None, "no profiles/pmc_sampler.json"
    rec = json.load(open(PMC_FILE))
    if rec.get("source_sha256") != sampler_source_hash():
        return None, "profiles/pmc_sampler.json is stale (collected from another build of sampler.hip): refused"
    key = f"{v}/{e}/bsz{args.batch_size}/hops{args.rw_hops}" + (f"/steps{steps_per_call}" if steps_per_call > 1 else "")
    ent = rec.get("workloads", {}).get(key)
    if ent is None:
        return None, f"profiles/pmc_sampler.json has no entry for workload {key}"
    return ent["induce_kernel_hbm_bytes_per_launch"], f"profiles/pmc_sampler.json[{key}] ({ent.get('correction', 'raw')})"


def committed_pmc_entry(args, v, e, steps_per_call=1):
    """Per-kernel HBM bytes per launch of the committed PMC passes for this build and workload, or {} (refused / absent)."""
    if args.pmc_traffic is not None or getattr(args, "hub_degree", 0) != 0 or not os.path.exists(PMC_FILE):
        return {}
    rec = json.load(open(PMC_FILE))
    if rec.get("source_sha256") != sampler_source_hash():
        return {}
    key = f"{v}/{e}/bsz{args.batch_size}/hops{args.rw_hops}" + (f"/steps{steps_per_call}" if steps_per_call > 1 else "")
    ent = rec.get("workloads", {}).get(key) or {}
    return {k: float(vv["hbm_bytes_per_launch"]) for k, vv in ent.get("kernels", {}).items()}


SAMPLER_INTERVALS = (   # (name in kernel_ms_isolated, byte-model key, kernels of the interval)
    ("rwr_walk_kernel+prefix_a_kernel+records_kernel", "walk", ("rwr_walk_kernel", "prefix_a_kernel", "records_kernel")),
    ("induce_kernel", "induce", ("induce_kernel",)),
    ("prefix_b_kernel+pack_kernel+hub_write_kernel", "pack", ("prefix_b_kernel", "pack_kernel", "hub_write_kernel")))


def hbm_roofline(kernel, alg_bytes, ms, traffic, **more):
    """SURVEY 8(d) roofline object of an HBM-bound launch: `frac` prices the ALGORITHMIC bytes (what the reference algorithm
    would read and write), `frac_moved` the bytes the kernels really moved (PMC traffic) -- side by side, because a kernel
    that skips work (the unscanned hub rows) scores high on the first and low on the second."""
    ach = alg_bytes / (ms * 1e-3) / 1e9
    out = dict(bound="hbm", kernel=kernel, achieved=ach, peak=HBM_PEAK_GBPS, unit="GB/s", frac=ach / HBM_PEAK_GBPS,
               algorithmic_bytes_per_launch=alg_bytes, ms_per_launch=ms, traffic=traffic,
               frac_moved=(traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None)
    out.update(more)
    return out


PAIR_CLASS = os.environ.get("GCC_POSEMB_PAIR", "1") != "0"      # the 65..128 class on four-wave teams (csrc/posemb.hip; 0: 1,024-thread workgroups)
TEAM_SHARE = 3                                                  # such workgroups per CU: 256 threads at 168 registers (GCC_POSEMB_QUAD_OCC 3), 33 KiB of LDS
SOLVER_KERNELS = {"mid": "posemb_direct_kernel<1, 65, 128, 256, true, true>" if PAIR_CLASS else "posemb_direct_kernel<1, 65, 128, 1024, false>", "sparse-block": "posemb_cheb_kernel", "wave48": "posemb_wave_kernel<6, 48>",
                  "wave64": "posemb_wave_kernel<7, 64>", "small": "posemb_direct_kernel<0, 0, 64, 256, false>", "slot": "posemb_direct_kernel<2, ...>",
                  "big": "posemb_direct_kernel<4, ...>", "krylov": "posemb_krylov_kernel"}


def solver_roofline(pe_probe):
    """The eigensolver class that holds the most CU-time per step, priced against ONE compute unit's exact-f32 rate (a solver
    workgroup owns one CU; MI355X_MICROARCH.md: 157.3 TFLOP/s / 256 CUs): executed f32 FLOPs (the kernels' own counters) over
    the workgroups' residency time."""
    cls = pe_probe.get("classes") or {}
    if not cls:
        return None
    name = max(cls, key=lambda k: cls[k]["cu_ms_per_item"] * cls[k]["items"])
    c = cls[name]
    cu_s = c["cu_ms_per_item"] * c["items"] / 1e3
    total = sum(v["cu_ms_per_item"] * v["items"] for v in cls.values()) / 1e3
    ach = c["gflop"] / cu_s / 1e3 if cu_s > 0 else 0.0
    peak = F32_PER_CU_GFLOPS / 1e3
    # the same FLOPs over the kernel's WALL time share of the call against the whole chip (what a reader of `unit: TFLOP/s` expects
    # next to a per-CU fraction): the class's launch keeps `workgroups` CUs busy for cu_s / workgroups seconds
    wall_s = pe_probe.get("call_ms", 0.0) / 1e3
    return dict(bound="mfma", kernel=SOLVER_KERNELS.get(name, name), solver_class=name, achieved=ach, peak=peak, unit="TFLOP/s",
                frac=ach / peak, peak_note="ONE compute unit's exact-f32 rate (157.3 TFLOP/s / 256): a solver workgroup owns one CU",
                frac_of_chip=(c["gflop"] / 1e3 / wall_s / F32_MFMA_PEAK_TFLOPS) if wall_s > 0 else None,
                frac_of_chip_note="the class's executed FLOPs over the WALL time of the whole multi-view eigensolver call (all classes "
                                  "run side by side in it) against the chip's 157.3 TFLOP/s",
                traffic=None, items=c["items"], cu_ms_per_item=c["cu_ms_per_item"],
                share_of_solver_cu_time=cu_s / total if total > 0 else None,
                note="f32 FLOP-bound latency chains, not an HBM stream (the solvers move < 30 GB/s: profiles/r4_posemb_traffic.txt): "
                     "executed f32 FLOPs of the class / its workgroups' residency, against the exact-f32 MFMA/vector rate of the one CU a "
                     "solver workgroup occupies; measured by an isolated multi-view call after the clock with the kernels' tick / FLOP "
                     "counters on (gcc_posemb_debug_ticks)")


def sampler_probe(sampler, rp, first_id, args, nsample, lt, Prof, torch, steps_per_call=1, world=1):
    """Isolated probe loop: HIP-event durations of the three sampler intervals and the exact algorithmic bytes of the
    batches they produced (same kernels, same batch ids as the timed steps, GPU otherwise idle).  A launch covers
    ``steps_per_call`` consecutive steps (gcc_sample_multi: what the producer lanes / the sampler mode issue); bytes
    are per LAUNCH, i.e. summed over its steps."""
    acc = dict(walk=0, induce=0, pack=0, total=0)
    iso = np.zeros(3)
    deg = np.diff(rp)
    shape = dict(nodes_q=0.0, nodes_k=0.0, edges_q=0.0, edges_k=0.0)
    S = max(1, min(steps_per_call, getattr(sampler, "max_steps", 1)))
    B = sampler.batch_size
    ncalls = max(2, nsample // S)
    for i in range(-2, ncalls):                   # two untimed probe warm-ups
        pr = Prof(4)
        step0 = max(i, 0) * S
        if S > 1:
            pairs = sampler.sample_multi(first_id(step0), S, world * B, prof=pr)
        else:
            pairs = [sampler.sample(first_id(step0), prof=pr)]
        torch.cuda.synchronize()
        if i < 0:
            continue
        iso += np.array([pr.elapsed_ms(j, j + 1) for j in range(3)]) / ncalls
        seeds = sampler.last_seeds().cpu().numpy()
        for t, (q, k) in enumerate(pairs):
            L = lt[deg[seeds[t * B:(t + 1) * B]]]
            cq, ck = q.csr_numpy(), k.csr_numpy()
            bts = sampler_algorithmic_bytes(rp, [(cq, L), (ck, L)])
            for key in acc:
                acc[key] += bts[key] / ncalls
            shape["nodes_q"] += len(cq["parent_nid"]) / (ncalls * S)
            shape["nodes_k"] += len(ck["parent_nid"]) / (ncalls * S)
            shape["edges_q"] += len(cq["col_idx"]) / (ncalls * S)
            shape["edges_k"] += len(ck["col_idx"]) / (ncalls * S)
    # the event marks sit around groups of launches: walk + prefix step A | induction alone | prefix step B + pack
    return {"rwr_walk_kernel+prefix_a_kernel+records_kernel": float(iso[0]), "induce_kernel": float(iso[1]),
            "prefix_b_kernel+pack_kernel+hub_write_kernel": float(iso[2])}, acc, shape, S


F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: exact-f32 MFMA = the f32 vector rate
F32_PER_CU_GFLOPS = 157300.0 / 256


def posemb_probe(sampler, posemb, first_id, nsteps, torch):
    """Isolated eigensolver call over the views of ``nsteps`` steps with the library's per-class tick and FLOP counters
    on (gcc_posemb_debug_ticks): wall time of the call, CU-time and executed f32 FLOPs per solver class."""
    from gcc_amd import _cabi

    lib = _cabi.load()
    NCLS = 8                                                     # GCC_POSEMB_TICK_CLASSES (include/gcc_amd.h)
    views = [g for s in range(nsteps) for g in sampler.sample(first_id(s))]
    posemb.multi(views)
    torch.cuda.synchronize()
    ticks = torch.zeros(NCLS * 16, dtype=torch.int64, device=views[0].node_off.device)
    lib.gcc_posemb_debug_ticks(ticks.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    posemb.multi(views)
    e1.record()
    torch.cuda.synchronize()
    lib.gcc_posemb_debug_ticks(None)
    t = ticks.cpu().numpy().reshape(NCLS, 16)
    out = dict(views=len(views), call_ms=float(e0.elapsed_time(e1)), classes={})
    for c, name in enumerate(["small", "mid", "slot", "krylov", "big", "sparse-block", "wave48", "wave64"]):
        items = int(t[c, 15])
        if items == 0:
            continue
        # 100 MHz ticks of one workgroup (one-wave classes: of one wave, four of which share a workgroup; the four-wave teams of the
        # 65..128 class: of one 256-thread workgroup, TEAM_SHARE of which share a CU)
        cu_s = float(t[c, :14].sum()) / 1e8 / (4 if name.startswith("wave") else TEAM_SHARE if (name == "mid" and PAIR_CLASS) else 1)
        flops = float(t[c, 14])
        out["classes"][name] = dict(items=items, cu_ms_per_item=cu_s * 1e3 / items, gflop=flops / 1e9,
                                    frac_of_cu_f32_peak=(flops / cu_s / 1e9 / F32_PER_CU_GFLOPS) if cu_s > 0 and flops > 0 else None)
    out["cu_seconds_per_step"] = sum(v["cu_ms_per_item"] * v["items"] for v in out["classes"].values()) / 1e3 / nsteps
    return out


def stage_rooflines(args, acc, kern_iso, stage_ms, shape, pe_probe):
    """What bounds the step, stage by stage (SURVEY.md 8d byte / FLOP models; the 8(d) `roofline` object prices one
    kernel only).  Sampler: isolated intervals.  Encoder / head: in-step HIP-event intervals on the training stream
    (they include contention with the producer lanes)."""
    H, B = 64, args.batch_size
    out = {}
    t_s = sum(kern_iso.values())
    out["sampler_end_to_end"] = dict(bound="hbm", algorithmic_bytes_per_step=acc["total"], ms_isolated=t_s,
                                     achieved=acc["total"] / (t_s * 1e-3) / 1e9, peak=HBM_PEAK_GBPS, unit="GB/s",
                                     frac=acc["total"] / (t_s * 1e-3) / 1e9 / HBM_PEAK_GBPS)
    L = 4                                                      # GIN layers with an MLP (num_layer 5 -> 4 GINConv)
    both = args.mode == "e2e"                                  # E2E: both views are differentiated
    nf, ef = shape["nodes_q"] + shape["nodes_k"], shape["edges_q"] + shape["edges_k"]
    nb, eb = (nf, ef) if both else (shape["nodes_q"], shape["edges_q"])

    def enc(n, e, bwd):
        # per layer: SpMM bytes = 2 N d s + 4 nnz + 4 (N + 1), flops 2 nnz d; MLP bytes = in + out rows of both
        # Linears, flops 2 N (d d + d d); the three BatchNorms add a read of z1 / z2 each in training mode.  Backward ~ 2x.
        by = L * ((2 * n * H * 4 + 4 * e + 4 * (n + 1)) + 4 * n * H * 4 + 3 * n * H * 4)
        fl = L * (2 * e * H + 2 * n * 2 * H * H)
        return (2 * by, 2 * fl) if bwd else (by, fl)
    for name, (by, fl), ms in (("gin_encoder_fwd", enc(nf, ef, False), stage_ms.get("gin_fwd")),
                               ("gin_encoder_bwd", enc(nb, eb, True), stage_ms.get("gin_bwd"))):
        if ms:
            out[name] = dict(bound="hbm", algorithmic_bytes=by, algorithmic_flops=fl, ms_in_step=ms,
                             achieved=by / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBPS, unit="GB/s",
                             frac=by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                             f32_mfma_frac=fl / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS,
                             note="latency-bound chain of small kernels: neither roof is near")
    K = B if args.mode == "e2e" else args.nce_k
    for name, mult in (("nce_fwd", 1), ("nce_bwd", 2)):
        ms = stage_ms.get(name)
        if ms:
            fl, by = mult * 2 * B * (K + 1) * H, mult * (K * H * 4 + 2 * B * H * 4)
            out["infonce_" + name[4:]] = dict(bound="hbm", algorithmic_bytes=by, algorithmic_flops=fl, ms_in_step=ms,
                                              achieved=by / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBPS, unit="GB/s",
                                              frac=by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                              f32_mfma_frac=fl / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS)
    # the whole step against the HBM roof: every stage's algorithmic bytes over the measured step time (the 8(d) `roofline`
    # object prices induce_kernel, ~2 % of the step; this is the figure that says where the step as a whole stands)
    out["_step_bytes"] = acc["total"] + sum(v["algorithmic_bytes"] for k, v in out.items()
                                            if k.startswith(("gin_encoder", "infonce")))
    if pe_probe:
        out["positional_embedding"] = dict(bound="f32 vector/MFMA rate of the CUs a solver workgroup occupies",
                                           peak_per_cu=F32_PER_CU_GFLOPS, unit="GFLOP/s per CU", **pe_probe)
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(subprocess.call(launcher_command(args.gpus, sys.argv[1:])))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.mode == "e2e" and world > 1:
        raise SystemExit("--mode e2e is single-GPU (train.py has no data-parallel E2E step)")
    assert torch.cuda.is_available(), "bench.py needs a GPU (run through gpurun)"
    ndev = torch.cuda.device_count()
    oversubscribed = ndev < world              # fewer devices than ranks (a 1-GPU box): ranks share devices, correctness only
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if oversubscribed:                     # RCCL refuses two ranks on one device; gloo moves the same buffers
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    elif args.collectives:                     # the multi-GPU step's launch path on a one-rank RCCL group
        if args.mode != "train":
            raise SystemExit("--collectives is the MoCo step's data-parallel path (--mode train)")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    from gcc_amd.graph import DeviceGraph
    from gcc_amd.graphgen import powerlaw_graph
    from gcc_amd.prof import Prof
    from gcc_amd.sampler import DeviceRWRSampler

    rp, ci = powerlaw_graph(args.nodes, args.edges, seed=0)
    V, E = int(len(rp) - 1), int(len(ci))
    graph = DeviceGraph(rp, ci, rw_hops=args.rw_hops, restart_prob=args.restart_prob, device=dev, validate=False, trusted=True)
    B = args.batch_size
    torch.manual_seed(0)

    def first_id(step):
        return (step * world + rank) * B

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    extra = {}
    if args.posemb_fork is not None and args.mode in ("train", "e2e") and torch.cuda.is_available():
        from gcc_amd import _cabi as _cabi_fork0

        _cabi_fork0.load().gcc_posemb_set_fork(args.posemb_fork)
        extra["posemb_fork"] = args.posemb_fork
    if args.mode == "sampler":
        S = max(1, min(args.sampler_steps, args.steps))
        sampler = DeviceRWRSampler(graph, B, run_seed=args.run_seed, num_buffers=max(2, S), scratch_entries=args.scratch_entries or None,
                                   edge_cap=args.edge_cap or None, max_steps=S, hub_degree=args.hub_degree)
        S = sampler.max_steps
        samplers = [sampler]

        def run_steps(first_step, count):            # `count` steps, S (or fewer) per call
            at = 0
            while at < count:
                n = min(S, count - at)
                sampler.sample_multi(first_id(first_step + at), n, world * B)
                at += n

        run_steps(0, args.warmup)
        barrier()
        t0 = time.perf_counter()
        run_steps(args.warmup, args.steps)
        barrier()
        dt = time.perf_counter() - t0
        extra["sampler_steps_per_call"] = S
        stages = ["seed-draw", "rwr-walk", "induce", "batch-pack"]
        produced = consumed = args.steps
        first_timed = args.warmup
        chunk = 1
    elif args.mode == "sample-ready":
        from gcc_amd.posemb import DevicePosEmb
        from gcc_amd import _cabi as _cabi_fork

        fork_mode = 1 if args.posemb_fork is None else args.posemb_fork
        if "GCC_POSEMB_FORK" not in os.environ:
            _cabi_fork.load().gcc_posemb_set_fork(fork_mode)
        extra["posemb_fork"] = fork_mode if "GCC_POSEMB_FORK" not in os.environ else int(os.environ["GCC_POSEMB_FORK"])

        # the producer lanes of the training modes without a consumer: lane l samples chunk c (c % lanes == l) -- `chunk`
        # steps per sampler call -- and runs ONE multi-view eigensolver call over its 2 * chunk views, on its own stream
        chunk = max(d for d in range(1, min(args.chunk, args.steps) + 1) if args.steps % d == 0)
        if args.warmup % chunk:
            chunk = max(d for d in range(1, chunk + 1) if args.steps % d == 0 and args.warmup % d == 0)
        samplers = [DeviceRWRSampler(graph, B, run_seed=args.run_seed, num_buffers=args.depth * chunk, scratch_entries=args.scratch_entries or None,
                                     edge_cap=args.edge_cap or None, max_steps=chunk, hub_degree=args.hub_degree) for _ in range(args.lanes)]
        sampler = samplers[0]
        posembs = [DevicePosEmb(B, sampler.node_cap, 32, device=dev, seed=args.run_seed, num_buffers=args.depth * chunk,
                                max_views=min(2 * chunk, 32)) for _ in range(args.lanes)]
        posemb = posembs[0]
        lane_streams = [torch.cuda.Stream(dev) for _ in range(args.lanes)]

        def run_steps(first_step, count):
            for cidx in range(first_step // chunk, (first_step + count) // chunk):
                lane = cidx % args.lanes
                with torch.cuda.stream(lane_streams[lane]):
                    pairs = samplers[lane].sample_multi(first_id(cidx * chunk), chunk, world * B)
                    posembs[lane].multi([g for pair in pairs for g in pair])

        run_steps(0, args.warmup)
        barrier()
        t0 = time.perf_counter()
        run_steps(args.warmup, args.steps)
        barrier()
        dt = time.perf_counter() - t0
        first_timed = args.warmup
        produced = consumed = args.steps
        stages = ["seed-draw", "rwr-walk", "induce", "batch-pack", "pos-emb:" + posemb.kind]
        sts = [p.status.cpu().tolist() for p in posembs]
        extra["posemb_status"] = dict(flags=int(np.bitwise_or.reduce([int(st[0]) for st in sts])),
                                      items_handed_on=int(sum(st[3] for st in sts)), items_failed=int(sum(st[4] for st in sts)))
        for pe_ in posembs:
            pe_.check_status(strict=not args.allow_posemb_flags)
        extra["producer_lanes"], extra["producer_chunk"] = args.lanes, chunk
    else:
        from gcc_amd.contrast import MemoryMoCo
        from gcc_amd.encoder import GraphEncoder
        from gcc_amd.misc import warmup_linear
        from gcc_amd.posemb import DevicePosEmb, PlaceholderPosEmb
        from gcc_amd.train_step import E2ETrainStep, MoCoTrainStep

        # the timed region must produce exactly what it consumes: whole chunks only
        chunk = max(d for d in range(1, min(args.chunk, args.steps) + 1) if args.steps % d == 0)
        nbuf = args.depth * chunk
        samplers = [DeviceRWRSampler(graph, B, run_seed=args.run_seed, num_buffers=nbuf, scratch_entries=args.scratch_entries or None,
                                     edge_cap=args.edge_cap or None, max_steps=chunk, hub_degree=args.hub_degree) for _ in range(args.lanes)]
        sampler = samplers[0]
        HS = args.hidden_size
        wide = HS > 64
        if wide and args.mode == "e2e":
            raise SystemExit("--hidden-size above 64: --mode train (the wide MoCo step); the wide E2E step is the API path (train.py)")
        enc_kw = dict(positional_embedding_size=32, max_node_freq=16, max_edge_freq=16, max_degree=512,
                      freq_embedding_size=16, degree_embedding_size=16, output_dim=HS, node_hidden_dim=HS,
                      edge_hidden_dim=HS, num_layers=5, num_step_set2set=6, num_layer_set2set=3, norm=True,
                      gnn_model="gin", degree_input=True)                 # train.py:601-618 with default flags (--hidden-size: :93)
        model, model_ema = GraphEncoder(**enc_kw).to(dev), GraphEncoder(**enc_kw).to(dev)
        model_ema.load_state_dict(model.state_dict())                     # moment_update(model, model_ema, 0), train.py:624
        contrast = MemoryMoCo(HS, None, args.nce_k, 0.07, use_softmax=True).to(dev)
        if args.posemb == "device":
            posembs = [DevicePosEmb(B, sampler.node_cap, 32, device=dev, seed=args.run_seed, num_buffers=nbuf,
                                    max_views=min(2 * chunk, 32)) for _ in range(args.lanes)]
        else:
            posembs = [PlaceholderPosEmb(sampler.node_cap, 32, device=dev)] * args.lanes
        posemb = posembs[0]
        # every producer lane: one sampler + one eigensolver workspace, `chunk` steps (2 * chunk views) per turn
        lanes = [(samplers[i], posembs[i]) for i in range(args.lanes)]
        if args.mode == "e2e":
            trainer = E2ETrainStep(model, sampler, posemb, nce_t=0.07, lanes=lanes, depth=args.depth, chunk=chunk,
                                   ahead=args.ahead, graph=args.e2e_graph and not args.no_graph)
            stages = ["seed-draw", "rwr-walk", "induce", "batch-pack", "pos-emb:" + posemb.kind, "gin-encoder fwd(q), fwd(k)",
                      "in-batch infonce (NS) fwd", "infonce bwd (dq, dk)", "gin-encoder bwd(q) + bwd(k)", "clip", "adam + meters (one launch)"]
        else:
            trainer = MoCoTrainStep(model, model_ema, contrast, sampler, posemb, world_size=world, rank=rank,
                                    lanes=lanes, depth=args.depth, chunk=chunk, reserved_cus=args.reserved_cus,
                                    cu_layout=args.cu_layout, ahead=args.ahead, graph=False if (args.no_graph or wide) else None,
                                    collectives=True if args.collectives else None)
            stages = ["seed-draw", "rwr-walk", "induce", "batch-pack", "pos-emb:" + posemb.kind, "gin-encoder q+k fwd",
                      "moco-infonce fwd", "infonce bwd", "gin-encoder bwd", "grad all-reduce" if (world > 1 or args.collectives) else "clip",
                      "adam + ema + meters (one launch)",
                      "key all-gather (overlapped since the encoder fwd) + enqueue" if (world > 1 or args.collectives) else "enqueue"]
        trainer.relaxed_streams = not args.strict_streams                  # results are read behind device synchronisations only
        n_batch = 2000 * 12 // 32                                          # train.py:356 with default flags
        total_steps = 100 * n_batch

        def lr_at(step):
            return 0.005 * warmup_linear(step / total_steps, 0.1)          # train.py:411-414

        names = ("gin_fwd", "nce_fwd", "nce_bwd", "gin_bwd")

        # The data pipeline is primed first (chunk 0 and the producers' look-ahead launched and complete: what a DataLoader's
        # workers do before the first iteration, no training step), then EXACTLY --warmup untimed steps run, then the clock
        # covers EXACTLY --steps steps.  `steps` is a multiple of `chunk`, so the timed region launches as many chunks as it
        # consumes (checked below: produced_steps == consumed_steps).  --steady-steps N (default 0) adds untimed steps up to
        # N: the first ~100 steps of a process run 2-3 % slower than the rest (profiles/r4_bench_window_warmup.txt); the
        # line's `warmup` field reports what was actually run.
        trainer.producer.prefill()
        torch.cuda.synchronize()
        warm = max(args.warmup, args.steady_steps)
        for i in range(warm):
            trainer.step(i, lr_at(i))
        first_timed = warm
        # with graph replay the step's launches carry no event marks (they are one graph launch): the timed steps mark the
        # producers only and the in-step stage intervals come from a few eager steps after the clock
        graph_on = bool(getattr(trainer, "use_graph", False))
        if wide:
            names = ()                                                     # (the any-width engine records no in-step marks)
        profs = [dict(sampler=Prof(4), **({} if graph_on else {n: Prof(2) for n in names})) for _ in range(args.steps)]
        if args.posemb == "device":
            for p in profs:
                p["posemb"] = Prof(2)
        barrier()
        launched0 = trainer.producer.launched
        late0 = (trainer.producer.late_chunks, trainer.producer.late_wait_s)
        trainer.graph_replays_at_clock = getattr(trainer, "graph_replays", 0)
        t0 = time.perf_counter()
        for i in range(args.steps):
            last = trainer.step(first_timed + i, lr_at(first_timed + i), prof=profs[i])
        barrier()
        dt = time.perf_counter() - t0
        produced = (trainer.producer.launched - launched0) * chunk
        consumed = args.steps
        extra["producer_late"] = dict(chunks=trainer.producer.late_chunks - late0[0], chunks_consumed=args.steps // chunk,
                                      wait_ms_per_step=(trainer.producer.late_wait_s - late0[1]) * 1e3 / args.steps,
                                      note="chunks whose data was not complete when the HOST reached their first step (it waits for the chunk's "
                                           "event before it reads its status word), and that host wait spread over the timed steps.  The host "
                                           "runs up to a chunk ahead of the device, so this is how the host paces itself, not a device stall: "
                                           "more look-ahead (--depth 3) does not change the step time, a third lane makes it slower "
                                           "(profiles/r6_producer_lateness.txt)")
        # two more windows of the same length right after the clock (diagnostics: `value` comes from the first window alone): a
        # 20-step window is 16 ms, and the line should say how much such a window moves from one to the next
        windows = [dt / args.steps * 1e3]
        nxt_step = first_timed + args.steps
        if world == 1 and args.steps <= 64:
            for _w in range(2):
                torch.cuda.synchronize()
                tw0 = time.perf_counter()
                for i in range(args.steps):
                    last = trainer.step(nxt_step + i, lr_at(nxt_step + i))
                torch.cuda.synchronize()
                windows.append((time.perf_counter() - tw0) / args.steps * 1e3)
                nxt_step += args.steps
        extra["ms_per_step_windows"] = windows
        extra["ms_per_step_windows_note"] = "the timed window (= ms_per_step) and two more windows of --steps steps run right after it"
        first_after = nxt_step
        extra["final_loss"] = float(last["loss"].item())
        extra["step_launch"] = ("hipGraph replay (one captured graph per ring slot)" if graph_on else "eager (launch by launch)") + \
            ("" if trainer.relaxed_streams else "; caller <-> step stream hand-offs around every step")
        stage_profs = profs
        if graph_on:
            extra["graph_capture_failures"] = int(getattr(trainer, "graph_capture_failures", 0))
            extra["graph_replays_in_timed_region"] = int(trainer.graph_replays - (trainer.graph_replays_at_clock if hasattr(trainer, "graph_replays_at_clock") else 0))
            trainer.use_graph = False                                     # stage intervals: eager steps right after the clock
            stage_profs = [{n: Prof(2) for n in names} for _ in range(chunk)]
            for i in range(chunk):
                trainer.step(first_after + i, lr_at(first_after + i), prof=stage_profs[i])
            torch.cuda.synchronize()
            trainer.use_graph = True
        extra["sampler_regrown"] = int(sum(getattr(sm, "regrown", 0) for sm in samplers))
        if hasattr(posemb, "status"):
            sts = [p.status.cpu().tolist() for p in posembs]
            flags = 0
            for st in sts:
                flags |= int(st[0])
            extra["posemb_status"] = dict(flags=flags, max_restart_cycles=int(max(st[1] for st in sts)),
                                          arnoldi_steps=int(sum(st[2] for st in sts)),
                                          items_handed_on=int(sum(st[3] for st in sts)),
                                          items_failed=int(sum(st[4] for st in sts)),
                                          first_failed=[st[5:14] for st in sts if st[4]][:1])
            for p in posembs:
                p.check_status(strict=not args.allow_posemb_flags)         # raises: a flagged eigen-solve is not a valid bench run

    for smp in samplers:
        smp.check_status()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if not oversubscribed else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        from oracle import sampler as O   # checker side only: L table for the byte count (after the clock)
        lt = O.max_nodes_table(int(np.diff(rp).max()), args.rw_hops, args.restart_prob)
        # roofline of the HBM-bound sampler kernel: durations AND byte counts from an ISOLATED probe loop (same
        # kernels, same batches as the timed steps, GPU otherwise idle).  Inside the timed region of --mode train the
        # producer streams overlap several sampler / eigensolver launches, so in-step marks measure contention.
        nsample = min(args.steps, 12)
        pe_probe = None
        # the launch shape the mode really issues: a producer chunk (training modes) / --sampler-steps (sampler mode)
        spc = chunk if args.mode != "sampler" else extra.get("sampler_steps_per_call", 1)
        nsample = max(nsample, 2 * spc)
        kern_iso, acc_call, probe_shape, spc = sampler_probe(sampler, rp, lambda i: first_id(first_timed + i), args, nsample, lt,
                                                             Prof, torch, steps_per_call=spc, world=world)
        acc = {k: v / spc for k, v in acc_call.items()}       # per STEP (stage_rooflines, algorithmic_bytes_per_step)
        if args.mode != "sampler" and (args.mode == "sample-ready" or args.posemb == "device"):
            pe_probe = posemb_probe(sampler, posemb, lambda i: first_id(first_timed + i), min(chunk, 8), torch)
        sampler.check_status()
        traffic, traffic_src = committed_pmc_traffic(args, V, E, spc)
        pmc = committed_pmc_entry(args, V, E, spc)
        ms_per_step = dt / args.steps * 1e3
        # one roofline object per sampler interval (HIP-event marks inside gcc_sample_multi): algorithmic bytes AND moved bytes
        samp_roofs = {}
        for iname, key, kerns in SAMPLER_INTERVALS:
            tr = sum(pmc.get(k, 0.0) for k in kerns) if all(k in pmc for k in kerns if k != "hub_write_kernel") else None
            if key == "induce" and tr is None:
                tr = traffic
            samp_roofs[key] = hbm_roofline(iname, acc_call[key], kern_iso[iname], tr, steps_per_launch=spc)
        samp_roofs["induce"].update(
            measured="isolated probe loop after the timed region: HIP events (gcc_prof marks on the launch stream) right before and after "
                     "the induction inside gcc_sample_multi (one call covers steps_per_launch consecutive batches, as the producer lanes "
                     "issue it; two launches of induce_kernel: the small and the big size class, the sum is what is timed); rocprofv3 "
                     "--kernel-trace --stats of the same launches alone: profiles/r5_kernel_stats_sampler_alone*.csv",
            note="algorithmic bytes = SURVEY 8(d): every member's parent row read once + the output.  The rows of the (at most 32) "
                 "highest-degree members of a subgraph are NOT read -- their induced rows are the mirror images of the other rows' hits "
                 "(symmetric graph) -- so `frac` is the rate of the reference algorithm's bytes (a replacement rate), `frac_moved` "
                 "the fraction of the HBM roof the kernel's own traffic reaches"
                 + ("" if args.hub_degree >= 0 else " (this run: --hub-degree -1, every row is read)"),
            traffic_source=traffic_src,
            traffic_note="committed constant from separate rocprofv3 --pmc passes of this build (hash-guarded), not a same-run measurement")
        sampler_dom = max(samp_roofs, key=lambda k: samp_roofs[k]["ms_per_launch"])
        sol_roof = solver_roofline(pe_probe) if pe_probe else None
        if sol_roof is not None:
            # training / sample-ready modes: the eigensolvers hold most of the GPU's kernel time (rocprofv3 --stats of this command:
            # profiles/r5_kernel_stats_default.csv); the line's `roofline` is THAT kernel, the sampler's objects follow
            roof = dict(sol_roof, dominant_by="CU-time per step among the solver classes (isolated multi-view call); share of all kernel "
                                              "time in the rocprofv3 --stats of this command: profiles/r5_kernel_stats_default.csv")
        else:
            roof = dict(samp_roofs[sampler_dom], dominant_by="largest isolated interval of a sampler call")
        out = {
            "metric": "sampled-subgraphs/sec", "value": 2 * B * world * args.steps / dt, "unit": "subgraphs/s",
            "n_gpus": world, "steps": args.steps, "warmup": first_timed, "warmup_requested": args.warmup,
            "untimed_steps": first_timed,
            "untimed_steps_note": "exactly the requested warm-up%s; the data pipeline is primed (first chunks sampled and embedded, no "
                                  "training step) before the warm-up steps" % (" extended by --steady-steps %d" % args.steady_steps if getattr(args, "steady_steps", 0) > args.warmup else ""),
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32" if args.mode == "sampler" else "f32", "data": "synthetic",
            "steps_per_sec": args.steps / dt,
            "produced_steps": produced, "consumed_steps": consumed,
            "config": {"workload": workload_name(args, world, V, E), "mode": args.mode,
                       "graph_nodes": V, "graph_edges": E,
                       "batch_size_per_gpu": B, "global_batch": B * world,
                       "nce_k": args.nce_k if args.mode == "train" else (B - 1 if args.mode == "e2e" else None),
                       "rw_hops": args.rw_hops, "restart_prob": args.restart_prob, "stages": stages,
                       "parallelism": f"dp{world} (seed batch sharded, graph replicated)" + ("; OVERSUBSCRIBED: %d ranks on %d device(s), gloo, correctness only" % (world, ndev) if oversubscribed else "")
                                      + ("; collectives forced on a 1-rank RCCL group (--collectives)" if args.collectives and world == 1 else "")},
            "kernel_ms_isolated": kern_iso, "kernel_ms_isolated_steps_per_launch": spc,
            "roofline": roof,
            "dominant_kernel": {"kernel": roof["kernel"], "bound": roof["bound"], "frac": roof["frac"], "by": roof["dominant_by"]},
            "sampler_rooflines": samp_roofs, "sampler_dominant_interval": sampler_dom,
            "algorithmic_bytes_per_step": acc,
        }
        if args.mode == "sampler":
            # the whole sampler (five kernels per call, `sampler_steps_per_call` steps per call) against the HBM roof
            e2e = acc["total"] / (ms_per_step * 1e-3) / 1e9
            moved = sum(v["traffic"] for v in samp_roofs.values()) / spc if all(v["traffic"] for v in samp_roofs.values()) else None
            out["stage_rooflines"] = {"sampler_end_to_end": dict(
                bound="hbm", algorithmic_bytes_per_step=acc["total"], ms_per_step=ms_per_step, achieved=e2e, peak=HBM_PEAK_GBPS,
                unit="GB/s", frac=e2e / HBM_PEAK_GBPS, moved_bytes_per_step=moved,
                frac_moved=(moved / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS) if moved else None,
                steps_per_call=extra.get("sampler_steps_per_call"), isolated_call_ms=sum(kern_iso.values()))}
        if args.mode == "sample-ready" and pe_probe:
            out["stage_rooflines"] = {"positional_embedding": dict(bound="f32 vector/MFMA rate of the CUs a solver workgroup occupies",
                                                                   peak_per_cu=F32_PER_CU_GFLOPS, unit="GFLOP/s per CU", **pe_probe)}
        if args.mode in ("train", "e2e"):
            out["config"].update(producer_lanes=args.lanes, producer_depth=args.depth, producer_chunk=chunk,
                                 producer_ahead=trainer.producer.ahead, reserved_cus=args.reserved_cus)
            used = [p for p in profs if p.get("used")]
            stage_ms = {n: float(np.mean([p[n].elapsed_ms(0, 1) for p in stage_profs])) for n in names}
            if used:
                k_ms = np.array([[p["sampler"].elapsed_ms(j, j + 1) for j in range(3)] for p in used])
                stage_ms["sampler"] = float(k_ms.sum(axis=1).mean())
                if args.posemb == "device":
                    stage_ms["posemb_chunk_of_%d_views" % min(2 * chunk, 32)] = float(
                        np.mean([p["posemb"].elapsed_ms(0, 1) for p in used]))
            out["stage_ms"] = stage_ms
            out["config"]["hidden_size"] = args.hidden_size
            if args.hidden_size > 64:
                # the any-width step: no per-stage marks; the line's roofline is the GEMM every Linear runs on (csrc/ginx.hip),
                # priced from an isolated replay of the step's training-stream launches on the last timed batch
                wr = wide_step_roofline(trainer, last, args, torch)
                out["solver_roofline"] = out["roofline"]
                out["roofline"] = wr
                out["dominant_kernel"] = {"kernel": wr["kernel"], "bound": wr["bound"], "frac": wr["frac"], "by": wr["dominant_by"]}
                out.update(extra)
                if not args.no_parity and world == 1:
                    out["parity"] = {"parity_step": parity_step(args, graph, dev)}
                sys.stdout.flush()
                print(json.dumps(out), flush=True)
                if world > 1:
                    dist.barrier()
                    dist.destroy_process_group()
                return
            out["stage_rooflines"] = stage_rooflines(args, acc, {k: v / spc for k, v in kern_iso.items()}, stage_ms, probe_shape, pe_probe)
            step_bytes = out["stage_rooflines"].pop("_step_bytes")
            step_gbps = step_bytes / (ms_per_step * 1e-3) / 1e9
            out["step_roofline"] = dict(
                bound="hbm", algorithmic_bytes_per_step=step_bytes, ms_per_step=ms_per_step, achieved=step_gbps,
                peak=HBM_PEAK_GBPS, unit="GB/s", frac=step_gbps / HBM_PEAK_GBPS,
                note="sum of the stages' SURVEY 8(d) algorithmic bytes (sampler + encoder fwd/bwd + head fwd/bwd; the "
                     "eigensolvers are FLOP/latency work and add none) over the timed step: the step is a latency-bound chain, "
                     "not an HBM stream -- stage_rooflines says which stage is how far from which roof")
        out.update(extra)
        if not args.no_cpu_baseline and world == 1:      # the CPU leg runs on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline_sampler(rp, ci, args) if args.mode == "sampler" else cpu_baseline(rp, ci, args)
            shaped = out["cpu_baseline"].get("reference_shaped") or {}
            if (shaped.get("nproc") or {}).get("value"):
                out["cpu_baseline"]["vs_reference_shaped_nproc"] = out["value"] / shaped["nproc"]["value"]
            if (shaped.get("best") or {}).get("value"):
                out["cpu_baseline"]["vs_reference_shaped_best"] = out["value"] / shaped["best"]["value"]
        if not args.no_parity and world == 1:
            # the checker legs, after the clock, whether or not the CPU baseline was timed (a line published with
            # --no-cpu-baseline carried no parity object before): the oracle as the checker, never as the thing measured
            par = {}
            if args.mode in ("train", "e2e") and args.posemb == "device":
                par["parity_step"] = parity_step(args, graph, dev)
            if args.mode in ("sampler", "sample-ready"):
                # the first batch the timed region produced, re-sampled and compared with the C oracle bit for bit
                par["parity_batch"] = parity_batch(sampler, rp, ci, args, first_id(first_timed))
            if args.mode == "sample-ready":
                q, _ = sampler.sample(first_id(first_timed))
                ev = torch.zeros(B, 32, device=dev)
                posemb(q, evals=ev)
                torch.cuda.synchronize()
                posemb.check_status(strict=True)
                par["parity_posemb"] = parity_posemb(q, q.pos_undirected.cpu().numpy(), ev.cpu().numpy())
            out["parity"] = par
            if "cpu_baseline" in out:            # (where rounds 3-5 published them)
                out["cpu_baseline"].update(par)
        try:                                     # (RCCL's version banner sits in the C library's stdout buffer until exit: out first,
            import ctypes                        #  so that the JSON line is the LAST line of stdout)
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
