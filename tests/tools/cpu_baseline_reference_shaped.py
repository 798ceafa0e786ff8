"""SURVEY.md section 8(d) baseline (i): the reference's data pipeline the way the reference runs it -- a Python loop per
sample mirroring graph_dataset.py:94-179 call for call, in DataLoader-style worker processes -- timed on the host cores.

    python tests/tools/cpu_baseline_reference_shaped.py [--nodes 1000000 --edges 10000000] [--samples 256] [--procs 1,8]

Per sample (both views, graph_dataset.py:104-106): seed drawn from deg^0.75 (:85-92); max_nodes_per_seed (:113-124);
the walker -- DGL's C++ random_walk_with_restart in the reference, here the C helper of oracle/sampler_oracle.c called
once per sample (:125-130), optionally preceded by the O(|V|) visit-count clear DGL 0.4 performs per seed
([DGL-recalled]; reported with and without); torch.unique + seed-first node list (data_util.py:221-226); the induced
subgraph by SciPy row/column slicing of the parent CSR (standing in for DGL's g.subgraph, data_util.py:230); the
positional embedding with SciPy ARPACK exactly as data_util.py:242-281 (oracle/posemb.py).  A batch is the Python list
of these (dgl.batch, data_util.py:26-32, is not timed: no DGL here).  bench.py's `cpu_baseline` is the stronger "port"
(C sampler with OpenMP over subgraphs); this file is the weaker, reference-shaped one.  Test infrastructure: it imports
oracle/."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

_G = {}


def _init(rp, ci, rw_hops, restart, clear):
    # one thread per worker process, as a DataLoader worker has: without this every worker's OpenMP runtime (the C walker's
    # library) starts one thread per core -- 256 x 256 spinning threads on the GPU box made 256 workers 4x SLOWER than 12
    os.environ["OMP_NUM_THREADS"] = "1"
    import scipy.sparse as sp

    if isinstance(rp, str):                                 # paths of .npy files (run_timed with many workers: the graph is
        rp, ci = np.load(rp, mmap_mode="r"), np.load(ci, mmap_mode="r")   # mapped, not pickled into every worker)
        rp, ci = np.asarray(rp), np.asarray(ci)
    import torch

    from oracle import posemb as P
    from oracle import sampler as O

    from threadpoolctl import threadpool_limits

    _G["blas_limit"] = threadpool_limits(1)                 # one ARPACK/BLAS thread per worker process, as a DataLoader worker has
    torch.set_num_threads(1)
    deg = np.diff(rp)
    _G.update(rp=rp, ci=ci, deg=deg, co=O.COracle(), cdf=O.seed_cdf(rp), lt=O.max_nodes_table(int(deg.max()), rw_hops, restart),
              thr=O.restart_threshold(restart), clear=clear, P=P, O=O, torch=torch,
              a=sp.csr_matrix((np.ones(len(ci), np.int8), ci, rp), shape=(len(rp) - 1,) * 2))


def _one_sample(sample_id):
    """LoadBalanceGraphDataset.__getitem__ (graph_dataset.py:94-130) + _rwr_trace_to_dgl_graph (data_util.py:218-239)."""
    g = _G
    torch = g["torch"]
    seed = int(g["co"].draw_seeds(g["cdf"], 0, sample_id, 1)[0])
    L = int(g["lt"][min(int(g["deg"][seed]), len(g["lt"]) - 1)])
    out = []
    for view in range(2):
        if g["clear"]:
            np.zeros(len(g["rp"]) - 1, dtype=np.int64)          # DGL 0.4: visit_counts cleared per seed
        trace = g["co"].rwr_trace(g["rp"], g["ci"], seed, L, 0, sample_id * 2 + view, g["thr"])
        subv = torch.unique(torch.from_numpy(trace.astype(np.int64))).tolist()          # data_util.py:221
        try:
            subv.remove(seed)                                                              # :222-225
        except ValueError:
            pass
        subv = [seed] + subv                                                               # :226
        idx = np.asarray(subv)
        sub = g["a"][idx][:, idx].tocsr()                                                  # g.subgraph(subv)  :230
        sub.sort_indices()
        pos = g["P"].positional_embedding(sub.indptr.astype(np.int64), sub.indices.astype(np.int64), 32)   # :232
        out.append((idx, sub.indptr, sub.indices, pos))
    return len(out[0][0]) + len(out[1][0])


def run(rp, ci, samples, procs, clear, rw_hops=256, restart=0.8):
    ids = list(range(10_000_000, 10_000_000 + samples))
    if procs == 1:
        _init(rp, ci, rw_hops, restart, clear)
        _one_sample(ids[0])
        t = time.time()
        for i in ids:
            _one_sample(i)
        dt = time.time() - t
    else:
        with mp.get_context("fork").Pool(procs, initializer=_init, initargs=(rp, ci, rw_hops, restart, clear)) as pool:
            pool.map(_one_sample, ids[:procs])                                             # warm the workers
            t = time.time()
            pool.map(_one_sample, ids, chunksize=max(1, samples // (procs * 4)))
            dt = time.time() - t
    return 2 * samples / dt, dt


def run_timed(rp, ci, procs, seconds, clear=True, rw_hops=256, restart=0.8, start="spawn"):
    """The same loop under a time budget (bench.py's cpu_baseline leg): worker processes keep drawing samples until
    ``seconds`` have passed.  ``start``: "spawn" when the parent holds a HIP context (fork + HIP is unsafe).
    -> (subgraphs/s, seconds, samples)"""
    import tempfile

    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=shm) as tmp:
        # the graph goes to the workers as two mapped .npy files, not pickled into every one of them (256 workers x 50 MB)
        np.save(os.path.join(tmp, "rp.npy"), rp)
        np.save(os.path.join(tmp, "ci.npy"), ci)
        with mp.get_context(start).Pool(procs, initializer=_init,
                                        initargs=(os.path.join(tmp, "rp.npy"), os.path.join(tmp, "ci.npy"), rw_hops, restart, clear)) as pool:
            return _timed_loop(pool, procs, seconds)


def _timed_loop(pool, procs, seconds):
    # (every wait is bounded: a worker that dies must not hang the caller)
    pool.map_async(_one_sample, list(range(9_000_000, 9_000_000 + procs))).get(timeout=300)   # warm the workers (imports, ARPACK)
    first, done = 10_000_000, 0
    t = time.time()
    while True:
        ids = list(range(first, first + 8 * procs))
        pool.map_async(_one_sample, ids, chunksize=2).get(timeout=120)
        first += len(ids)
        done += len(ids)
        dt = time.time() - t
        if dt >= seconds:
            break
    return 2 * done / dt, dt, done


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--edges", type=int, default=10_000_000)
    ap.add_argument("--samples", type=int, default=256)
    ap.add_argument("--procs", default="1,%d" % (os.cpu_count() or 1))
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from gcc_amd.graphgen import powerlaw_graph

    rp, ci = powerlaw_graph(args.nodes, args.edges, seed=0)
    res = dict(metric="sampled-subgraphs/sec", unit="subgraphs/s", kind="reference-shaped (SURVEY 8d-i)",
               host_cores=os.cpu_count(), graph_nodes=int(len(rp) - 1), graph_edges=int(len(ci)), samples=args.samples, runs=[])
    for procs in [int(x) for x in args.procs.split(",")]:
        for clear in (False, True):
            v, dt = run(rp, ci, args.samples, procs, clear)
            res["runs"].append(dict(processes=procs, per_seed_visit_count_clear=clear, value=v, seconds=dt))
            print(f"processes {procs:3d}  O(|V|) clear per seed {str(clear):5s}: {v:9.1f} subgraphs/s ({dt:.1f} s)", flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
