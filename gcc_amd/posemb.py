"""Positional embedding of the sampled subgraphs (SURVEY.md §8 row a-6,
/root/reference/gcc/datasets/data_util.py:242-281).

``PlaceholderPosEmb`` is NOT the reference computation: it fills
``pos_undirected`` with fixed pseudo-random unit rows so that the rest of the
step can be exercised and timed while the device eigensolver is being built;
bench.py labels its output accordingly.
"""
from __future__ import annotations

import torch


class PlaceholderPosEmb:
    def __init__(self, node_cap, hidden_size=32, device="cuda", seed=0):
        g = torch.Generator(device="cpu").manual_seed(seed)
        x = torch.randn(node_cap, hidden_size, generator=g)
        self.table = torch.nn.functional.normalize(x, dim=1).to(device)
        self.kind = "placeholder-random-unit-rows"

    def __call__(self, graph, prof=None):
        graph.pos_undirected = self.table
        return graph


class HeavyGate:
    """Shared by the producer lanes' positional-embedding calls: the LDS-heavy launches of concurrent calls (sparse block,
    Krylov, 1024-thread dense classes) take turns -- each call waits for the previous call's heavy phase and records the
    end of its own (gcc_posemb_multi_gated) -- while the light launches of all lanes overlap freely."""

    def __init__(self):
        self.last = None            # torch.cuda.Event recorded after the most recent heavy phase

    def next_pair(self, stream):
        """-> (raw hipEvent_t to wait for or None, raw hipEvent_t to record, keep-alive)."""
        wait = self.last
        rec = torch.cuda.Event()
        rec.record(stream)          # materialises the handle; the library records it again after its heavy launches
        self.last = rec
        return (wait.cuda_event if wait is not None else None), rec.cuda_event, (wait, rec)


class DevicePosEmb:
    """``_add_undirected_graph_positional_embedding`` (data_util.py:266-281) for a whole
    batched graph in one launch set (gcc_amd/csrc/posemb.hip).  ``lib``/``ptr`` are
    injectable for the emulator tests only."""

    kind = "device-direct+krylov-schur"

    def __init__(self, batch_size, node_cap, hidden_size=32, device="cuda", seed=0, num_buffers=2, lib=None, ptr=None,
                 max_views=1):
        import ctypes

        from . import _cabi

        self._ct, self._cabi = ctypes, _cabi
        self.lib = lib if lib is not None else _cabi.load()
        self.ptr = ptr if ptr is not None else _cabi.dev_ptr
        self.B, self.hidden, self.seed = int(batch_size), int(hidden_size), int(seed)
        self.node_cap, self.max_views = int(node_cap), int(max_views)
        nbytes = self.lib.gcc_posemb_multi_workspace_bytes(self.max_views, self.B, node_cap, self.hidden)
        if nbytes < 0:
            raise RuntimeError(self.lib.gcc_last_error().decode())
        self.nbytes = nbytes
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.status = torch.zeros(16, dtype=torch.int32, device=device)   # [flags, max cycles, arnoldi steps, handed on, failed, first failed item ...]
        # one output buffer per in-flight batch view (q and k of each ring slot)
        self._ring = [torch.zeros(node_cap, self.hidden, dtype=torch.float32, device=device)
                      for _ in range(2 * num_buffers)]
        self._next = 0

    def __call__(self, graph, evals=None, raw=None, prof=None):
        out = self._ring[self._next]
        self._next = (self._next + 1) % len(self._ring)
        c = self._cabi.GccBatchOut(node_off=self.ptr(graph.node_off), edge_off=0, parent_nid=0, graph_id=0,
                                   row_ptr=self.ptr(graph.row_ptr), col_idx=self.ptr(graph.col_idx),
                                   node_cap=out.shape[0], edge_cap=graph.col_idx.numel())
        st = torch.cuda.current_stream(out.device).cuda_stream if out.is_cuda else None
        rc = self.lib.gcc_posemb(self._ct.byref(c), self.B, self.hidden, self.ptr(out),
                                 self.ptr(evals) if evals is not None else None,
                                 self.ptr(raw) if raw is not None else None, self.seed,
                                 self.ptr(self.workspace), self.nbytes, self.ptr(self.status),
                                 prof.handle if prof is not None else None, st)
        if rc != 0:
            raise RuntimeError(f"gcc_posemb failed ({rc}): {self.lib.gcc_last_error().decode()}")
        graph.pos_undirected = out
        return graph

    def multi(self, graphs, prof=None, evals=None, raws=None, gate=None):
        """Embed several batched graphs (the views of several future steps) in one set of kernel launches.
        ``gate``: a :class:`HeavyGate` shared with the other producer lanes (device streams only)."""
        if len(graphs) > self.max_views:          # more views than the workspace was sized for: several calls
            for i in range(0, len(graphs), self.max_views):
                self.multi(graphs[i:i + self.max_views], prof=prof if i == 0 else None,
                           evals=evals[i:i + self.max_views] if evals is not None else None,
                           raws=raws[i:i + self.max_views] if raws is not None else None, gate=gate)
            return graphs
        views = (self._cabi.GccPosembView * len(graphs))()
        keep = []
        for i, graph in enumerate(graphs):
            out = self._ring[self._next]
            self._next = (self._next + 1) % len(self._ring)
            c = self._cabi.GccBatchOut(node_off=self.ptr(graph.node_off), edge_off=0, parent_nid=0, graph_id=0,
                                       row_ptr=self.ptr(graph.row_ptr), col_idx=self.ptr(graph.col_idx),
                                       node_cap=out.shape[0], edge_cap=graph.col_idx.numel())
            keep.append(c)
            views[i] = self._cabi.GccPosembView(g=self._ct.addressof(c), pos=self.ptr(out),
                                                evals=self.ptr(evals[i]) if evals is not None else None,
                                                raw=self.ptr(raws[i]) if raws is not None else None)
            graph.pos_undirected = out
        dev = self._ring[0].device
        st = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None
        if gate is not None and dev.type == "cuda":
            wait, rec, self._gate_keep = gate.next_pair(torch.cuda.current_stream(dev))
            rc = self.lib.gcc_posemb_multi_gated(views, len(graphs), self.B, self.node_cap, self.hidden, self.seed,
                                                 self.ptr(self.workspace), self.nbytes, self.ptr(self.status),
                                                 prof.handle if prof is not None else None, wait, rec, st)
        else:
            rc = self.lib.gcc_posemb_multi(views, len(graphs), self.B, self.node_cap, self.hidden, self.seed,
                                           self.ptr(self.workspace), self.nbytes, self.ptr(self.status),
                                           prof.handle if prof is not None else None, st)
        if rc != 0:
            raise RuntimeError(f"gcc_posemb_multi failed ({rc}): {self.lib.gcc_last_error().decode()}")
        return graphs

    def check_status(self, strict=False):
        """Bit 8 = some large subgraph hit the Krylov restart cap with a Ritz residual above 1e-3 (its
        embedding is still written).  The reference swallows ARPACK failures too (data_util.py:249-259:
        retry, then zeros), so this only raises with ``strict=True``; returns the flag word."""
        s = int(self.status[0].item())
        if s & 16:                                   # zeros were written for a subgraph: a sizing error, never silent
            raise RuntimeError("gcc_posemb: a large subgraph has more nodes than node_cap / batch_size (size node_cap as "
                               "batch_size * (largest subgraph + 1))")
        if s and strict:
            raise RuntimeError(f"gcc_posemb: status {s} (8 = an eigen-iteration hit its restart cap)")
        return s
