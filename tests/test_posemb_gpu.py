"""Positional embedding on a real MI355X: batched Jacobi (n <= 128) and Krylov-Schur (n > 128)
kernels vs dense float64 eigendecompositions of every sampled subgraph (invariants, see
tests/test_posemb_emu.py)."""
import numpy as np
import pytest
import torch

from tests.test_posemb_emu import HID, _check, _check_krylov

pytestmark = pytest.mark.gpu


def _device_posemb(q, B):
    from gcc_amd.posemb import DevicePosEmb

    pe = DevicePosEmb(B, q.parent_nid.numel(), HID, device="cuda", seed=7)
    evals = torch.zeros(B, HID, device="cuda")
    raw = torch.zeros(q.parent_nid.numel(), HID, device="cuda")
    pe(q, evals=evals, raw=raw)
    pe.check_status(strict=True)
    c = q.csr_numpy()
    n = c["node_off"][-1]
    view = dict(node_off=torch.from_numpy(c["node_off"].astype(np.int64)),
                row_ptr=torch.from_numpy(c["row_ptr"].astype(np.int64)),
                col_idx=torch.from_numpy(c["col_idx"].astype(np.int64)))
    return view, q.pos_undirected[:n].cpu().numpy(), evals.cpu().numpy(), raw[:n].cpu().numpy()


def test_sampled_batch_on_g1_like_graph():
    from gcc_amd.graph import DeviceGraph
    from gcc_amd.graphgen import powerlaw_graph
    from gcc_amd.sampler import DeviceRWRSampler

    rp, ci = powerlaw_graph(200000, 2000000, 5)
    g = DeviceGraph(rp, ci, rw_hops=256)
    B = 48
    s = DeviceRWRSampler(g, B, run_seed=9)
    q, _ = s.sample(0)
    s.check_status()
    view, x, evals, raw = _device_posemb(q, B)
    sizes = np.diff(view["node_off"].numpy())
    assert (sizes <= 128).any()
    small = dict(view)
    # Jacobi path: strict invariants on the subgraphs with n <= 128 (the helper skips nothing, so mask big ones)
    keep = sizes <= 128
    idx = np.where(keep)[0]
    no = view["node_off"].numpy()
    for b in idx[:24]:
        lo, hi = no[b], no[b + 1]
        sub = dict(node_off=torch.tensor([0, hi - lo]),
                   row_ptr=view["row_ptr"][lo:hi + 1] - view["row_ptr"][lo],
                   col_idx=view["col_idx"][view["row_ptr"][lo]:view["row_ptr"][hi]] - lo)
        _check(sub, x[lo:hi], evals[b:b + 1], raw[lo:hi])
    _check_krylov(view, x, evals, raw)


def test_hub_seeds_take_the_krylov_path():
    from gcc_amd.graph import DeviceGraph
    from gcc_amd.graphgen import powerlaw_graph
    from gcc_amd.sampler import DeviceRWRSampler

    rp, ci = powerlaw_graph(200000, 4000000, 1)
    hubs = np.argsort(np.diff(rp))[-6:].astype(np.int32)
    g = DeviceGraph(rp, ci, rw_hops=256)
    s = DeviceRWRSampler(g, 6, run_seed=2)
    q, _ = s.sample(0, seeds=torch.from_numpy(hubs).cuda())
    s.check_status()
    view, x, evals, raw = _device_posemb(q, 6)
    assert np.diff(view["node_off"].numpy()).min() > 128
    _check_krylov(view, x, evals, raw)
