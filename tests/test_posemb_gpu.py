"""Positional embedding on a real MI355X: the direct eigensolver (deflated size <= 384: LDS and workspace
classes) and the Krylov-Schur fallback vs dense float64 eigendecompositions of every sampled subgraph
(invariants, see tests/test_posemb_emu.py)."""
import numpy as np
import pytest
import torch

from tests.test_posemb_emu import DIRECT_MAX, HID, _check_krylov, check_by_path, reduced_sizes

pytestmark = pytest.mark.gpu


def _device_posemb(q, B):
    from gcc_amd.posemb import DevicePosEmb

    pe = DevicePosEmb(B, q.parent_nid.numel(), HID, device="cuda", seed=7)
    evals = torch.zeros(B, HID, device="cuda")
    raw = torch.zeros(q.parent_nid.numel(), HID, device="cuda")
    pe(q, evals=evals, raw=raw)
    pe.check_status(strict=True)
    _device_posemb.arnoldi_steps = int(pe.status[2].item())
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
    red = reduced_sizes(view)
    assert (red <= 64).any() and ((red > 64) & (red <= 128)).any() and ((red > 128) & (red <= DIRECT_MAX)).any(), red
    # strict invariants (all multiplicities) wherever the direct solver ran: a mix of small and all large subgraphs
    idx = np.r_[np.where(sizes <= 128)[0][:20], np.where(sizes > 128)[0]]
    check_by_path(view, x, evals, raw, only=idx)


def test_hub_seeds():
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
    check_by_path(view, x, evals, raw)


def test_krylov_fallback_on_device():
    """No twin leaves, n = 700 > GCC_POSEMB_DIRECT_MAX: the Krylov-Schur kernel runs (same case as the emulator test)."""
    import scipy.sparse as sp

    from gcc_amd.posemb import DevicePosEmb
    from gcc_amd.sampler import BatchedCSR

    rng = np.random.RandomState(1)
    n = 700
    w = 1.0 / np.arange(1, n + 1) ** 0.5
    pr = np.minimum(1.0, 6.0 * np.outer(w, w) / w.mean())
    up = np.triu(rng.rand(n, n) < pr, 1)
    up[np.arange(n - 1), np.arange(1, n)] = True
    a = sp.csr_matrix((up | up.T).astype(np.float64))
    a.sort_indices()
    i32 = dict(dtype=torch.int32, device="cuda")
    q = BatchedCSR(1, torch.tensor([0, n], **i32), torch.tensor([0, a.nnz], **i32),
                   torch.zeros(n, **i32), torch.zeros(n, **i32), torch.from_numpy(a.indptr.astype(np.int32)).cuda(),
                   torch.from_numpy(a.indices.astype(np.int32)).cuda())
    q.pos_undirected = torch.zeros(n, HID, device="cuda")
    view, x, evals, raw = _device_posemb(q, 1)
    assert _device_posemb.arnoldi_steps > 0
    _check_krylov(view, x, evals, raw)
