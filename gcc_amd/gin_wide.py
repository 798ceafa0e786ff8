"""Wide (hidden 256) GIN layers in bf16 on the matrix cores: the host side of ``gcc_ginw_forward``
(include/gcc_amd.h), BASELINE.json configs[4].

``FoldedWideGIN`` holds the layer stack of UnsupervisedGIN (gcc/models/gin.py:160-221) for inference
(generate.py:71 ``model.eval()``): Linear weights as bf16, Linear biases and BatchNorm running statistics folded
into per-channel scale/shift pairs.  Device only: there is no CPU path.
"""
from __future__ import annotations

import ctypes

import torch

from . import _cabi

HIDDEN = 256
MAX_NODES = 128


def fold_bn(bn, bias=None):
    """(scale, shift) of eval-mode BatchNorm1d applied to ``x + bias``."""
    s = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    b = torch.zeros_like(s) if bias is None else bias.detach().double()
    return s.float(), ((b - bn.running_mean.detach().double()) * s + bn.bias.detach().double()).float()


class FoldedWideGIN:
    def __init__(self, layers, device):
        """layers: list of dicts with float32 tensors w0, w1 [256, 256] (torch Linear layout) and s0..t2 [256]."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("gcc_amd kernels run on the GPU only; there is no CPU path")
        if not 1 <= len(layers) <= 8:
            raise ValueError("1 to 8 layers")
        self.lib = _cabi.load()
        self.layers = []
        for ly in layers:
            d = {}
            for k in ("w0", "w1"):
                w = torch.as_tensor(ly[k], dtype=torch.float32)
                if tuple(w.shape) != (HIDDEN, HIDDEN):
                    raise ValueError(f"{k} must be [{HIDDEN}, {HIDDEN}]")
                d[k] = w.to(self.device).to(torch.bfloat16).contiguous()          # round to nearest even
            for k in ("s0", "t0", "s1", "t1", "s2", "t2"):
                v = torch.as_tensor(ly[k], dtype=torch.float32)
                if tuple(v.shape) != (HIDDEN,):
                    raise ValueError(f"{k} must be [{HIDDEN}]")
                d[k] = v.to(self.device).contiguous()
            # the same weights in the order the kernel's waves request them (include/gcc_amd.h: gcc_ginw_pack_weights)
            st = torch.cuda.current_stream(self.device).cuda_stream
            for which, k in enumerate(("w0", "w1")):
                d[k + "_frag"] = torch.empty_like(d[k])
                _cabi.check(self.lib.gcc_ginw_pack_weights(_cabi.dev_ptr(d[k]), _cabi.dev_ptr(d[k + "_frag"]), which, st),
                            "gcc_ginw_pack_weights")
            self.layers.append(d)
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._scratch = None             # rows in transit + work list of subgraphs over 128 nodes (gcc_ginw_scratch_bytes)

    @classmethod
    def from_gin(cls, gin, device):
        """gin: a module with the reference's attribute names (``ginlayers[i].apply_func.{mlp,bn}``,
        ``batch_norms[i]``; gin.py:160-199) whose hidden size is 256."""
        layers = []
        for i, layer in enumerate(gin.ginlayers):
            mlp = layer.apply_func.mlp
            s0, t0 = fold_bn(mlp.batch_norms[0], mlp.linears[0].bias)
            s1, t1 = fold_bn(layer.apply_func.bn, mlp.linears[1].bias)
            s2, t2 = fold_bn(gin.batch_norms[i])
            layers.append(dict(w0=mlp.linears[0].weight.detach(), w1=mlp.linears[1].weight.detach(),
                               s0=s0, t0=t0, s1=s1, t1=t1, s2=s2, t2=t2))
        return cls(layers, device)

    def forward(self, node_off, row_ptr, col_idx, x, num_layers=None, first_layer=0, want_rows=True, want_pooled=True,
                prof=None, big=True):
        """x: bf16 [N, 256] on the device; the CSR is the batched graph of the sampler (int32, row v = in-neighbours
        of v, global ids).  Runs layers first_layer .. first_layer + num_layers - 1 in one launch.  Returns (rows bf16 [N, 256] or None, pooled f32 [B, L + 1, 256] or None); call
        ``check_status()`` after synchronising."""
        L = len(self.layers) - first_layer if num_layers is None else int(num_layers)
        if not (0 <= first_layer and 1 <= L and first_layer + L <= len(self.layers)):
            raise ValueError("layer range out of bounds")
        if x.dtype != torch.bfloat16 or x.dim() != 2 or x.shape[1] != HIDDEN:
            raise TypeError(f"x must be bfloat16 [N, {HIDDEN}]")
        B = node_off.numel() - 1
        rows = torch.empty_like(x) if want_rows else None
        pooled = torch.empty(B, L + 1, HIDDEN, dtype=torch.float32, device=self.device) if want_pooled else None
        a = _cabi.GccGinwArgs(node_off=_cabi.dev_ptr(node_off, torch.int32), row_ptr=_cabi.dev_ptr(row_ptr, torch.int32),
                              col_idx=_cabi.dev_ptr(col_idx, torch.int32), x_in=_cabi.dev_ptr(x),
                              x_out=_cabi.dev_ptr(rows), pooled=_cabi.dev_ptr(pooled), batch_size=B, num_layers=L)
        for i in range(L):
            for k, v in self.layers[first_layer + i].items():
                setattr(a.layers[i], k, _cabi.dev_ptr(v))
        if big:
            # subgraphs over 128 nodes run block by block (one launch per layer for them); without the scratch they are
            # refused through the status word
            need = int(self.lib.gcc_ginw_scratch_bytes(x.shape[0], B))
            if self._scratch is None or self._scratch.numel() < need:
                self._scratch = torch.empty(need, dtype=torch.uint8, device=self.device)
            a.scratch, a.scratch_bytes, a.num_nodes = _cabi.dev_ptr(self._scratch), need, x.shape[0]
        st = torch.cuda.current_stream(self.device).cuda_stream
        _cabi.check(self.lib.gcc_ginw_forward(ctypes.byref(a), _cabi.dev_ptr(self.status),
                                              prof.handle if prof is not None else None, st), "gcc_ginw_forward")
        return rows, pooled

    def check_status(self):
        s = int(self.status[0].item())
        if s & 32:
            raise RuntimeError(f"gcc_ginw_forward: a subgraph has more than {MAX_NODES} nodes and the call had no scratch "
                               "(big=False), or more row blocks than the work list holds: its outputs are zero")
        if s & 64:
            raise RuntimeError("gcc_ginw_forward: a neighbour id lies outside its subgraph")
        return s
