"""GraphEncoder at hidden / output sizes above 64 (``--hidden-size``, train.py:93; graph_encoder.py:44-63): host side of
csrc/ginx.hip -- one C-ABI call for the forward pass (training or eval mode), one for the backward pass, plus the
autograd glue of the API path.  Same module tree, state_dict keys and arithmetic as the 64-channel path; what differs
is the kernels underneath (unfused, one launch per operator) and that dropout masks are drawn on the host side of the
call (torch.rand on the device, as torch.nn.Dropout draws them: gin.py:202,230).
"""
from __future__ import annotations

import ctypes

import torch

from . import _cabi


class WideGinEngine:
    """Workspaces + C-ABI calls of gcc_ginx_forward / gcc_ginx_backward.  ``lib`` / ``ptr`` are injectable only so that the
    tests can run the same host code against the emulator build."""

    def __init__(self, lib=None, ptr=None):
        self.lib = lib if lib is not None else _cabi.load()
        self.ptr = ptr if ptr is not None else _cabi.dev_ptr
        self._ws = {}
        self._gen = {}                      # workspace key -> generation: a forward stamps its slot, a backward checks the stamp

    def make_pass(self, enc, g, training, keep=None, slot=0, want_pooled=False):
        from .encoder import fill_weights

        ptr = self.ptr
        L = len(enc.gnn.ginlayers)
        node_cap = g.parent_nid.numel() if hasattr(g, "parent_nid") else g.graph_id.numel()
        B, dev = g.batch_size, g.node_off.device
        d_in = enc.positional_embedding_size + enc.degree_embedding_size + 1
        nbytes = self.lib.gcc_ginx_workspace_bytes(node_cap, B, L, d_in, enc.hidden, enc.output_dim)
        if nbytes < 0:
            raise RuntimeError(self.lib.gcc_last_error().decode())
        key = (slot, nbytes, str(dev))
        if key not in self._ws:
            self._ws[key] = dict(ws=torch.zeros(nbytes, dtype=torch.uint8, device=dev),
                                 feat=torch.zeros(B, enc.output_dim, dtype=torch.float32, device=dev),
                                 pooled=torch.zeros(L, B, enc.hidden, dtype=torch.float32, device=dev))
        buf = self._ws[key]
        self._gen[key] = self._gen.get(key, 0) + 1
        if g.pos_undirected is None:
            raise RuntimeError("the batch has no pos_undirected (run the positional embedding first)")
        p = _cabi.GccGinxPass()
        p.node_off, p.row_ptr, p.col_idx, p.graph_id = ptr(g.node_off), ptr(g.row_ptr), ptr(g.col_idx), ptr(g.graph_id)
        p.pos = ptr(g.pos_undirected)
        seed_local = getattr(g, "seed_local", None)
        p.seed_local = ptr(seed_local) if seed_local is not None else None
        p.batch_size, p.training, p.update_running_stats, p.normalize = B, int(training), int(training), int(enc.norm)
        p.dropout_keep = ptr(keep) if keep is not None else None
        p.hidden, p.out_dim = enc.hidden, enc.output_dim
        p.edge_multiplicity = int(getattr(g, "edge_multiplicity", 1))
        p.node_cap = node_cap
        p.w = fill_weights(enc, ptr)
        p.workspace, p.workspace_bytes = ptr(buf["ws"]), nbytes
        p.feat = ptr(buf["feat"])
        p.pooled_out = ptr(buf["pooled"]) if want_pooled else None
        out = dict(buf)
        out["_keepalive"] = (g, keep, enc)          # the struct holds raw pointers into these
        out["_slot"] = (key, self._gen[key])        # which workspace this pass's activations live in, and its generation
        return p, out

    def forward(self, p, stream=None):
        rc = self.lib.gcc_ginx_forward(ctypes.byref(p), stream)
        if rc != 0:
            raise RuntimeError(f"gcc_ginx_forward failed ({rc}): {self.lib.gcc_last_error().decode()}")

    def slot_is_current(self, buf):
        """False once a later forward has reused the workspace slot this pass's activations were stored in."""
        key, gen = buf["_slot"]
        return self._gen.get(key) == gen

    def backward(self, enc, p, dfeat, targets, stream=None):
        from .encoder import grad_params

        grads = _cabi.GccGinGrads()
        for (name, idx, _), tgt in zip(grad_params(enc), targets):
            if idx is None:
                setattr(grads, name, self.ptr(tgt))
            else:
                getattr(grads, name)[idx] = self.ptr(tgt)
        dfeat = dfeat.contiguous()
        rc = self.lib.gcc_ginx_backward(ctypes.byref(p), self.ptr(dfeat), ctypes.byref(grads), stream)
        if rc != 0:
            raise RuntimeError(f"gcc_ginx_backward failed ({rc}): {self.lib.gcc_last_error().decode()}")
        return targets


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else None


class _GinxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, g, keep, want_pooled, *params):
        eng = enc.wide_engine()
        training = enc.bn_training()
        enc._calls += 1
        slot = (enc._slot, enc._calls % 2)          # two passes may be in flight (E2E: model(q), model(k))
        p, buf = eng.make_pass(enc, g, training=training, keep=keep, slot=slot, want_pooled=want_pooled)
        eng.forward(p, stream=_stream(g.node_off))
        ctx.enc, ctx.p, ctx.buf = enc, p, buf
        L = len(enc.gnn.ginlayers)
        outs = [buf["feat"].clone()] + ([buf["pooled"][i].clone() for i in range(L)] if want_pooled else [])
        ctx.mark_non_differentiable(*outs[1:])
        return tuple(outs)

    @staticmethod
    def backward(ctx, dfeat, *_unused):
        from .encoder import grad_params

        enc = ctx.enc
        if not ctx.p.training:
            raise RuntimeError("backward through an eval-mode (running statistics) pass is not supported")
        if not enc.wide_engine().slot_is_current(ctx.buf):
            # the activations of a pending backward live in one of two workspace slots per encoder (model(q), model(k) of an
            # E2E step); a third forward before this backward has overwritten them -- gradients from the wrong activations
            # would be silent
            raise RuntimeError("backward of a GraphEncoder forward whose activations were overwritten: at most two forward passes "
                               "of one wide encoder may be pending a backward (run backward, or wrap the extra passes in torch.no_grad())")
        targets = [torch.zeros_like(param) for _, _, param in grad_params(enc)]
        enc.wide_engine().backward(enc, ctx.p, dfeat, targets, stream=_stream(dfeat))
        return (None, None, None, None, *targets)


def ginx_apply(enc, g, return_all_outputs=False):
    """GraphEncoder.forward (graph_encoder.py:132-200) on a BatchedCSR, any width."""
    from .encoder import grad_params

    keep = None
    if enc.gnn.drop.training and enc.gnn.drop.p > 0:          # gin.py:202,230 nn.Dropout(0.5)
        L = len(enc.gnn.ginlayers)
        keep = (torch.rand(L + 1, g.batch_size, enc.output_dim, device=g.node_off.device) >= enc.gnn.drop.p).float()
    params = [param for _, _, param in grad_params(enc)]
    outs = _GinxFn.apply(enc, g, keep, bool(return_all_outputs), *params)
    if return_all_outputs:
        return outs[0], list(outs[1:])
    return outs[0]
