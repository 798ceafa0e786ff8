"""torch.autograd glue for the drop-in ``GraphEncoder.forward`` (API path).

The bench / train.py fast path (gcc_amd/train_step.py) drives the same C-ABI
calls directly and skips autograd.
"""
from __future__ import annotations

import torch

from .encoder import H, grad_params


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else None


class _GinFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, g, keep, nparams, *params):
        eng = enc.engine()
        bn_training = enc.bn_training()
        enc._calls += 1
        slot = (enc._slot, enc._calls % 2)          # two passes may be in flight (E2E: model(q), model(k))
        p, buf = eng.make_pass(enc, g, training=bn_training, keep=keep, slot=slot)
        if not bn_training and getattr(enc, "fused_eval", True):
            eng.eval_fused([p], stream=_stream(g.node_off))     # eval mode: one launch, one workgroup per subgraph
        else:
            eng.forward([p], stream=_stream(g.node_off))
        ctx.enc, ctx.p, ctx.buf = enc, p, buf
        L = len(enc.gnn.ginlayers)
        outs = [buf["feat"].clone()] + [buf["pooled"][i + 1].float() for i in range(L)]
        ctx.mark_non_differentiable(*outs[1:])
        return tuple(outs)

    @staticmethod
    def backward(ctx, dfeat, *_unused):
        enc = ctx.enc
        if not ctx.p.training:
            raise RuntimeError("backward through an eval-mode (running statistics) pass is not supported")
        targets = [enc.padded_zeros_like(param) for _, _, param in grad_params(enc)]
        enc.engine().backward(enc, ctx.p, ctx.buf, dfeat, targets=targets, stream=_stream(dfeat))
        return (None, None, None, None, *targets)


def gin_apply(enc, g, return_all_outputs=False):
    """GraphEncoder.forward (graph_encoder.py:132-200) on a BatchedCSR."""
    keep = None
    if enc.gnn.drop.training and enc.gnn.drop.p > 0:          # gin.py:202,230 nn.Dropout(0.5)
        L = len(enc.gnn.ginlayers)
        keep = (torch.rand(L + 1, g.batch_size, H, device=g.node_off.device) >= enc.gnn.drop.p).float()
    params = [param for _, _, param in grad_params(enc)]
    outs = _GinFn.apply(enc, g, keep, len(params), *params)
    x = outs[0]
    if return_all_outputs:
        return x, list(outs[1:])
    return x
