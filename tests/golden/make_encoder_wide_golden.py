"""Generates tests/golden/encoder_wide_golden.pt by EXECUTING THE REFERENCE'S OWN CODE at --hidden-size 128 and 256
(train.py:93,601-629: GraphEncoder(output_dim = node_hidden_dim = edge_hidden_dim = hidden_size), MemoryMoCo(hidden_size, ...)),
CPU, DGL replaced by tests/golden/dgl_stub.py -- the same recipe as make_encoder_golden.py, which pins width 64.

    python tests/golden/make_encoder_wide_golden.py

Per width: one MoCo step (train.py:378-431) on two small batched graphs: embeddings, logits, loss, gradients (the fp32 run the
reference performs, and the SAME reference modules run in float64 -- `grads64`, the exact values for this purpose, stored rounded
to f32), grad-norm, post-Adam weights, EMA weights, queue after the enqueue, BatchNorm running statistics.  The initial weights
are NOT stored: tests/golden/wide_init.py derives every tensor from its name (3 MB per copy at width 256); at width 256 the
post-step state is stored for a sample of the tensors only (the update is elementwise in the gradient, pinned in full at 128).
"""
import copy
import os
import sys

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_encoder_golden as G  # noqa: E402  (installs the DGL stub, puts /root/reference on the path)
import wide_init  # noqa: E402

from gcc.contrastive.criterions import NCESoftmaxLoss  # noqa: E402
from gcc.contrastive.memory_moco import MemoryMoCo  # noqa: E402
from gcc.models import GraphEncoder  # noqa: E402
from gcc.utils.misc import warmup_linear  # noqa: E402

import dgl_stub  # noqa: E402

MEMORY_SCALE = 3.0 ** 0.5
SAMPLED_AFTER = ("gnn.ginlayers.0.apply_func.mlp.linears.0.weight", "gnn.ginlayers.3.apply_func.mlp.linears.1.weight",
                 "gnn.linears_prediction.4.weight", "gnn.linears_prediction.0.weight", "degree_embedding.weight")


def build_encoder(hidden):
    # train.py:601-620 with --hidden-size `hidden`
    return GraphEncoder(positional_embedding_size=32, max_node_freq=16, max_edge_freq=16, max_degree=512,
                        freq_embedding_size=16, degree_embedding_size=16, output_dim=hidden, node_hidden_dim=hidden,
                        edge_hidden_dim=hidden, num_layers=5, num_step_set2set=6, num_layer_set2set=3,
                        norm=True, gnn_model="gin", degree_input=True)


class ReplayDropout(nn.Module):
    """nn.Dropout(p) replaying recorded keep-masks (the float64 run sees the fp32 run's masks)."""

    def __init__(self, p, masks):
        super().__init__()
        self.p, self.masks, self.i = p, masks, 0

    def forward(self, x):
        keep = self.masks[self.i].to(x.dtype)
        self.i += 1
        return x * keep / (1.0 - self.p)


def moco_case(views, hidden, K, gen, full_after):
    model, model_ema = wide_init.fill_(build_encoder(hidden), 0), wide_init.fill_(build_encoder(hidden), 1)   # EMA != model
    contrast = MemoryMoCo(hidden, None, K, 0.07, use_softmax=True)          # train.py:627-629
    with torch.no_grad():
        contrast.memory.copy_(wide_init.tensor_for("contrast.memory", contrast.memory) * MEMORY_SCALE)     # = memory_moco.py:21-23's +-1/sqrt(D/3)
    memory0 = contrast.memory.clone()
    criterion = NCESoftmaxLoss()
    optimizer = torch.optim.Adam(model.parameters(), lr=0.005, betas=(0.9, 0.999), weight_decay=1e-5)
    m64, e64 = copy.deepcopy(model).double(), copy.deepcopy(model_ema).double()

    gq = dgl_stub.StubBatchedGraph(**views[0])
    gk = dgl_stub.StubBatchedGraph(**views[1])
    model.train()                                                        # train.py:357-365
    model_ema.eval()
    for m in model_ema.modules():
        if m.__class__.__name__.find("BatchNorm") != -1:
            m.train()
    model.gnn.drop = G.RecordedDropout(0.5, gen)
    feat_q, all_q = model(gq, return_all_outputs=True)                  # train.py:389
    with torch.no_grad():
        feat_k = model_ema(gk)                                           # train.py:390-391
    out = contrast(feat_q, feat_k)                                       # train.py:393
    prob = out[:, 0].mean()
    optimizer.zero_grad()
    loss = criterion(out)                                                # train.py:407
    feat_q.retain_grad()
    loss.backward()
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)  # train.py:409
    lr = 0.005 * warmup_linear(3 / 7500.0, 0.1)                          # train.py:411-416
    for g in optimizer.param_groups:
        g["lr"] = lr
    optimizer.step()
    G.moment_update(model, model_ema, 0.999)                             # train.py:430-431
    masks = torch.stack(model.gnn.drop.masks)

    # the same reference modules in float64 on the same inputs, masks, queue: the exact gradients
    v64 = [dict(v, pos_undirected=v["pos_undirected"].double()) for v in views]
    m64.train()
    e64.eval()
    for m in e64.modules():
        if m.__class__.__name__.find("BatchNorm") != -1:
            m.train()
    m64.gnn.drop = ReplayDropout(0.5, list(masks))
    c64 = MemoryMoCo(hidden, None, K, 0.07, use_softmax=True).double()
    with torch.no_grad():
        c64.memory.copy_(memory0.double())
    fq64 = m64(dgl_stub.StubBatchedGraph(**v64[0]))
    with torch.no_grad():
        fk64 = e64(dgl_stub.StubBatchedGraph(**v64[1]))
    loss64 = criterion(c64(fq64, fk64))
    loss64.backward()
    grads64 = {n: p.grad.float() for n, p in m64.named_parameters() if p.grad is not None}

    keep_after = (lambda k: True) if full_after else (lambda k: k in SAMPLED_AFTER or k.split(".")[-1] in ("running_mean", "running_var", "num_batches_tracked") or k.endswith("bias"))
    after_model = {k: v for k, v in G.sd(model).items() if keep_after(k) and not k.startswith(("set2set", "lin_readout"))}
    after_ema = {k: v for k, v in G.sd(model_ema).items() if keep_after(k) and not k.startswith(("set2set", "lin_readout"))}
    ref_err = max(float((grads[n] - grads64[n]).abs().max()) / max(float(grads64[n].abs().max()), 1e-3) for n in grads)
    if not full_after:
        grads = {n: g for n, g in grads.items() if n in SAMPLED_AFTER or g.dim() == 1}      # (width 256: the fp32 run's matrices are sampled too)
    return dict(hidden=hidden, K=K, T=0.07, lr=lr, memory0_scale=MEMORY_SCALE, masks=masks,
                feat_q=feat_q.detach(), feat_k=feat_k, all_outputs_q=[a.detach() for a in all_q],
                out=out.detach(), prob=prob.detach(), loss=loss.detach(), loss64=loss64.detach().float(), dfeat_q=feat_q.grad.clone(),
                grads=grads, grads64=grads64, ref_fp32_vs_f64=ref_err, grad_norm=torch.as_tensor(grad_norm),
                after=dict(model=after_model, model_ema=after_ema, memory=contrast.memory.clone(), index=contrast.index))


def main():
    gen = torch.Generator().manual_seed(4321)
    views = G.make_inputs(B=8, rw_hops=48, run_seed=9)
    gold = dict(views=views, cases={128: moco_case(views, 128, 96, gen, True), 256: moco_case(views, 256, 160, gen, False)})
    path = os.path.join(HERE, "encoder_wide_golden.pt")
    torch.save(gold, path)
    for h, c in gold["cases"].items():
        worst = c["ref_fp32_vs_f64"]
        print(f"hidden {h}: loss {float(c['loss']):.6f} (float64 {float(c['loss64']):.6f}) gnorm {float(c['grad_norm']):.4f}; "
              f"the reference's own fp32 gradients are within {worst:.2e} of their float64 run (of the tensor's largest entry)")
    print("wrote", path, os.path.getsize(path), "bytes; N =", int(views[0]["node_off"][-1]), int(views[1]["node_off"][-1]))


if __name__ == "__main__":
    main()
