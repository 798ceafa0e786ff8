"""Generates tests/golden/encoder_golden.pt by EXECUTING THE REFERENCE'S OWN CODE
(/root/reference/gcc/models/{gin,graph_encoder}.py, gcc/contrastive/*.py,
gcc/utils/misc.py) on CPU in this container, with DGL replaced by
tests/golden/dgl_stub.py.  /root/reference does not exist on the GPU box, so
the vectors are committed.  Run from the repo root:

    python tests/golden/make_encoder_golden.py

Contents: one MoCo step (train.py:378-431) and one E2E step (train.py:397-401)
on two small batched graphs: inputs, initial weights, dropout keep-masks,
embeddings, logits, loss, gradients, grad-norm, post-Adam weights, EMA weights,
queue after enqueue, BatchNorm running statistics, eval-mode embeddings.
"""
import copy
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import dgl_stub  # noqa: E402

dgl_stub.install()
sys.path.insert(0, "/root/reference")
torch.Tensor.cuda = lambda self, *a, **k: self          # memory_moco.py:56, criterions.py:15 call .cuda()

from gcc.contrastive.criterions import NCESoftmaxLoss, NCESoftmaxLossNS  # noqa: E402
from gcc.contrastive.memory_moco import MemoryMoCo  # noqa: E402
from gcc.models import GraphEncoder  # noqa: E402
from gcc.utils.misc import warmup_linear  # noqa: E402

from gcc_amd.graphgen import powerlaw_graph  # noqa: E402
from oracle import posemb as P  # noqa: E402
from oracle import sampler as O  # noqa: E402


class RecordedDropout(nn.Module):
    """nn.Dropout(p) with the Bernoulli keep-mask drawn here and recorded, so that
    other implementations can replay it (gin.py:202,230)."""

    def __init__(self, p, gen):
        super().__init__()
        self.p, self.gen, self.masks = p, gen, []

    def forward(self, x):
        if not self.training:
            return x
        keep = (torch.rand(x.shape, generator=self.gen) >= self.p).float()
        self.masks.append(keep)
        return x * keep / (1.0 - self.p)


def make_inputs(B, rw_hops, run_seed):
    rp, ci = powerlaw_graph(3000, 30000, 3)
    c = O.COracle()
    seeds = c.draw_seeds(O.seed_cdf(rp), run_seed, 0, B)
    L = O.max_nodes_table(int(np.diff(rp).max()), rw_hops, 0.8)[np.diff(rp)[seeds]]
    views = []
    for v in range(2):
        r = c.sample_batch(rp, ci, seeds, L, v, run_seed, 0, O.restart_threshold(0.8))
        pos = P.batched_positional_embedding(r["node_off"], r["row_ptr"], r["col_idx"], 32, seed=run_seed + v)
        views.append(dict(node_off=torch.from_numpy(r["node_off"].astype(np.int64)),
                          row_ptr=torch.from_numpy(r["row_ptr"].astype(np.int64)),
                          col_idx=torch.from_numpy(r["col_idx"].astype(np.int64)),
                          pos_undirected=torch.from_numpy(pos)))
    return views


def build_encoder():
    # exactly train.py:601-620 with the default flags (train.py:79,93-99,83)
    return GraphEncoder(positional_embedding_size=32, max_node_freq=16, max_edge_freq=16, max_degree=512,
                        freq_embedding_size=16, degree_embedding_size=16, output_dim=64, node_hidden_dim=64,
                        edge_hidden_dim=64, num_layers=5, num_step_set2set=6, num_layer_set2set=3,
                        norm=True, gnn_model="gin", degree_input=True)


def moment_update(model, model_ema, m):          # train.py:169-172 (train.py itself needs tensorboard to import)
    for p1, p2 in zip(model.parameters(), model_ema.parameters()):
        p2.data.mul_(m).add_(p1.detach().data, alpha=1 - m)


def sd(module):
    return {k: v.clone() for k, v in module.state_dict().items()}


def moco_case(views, K, gen):
    torch.manual_seed(0)
    model, model_ema = build_encoder(), build_encoder()
    moment_update(model, model_ema, 0)                                  # train.py:624
    # non-trivial BN affine parameters so that their gradients are exercised
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5, generator=gen)
                m.bias.uniform_(-0.3, 0.3, generator=gen)
        moment_update(model, model_ema, 0)
        for p in model_ema.parameters():                                # make EMA != model
            p.add_(0.01 * torch.randn(p.shape, generator=gen))
    contrast = MemoryMoCo(64, None, K, 0.07, use_softmax=True)          # train.py:627-629
    criterion = NCESoftmaxLoss()
    optimizer = torch.optim.Adam(model.parameters(), lr=0.005, betas=(0.9, 0.999), weight_decay=1e-5)
    init = dict(model=sd(model), model_ema=sd(model_ema), memory=contrast.memory.clone())

    gq = dgl_stub.StubBatchedGraph(**views[0])
    gk = dgl_stub.StubBatchedGraph(**views[1])
    model.train()                                                        # train.py:357-365
    model_ema.eval()
    for m in model_ema.modules():
        if m.__class__.__name__.find("BatchNorm") != -1:
            m.train()
    model.gnn.drop = RecordedDropout(0.5, gen)
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
    lr = 0.005 * warmup_linear(3 / 7500.0, 0.1)                          # train.py:411-416 (some early step)
    for g in optimizer.param_groups:
        g["lr"] = lr
    optimizer.step()
    moment_update(model, model_ema, 0.999)                               # train.py:430-431
    model.eval()
    with torch.no_grad():
        feat_eval = model(gq)                                            # generate.py:38-49 style
    return dict(K=K, T=0.07, lr=lr, init=init, masks=torch.stack(model.gnn.drop.masks),
                feat_q=feat_q.detach(), feat_k=feat_k, all_outputs_q=[a.detach() for a in all_q],
                out=out.detach(), prob=prob.detach(), loss=loss.detach(), dfeat_q=feat_q.grad.clone(),
                grads=grads, grad_norm=torch.as_tensor(grad_norm),
                after=dict(model=sd(model), model_ema=sd(model_ema), memory=contrast.memory.clone(),
                           index=contrast.index),
                feat_eval=feat_eval)


def e2e_case(views, gen):
    torch.manual_seed(1)
    model = build_encoder()
    init = dict(model=sd(model))
    gq = dgl_stub.StubBatchedGraph(**views[0])
    gk = dgl_stub.StubBatchedGraph(**views[1])
    model.train()
    model.gnn.drop = RecordedDropout(0.5, gen)
    feat_q = model(gq)                                                   # train.py:397-398
    feat_k = model(gk)
    out = torch.matmul(feat_k, feat_q.t()) / 0.07                        # train.py:400
    prob = out[range(gq.batch_size), range(gq.batch_size)].mean()
    loss = NCESoftmaxLossNS()(out)
    loss.backward()
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    return dict(init=init, masks=torch.stack(model.gnn.drop.masks), feat_q=feat_q.detach(),
                feat_k=feat_k.detach(), out=out.detach(), prob=prob.detach(), loss=loss.detach(), grads=grads)


def main():
    gen = torch.Generator().manual_seed(1234)
    views = make_inputs(B=6, rw_hops=32, run_seed=5)
    gold = dict(views=views, moco=moco_case(views, K=48, gen=gen), e2e=e2e_case(views, gen=gen),
                param_names=[n for n, _ in build_encoder().named_parameters()],
                state_dict_shapes={k: tuple(v.shape) for k, v in build_encoder().state_dict().items()})
    path = os.path.join(HERE, "encoder_golden.pt")
    torch.save(gold, path)
    print("wrote", path, os.path.getsize(path), "bytes; N =", int(views[0]["node_off"][-1]),
          int(views[1]["node_off"][-1]), "loss", float(gold["moco"]["loss"]), "gnorm", float(gold["moco"]["grad_norm"]))


if __name__ == "__main__":
    main()
