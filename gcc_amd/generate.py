"""``test_moco`` of the reference's generate.py:33-53 -- eval-mode encoder over every node's two views, embedding =
(f(q) + f(k)) / 2 -- on the device pipeline."""
from __future__ import annotations

import torch


def test_moco(dataset, model, posemb, opt=None):
    """dataset: gcc_amd.datasets.NodeClassificationDataset; model: gcc_amd.encoder.GraphEncoder;
    posemb: gcc_amd.posemb.DevicePosEmb (max_views >= 2).  Returns a CPU tensor [len(dataset), hidden]."""
    model.eval()                                                   # generate.py:38
    emb_list = []
    for graph_q, graph_k in dataset:
        bsz = graph_q.batch_size
        views = [graph_q] if graph_k is graph_q else [graph_q, graph_k]       # entire_graph: both views are one graph
        posemb.multi(views) if hasattr(posemb, "multi") else [posemb(v) for v in views]
        with torch.no_grad():
            if getattr(model, "fused_eval", False):
                # both views through the encoder AND (feat_q + feat_k) / 2 in one launch (gcc_gin_eval_fused)
                emb = model.embed_views(graph_q, graph_k)
            else:
                feat_q = model(graph_q)
                feat_k = feat_q if graph_k is graph_q else model(graph_k)
                emb = (feat_q + feat_k) / 2
        if opt is not None:
            assert emb.shape == (bsz, opt.hidden_size)             # generate.py:51
        emb_list.append(emb[: graph_q.valid].detach().cpu())
    return torch.cat(emb_list)
