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
        posemb.multi([graph_q, graph_k]) if hasattr(posemb, "multi") else (posemb(graph_q), posemb(graph_k))
        with torch.no_grad():
            feat_q = model(graph_q)
            feat_k = model(graph_k)
        if opt is not None:
            assert feat_q.shape == (bsz, opt.hidden_size)          # generate.py:51
        emb_list.append(((feat_q + feat_k) / 2)[: graph_q.valid].detach().cpu())
    return torch.cat(emb_list)
