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

    def __call__(self, graph):
        graph.pos_undirected = self.table
        return graph
