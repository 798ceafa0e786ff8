"""Test-side helpers: run gcc_amd's encoder host code (GinEngine) against the
emulator build with CPU tensors.  TEST INFRASTRUCTURE ONLY."""
import torch

from gcc_amd.encoder import GinEngine, GraphEncoder
from tests.hipemu.emu_driver import emu_lib


class CpuBatch:
    """CPU-tensor stand-in for gcc_amd.sampler.BatchedCSR (emulator runs only)."""

    def __init__(self, view, node_cap=None):
        n = int(view["node_off"][-1])
        self.batch_size = len(view["node_off"]) - 1
        cap = node_cap or n + 37                      # capacity > N: kernels must honour node_off[B]
        self.node_off = view["node_off"].to(torch.int32).contiguous()
        self.row_ptr = torch.zeros(cap + 1, dtype=torch.int32)
        self.row_ptr[: n + 1] = view["row_ptr"].to(torch.int32)
        self.col_idx = view["col_idx"].to(torch.int32).contiguous()
        self.edge_off = view["row_ptr"].to(torch.int32)[view["node_off"].long()].contiguous()   # dgl.batch edge offsets
        self.graph_id = torch.zeros(cap, dtype=torch.int32)
        self.graph_id[:n] = torch.repeat_interleave(torch.arange(self.batch_size, dtype=torch.int32),
                                                    (view["node_off"][1:] - view["node_off"][:-1]))
        self.parent_nid = torch.zeros(cap, dtype=torch.int32)
        self.pos_undirected = torch.zeros(cap, view["pos_undirected"].shape[1])
        self.pos_undirected[:n] = view["pos_undirected"]
        self.n = n


def reference_encoder():
    """GraphEncoder built exactly as train.py:601-618 does with default flags."""
    return GraphEncoder(positional_embedding_size=32, max_node_freq=16, max_edge_freq=16, max_degree=512,
                        freq_embedding_size=16, degree_embedding_size=16, output_dim=64, node_hidden_dim=64,
                        edge_hidden_dim=64, num_layers=5, num_step_set2set=6, num_layer_set2set=3,
                        norm=True, gnn_model="gin", degree_input=True)


def emu_engine():
    return GinEngine(lib=emu_lib(), ptr=lambda t: 0 if t is None else t.data_ptr())
