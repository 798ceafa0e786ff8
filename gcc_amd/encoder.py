"""GraphEncoder -- drop-in for /root/reference/gcc/models/graph_encoder.py:19-200
on the ``gnn_model="gin"`` path (the only one train.py's defaults and the
README commands select), computed by the HIP kernels of gcc_amd/csrc/encoder*.hip.

Same constructor signature, same ``forward(g, return_all_outputs=False)``, same
``state_dict()`` keys and shapes so reference checkpoints load.
``g`` is a :class:`gcc_amd.sampler.BatchedCSR` with ``pos_undirected`` attached.
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn

from . import _cabi

H = _cabi.GIN_HIDDEN
STATS_REPLICAS = 16        # kRep of gcc_amd/csrc/encoder_common.h (GCC_GIN_STAT_REPLICAS)


# ---------------------------------------------------------------------------
# parameter containers with the reference's module tree (names = state_dict keys)
class _MLP(nn.Module):                       # gin.py:61-105 (num_mlp_layers == 2)
    def __init__(self, d_in, d_hid, d_out):
        super().__init__()
        self.linears = nn.ModuleList([nn.Linear(d_in, d_hid), nn.Linear(d_hid, d_out)])
        self.batch_norms = nn.ModuleList([nn.BatchNorm1d(d_hid)])


class _ApplyNodeFunc(nn.Module):             # gin.py:42-52
    def __init__(self, mlp, d):
        super().__init__()
        self.mlp = mlp
        self.bn = nn.BatchNorm1d(d)


class _GINConv(nn.Module):                   # DGL GINConv(apply_func, "sum", 0, learn_eps=False)
    def __init__(self, apply_func):
        super().__init__()
        self.apply_func = apply_func
        self.register_buffer("eps", torch.FloatTensor([0]))


class _UnsupervisedGIN(nn.Module):           # gin.py:119-211
    def __init__(self, num_layers, input_dim, hidden_dim, output_dim, final_dropout):
        super().__init__()
        self.num_layers = num_layers
        self.ginlayers = nn.ModuleList()
        self.batch_norms = nn.ModuleList()
        for layer in range(num_layers - 1):
            mlp = _MLP(input_dim if layer == 0 else hidden_dim, hidden_dim, hidden_dim)
            self.ginlayers.append(_GINConv(_ApplyNodeFunc(mlp, hidden_dim)))
            self.batch_norms.append(nn.BatchNorm1d(hidden_dim))
        self.linears_prediction = nn.ModuleList(
            [nn.Linear(input_dim if layer == 0 else hidden_dim, output_dim) for layer in range(num_layers)])
        self.drop = nn.Dropout(final_dropout)


class _Set2Set(nn.Module):                   # allocated by the reference, never used on the GIN path
    def __init__(self, d, n_layers):
        super().__init__()
        self.lstm = nn.LSTM(2 * d, d, n_layers)


# ---------------------------------------------------------------------------
def fill_weights(enc: "GraphEncoder", ptr) -> _cabi.GccGinWeights:
    """state tensors -> gcc_gin_weights.  ``ptr`` maps a tensor to its address
    (the product passes :func:`_cabi.dev_ptr`, which refuses CPU tensors)."""
    w = _cabi.GccGinWeights()
    g = enc.gnn
    L = len(g.ginlayers)
    w.num_gin_layers = L
    w.pos_dim = enc.positional_embedding_size
    w.deg_emb_dim = enc.degree_embedding_size
    w.max_degree = enc.max_degree
    w.degree_embedding = ptr(enc.degree_embedding.weight)

    def bn(dst, m):
        dst.weight, dst.bias = ptr(m.weight), ptr(m.bias)
        dst.running_mean, dst.running_var = ptr(m.running_mean), ptr(m.running_var)
        dst.num_batches_tracked = ptr(m.num_batches_tracked)

    for i, layer in enumerate(g.ginlayers):
        mlp = layer.apply_func.mlp
        w.lin0_w[i], w.lin0_b[i] = ptr(mlp.linears[0].weight), ptr(mlp.linears[0].bias)
        w.lin1_w[i], w.lin1_b[i] = ptr(mlp.linears[1].weight), ptr(mlp.linears[1].bias)
        bn(w.bn_a[i], mlp.batch_norms[0])
        bn(w.bn_b[i], layer.apply_func.bn)
        bn(w.bn_c[i], g.batch_norms[i])
    for i, lin in enumerate(g.linears_prediction):
        w.pred_w[i], w.pred_b[i] = ptr(lin.weight), ptr(lin.bias)
    bn0 = g.batch_norms[0]
    w.bn_eps, w.bn_momentum = bn0.eps, bn0.momentum
    w.dropout_p = g.drop.p
    w.norm_eps = 1e-5                                   # graph_encoder.py:196
    w.hidden = 0 if enc.hidden == H else enc.hidden     # narrower models run zero-padded (see GraphEncoder.ensure_padded)
    return w


def grad_params(enc: "GraphEncoder"):
    """Parameters that receive gradients, in gcc_gin_grads field order."""
    g = enc.gnn
    out = [("degree_embedding", None, enc.degree_embedding.weight)]
    for i, layer in enumerate(g.ginlayers):
        mlp = layer.apply_func.mlp
        out += [("lin0_w", i, mlp.linears[0].weight), ("lin0_b", i, mlp.linears[0].bias),
                ("lin1_w", i, mlp.linears[1].weight), ("lin1_b", i, mlp.linears[1].bias),
                ("bn_a_w", i, mlp.batch_norms[0].weight), ("bn_a_b", i, mlp.batch_norms[0].bias),
                ("bn_b_w", i, layer.apply_func.bn.weight), ("bn_b_b", i, layer.apply_func.bn.bias),
                ("bn_c_w", i, g.batch_norms[i].weight), ("bn_c_b", i, g.batch_norms[i].bias)]
    for i, lin in enumerate(g.linears_prediction):
        out += [("pred_w", i, lin.weight), ("pred_b", i, lin.bias)]
    return out


class GinEngine:
    """Buffers + C-ABI calls for encoder passes.  ``lib``/``ptr`` are injectable
    only so that tests can run the same host code against the emulator build;
    :class:`GraphEncoder` always uses the HIP library and device pointers."""

    def __init__(self, lib=None, ptr=None):
        self.lib = lib if lib is not None else _cabi.load()
        self.ptr = ptr if ptr is not None else _cabi.dev_ptr
        self._bufs = {}
        self.rows_hint = None       # gcc_gin_pass.rows_hint of every pass made from here on (None: launch for the capacity)

    def hint_rows(self, n, margin=1.10):
        """``n`` live rows were seen in a batch: size the tile kernels' grids for ``margin`` times the largest batch so far
        (a batch beyond it is still correct -- some workgroups walk two tiles --, a grid for the CAPACITY launches about
        twice the workgroups a batch needs, and the idle ones' requests cost 10 % of the step)."""
        want = int(n * margin) + 64
        if self.rows_hint is None or want > self.rows_hint:
            self.rows_hint = want

    def _buffers(self, key, node_cap, B, L, device):
        k = (key, node_cap, B, L, str(device))
        if k not in self._bufs:
            f32 = dict(dtype=torch.float32, device=device)
            self._bufs[k] = dict(
                x0=torch.zeros(node_cap, H, **f32),
                agg=[torch.zeros(node_cap, H, **f32) for _ in range(L)],
                z1=[torch.zeros(node_cap, H, **f32) for _ in range(L)],
                z2=[torch.zeros(node_cap, H, **f32) for _ in range(L)],
                stats=torch.zeros(L, 3, STATS_REPLICAS, 2, H, dtype=torch.float64, device=device),
                # [L][3][2][64] totals of the replicas, written by the forward pass for the backward pass (gcc_gin_pass.bn_totals)
                bn_totals=torch.zeros(L * 3 * 2 * H, dtype=torch.float64, device=device),
                pooled=torch.zeros(L + 1, B, H, dtype=torch.float64, device=device),
                score=torch.zeros(B, H, **f32), feat=torch.zeros(B, H, **f32))
        return self._bufs[k]

    def make_pass(self, enc, g, training, keep=None, slot=0, dropout_seed=None, scalars=None, backward=True):
        """-> (GccGinPass, buffers).  ``g`` needs node_off,row_ptr,col_idx,graph_id,pos_undirected,batch_size.
        ``backward=False``: a pass nobody differentiates (MoCo's key encoder): ``agg`` -- kept only for the weight gradient of
        linears.0 -- is not stored (6.4 MB per layer at bsz 256)."""
        ptr = self.ptr
        L = len(enc.gnn.ginlayers)
        enc.ensure_padded()
        node_cap = g.parent_nid.numel() if hasattr(g, "parent_nid") else g.graph_id.numel()
        buf = self._buffers(slot, node_cap, g.batch_size, L, g.node_off.device)
        p = _cabi.GccGinPass()
        p.node_off, p.row_ptr, p.col_idx, p.graph_id = ptr(g.node_off), ptr(g.row_ptr), ptr(g.col_idx), ptr(g.graph_id)
        if g.pos_undirected is None:
            raise RuntimeError("the batch has no pos_undirected (run the positional embedding first)")
        p.pos = ptr(g.pos_undirected)
        p.batch_size = g.batch_size
        p.training = int(training)
        p.update_running_stats = int(training)
        p.normalize = int(enc.norm)
        p.dropout_keep = ptr(keep) if keep is not None else None
        p.dropout_philox = int(keep is None and dropout_seed is not None)      # in-kernel Philox masks
        p.dropout_seed = int(dropout_seed or 0) & 0xFFFFFFFFFFFFFFFF
        p.w = fill_weights(enc, ptr)
        p.x0 = ptr(buf["x0"])
        for i in range(L):
            p.agg[i], p.z1[i], p.z2[i] = (ptr(buf["agg"][i]) if backward else None), ptr(buf["z1"][i]), ptr(buf["z2"][i])
        p.stats, p.pooled, p.score, p.feat = ptr(buf["stats"]), ptr(buf["pooled"]), ptr(buf["score"]), ptr(buf["feat"])
        p.edge_multiplicity = int(getattr(g, "edge_multiplicity", 1))
        p.bn_totals = ptr(buf["bn_totals"]) if training else None
        seed_local = getattr(g, "seed_local", None)
        p.seed_local = ptr(seed_local) if seed_local is not None else None
        # replayed step (hipGraph): the Philox key of the dropout masks is read from the device struct (gcc_step_scalars)
        p.scalars = ptr(scalars) if scalars is not None else None
        p.node_cap = node_cap
        if self.rows_hint is None:               # API path / tests: the first batch this engine sees sizes the grids (one host read, once;
            self.hint_rows(int(g.node_off[g.batch_size].item()))      # the fused steps set it before their first pass, outside any capture)
        p.rows_hint = int(self.rows_hint or 0)   # grid of the tile kernels: an upper estimate of the live rows (0: the capacity)
        buf = dict(buf)
        buf["_keepalive"] = (g, keep, enc)      # the struct holds raw pointers into these
        return p, buf

    def forward(self, passes, stream=None, prof=None):
        arr = (_cabi.GccGinPass * len(passes))(*passes)
        rc = self.lib.gcc_gin_forward(arr, len(passes), prof.handle if prof is not None else None, stream)
        if rc != 0:
            raise RuntimeError(f"gcc_gin_forward failed ({rc}): {self.lib.gcc_last_error().decode()}")


    def eval_fused(self, passes, mean_out=None, stream=None):
        """gcc_gin_eval_fused: eval-mode passes (running statistics) as one launch, one workgroup per subgraph; with
        ``mean_out`` [B, 64] the mean of the passes' embeddings (generate.py:52)."""
        arr = (_cabi.GccGinPass * len(passes))(*passes)
        rc = self.lib.gcc_gin_eval_fused(arr, len(passes), self.ptr(mean_out) if mean_out is not None else None, stream)
        if rc != 0:
            raise RuntimeError(f"gcc_gin_eval_fused failed ({rc}): {self.lib.gcc_last_error().decode()}")

    def backward(self, enc, p, buf, dfeat, targets=None, accumulate=False, stream=None, prof=None):
        """Backward of a training-mode pass.  Gradients are written (or added, ``accumulate``) into
        ``targets`` (tensors in :func:`grad_params` order); by default into each ``param.grad``."""
        ptr = self.ptr
        L = len(enc.gnn.ginlayers)
        g = buf["_keepalive"][0]
        node_cap = g.parent_nid.numel() if hasattr(g, "parent_nid") else g.graph_id.numel()
        nbytes = self.lib.gcc_gin_backward_workspace_bytes(node_cap, p.batch_size, L)
        key = ("bwd", nbytes, str(dfeat.device))
        if key not in self._bufs:
            self._bufs[key] = torch.empty(nbytes, dtype=torch.uint8, device=dfeat.device)
        ws = self._bufs[key]
        grads = _cabi.GccGinGrads()
        plist = grad_params(enc)
        if targets is None:
            targets = []
            for _, _, param in plist:
                if param.grad is None or (enc.is_padded() and not getattr(param.grad, "_gcc_padded", False)):
                    param.grad = enc.padded_zeros_like(param)
                targets.append(param.grad)
        for (name, idx, _), tgt in zip(plist, targets):
            if idx is None:
                setattr(grads, name, ptr(tgt))
            else:
                getattr(grads, name)[idx] = ptr(tgt)
        dfeat = dfeat.contiguous()
        rc = self.lib.gcc_gin_backward(ctypes.byref(p), ptr(dfeat), ctypes.byref(grads), int(accumulate),
                                       ptr(ws), nbytes, node_cap, prof.handle if prof is not None else None, stream)
        if rc != 0:
            raise RuntimeError(f"gcc_gin_backward failed ({rc}): {self.lib.gcc_last_error().decode()}")
        return targets


class GraphEncoder(nn.Module):
    """graph_encoder.py:44-63 signature; only gnn_model="gin" with degree_input=True
    (train.py:601-618) is implemented -- the other backbones are out of scope."""

    def __init__(self, positional_embedding_size=32, max_node_freq=8, max_edge_freq=8, max_degree=128,
                 freq_embedding_size=32, degree_embedding_size=32, output_dim=32, node_hidden_dim=32,
                 edge_hidden_dim=32, num_layers=6, num_heads=4, num_step_set2set=6, num_layer_set2set=3,
                 norm=False, gnn_model="mpnn", degree_input=False, lstm_as_gate=False):
        super().__init__()
        if gnn_model != "gin":
            raise NotImplementedError("gcc_amd accelerates the GIN path only (train.py:77 default)")
        if not degree_input:
            raise NotImplementedError("train.py:617 always passes degree_input=True")
        if node_hidden_dim < 1 or output_dim < 1:
            raise ValueError(f"hidden / output size must be positive (got {node_hidden_dim} / {output_dim})")
        node_input_dim = positional_embedding_size + degree_embedding_size + 1      # graph_encoder.py:66-67
        # --hidden-size up to 64 runs on the fused 64-channel kernels, zero-padded and exact (ensure_padded); anything wider
        # (or a wider input) on the any-width kernels of csrc/ginx.hip (gcc_amd/encoder_wide.py): same arithmetic, unfused
        self.wide = node_hidden_dim > H or output_dim > H or node_input_dim > H
        if num_layers - 1 > _cabi.GIN_MAX_LAYERS:
            raise NotImplementedError("too many GIN layers")
        self.gnn = _UnsupervisedGIN(num_layers, node_input_dim, node_hidden_dim, output_dim, final_dropout=0.5)
        self.gnn_model = gnn_model
        self.max_node_freq, self.max_edge_freq = max_node_freq, max_edge_freq
        self.max_degree = max_degree
        self.degree_input = degree_input
        self.positional_embedding_size = positional_embedding_size
        self.degree_embedding_size = degree_embedding_size
        self.degree_embedding = nn.Embedding(max_degree + 1, degree_embedding_size)   # :116-118
        self.set2set = _Set2Set(node_hidden_dim, num_layer_set2set)                   # :124 (unused by GIN)
        self.lin_readout = nn.Sequential(nn.Linear(2 * node_hidden_dim, node_hidden_dim), nn.ReLU(),
                                         nn.Linear(node_hidden_dim, output_dim))     # :125-129 (unused by GIN)
        self.norm = norm
        self.hidden, self.output_dim = int(node_hidden_dim), int(output_dim)
        self._pad_ptrs = {}          # name -> data_ptr of the tensor's padded home (ensure_padded)
        self.fused_eval = True       # eval-mode forward as one launch (gcc_gin_eval_fused); False: the 15-launch chain
        self._engine = None
        self._wide_engine = None
        self._slot = id(self)
        self._calls = 0

    # ---- hidden / output sizes below 64: the kernels always compute 64 channels.  Every tensor they index by channel
    # (Linear rows and biases, BatchNorm weight / bias / running statistics, prediction layers) lives as the PREFIX of a
    # zero-padded block: a [h, k] weight is the first h rows of a [64, k] block, a [h] vector the first h entries of a [64]
    # block.  A zero channel stays exactly zero through Linear (zero rows, zero bias), BatchNorm (0 * scale + 0), ReLU and
    # their backward, and Adam never moves a weight whose gradient and value are zero -- so the padded model IS the narrow
    # model, and state_dict() / load_state_dict() see tensors of the reference's shapes (graph_encoder.py:44-63).
    def is_padded(self) -> bool:
        return not self.wide and (self.hidden != H or self.output_dim != H)

    def _channel_tensors(self):
        """(name, tensor holder, attribute, is_parameter) of every tensor whose leading dimension is a channel count."""
        out = []
        g = self.gnn

        def bn(prefix, m):
            for a in ("weight", "bias"):
                out.append((f"{prefix}.{a}", m, a, True))
            for a in ("running_mean", "running_var"):
                out.append((f"{prefix}.{a}", m, a, False))

        for i, layer in enumerate(g.ginlayers):
            mlp = layer.apply_func.mlp
            for j in (0, 1):
                out.append((f"gin{i}.lin{j}.weight", mlp.linears[j], "weight", True))
                out.append((f"gin{i}.lin{j}.bias", mlp.linears[j], "bias", True))
            bn(f"gin{i}.bn_a", mlp.batch_norms[0])
            bn(f"gin{i}.bn_b", layer.apply_func.bn)
            bn(f"gin{i}.bn_c", g.batch_norms[i])
        for i, lin in enumerate(g.linears_prediction):
            out.append((f"pred{i}.weight", lin, "weight", True))
            out.append((f"pred{i}.bias", lin, "bias", True))
        return out

    def padded_numel(self, t) -> int:
        """elements of the zero-padded block ``t`` is the prefix of (``t.numel()`` for tensors that are not padded)."""
        if not self.is_padded() or t.dim() == 0 or t.shape[0] not in (self.hidden, self.output_dim) or t.shape[0] == H:
            return t.numel()
        ids = getattr(self, "_chan_ids", None)
        if ids is None:          # the channel-indexed PARAMETERS (nn.Parameter objects keep their identity; only .data is re-homed)
            self._chan_ids = ids = {id(getattr(m, a)) for _, m, a, is_param in self._channel_tensors() if is_param}
        return (t.numel() // t.shape[0]) * H if id(t) in ids else t.numel()

    def padded_zeros_like(self, t):
        """a tensor shaped like ``t`` that is the prefix of a zeroed padded block (gradient targets of the kernels)."""
        n = self.padded_numel(t)
        block = torch.zeros(n, dtype=t.dtype, device=t.device)
        v = block[: t.numel()].view(t.shape)
        v._gcc_padded = True
        return v

    def mark_padded(self):
        """The channel tensors' CURRENT storage is padded (the trainer's flat buffers lay them out that way)."""
        self._pad_ptrs = {name: getattr(m, a).data_ptr() for name, m, a, _ in self._channel_tensors()}

    def ensure_padded(self):
        """(Re-)home every channel tensor as the prefix of a zero-padded block.  nn.Module.to() / .cuda() and
        load-time re-materialisation replace ``.data`` by exactly sized tensors, so this runs before every pass and
        compares data pointers (a few microseconds; nothing at all for hidden = output = 64)."""
        if not self.is_padded():
            return
        for name, m, a, is_param in self._channel_tensors():
            t = getattr(m, a)
            if self._pad_ptrs.get(name) == t.data_ptr():
                continue
            n = (t.numel() // t.shape[0]) * H
            block = torch.zeros(n, dtype=t.dtype, device=t.device)
            block[: t.numel()].copy_(t.detach().reshape(-1))
            home = block[: t.numel()].view(t.shape)
            if is_param:
                t.data = home
            else:
                m._buffers[a] = home
            self._pad_ptrs[name] = home.data_ptr()

    def engine(self) -> GinEngine:
        if self.wide:
            raise NotImplementedError(f"the fused 64-channel kernels serve hidden / output sizes up to {H}; this model "
                                      f"({self.hidden} / {self.output_dim}) runs through GraphEncoder.forward (csrc/ginx.hip)")
        if self._engine is None:
            self._engine = GinEngine()
        return self._engine

    def wide_engine(self):
        if self._wide_engine is None:
            from .encoder_wide import WideGinEngine

            self._wide_engine = WideGinEngine()
        return self._wide_engine

    def bn_training(self) -> bool:
        """train.py:357-365: model_ema is in eval() but its BatchNorm layers are switched back to train()."""
        return self.gnn.batch_norms[0].training

    def embed_views(self, graph_q, graph_k):
        """generate.py:45-52 in one launch: ``(model(graph_q) + model(graph_k)) / 2`` in eval mode (both views' subgraphs
        as workgroups of the same gcc_gin_eval_fused call, the mean taken on the device).  -> [B, output_dim]"""
        if self.bn_training():
            raise RuntimeError("embed_views is the eval-mode path (generate.py:38 calls model.eval())")
        if self.wide:                                   # two eval passes and their mean (generate.py:48-52 as written)
            with torch.no_grad():
                fq = self(graph_q)
                return fq.clone() if graph_k is graph_q else (fq + self(graph_k)) / 2
        eng = self.engine()
        st = torch.cuda.current_stream(graph_q.node_off.device).cuda_stream if graph_q.node_off.is_cuda else None
        views = [graph_q] if graph_k is graph_q else [graph_q, graph_k]        # entire_graph: both views are one graph
        passes, keep = [], []
        for i, g in enumerate(views):
            p, buf = eng.make_pass(self, g, training=False, keep=None, slot=(self._slot, "embed", i))
            passes.append(p)
            keep.append(buf)
        mean = eng._buffers((self._slot, "embed", "mean"), 1, graph_q.batch_size, 1, graph_q.node_off.device)["feat"]
        eng.eval_fused(passes, mean_out=mean, stream=st)
        return mean[:, : self.output_dim].clone()

    def forward(self, g, return_all_outputs=False):
        if self.wide:
            from .encoder_wide import ginx_apply

            return ginx_apply(self, g, return_all_outputs)
        from .autograd import gin_apply

        out = gin_apply(self, g, return_all_outputs)
        if not self.is_padded():
            return out
        if return_all_outputs:                       # the kernels' 64 channels -> the model's own widths
            return out[0][:, : self.output_dim], [t[:, : self.hidden] for t in out[1]]
        return out[:, : self.output_dim]
