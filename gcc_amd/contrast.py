"""MoCo / InfoNCE head behind the reference's API.

Mirrors /root/reference/gcc/contrastive/memory_moco.py:7-63 (``MemoryMoCo``) and
gcc/contrastive/criterions.py:5-33 (``NCESoftmaxLoss``, ``NCESoftmaxLossNS``).
The logits [B, K+1] are not materialised: ``MemoryMoCo.forward`` returns a
:class:`NCELogits` that carries the fused loss (HIP online-softmax kernels of
gcc_amd/csrc/nce.hip) and supports exactly what train.py does with ``out``:
``out[:, 0]`` (train.py:394), ``criterion(out)`` (train.py:407), ``out.shape``;
``out.dense()`` builds the full tensor on demand.
"""
from __future__ import annotations

import ctypes
import math

import torch
from torch import nn

from . import _cabi

D = 64


class NceEngine:
    """C-ABI calls of the head.  ``lib``/``ptr`` are injectable for the emulator tests only."""

    def __init__(self, lib=None, ptr=None, dtype="f32"):
        self.lib = lib if lib is not None else _cabi.load()
        self.ptr = ptr if ptr is not None else _cabi.dev_ptr
        self._ws = {}
        self.dtype = dtype            # "f32": exact-fp32 MFMA (parity mode); "bf16": operands rounded to bf16 (GCC_NCE_BF16)

    def _workspace(self, B, K, device):
        nbytes = self.lib.gcc_nce_workspace_bytes(B, K)
        key = (nbytes, str(device))
        if key not in self._ws:
            self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self._ws[key], nbytes

    def _args(self, q, k, mem, inv_T, pos_mode, patch, patch_index, outs, dense):
        ptr = self.ptr
        a = _cabi.GccNceArgs()
        a.q, a.k, a.mem = ptr(q), ptr(k) if k is not None else None, ptr(mem)
        a.patch = ptr(patch) if patch is not None else None
        a.patch_index = int(patch_index)
        a.patch_rows = int(patch.shape[0]) if patch is not None else 0
        a.B, a.K, a.pos_mode, a.inv_T = q.shape[0], mem.shape[0], pos_mode, inv_T
        a.lse, a.pos, a.loss, a.prob = ptr(outs["lse"]), ptr(outs["pos"]), ptr(outs["loss"]), ptr(outs["prob"])
        a.out_dense = ptr(dense) if dense is not None else None
        a.dtype = {"f32": 0, "bf16": 1}[self.dtype]
        return a

    def forward(self, q, k, mem, T, pos_mode, dense=False, lse_rows=None, patch=None, patch_index=0, stream=None,
                prof=None):
        B, K = q.shape[0], mem.shape[0]
        f32 = dict(dtype=torch.float32, device=q.device)
        outs = dict(lse=torch.empty(lse_rows or B, **f32), pos=torch.empty(B, **f32), loss=torch.empty(1, **f32),
                    prob=torch.empty(1, **f32))
        out = torch.empty(B, K + (1 if pos_mode == 0 else 0), **f32) if dense else None
        ws, nbytes = self._workspace(B, K, q.device)
        a = self._args(q, k, mem, 1.0 / T, pos_mode, patch, patch_index, outs, out)
        rc = self.lib.gcc_nce_forward(ctypes.byref(a), self.ptr(ws), nbytes, prof.handle if prof else None, stream)
        if rc != 0:
            raise RuntimeError(f"gcc_nce_forward failed ({rc}): {self.lib.gcc_last_error().decode()}")
        outs["out"] = out
        return outs

    def backward(self, q, k, mem, T, pos_mode, outs, dloss, patch=None, patch_index=0, by_mem_row=False,
                 stream=None, prof=None):
        B, K = q.shape[0], mem.shape[0]
        dq = torch.empty_like(q)
        ws, nbytes = self._workspace(B, K, q.device)
        a = self._args(q, k, mem, 1.0 / T, pos_mode, patch, patch_index, outs, None)
        dloss = dloss.reshape(1).to(torch.float32).contiguous()
        rc = self.lib.gcc_nce_backward(ctypes.byref(a), self.ptr(dloss), int(by_mem_row), self.ptr(dq), self.ptr(ws),
                                       nbytes, prof.handle if prof else None, stream)
        if rc != 0:
            raise RuntimeError(f"gcc_nce_backward failed ({rc}): {self.lib.gcc_last_error().decode()}")
        return dq

    def set_scalars(self, scalars, lr, betas, adam_step, enqueue_index, dropout_seed, stream=None):
        """gcc_step_scalars_set: the per-step scalars of a replayed step into their device struct (uint8[24] tensor)."""
        rc = self.lib.gcc_step_scalars_set(self.ptr(scalars), float(lr), float(betas[0]), float(betas[1]), int(adam_step),
                                           int(enqueue_index), int(dropout_seed) & 0xFFFFFFFFFFFFFFFF, stream)
        if rc != 0:
            raise RuntimeError(f"gcc_step_scalars_set failed ({rc}): {self.lib.gcc_last_error().decode()}")

    def fill_scalars(self, ring, slot, lr, betas, adam_step, enqueue_index, dropout_seed):
        """gcc_step_scalars_fill: entry ``slot`` of the host-pinned ring (a CPU uint8 tensor), no device work."""
        self.lib.gcc_step_scalars_fill(ring.data_ptr() + 24 * int(slot), float(lr), float(betas[0]), float(betas[1]),
                                       int(adam_step), int(enqueue_index), int(dropout_seed) & 0xFFFFFFFFFFFFFFFF)

    def fetch_scalars(self, scalars, ring, ring_len, counter, stream=None):
        """gcc_step_scalars_fetch: the step's first launch (captured with it): ring[counter % ring_len] -> device struct."""
        rc = self.lib.gcc_step_scalars_fetch(self.ptr(scalars), ring.data_ptr(), int(ring_len), self.ptr(counter), stream)
        if rc != 0:
            raise RuntimeError(f"gcc_step_scalars_fetch failed ({rc}): {self.lib.gcc_last_error().decode()}")

    def enqueue(self, mem, keys, index, save=True, stream=None, scalars=None):
        if scalars is not None:                      # ring pointer from the device struct (replayed step); no saved rows
            rc = self.lib.gcc_queue_enqueue_scalars(self.ptr(mem), mem.shape[0], self.ptr(keys), keys.shape[0],
                                                    self.ptr(scalars), stream)
            if rc != 0:
                raise RuntimeError(f"gcc_queue_enqueue_scalars failed ({rc}): {self.lib.gcc_last_error().decode()}")
            return None
        saved = torch.empty_like(keys) if save else None
        rc = self.lib.gcc_queue_enqueue(self.ptr(mem), mem.shape[0], self.ptr(keys), keys.shape[0], int(index),
                                        self.ptr(saved) if saved is not None else None, stream)
        if rc != 0:
            raise RuntimeError(f"gcc_queue_enqueue failed ({rc}): {self.lib.gcc_last_error().decode()}")
        return saved

    def adam(self, param, grad, exp_avg, exp_avg_sq, lr, betas, eps, weight_decay, step, max_norm, grad_norm,
             scratch, stream=None, grad_scale=1.0):
        rc = self.lib.gcc_adam_step(self.ptr(param), self.ptr(grad), self.ptr(exp_avg), self.ptr(exp_avg_sq),
                                    param.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                    float(weight_decay), int(step), float(max_norm), float(grad_scale), self.ptr(grad_norm),
                                    self.ptr(scratch), stream)
        if rc != 0:
            raise RuntimeError(f"gcc_adam_step failed ({rc}): {self.lib.gcc_last_error().decode()}")

    def adam_ema(self, param, grad, exp_avg, exp_avg_sq, lr, betas, eps, weight_decay, step, max_norm, grad_norm,
                 scratch, stream=None, grad_scale=1.0, ema=None, ema_src=None, ema_m=0.0, meters=None, scalars=None):
        """gcc_adam_ema_step: clip + Adam over ``param`` (the live prefix of the flat buffer ``ema_src``), the EMA copy
        ``ema`` of all of ``ema_src`` and one step of the meters ``(acc, mx, loss, prob, q, k)`` in the Adam launch."""
        ma = None
        if meters is not None:
            acc, mx, loss, prob, q, k = meters
            ma = _cabi.GccStepMetersArgs(self.ptr(acc), self.ptr(mx), self.ptr(loss), self.ptr(prob), self.ptr(q.node_off),
                                         self.ptr(q.edge_off), self.ptr(k.node_off), int(q.batch_size))
        if ema is not None and (ema_src is None or ema_src.data_ptr() != param.data_ptr() or ema.numel() != ema_src.numel()):
            raise ValueError("adam_ema: param must be a prefix of ema_src, and ema the same size as ema_src")
        if scalars is not None:                      # lr / bias corrections from the device struct (replayed step)
            rc = self.lib.gcc_adam_ema_step_scalars(
                self.ptr(param), self.ptr(grad), self.ptr(exp_avg), self.ptr(exp_avg_sq), param.numel(), float(betas[0]),
                float(betas[1]), float(eps), float(weight_decay), float(max_norm), float(grad_scale), self.ptr(grad_norm),
                self.ptr(scratch), self.ptr(ema) if ema is not None else None, ema.numel() if ema is not None else 0,
                float(ema_m), ctypes.byref(ma) if ma is not None else None, self.ptr(scalars), stream)
            if rc != 0:
                raise RuntimeError(f"gcc_adam_ema_step_scalars failed ({rc}): {self.lib.gcc_last_error().decode()}")
            return
        rc = self.lib.gcc_adam_ema_step(self.ptr(param), self.ptr(grad), self.ptr(exp_avg), self.ptr(exp_avg_sq),
                                        param.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                        float(weight_decay), int(step), float(max_norm), float(grad_scale),
                                        self.ptr(grad_norm), self.ptr(scratch),
                                        self.ptr(ema) if ema is not None else None, ema.numel() if ema is not None else 0,
                                        float(ema_m), ctypes.byref(ma) if ma is not None else None, stream)
        if rc != 0:
            raise RuntimeError(f"gcc_adam_ema_step failed ({rc}): {self.lib.gcc_last_error().decode()}")

    def meters(self, acc, mx, loss, prob, grad_norm, q, k, stream=None):
        rc = self.lib.gcc_step_meters(self.ptr(acc), self.ptr(mx), self.ptr(loss), self.ptr(prob), self.ptr(grad_norm),
                                      self.ptr(q.node_off), self.ptr(q.edge_off), self.ptr(k.node_off),
                                      int(q.batch_size), stream)
        if rc != 0:
            raise RuntimeError(f"gcc_step_meters failed ({rc}): {self.lib.gcc_last_error().decode()}")

    def ema(self, ema, p, m, stream=None):
        rc = self.lib.gcc_ema_update(self.ptr(ema), self.ptr(p), ema.numel(), float(m), stream)
        if rc != 0:
            raise RuntimeError(f"gcc_ema_update failed ({rc}): {self.lib.gcc_last_error().decode()}")


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else None


class _MoCoLoss(torch.autograd.Function):
    """Attaches the HIP backward (recompute with the pre-enqueue queue rows) to the fused loss."""

    @staticmethod
    def forward(ctx, q, k, module, outs, patch, patch_index):
        ctx.module, ctx.outs, ctx.patch, ctx.patch_index = module, outs, patch, patch_index
        ctx.save_for_backward(q, k)
        return outs["loss"].reshape(())

    @staticmethod
    def backward(ctx, dloss):
        q, k = ctx.saved_tensors
        m = ctx.module
        dq = m.engine().backward(q, k, m.kernel_memory(), m.T, 0, ctx.outs, dloss, patch=ctx.patch,
                                 patch_index=ctx.patch_index, stream=_stream(q))
        return dq, None, None, None, None, None


class _NSLoss(torch.autograd.Function):
    """CE(feat_k feat_q^T / T, arange) of train.py:400 + criterions.py:27-33; gradients to both views."""

    @staticmethod
    def forward(ctx, feat_q, feat_k, T, eng):
        fq, fk = feat_q.contiguous(), feat_k.contiguous()
        outs = eng.forward(fk, None, fq, T, 1, stream=_stream(fq))     # rows = feat_k, columns = feat_q
        ctx.eng, ctx.T, ctx.outs = eng, T, outs
        ctx.save_for_backward(fq, fk)
        ctx.mark_non_differentiable(outs["prob"])
        return outs["loss"].reshape(()), outs["prob"].reshape(())

    @staticmethod
    def backward(ctx, dloss, _dprob):
        fq, fk = ctx.saved_tensors
        eng, T, outs = ctx.eng, ctx.T, ctx.outs
        dk = eng.backward(fk, None, fq, T, 1, outs, dloss, stream=_stream(fq))
        dq = eng.backward(fq, None, fk, T, 1, outs, dloss, by_mem_row=True, stream=_stream(fq))
        return dq, dk, None, None


class WideNceEngine:
    """The head at feature sizes above 64 (csrc/ginx.hip: gcc_ncex_forward / gcc_queue_enqueue_x): dense logits, and the
    gradient of the loss taken in the forward call -- against the queue before the step's keys overwrite rows of it."""

    def __init__(self, lib=None, ptr=None):
        self.lib = lib if lib is not None else _cabi.load()
        self.ptr = ptr if ptr is not None else _cabi.dev_ptr

    def forward(self, rows, k, mem, T, mode, stream=None):
        """mode 0: MoCo (rows = q, k = keys, mem = queue [K, D]); mode 1: in-batch (rows = feat_k, mem = feat_q).
        -> dict(out, dlog, grad_rows, grad_mem, loss, prob)"""
        B, D = rows.shape
        K = mem.shape[0]
        f32 = dict(dtype=torch.float32, device=rows.device)
        ncols = K + 1 if mode == 0 else K
        o = dict(out=torch.empty(B, ncols, **f32), dlog=torch.empty(B, ncols, **f32), grad_rows=torch.empty(B, D, **f32),
                 grad_mem=torch.empty(K, D, **f32) if mode == 1 else None, loss=torch.empty(1, **f32), prob=torch.empty(1, **f32),
                 acc=torch.zeros(2 + B * D, dtype=torch.float64, device=rows.device))     # [2] loss / prob sums + [B, D] accumulator of grad_rows
        rc = self.lib.gcc_ncex_forward(self.ptr(rows), self.ptr(k) if k is not None else None, self.ptr(mem), B, K, D, 1.0 / T, mode,
                                       self.ptr(o["out"]), self.ptr(o["dlog"]), self.ptr(o["grad_rows"]),
                                       self.ptr(o["grad_mem"]) if o["grad_mem"] is not None else None, self.ptr(o["loss"]),
                                       self.ptr(o["prob"]), self.ptr(o["acc"]), stream)
        if rc != 0:
            raise RuntimeError(f"gcc_ncex_forward failed ({rc}): {self.lib.gcc_last_error().decode()}")
        return o

    def enqueue(self, mem, keys, index, stream=None):
        rc = self.lib.gcc_queue_enqueue_x(self.ptr(mem), mem.shape[0], mem.shape[1], self.ptr(keys), keys.shape[0], int(index), stream)
        if rc != 0:
            raise RuntimeError(f"gcc_queue_enqueue_x failed ({rc}): {self.lib.gcc_last_error().decode()}")


class _WideLoss(torch.autograd.Function):
    """loss of a WideNceEngine.forward result; backward scales the gradients that call already produced."""

    @staticmethod
    def forward(ctx, rows, other, outs):
        ctx.outs, ctx.has_other = outs, other is not None
        return outs["loss"].reshape(())

    @staticmethod
    def backward(ctx, dloss):
        o = ctx.outs
        return o["grad_rows"] * dloss, (o["grad_mem"] * dloss if ctx.has_other else None), None


class NCELogits:
    """What ``MemoryMoCo.forward`` returns in place of the dense [B, K+1] tensor."""

    def __init__(self, loss, prob, pos, shape, dense_fn):
        self.loss, self.prob, self.pos = loss, prob, pos
        self.shape = torch.Size(shape)
        self._dense_fn = dense_fn

    def __getitem__(self, idx):
        if isinstance(idx, tuple) and len(idx) == 2 and idx[0] == slice(None) and idx[1] == 0:
            return self.pos                                             # out[:, 0]  (train.py:394)
        return self.dense()[idx]

    def squeeze(self):
        return self

    def dense(self):
        return self._dense_fn()


class MemoryMoCo(nn.Module):
    """memory_moco.py:7-24: fixed-size queue; buffers ``params`` and ``memory`` (checkpoint["contrast"])."""

    def __init__(self, inputSize, outputSize, K, T=0.07, use_softmax=False, nce_dtype="f32"):
        super().__init__()
        self.nce_dtype = nce_dtype         # not in the reference: "bf16" selects the throughput mode of the head
        if inputSize < 1:
            raise ValueError("feature size must be positive")
        # up to 64: the fused head of csrc/nce.hip (narrower than 64: zero-padded, exactly); above: the dense any-size head of
        # csrc/ginx.hip (--hidden-size above 64, train.py:93,627-629)
        self.wide = inputSize > D
        if self.wide and nce_dtype != "f32":
            raise NotImplementedError(f"nce_dtype={nce_dtype!r} is the 64-channel head's throughput mode (csrc/nce.hip); the any-size head "
                                      f"(feature size {inputSize} > {D}) computes in f32 -- drop --nce-dtype or use --hidden-size <= {D}")
        if not use_softmax:
            raise NotImplementedError("train.py:628 always passes use_softmax=True (the exp/Z branch is dead)")
        self.outputSize = outputSize
        self.inputSize = inputSize
        self.queueSize = K
        self.T = T
        self.index = 0                     # Python int, not saved -- as in the reference (memory_moco.py:16,61)
        self.use_softmax = use_softmax
        self.register_buffer("params", torch.tensor([-1]))
        stdv = 1.0 / math.sqrt(inputSize / 3)
        self.register_buffer("memory", torch.rand(self.queueSize, inputSize).mul_(2 * stdv).add_(-stdv))
        self.gather_keys = None            # multi-GPU: all-gather of keys before the enqueue
        self._engine = None
        print("using queue shape: ({},{})".format(self.queueSize, inputSize))

    def engine(self):
        if self._engine is None:
            self._engine = WideNceEngine() if self.wide else NceEngine(dtype=self.nce_dtype)
        return self._engine

    # ---- feature sizes below 64 (--hidden-size): the kernels' rows are 64 floats.  The ``memory`` buffer keeps the
    # reference's shape [K, inputSize] (checkpoint["contrast"]) as the column slice of a zero-padded [K, 64] block that the
    # kernels read and write; zero columns change no dot product, norm or gradient.
    def kernel_memory(self):
        if self.inputSize >= D:
            return self.memory
        m = self.memory
        big = getattr(self, "_mem64", None)
        if big is None or m.data_ptr() != big.data_ptr() or m.stride(0) != D or big.device != m.device:
            big = torch.zeros(self.queueSize, D, dtype=m.dtype, device=m.device)   # (.to() / load re-materialised the buffer)
            big[:, : self.inputSize].copy_(m)
            self._mem64 = big
            self._buffers["memory"] = big[:, : self.inputSize]
        return big

    def _pad(self, t):
        return t if t.shape[1] == D else torch.nn.functional.pad(t, (0, D - t.shape[1]))

    def _forward_wide(self, q, k):
        eng = self.engine()
        qc, kc = q.contiguous(), k.detach().contiguous()               # memory_moco.py:28
        st = _stream(qc)
        o = eng.forward(qc.detach(), kc, self.memory, self.T, 0, stream=st)   # logits AND gradient vs the queue before the update
        keys = self.gather_keys(kc) if self.gather_keys is not None else kc
        with torch.no_grad():                                          # memory_moco.py:55-61
            eng.enqueue(self.memory, keys, self.index, stream=st)
        self.index = (self.index + keys.shape[0]) % self.queueSize
        loss = _WideLoss.apply(qc, None, o)
        return NCELogits(loss, o["prob"].reshape(()), o["out"][:, 0], (q.shape[0], self.queueSize + 1), lambda: o["out"])

    def forward(self, q, k):
        if self.wide:
            return self._forward_wide(q, k)
        eng = self.engine()
        mem = self.kernel_memory()
        qc = self._pad(q).contiguous()                                 # (differentiable: the gradient comes back sliced)
        kc = self._pad(k.detach()).contiguous()                        # memory_moco.py:28
        st = _stream(qc)
        outs = eng.forward(qc.detach(), kc, mem, self.T, 0, stream=st)   # logits vs the queue BEFORE the update
        keys = self.gather_keys(kc) if self.gather_keys is not None else kc
        index = self.index
        with torch.no_grad():                                          # memory_moco.py:55-61
            saved = eng.enqueue(mem, keys, index, save=True, stream=st)
        self.index = (index + keys.shape[0]) % self.queueSize
        loss = _MoCoLoss.apply(qc, kc, self, outs, saved, index)

        def dense():
            return eng.forward(qc.detach(), kc, mem, self.T, 0, dense=True, patch=saved, patch_index=index,
                               stream=st)["out"]

        return NCELogits(loss, outs["prob"].reshape(()), outs["pos"], (q.shape[0], self.queueSize + 1), dense)

    def logits(self, q, k):
        """Dense ``out`` of memory_moco.py:40-44 without the enqueue side effect (tests, debugging)."""
        if self.wide:
            return self.engine().forward(q.detach().contiguous(), k.detach().contiguous(), self.memory, self.T, 0, stream=_stream(q))["out"]
        outs = self.engine().forward(self._pad(q.detach()).contiguous(), self._pad(k.detach()).contiguous(), self.kernel_memory(), self.T, 0,
                                     dense=True, stream=_stream(q))
        return outs["out"]


class NCESoftmaxLoss(nn.Module):
    """criterions.py:5-17 (label 0)."""

    def forward(self, x):
        if isinstance(x, NCELogits):
            return x.loss
        raise TypeError("NCESoftmaxLoss expects the NCELogits returned by gcc_amd.contrast.MemoryMoCo")


class NCESoftmaxLossNS(nn.Module):
    """criterions.py:20-33 (labels on the diagonal).  ``x`` comes from :func:`e2e_logits`."""

    def forward(self, x):
        if isinstance(x, NCELogits):
            return x.loss
        raise TypeError("NCESoftmaxLossNS expects the NCELogits returned by gcc_amd.contrast.e2e_logits")


def e2e_logits(feat_q, feat_k, T, engine=None):
    """``torch.matmul(feat_k, feat_q.t()) / T`` of train.py:400, fused with its loss."""
    if feat_q.shape[1] > D:                        # --hidden-size above 64: the dense any-size head (csrc/ginx.hip)
        eng = engine if engine is not None else WideNceEngine()
        fq, fk = feat_q.contiguous(), feat_k.contiguous()
        o = eng.forward(fk.detach(), None, fq.detach(), T, 1, stream=_stream(fq))      # rows = feat_k, columns = feat_q
        loss = _WideLoss.apply(fk, fq, o)
        return NCELogits(loss, o["prob"].reshape(()), None, (fq.shape[0], fq.shape[0]), lambda: o["out"])
    eng = engine if engine is not None else NceEngine()
    if feat_q.shape[1] != D:                       # --hidden-size below 64: zero columns change no dot product
        feat_q = torch.nn.functional.pad(feat_q, (0, D - feat_q.shape[1]))
        feat_k = torch.nn.functional.pad(feat_k, (0, D - feat_k.shape[1]))
    loss, prob = _NSLoss.apply(feat_q, feat_k, T, eng)

    def dense():
        return eng.forward(feat_k.detach().contiguous(), None, feat_q.detach().contiguous(), T, 1, dense=True,
                           stream=_stream(feat_q))["out"]

    B = feat_q.shape[0]
    return NCELogits(loss, prob, None, (B, B), dense)
