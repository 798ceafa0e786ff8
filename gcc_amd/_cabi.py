"""ctypes binding of libgcc_amd.so (include/gcc_amd.h).

This is the stub a maintainer of the reference would add: the
reference is pure Python, so the FFI is ctypes.  The product path has NO CPU
fallback: :func:`load` raises if the HIP library is missing, and every wrapper
in this package refuses non-CUDA tensors.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgcc_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "gcc_amd.h")

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)

STATUS_SCRATCH_OVERFLOW = 1
STATUS_NODE_OVERFLOW = 2
STATUS_EDGE_OVERFLOW = 4


class GccGraph(ctypes.Structure):
    _fields_ = [
        ("row_ptr", ctypes.c_void_p),
        ("col_idx", ctypes.c_void_p),
        ("seed_cdf", ctypes.c_void_p),
        ("ltab", ctypes.c_void_p),
        ("num_nodes", ctypes.c_int64),
        ("num_edges", ctypes.c_int64),
        ("ltab_len", ctypes.c_int32),
        ("lmax", ctypes.c_int32),
        ("shard_off", ctypes.c_void_p),
        ("num_shards", ctypes.c_int32),
        ("flags", ctypes.c_int32),
        ("hub_index", ctypes.c_void_p), ("hub_adj", ctypes.c_void_p),
        ("num_hubs", ctypes.c_int32), ("hub_words", ctypes.c_int32),
        ("hub_table_degree", ctypes.c_int32), ("reserved_", ctypes.c_int32),
    ]


class GccSampleParams(ctypes.Structure):
    _fields_ = [
        ("run_seed", ctypes.c_uint64),
        ("first_sample_id", ctypes.c_int64),
        ("batch_size", ctypes.c_int32),
        ("restart_u32", ctypes.c_uint32),
        ("seeds", ctypes.c_void_p),
        ("prof", ctypes.c_void_p),
        ("hub_degree", ctypes.c_int32),
        ("max_hubs", ctypes.c_int32),
    ]


class GccPosembView(ctypes.Structure):
    _fields_ = [("g", ctypes.c_void_p), ("pos", ctypes.c_void_p), ("evals", ctypes.c_void_p), ("raw", ctypes.c_void_p)]


class GccBatchOut(ctypes.Structure):
    _fields_ = [
        ("node_off", ctypes.c_void_p),
        ("edge_off", ctypes.c_void_p),
        ("parent_nid", ctypes.c_void_p),
        ("graph_id", ctypes.c_void_p),
        ("row_ptr", ctypes.c_void_p),
        ("col_idx", ctypes.c_void_p),
        ("node_cap", ctypes.c_int64),
        ("edge_cap", ctypes.c_int64),
    ]


GIN_MAX_LAYERS = 8
GIN_HIDDEN = 64
_VP = ctypes.c_void_p


class GccBn(ctypes.Structure):
    _fields_ = [("weight", _VP), ("bias", _VP), ("running_mean", _VP), ("running_var", _VP),
                ("num_batches_tracked", _VP)]


class GccGinWeights(ctypes.Structure):
    _fields_ = [
        ("num_gin_layers", ctypes.c_int32), ("pos_dim", ctypes.c_int32), ("deg_emb_dim", ctypes.c_int32),
        ("max_degree", ctypes.c_int32),
        ("degree_embedding", _VP),
        ("lin0_w", _VP * GIN_MAX_LAYERS), ("lin0_b", _VP * GIN_MAX_LAYERS),
        ("lin1_w", _VP * GIN_MAX_LAYERS), ("lin1_b", _VP * GIN_MAX_LAYERS),
        ("bn_a", GccBn * GIN_MAX_LAYERS), ("bn_b", GccBn * GIN_MAX_LAYERS), ("bn_c", GccBn * GIN_MAX_LAYERS),
        ("pred_w", _VP * (GIN_MAX_LAYERS + 1)), ("pred_b", _VP * (GIN_MAX_LAYERS + 1)),
        ("bn_eps", ctypes.c_float), ("bn_momentum", ctypes.c_float), ("dropout_p", ctypes.c_float),
        ("norm_eps", ctypes.c_float),
        ("hidden", ctypes.c_int32),
    ]


class GccGinPass(ctypes.Structure):
    _fields_ = [
        ("node_off", _VP), ("row_ptr", _VP), ("col_idx", _VP), ("graph_id", _VP), ("pos", _VP),
        ("batch_size", ctypes.c_int32), ("training", ctypes.c_int32), ("update_running_stats", ctypes.c_int32),
        ("normalize", ctypes.c_int32),
        ("dropout_keep", _VP),
        ("dropout_seed", ctypes.c_uint64), ("dropout_philox", ctypes.c_int32),
        ("w", GccGinWeights),
        ("x0", _VP), ("agg", _VP * GIN_MAX_LAYERS), ("z1", _VP * GIN_MAX_LAYERS), ("z2", _VP * GIN_MAX_LAYERS),
        ("stats", _VP), ("pooled", _VP), ("score", _VP), ("feat", _VP),
        ("edge_multiplicity", ctypes.c_int32),
        ("bn_totals", _VP),
        ("seed_local", _VP),
        ("scalars", _VP),
        ("node_cap", ctypes.c_int64),
        ("rows_hint", ctypes.c_int64),
    ]


class GccGinxPass(ctypes.Structure):          # gcc_ginx_pass: the encoder at any width (csrc/ginx.hip)
    _fields_ = [
        ("node_off", _VP), ("row_ptr", _VP), ("col_idx", _VP), ("graph_id", _VP), ("pos", _VP), ("seed_local", _VP),
        ("batch_size", ctypes.c_int32), ("training", ctypes.c_int32), ("update_running_stats", ctypes.c_int32),
        ("normalize", ctypes.c_int32),
        ("dropout_keep", _VP),
        ("hidden", ctypes.c_int32), ("out_dim", ctypes.c_int32), ("edge_multiplicity", ctypes.c_int32), ("reserved_", ctypes.c_int32),
        ("node_cap", ctypes.c_int64),
        ("w", GccGinWeights),
        ("workspace", _VP), ("workspace_bytes", ctypes.c_int64),
        ("feat", _VP), ("pooled_out", _VP),
    ]


class GccGinGrads(ctypes.Structure):
    _fields_ = [
        ("degree_embedding", _VP),
        ("lin0_w", _VP * GIN_MAX_LAYERS), ("lin0_b", _VP * GIN_MAX_LAYERS),
        ("lin1_w", _VP * GIN_MAX_LAYERS), ("lin1_b", _VP * GIN_MAX_LAYERS),
        ("bn_a_w", _VP * GIN_MAX_LAYERS), ("bn_a_b", _VP * GIN_MAX_LAYERS),
        ("bn_b_w", _VP * GIN_MAX_LAYERS), ("bn_b_b", _VP * GIN_MAX_LAYERS),
        ("bn_c_w", _VP * GIN_MAX_LAYERS), ("bn_c_b", _VP * GIN_MAX_LAYERS),
        ("pred_w", _VP * (GIN_MAX_LAYERS + 1)), ("pred_b", _VP * (GIN_MAX_LAYERS + 1)),
    ]


class GccNceArgs(ctypes.Structure):
    _fields_ = [
        ("q", _VP), ("k", _VP), ("mem", _VP), ("patch", _VP),
        ("patch_index", ctypes.c_int32), ("patch_rows", ctypes.c_int32),
        ("B", ctypes.c_int32), ("K", ctypes.c_int32), ("pos_mode", ctypes.c_int32), ("inv_T", ctypes.c_float),
        ("lse", _VP), ("pos", _VP), ("loss", _VP), ("prob", _VP), ("out_dense", _VP),
        ("dtype", ctypes.c_int32),
    ]


class GccStepMetersArgs(ctypes.Structure):
    _fields_ = [("acc", _VP), ("mx", _VP), ("loss", _VP), ("prob", _VP), ("node_off_q", _VP), ("edge_off_q", _VP),
                ("node_off_k", _VP), ("batch_size", ctypes.c_int32)]


class GccGinwLayer(ctypes.Structure):
    _fields_ = [("w0", _VP), ("w1", _VP), ("s0", _VP), ("t0", _VP), ("s1", _VP), ("t1", _VP), ("s2", _VP), ("t2", _VP),
                ("w0_frag", _VP), ("w1_frag", _VP)]


class GccGinwArgs(ctypes.Structure):
    _fields_ = [
        ("node_off", _VP), ("row_ptr", _VP), ("col_idx", _VP), ("x_in", _VP), ("x_out", _VP), ("pooled", _VP),
        ("batch_size", ctypes.c_int32), ("num_layers", ctypes.c_int32),
        ("layers", GccGinwLayer * 8),
        ("scratch", _VP), ("scratch_bytes", ctypes.c_int64), ("num_nodes", ctypes.c_int64),
    ]


ABI_VERSION = 3          # GCC_AMD_ABI_VERSION of the include/gcc_amd.h these structs mirror (checked in load())
GRAPH_CONTRACT_CHECKED = 1   # gcc_graph.flags

# name -> (restype, argtypes); the single source of truth for the symbol test
SIGNATURES = {
    "gcc_abi_version": (ctypes.c_int32, []),
    "gcc_last_error": (ctypes.c_char_p, []),
    "gcc_stream_create_cu_mask": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]),
    "gcc_stream_destroy": (ctypes.c_int32, [ctypes.c_void_p]),
    "gcc_prof_create": (ctypes.c_void_p, [ctypes.c_int32]),
    "gcc_prof_destroy": (None, [ctypes.c_void_p]),
    "gcc_prof_elapsed_ms": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, c_f32p]),
    "gcc_sampler_workspace_bytes": (ctypes.c_int64, [ctypes.POINTER(GccGraph), ctypes.c_int32, ctypes.c_int64]),
    "gcc_sampler_workspace_bytes_multi": (ctypes.c_int64, [ctypes.POINTER(GccGraph), ctypes.c_int32, ctypes.c_int32, ctypes.c_int64]),
    "gcc_sample_multi": (ctypes.c_int32, [
        ctypes.POINTER(GccGraph), ctypes.POINTER(GccSampleParams), ctypes.c_int32, ctypes.c_int64, ctypes.POINTER(GccBatchOut),
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "gcc_sample_batch": (ctypes.c_int32, [
        ctypes.POINTER(GccGraph), ctypes.POINTER(GccSampleParams), ctypes.POINTER(GccBatchOut),
        ctypes.POINTER(GccBatchOut), ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
        ctypes.c_void_p]),
    "gcc_posemb_workspace_bytes": (ctypes.c_int64, [ctypes.c_int32, ctypes.c_int64, ctypes.c_int32]),
    "gcc_posemb_debug_ticks": (None, [ctypes.c_void_p]),
    "gcc_posemb_set_fork": (None, [ctypes.c_int32]),
    "gcc_sampler_debug_ticks": (None, [ctypes.c_void_p]),
    "gcc_sampler_debug_grids": (None, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "gcc_debug_load": (ctypes.c_int32, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32,
                                        ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "gcc_gin_debug_ticks": (None, [ctypes.c_void_p]),
    "gcc_ginw_debug_ticks": (None, [ctypes.c_void_p]),
    "gcc_posemb_multi_workspace_bytes": (ctypes.c_int64, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32]),
    "gcc_posemb_multi": (ctypes.c_int32, [ctypes.POINTER(GccPosembView), ctypes.c_int32, ctypes.c_int32, ctypes.c_int64,
                                          ctypes.c_int32, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p]),
    "gcc_posemb_multi_gated": (ctypes.c_int32, [ctypes.POINTER(GccPosembView), ctypes.c_int32, ctypes.c_int32, ctypes.c_int64,
                                          ctypes.c_int32, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "gcc_posemb": (ctypes.c_int32, [ctypes.POINTER(GccBatchOut), ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_void_p]),
    "gcc_gin_forward": (ctypes.c_int32, [ctypes.POINTER(GccGinPass), ctypes.c_int32, ctypes.c_void_p,
                                         ctypes.c_void_p]),
    "gcc_gin_backward_workspace_bytes": (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]),
    "gcc_gin_backward": (ctypes.c_int32, [ctypes.POINTER(GccGinPass), ctypes.c_void_p, ctypes.POINTER(GccGinGrads),
                                          ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                          ctypes.c_void_p, ctypes.c_void_p]),
    "gcc_ginx_workspace_bytes": (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "gcc_ginx_forward": (ctypes.c_int32, [ctypes.POINTER(GccGinxPass), ctypes.c_void_p]),
    "gcc_ginx_backward": (ctypes.c_int32, [ctypes.POINTER(GccGinxPass), ctypes.c_void_p, ctypes.POINTER(GccGinGrads), ctypes.c_void_p]),
    "gcc_ncex_forward": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_float, ctypes.c_int32] + [ctypes.c_void_p] * 8),
    "gcc_queue_enqueue_x": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_void_p]),
    "gcc_ginw_forward": (ctypes.c_int32, [ctypes.POINTER(GccGinwArgs), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "gcc_ginw_pack_weights": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]),
    "gcc_nce_workspace_bytes": (ctypes.c_int64, [ctypes.c_int32, ctypes.c_int32]),
    "gcc_nce_forward": (ctypes.c_int32, [ctypes.POINTER(GccNceArgs), ctypes.c_void_p, ctypes.c_int64,
                                         ctypes.c_void_p, ctypes.c_void_p]),
    "gcc_nce_backward": (ctypes.c_int32, [ctypes.POINTER(GccNceArgs), ctypes.c_void_p, ctypes.c_int32,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                          ctypes.c_void_p]),
    "gcc_queue_enqueue": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                           ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "gcc_adam_step": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                       ctypes.c_float, ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p]),
    "gcc_adam_ema_step": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                           ctypes.c_float, ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_float,
                                           ctypes.POINTER(GccStepMetersArgs), ctypes.c_void_p]),
    "gcc_adam_ema_step_scalars": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                                   ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p,
                                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_float,
                                                   ctypes.POINTER(GccStepMetersArgs), ctypes.c_void_p, ctypes.c_void_p]),
    "gcc_queue_enqueue_scalars": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                                   ctypes.c_void_p, ctypes.c_void_p]),
    "gcc_ginw_scratch_bytes": (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int32]),
    "gcc_gin_eval_debug_ticks": (None, [ctypes.c_void_p]),
    "gcc_gin_eval_fused": (ctypes.c_int32, [ctypes.POINTER(GccGinPass), ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "gcc_step_scalars_fill": (None, [ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int32,
                                      ctypes.c_int32, ctypes.c_uint64]),
    "gcc_step_scalars_fetch": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "gcc_step_scalars_set": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int32,
                                              ctypes.c_int32, ctypes.c_uint64, ctypes.c_void_p]),
    "gcc_step_meters": (ctypes.c_int32, [ctypes.c_void_p] * 8 + [ctypes.c_int32, ctypes.c_void_p]),
    "gcc_ema_update": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_float,
                                        ctypes.c_void_p]),
}
# symbols declared in the header but not built yet are listed here while the build is in progress
PENDING = set()


def declare(lib: ctypes.CDLL) -> ctypes.CDLL:
    """Attach restype/argtypes for every symbol of include/gcc_amd.h."""
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def load() -> ctypes.CDLL:
    """Open the gfx950 library built in-tree by ``__graft_entry__.build()``."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  gcc_amd has no CPU fallback.")
        # PyTorch first: its wheel bundles its own libamdhip64.  If this library were opened before torch, the loader would
        # satisfy its libamdhip64 dependency from /opt/rocm and the process would hold TWO HIP runtimes -- kernels
        # registered with one, torch's streams and allocations owned by the other ("no ROCm-capable device is detected"
        # at the first launch; seen when build() and smoke() ran in one process).
        import torch  # noqa: F401

        lib = declare(ctypes.CDLL(LIB_PATH))
        got = lib.gcc_abi_version()
        if got != ABI_VERSION:             # the ctypes structs below mirror ONE layout of include/gcc_amd.h
            raise RuntimeError(f"{LIB_PATH} reports C-ABI version {got}, this binding was written for {ABI_VERSION}: "
                               "rebuild the library (python -c 'import __graft_entry__ as g; g.build()')")
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {load().gcc_last_error().decode()}")


def dev_ptr(t, dtype=None) -> int:
    """data_ptr() of a contiguous CUDA (HIP) tensor; anything else is refused."""
    import torch

    if t is None:
        return 0
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("gcc_amd kernels take device (HIP) tensors only; there is no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return t.data_ptr()
