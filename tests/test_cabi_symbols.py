"""The C-ABI library builds, loads and exports every symbol include/gcc_amd.h
declares (no compute calls: there is no GPU in the CPU test tier)."""
import ctypes
import os
import re

from gcc_amd import _cabi


def _declared_functions():
    src = open(_cabi.HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gcc_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert _declared_functions() == sorted(set(_cabi.SIGNATURES) | _cabi.PENDING)


def test_library_exports_every_symbol():
    import __graft_entry__ as ge

    ge.build()
    assert os.path.exists(_cabi.LIB_PATH)
    lib = ctypes.CDLL(_cabi.LIB_PATH)
    for name in _declared_functions():
        if name not in _cabi.PENDING:
            assert hasattr(lib, name), name
    _cabi.declare(lib)
    assert lib.gcc_abi_version() == 3 == _cabi.ABI_VERSION


def test_product_refuses_cpu_tensors():
    import pytest
    import torch

    with pytest.raises(RuntimeError):
        _cabi.dev_ptr(torch.zeros(4))
