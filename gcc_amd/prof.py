"""hipEvent marks recorded between the kernels of one C-ABI call (bench.py's
live per-kernel durations)."""
from __future__ import annotations

import ctypes

from . import _cabi


class Prof:
    def __init__(self, num_marks: int):
        self.lib = _cabi.load()
        self.handle = self.lib.gcc_prof_create(num_marks)
        if not self.handle:
            raise RuntimeError("gcc_prof_create failed")

    def elapsed_ms(self, a: int, b: int) -> float:
        ms = ctypes.c_float(0.0)
        _cabi.check(self.lib.gcc_prof_elapsed_ms(self.handle, a, b, ctypes.byref(ms)), "gcc_prof_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            if self.handle:
                self.lib.gcc_prof_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
