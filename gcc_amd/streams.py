"""HIP streams restricted to a subset of the compute units (gcc_stream_create_cu_mask).

The data pipeline (sampler + eigensolvers) and the training step share one GPU.  Eigensolver workgroups run for
milliseconds and fill a CU's registers / LDS, so the short kernels of a training step wait for CUs to drain.  The
few producer streams can be confined to a CU subset, which leaves the other CUs to the training step alone.  (Only a
handful of masked streams: every masked stream owns a hardware queue, and the command processor serves few queues
well.)"""
import ctypes

import torch

from . import _cabi

TOTAL_CUS = 256          # MI355X


class MaskedStream:
    """torch.cuda.ExternalStream over a CU-masked hipStream_t; destroys the stream with the object."""

    def __init__(self, device, cus, lib=None):
        self.lib = lib or _cabi.load()
        words = (TOTAL_CUS + 31) // 32
        m = (ctypes.c_uint32 * words)()
        for i in cus:
            m[i // 32] |= 1 << (i % 32)
        h = ctypes.c_void_p()
        with torch.cuda.device(device):
            rc = self.lib.gcc_stream_create_cu_mask(ctypes.cast(m, ctypes.c_void_p), words, ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(self.lib.gcc_last_error().decode())
        self.handle = h.value
        self.stream = torch.cuda.ExternalStream(self.handle, device=device)

    def __del__(self):
        try:
            self.lib.gcc_stream_destroy(ctypes.c_void_p(self.handle))
        except Exception:
            pass


def producer_cus(reserved, layout="interleaved"):
    """CU indices the producers may use when ``reserved`` CUs are kept for the training step.
    ``interleaved``: the reserved CUs are spread evenly over the mask (every XCD / shader engine gives up some,
    whatever the bit order means); ``block``: the first ``reserved`` bits."""
    if layout == "block":
        return list(range(reserved, TOTAL_CUS))
    keep_every = TOTAL_CUS / max(reserved, 1)
    res = {int(i * keep_every) for i in range(reserved)}
    return [i for i in range(TOTAL_CUS) if i not in res]
