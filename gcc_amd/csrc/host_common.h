// gcc_amd/csrc/host_common.h -- host-side helpers shared by the C-ABI
// translation units: last-error buffer and the profiling-mark ring.
#pragma once
#include "device_compat.h"
#include "../../include/gcc_amd.h"

#include <stdio.h>
#include <string.h>

constexpr int kErrLen = 256;
extern thread_local char g_err[kErrLen];

struct gcc_prof {
    int32_t n;
    hipEvent_t *ev;
};

static inline void prof_mark(gcc_prof *p, int idx, hipStream_t s)
{
    if (p && idx >= 0 && idx < p->n) (void)hipEventRecord(p->ev[idx], s);
}
