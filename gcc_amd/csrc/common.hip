// gcc_amd/csrc/common.hip -- ABI version, error string, profiling marks.
#include "device_compat.h"
#include "host_common.h"

#include <stdlib.h>

namespace {
// A synthetic co-tenant for contention experiments (tools/load_probe.py): `workgroups` resident workgroups that do ONE kind of
// thing for `ticks` of the 100 MHz clock -- 0: barriers and a little LDS traffic (occupancy: wave slots + LDS, no memory, no
// arithmetic: what a latency-bound solver workgroup looks like to its neighbours), 1: stream float4 reads over `buf` (memory
// system: L2 when the buffer is small, Infinity Cache / HBM when it is large), 2: independent FMA chains (issue slots, power).
__global__ void debug_load_kernel(int32_t kind, long long ticks, int32_t max_iters, const float4 *buf, long long n4, float *sink)
{
    DYN_SMEM(smem);
    const long long t0 = device_ticks();
    const int tid = (int)threadIdx.x;
    float a0 = (float)tid, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    long long idx = n4 > 0 ? ((long long)blockIdx.x * blockDim.x + tid) % n4 : 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (int it = 0; it < max_iters; ++it) {
        if (kind == 0) {
            ((volatile float *)smem)[tid] = a0;
            __syncthreads();
            a0 += ((volatile float *)smem)[(tid + 64) % (int)blockDim.x];
            __syncthreads();
        } else if (kind == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 v = buf[idx];
                a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
                idx += stride;
                if (idx >= n4) idx -= n4;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 64; ++u) {
                a0 = fmaf(a0, 1.0001f, 0.5f); a1 = fmaf(a1, 0.9999f, 0.25f);
                a2 = fmaf(a2, 1.0002f, 0.125f); a3 = fmaf(a3, 0.9998f, 0.0625f);
            }
        }
        if ((it & 7) == 7 && device_ticks() - t0 >= ticks) break;
    }
    if (a0 + a1 + a2 + a3 == 12345.678f) sink[0] = a0;      // (keeps the work alive)
}
}  // namespace

thread_local char g_err[kErrLen] = "";

extern "C" {

int32_t gcc_abi_version(void) { return GCC_AMD_ABI_VERSION; }
const char *gcc_last_error(void) { return g_err; }

gcc_prof *gcc_prof_create(int32_t num_marks)
{
    if (num_marks <= 0 || num_marks > 4096) return nullptr;
    gcc_prof *p = (gcc_prof *)malloc(sizeof(gcc_prof));
    p->n = num_marks;
    p->ev = (hipEvent_t *)malloc(sizeof(hipEvent_t) * (size_t)num_marks);
    for (int i = 0; i < num_marks; ++i) (void)hipEventCreate(&p->ev[i]);
    return p;
}

void gcc_prof_destroy(gcc_prof *p)
{
    if (!p) return;
    for (int i = 0; i < p->n; ++i) (void)hipEventDestroy(p->ev[i]);
    free(p->ev);
    free(p);
}

int32_t gcc_prof_elapsed_ms(gcc_prof *p, int32_t from_mark, int32_t to_mark, float *ms)
{
    if (!p || !ms || from_mark < 0 || to_mark < 0 || from_mark >= p->n || to_mark >= p->n) {
        snprintf(g_err, kErrLen, "gcc_prof_elapsed_ms: bad argument");
        return -1;
    }
    (void)hipEventSynchronize(p->ev[to_mark]);
    hipError_t e = hipEventElapsedTime(ms, p->ev[from_mark], p->ev[to_mark]);
    if (e != hipSuccess) {
        snprintf(g_err, kErrLen, "gcc_prof_elapsed_ms: %s", hipGetErrorString(e));
        return -2;
    }
    return 0;
}

int32_t gcc_stream_create_cu_mask(const uint32_t *cu_mask, int32_t words, void **stream)
{
    if (!cu_mask || !stream || words < 1 || words > 64) {
        snprintf(g_err, kErrLen, "gcc_stream_create_cu_mask: bad argument");
        return -1;
    }
#ifdef GCC_AMD_HIPEMU
    snprintf(g_err, kErrLen, "gcc_stream_create_cu_mask: not available on the emulator");
    return -2;
#else
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, cu_mask);
    if (e != hipSuccess) {
        snprintf(g_err, kErrLen, "gcc_stream_create_cu_mask: %s", hipGetErrorString(e));
        return -10;
    }
    *stream = (void *)s;
    return 0;
#endif
}

int32_t gcc_debug_load(int32_t kind, int32_t workgroups, int32_t threads, int32_t lds_bytes, int64_t ticks, int32_t max_iters,
                       const float *buf, int64_t buf_floats, float *sink, void *stream)
{
    if (kind < 0 || kind > 2 || workgroups < 1 || threads < 64 || threads > 1024 || (threads & 63) || lds_bytes < 4 * threads ||
        lds_bytes > 160 * 1024 - 512 || max_iters < 1 || !sink || (kind == 1 && (!buf || buf_floats < 4 * (int64_t)workgroups * threads))) {
        snprintf(g_err, kErrLen, "gcc_debug_load: bad argument");
        return -1;
    }
#ifndef GCC_AMD_HIPEMU
    if (lds_bytes > 64 * 1024) (void)hipFuncSetAttribute((const void *)debug_load_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
#endif
    hipLaunchKernelGGL(debug_load_kernel, dim3(workgroups), dim3(threads), (size_t)lds_bytes, (hipStream_t)stream, kind, (long long)ticks,
                       max_iters, (const float4 *)buf, (long long)(buf_floats / 4), sink);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, kErrLen, "gcc_debug_load: launch failed: %s", hipGetErrorString(e));
        return -10;
    }
    return 0;
}

int32_t gcc_stream_destroy(void *stream)
{
#ifndef GCC_AMD_HIPEMU
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) {
        snprintf(g_err, kErrLen, "gcc_stream_destroy: %s", hipGetErrorString(e));
        return -10;
    }
#endif
    return 0;
}

}  // extern "C"
