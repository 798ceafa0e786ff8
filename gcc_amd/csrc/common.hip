// gcc_amd/csrc/common.hip -- ABI version, error string, profiling marks.
#include "host_common.h"

#include <stdlib.h>

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
