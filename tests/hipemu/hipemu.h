// tests/hipemu/hipemu.h -- a lock-step wave64 emulator for kernel-logic tests.
//
// TEST INFRASTRUCTURE ONLY.  There is no GPU in the development container and
// only a handful of gpurun calls per round, so the HIP kernels in
// gcc_amd/csrc/*.hip are additionally compiled with g++ against this header
// (-DGCC_AMD_HIPEMU) into tests/hipemu/_build/libgcc_amd_emu.so and exercised by
// the "not gpu" tests through the same C ABI.  The product package never loads
// that library: gcc_amd/_cabi.py only opens gcc_amd/csrc/libgcc_amd.so and
// raises if it or the GPU is missing.
//
// Model: one OS thread; every HIP thread of a block is a fiber; blocks run one
// after another.  Fibers switch only inside collectives (__syncthreads, wave
// shuffles/ballots/MFMA), so plain memory operations and atomics are trivially
// atomic.  A collective reached by only part of a wave (divergent control
// flow, which is undefined behaviour on hardware) aborts with a message.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

namespace hipemu {

struct Dim3 {
    unsigned x, y, z;
    Dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct Fiber {
    void *sp;
    char *stack;
    Dim3 tid;
    int lane, wave;
    bool done;
};

extern Fiber *cur;
extern Dim3 g_blockIdx, g_blockDim, g_gridDim;
extern unsigned char *g_dyn_smem;

[[noreturn]] void fail(const char *msg);      // message + block / thread, abort
void sync_block();
void wave_barrier_only();
// every lane deposits `size` bytes; returns pointer to a [64][size] table valid
// until this wave's next collective.  tag identifies the call site kind.
const unsigned char *wave_gather(const void *in, size_t size, unsigned tag);
void launch(Dim3 grid, Dim3 block, size_t smem, const std::function<void()> &body);

template <class T> inline T gather_at(const unsigned char *tab, int lane)
{
    T v;
    memcpy(&v, tab + (size_t)lane * sizeof(T), sizeof(T));
    return v;
}

}  // namespace hipemu

// ---------------------------------------------------------------- HIP surface
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define threadIdx (hipemu::cur->tid)
#define blockIdx hipemu::g_blockIdx
#define blockDim hipemu::g_blockDim
#define gridDim hipemu::g_gridDim
typedef hipemu::Dim3 dim3;
typedef void *hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char *hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
enum { hipMemcpyDeviceToDevice = 3 };
static inline hipError_t hipMemcpyAsync(void *d, const void *src, size_t n, int, hipStream_t) { memcpy(d, src, n); return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
typedef void *hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }

#define hipLaunchKernelGGL(kern, grid, block, smem, stream, ...) \
    hipemu::launch((grid), (block), (smem), [=]() { kern(__VA_ARGS__); })

static inline void __syncthreads() { hipemu::sync_block(); }

template <class T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T *p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }

static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __clz(int x) { return x == 0 ? 32 : __builtin_clz((unsigned)x); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline double rsqrt(double x) { return 1.0 / sqrt(x); }

template <class A, class B> static inline auto min(A a, B b) -> decltype(a + b) { return a < b ? a : b; }
template <class A, class B> static inline auto max(A a, B b) -> decltype(a + b) { return a > b ? a : b; }

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
