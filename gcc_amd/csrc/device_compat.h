// gcc_amd/csrc/device_compat.h -- the ONLY place that knows about the two ways
// the kernels are compiled:
//   * hipcc --offload-arch=gfx950  -> gcc_amd/csrc/libgcc_amd.so   (the product)
//   * g++ -DGCC_AMD_HIPEMU         -> tests/hipemu/_build/libgcc_amd_emu.so
//     (lock-step wave64 emulator; kernel-logic tests on machines without a GPU)
// Kernels use the wave_* / mfma_* wrappers below instead of raw builtins.
#pragma once
#include <stdint.h>

#ifdef GCC_AMD_HIPEMU
#define TRAIN_STEP_WAVE_PRIORITY() ((void)0)
static inline long long device_ticks() { return 0; }
static inline float fast_rcp(float x) { return 1.0f / x; }
static inline void device_fence() {}
#define lds_barrier() __syncthreads()
#define SCHED_FENCE() ((void)0)
static inline float load_fresh(const float *p) { return *p; }
static inline double load_fresh_f64(const double *p) { return *p; }
static inline int load_fresh_i32(const int *p) { return *p; }
static inline uint32_t load_system_u32(const uint32_t *p) { return *p; }
#include "hipemu.h"

#define DYN_SMEM(name) unsigned char *name = hipemu::g_dyn_smem
typedef float f32x4 __attribute__((vector_size(16)));

static inline int lane_id() { return hipemu::cur->lane; }
static inline void wave_sync() { hipemu::wave_barrier_only(); }
static inline unsigned long long wave_ballot(bool p)
{
    unsigned char v = p ? 1 : 0;
    const unsigned char *t = hipemu::wave_gather(&v, 1, 0xBA1107u);
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) m |= (unsigned long long)(t[l] & 1) << l;
    return m;
}
template <class T> static inline T wave_shfl(T v, int src)
{
    const unsigned char *t = hipemu::wave_gather(&v, sizeof(T), 0x5F1u + (unsigned)sizeof(T));
    return hipemu::gather_at<T>(t, src & 63);
}
template <class T> static inline T wave_shfl_xor(T v, int mask) { return wave_shfl(v, lane_id() ^ mask); }
template <class T> static inline T wave_shfl_up(T v, int delta)
{
    int src = lane_id() - delta;
    T r = wave_shfl(v, src < 0 ? lane_id() : src);
    return src < 0 ? v : r;
}
template <class T> static inline T wave_shfl_down(T v, int delta)
{
    int src = lane_id() + delta;
    T r = wave_shfl(v, src > 63 ? lane_id() : src);
    return src > 63 ? v : r;
}
template <class T> static inline T wave_bcast_first(T v) { return wave_shfl(v, 0); }
// value of lane `src` (src must be wave-uniform): v_readlane_b32 on the device
// sum over each half of the wave (lanes 0-31 / 32-63), returned in every lane of the half
static inline float half32_sum(float v)
{
    for (int d = 1; d <= 16; d <<= 1) v += wave_shfl_xor(v, d);
    return v;
}
static inline void opaque_u64(uint64_t &) {}
static inline void opaque_u32(uint32_t &) {}
#define COMPILER_MEMORY_FENCE() ((void)0)
static inline float wave_readlane(float v, int src) { return wave_shfl(v, src); }
static inline double wave_readlane(double v, int src) { return wave_shfl(v, src); }
static inline int wave_readlane(int v, int src) { return wave_shfl(v, src); }
// a value the caller knows to be wave-uniform (device: moved to a scalar register) / lane 63's value as a uniform
// (the emulator CHECKS the claim among the lanes that are still running: on the device a non-uniform value silently becomes the
// first active lane's -- round 6's work-list kernel passed every emulator test that way and was wrong on the GPU)
static inline int wave_uniform(int v)
{
    struct { int live, v; } me = {1, v};
    const unsigned char *t = hipemu::wave_gather(&me, sizeof(me), 0x0F1E1Du);
    for (int l = 0; l < 64; ++l) {
        const auto o = hipemu::gather_at<decltype(me)>(t, l);
        if (o.live && o.v != v) hipemu::fail("wave_uniform() of a value that differs between the lanes of a wave");
    }
    return v;
}
static inline void keep_alive(double) {}
static inline int wave_last(int v) { return wave_shfl(v, 63); }
// D = A(16x4) * B(4x16) + C; lane l: a = A[l&15][l>>4], b = B[l>>4][l&15],
// c/d[r] = C[(l>>4)*4 + r][l&15]  (cdna_hip_programming.md §3)
static inline f32x4 mfma_16x16x4_f32(float a, float b, f32x4 c)
{
    struct AB { float a, b; } ab = {a, b};
    const unsigned char *t = hipemu::wave_gather(&ab, sizeof(ab), 0x3F3Au);
    int l = lane_id();
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r, col = l & 15;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            AB x = hipemu::gather_at<AB>(t, k * 16 + row);
            AB y = hipemu::gather_at<AB>(t, k * 16 + col);
            acc = fmaf(x.a, y.b, acc);
        }
        d[r] = acc;
    }
    return d;
}

// v_mfma_f64_16x16x4_f64: A / B as the f32 form (one f64 per lane); C / D: c/d[r] = C[(l>>4) + 4 * r][l&15] -- NOT the f32 row map
// (cdna_hip_programming.md, "f64 MFMA does NOT use these maps")
typedef double f64x4 __attribute__((vector_size(32)));
static inline f64x4 mfma_16x16x4_f64(double a, double b, f64x4 c)
{
    struct AB { double a, b; } ab = {a, b};
    const unsigned char *t = hipemu::wave_gather(&ab, sizeof(ab), 0xF64Au);
    int l = lane_id();
    f64x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) + 4 * r, col = l & 15;
        double acc = c[r];
        for (int k = 0; k < 4; ++k) {
            AB x = hipemu::gather_at<AB>(t, k * 16 + row);
            AB y = hipemu::gather_at<AB>(t, k * 16 + col);
            acc = fma(x.a, y.b, acc);
        }
        d[r] = acc;
    }
    return d;
}

// D(16x16) = A(16x32) * B(32x16) + C with bf16 operands: lane l passes 8 bf16 of row (l & 15) of A and of COLUMN
// (l & 15) of B, both for the same 8 values of k (the (l >> 4)-th group); c/d as mfma_16x16x4_f32.
typedef unsigned u32x4 __attribute__((vector_size(16)));
typedef unsigned u32x2 __attribute__((vector_size(8)));
static inline f32x4 mfma_16x16x32_bf16(u32x4 a, u32x4 b, f32x4 c)
{
    struct AB { unsigned a[4], b[4]; } ab;
    for (int i = 0; i < 4; ++i) { ab.a[i] = a[i]; ab.b[i] = b[i]; }
    const unsigned char *t = hipemu::wave_gather(&ab, sizeof(ab), 0xBF16u);
    int l = lane_id();
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r, col = l & 15;
        float acc = c[r];
        for (int g = 0; g < 4; ++g) {
            AB x = hipemu::gather_at<AB>(t, g * 16 + row);
            AB y = hipemu::gather_at<AB>(t, g * 16 + col);
            for (int e = 0; e < 8; ++e) {
                unsigned ua = (x.a[e >> 1] >> ((e & 1) * 16)) << 16, ub = (y.b[e >> 1] >> ((e & 1) * 16)) << 16;
                float fa, fb;
                memcpy(&fa, &ua, 4); memcpy(&fb, &ub, 4);
                acc = fmaf(fa, fb, acc);
            }
        }
        d[r] = acc;
    }
    return d;
}

// sum over each group of 16 consecutive lanes; valid in the LAST lane of the group (lane & 15) == 15
static inline float row16_sum_last(float v)
{
    for (int d = 1; d < 16; d <<= 1) { float t = wave_shfl_up(v, d); if ((lane_id() & 15) >= d) v += t; }
    return v;
}
// ---- wave reductions / scan (shuffle butterflies; the gfx950 build uses DPP row operations)
// inclusive wave prefix sum (all 64 lanes must call)
static inline int wave_scan_incl(int v)
{
    int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = wave_shfl_up(v, d);
        if (l >= d) v += t;
    }
    return v;
}

// inclusive wave prefix maximum of non-negative values
static inline int wave_scan_max_incl(int v)
{
    int l = lane_id();
    for (int d = 1; d < 64; d <<= 1) {
        int t = wave_shfl_up(v, d);
        if (l >= d) v = t > v ? t : v;
    }
    return v;
}
static inline uint32_t umul24(uint32_t a, uint32_t b) { return (uint32_t)((uint64_t)(a & 0xFFFFFFu) * (uint64_t)(b & 0xFFFFFFu)); }

static inline float wave_sum(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += wave_shfl_xor(v, d);
    return v;
}

static inline double wave_sum(double v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += wave_shfl_xor(v, d);
    return v;
}

static inline float wave_max(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { float t = wave_shfl_xor(v, d); v = t > v ? t : v; }
    return v;
}

#else  // ------------------------------------------------------------ gfx950
#include <hip/hip_runtime.h>

// Waves of the training step share SIMDs with the data pipeline's long-running eigensolver waves; the
// step's kernels are short and on the critical path, so their waves take the issue slots first.
#define TRAIN_STEP_WAVE_PRIORITY() __builtin_amdgcn_s_setprio(3)
// the instruction scheduler moves nothing across this point
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ long long device_ticks() { return (long long)wall_clock64(); }   // 100 MHz
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }        // v_rcp_f32, 1 ulp
__device__ __forceinline__ void device_fence() { __threadfence(); }                            // release + acquire, agent scope
// workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not wait for outstanding global
// stores / atomics (s_waitcnt vmcnt(0)), whose acknowledgement takes microseconds for device-scope fp64 atomics
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ float load_fresh(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double load_fresh_f64(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int load_fresh_i32(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// host-written (pinned, device-visible) memory: never from a cache line of an earlier launch
__device__ __forceinline__ uint32_t load_system_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

#define DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }
// LDS hand-off between lanes of ONE wave: LDS operations of a wave execute in
// order, so only the compiler has to be stopped from reordering them.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __ballot(p); }
template <class T> __device__ __forceinline__ T wave_shfl(T v, int src) { return __shfl(v, src, 64); }
template <class T> __device__ __forceinline__ T wave_shfl_xor(T v, int mask) { return __shfl_xor(v, mask, 64); }
template <class T> __device__ __forceinline__ T wave_shfl_up(T v, int delta) { return __shfl_up(v, delta, 64); }
template <class T> __device__ __forceinline__ T wave_shfl_down(T v, int delta) { return __shfl_down(v, delta, 64); }
template <class T> __device__ __forceinline__ T wave_bcast_first(T v) { return __shfl(v, 0, 64); }
#define COMPILER_MEMORY_FENCE() asm volatile("" ::: "memory")
__device__ __forceinline__ void opaque_u32(uint32_t &x) { asm volatile("" : "+v"(x)); }
// hides how a 64-bit value was computed from the optimiser (it stays in two vector registers)
__device__ __forceinline__ void opaque_u64(uint64_t &x)
{
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    x = ((uint64_t)hi << 32) | lo;
}
// value of lane `src` (src must be wave-uniform): one v_readlane_b32 with the lane in a scalar register, no LDS crossbar
__device__ __forceinline__ float wave_readlane(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), __builtin_amdgcn_readfirstlane(src)));
}
__device__ __forceinline__ int wave_readlane(int v, int src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)); }
__device__ __forceinline__ double wave_readlane(double v, int src)
{
    const int s = __builtin_amdgcn_readfirstlane(src);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), s), __builtin_amdgcn_readlane(__double2loint(v), s));
}
// a value the caller knows to be wave-uniform, moved to a scalar register (the compiler cannot tell for values that
// come from threadIdx or a vector load: everything derived from them would stay in VGPRs and on the VALU)
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// forces a value (e.g. the result of a returning atomic) to be waited for
__device__ __forceinline__ void keep_alive(double v) { asm volatile("" ::"v"(v)); }
__device__ __forceinline__ int wave_last(int v) { return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ f32x4 mfma_16x16x4_f32(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// v_mfma_f64_16x16x4_f64 (lane l: a = A[l&15][l>>4], b = B[l>>4][l&15]; c/d[r] = C[(l>>4) + 4 r][l&15]): fp64 products and
// accumulation on the matrix cores -- 128 FLOP per cycle and CU on gfx950, the vector fp64 rate, but without a register / LDS
// operand fetch per FMA (the eigensolver's Gram matrices: gcc_amd/csrc/posemb.hip)
typedef double f64x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f64x4 mfma_16x16x4_f64(double a, double b, f64x4 c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// v_mfma_f32_16x16x32_bf16: lane l passes 8 bf16 of row (l & 15) of A and of column (l & 15) of B for the k-group
// (l >> 4); the products pair element e of group g of A with element e of group g of B, so any operand layout
// with k contiguous per lane works as long as A and B use the same one.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma_16x16x32_bf16(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// ---- wave reductions / scan on DPP row operations (ALU speed; __shfl_* goes through the LDS crossbar, ~60 cycles
// per step).  row_shr:1,2,4,8 leave an inclusive scan in every row of 16 lanes; row_bcast:15 / row_bcast:31 carry
// the row totals on.  All 64 lanes must be active.
template <int kCtrl, int kRowMask>
__device__ __forceinline__ int dpp_or_zero(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, kCtrl, kRowMask, 0xF, true);
}
template <int kCtrl, int kRowMask> __device__ __forceinline__ float dpp_or_zero(float v)
{
    return __int_as_float(dpp_or_zero<kCtrl, kRowMask>(__float_as_int(v)));
}
template <int kCtrl, int kRowMask> __device__ __forceinline__ double dpp_or_zero(double v)
{
    const int lo = dpp_or_zero<kCtrl, kRowMask>(__double2loint(v)), hi = dpp_or_zero<kCtrl, kRowMask>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <class T> __device__ __forceinline__ T wave_scan_incl_dpp(T v)
{
    v += dpp_or_zero<0x111, 0xF>(v);      // row_shr:1
    v += dpp_or_zero<0x112, 0xF>(v);      // row_shr:2
    v += dpp_or_zero<0x114, 0xF>(v);      // row_shr:4
    v += dpp_or_zero<0x118, 0xF>(v);      // row_shr:8
    v += dpp_or_zero<0x142, 0xA>(v);      // row_bcast:15 -> rows 1, 3
    v += dpp_or_zero<0x143, 0xC>(v);      // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ int wave_scan_incl(int v) { return wave_scan_incl_dpp(v); }
// inclusive wave prefix maximum of non-negative values (the lanes a DPP step does not reach read 0)
__device__ __forceinline__ int wave_scan_max_incl(int v)
{
    v = max(v, dpp_or_zero<0x111, 0xF>(v));
    v = max(v, dpp_or_zero<0x112, 0xF>(v));
    v = max(v, dpp_or_zero<0x114, 0xF>(v));
    v = max(v, dpp_or_zero<0x118, 0xF>(v));
    v = max(v, dpp_or_zero<0x142, 0xA>(v));
    v = max(v, dpp_or_zero<0x143, 0xC>(v));
    return v;
}
__device__ __forceinline__ uint32_t umul24(uint32_t a, uint32_t b) { return __umul24(a, b); }   // v_mul_u32_u24: full rate
__device__ __forceinline__ float wave_sum(float v)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_scan_incl_dpp(v)), 63));
}
__device__ __forceinline__ double wave_sum(double v)
{
    v = wave_scan_incl_dpp(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// sum over each half of the wave (lanes 0-31 / 32-63), returned in every lane of the half: one DPP scan, two v_readlane
__device__ __forceinline__ float half32_sum(float v)
{
    const float sc = wave_scan_incl_dpp(v);
    const float t31 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sc), 31));
    const float t63 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sc), 63));
    return lane_id() < 32 ? t31 : t63 - t31;
}
template <int kCtrl, int kRowMask> __device__ __forceinline__ float dpp_or_self(float v)
{
    const int i = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(i, i, kCtrl, kRowMask, 0xF, false));
}
// sum over each group of 16 consecutive lanes (a DPP row); valid in the LAST lane of the group
__device__ __forceinline__ float row16_sum_last(float v)
{
    v += dpp_or_zero<0x111, 0xF>(v);
    v += dpp_or_zero<0x112, 0xF>(v);
    v += dpp_or_zero<0x114, 0xF>(v);
    v += dpp_or_zero<0x118, 0xF>(v);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
    v = fmaxf(v, dpp_or_self<0x111, 0xF>(v));
    v = fmaxf(v, dpp_or_self<0x112, 0xF>(v));
    v = fmaxf(v, dpp_or_self<0x114, 0xF>(v));
    v = fmaxf(v, dpp_or_self<0x118, 0xF>(v));
    v = fmaxf(v, dpp_or_self<0x142, 0xA>(v));
    v = fmaxf(v, dpp_or_self<0x143, 0xC>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#endif


// ------------------------------------------------------------------ common
__device__ __forceinline__ unsigned long long lanemask_lt()
{
    return (1ull << lane_id()) - 1ull;
}

// Philox4x32-10 (Salmon et al., Random123); identical to oracle/sampler_oracle.c
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// bf16 <-> f32 bit helpers: round to nearest even, the same integer formula in the kernels, in the emulator and
// in oracle/gin_wide.py (finite inputs only)
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float x)
{
    uint32_t u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
// min(max(x, lo), hi) for lo <= hi (one v_med3_f32 on gfx950)
__device__ __forceinline__ float clamp_f32(float x, float lo, float hi)
{
#ifdef GCC_AMD_HIPEMU
    return fminf(fmaxf(x, lo), hi);
#else
    return __builtin_amdgcn_fmed3f(x, lo, hi);
#endif
}
// acc + both bf16 halves of a word (v_dot2c_f32_bf16 with a pair of ones on gfx950: 1 instruction instead of 4)
__device__ __forceinline__ float add2_bf16(uint32_t pk, float acc)
{
#ifdef GCC_AMD_HIPEMU
    return acc + (bf16_bits_to_f32(pk & 0xFFFFu) + bf16_bits_to_f32(pk >> 16));
#else
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, pk), __builtin_bit_cast(bf16x2_t, 0x3F803F80u), acc, false);
#endif
}
// two floats -> two bf16 in one word (a in the low half), round to nearest even: one v_cvt_pk_bf16_f32 on gfx950 (the
// integer formula above is ~5 VALU instructions per value, which made the epilogues of the bf16 GIN products as long as
// their matrix instructions)
__device__ __forceinline__ uint32_t pack2_bf16(float a, float b)
{
#ifdef GCC_AMD_HIPEMU
    return f32_to_bf16_bits(a) | (f32_to_bf16_bits(b) << 16);
#else
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
#endif
}
