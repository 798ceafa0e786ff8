// tools/probes/grid_barrier_probe.hip -- stand-alone micro-benchmark (not part of libgcc_amd.so): what does a
// device-wide barrier between the phases of ONE resident kernel cost on MI355X (8 XCDs, one L2 each), next to the kernel
// boundary the GIN encoder pays today between its BatchNorm phases?
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/grid_barrier_probe tools/probes/grid_barrier_probe.hip
//   tools/probes/grid_barrier_probe
//
// A phase = every workgroup writes its own 64 x 64 f32 tile (16 KiB, what gin_mid_kernel stores), the barrier, then every
// workgroup reads a tile ANOTHER workgroup (another XCD) wrote in this phase and checks it -- so the numbers include what
// makes the data visible across the L2s (release: write back; acquire: invalidate), and a wrong barrier shows as errors.
//   flat      one arrival counter (monotonic), every workgroup's thread 0: fence, atomic add, spin, fence
//   tree      eight counters on separate cache lines (workgroup % 8 = the XCD the dispatcher places it on), the last
//             arrival of each adds to a root counter; everybody spins on the root
//   elect     as tree, but only the LAST arrival of an XCD pays the agent-scope release (one L2 write-back per XCD instead of
//             one per workgroup): the others wait for their own stores (s_waitcnt) and release at workgroup scope; every
//             workgroup still invalidates on the way out.  Relies on blockIdx % 8 == XCD (checked: `xcd mismatches`)
//   launches  the same phases as separate kernel launches on one stream (the kernel boundary IS the barrier)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kThreads = 256, kTileFloats = 64 * 64;

struct Bar { unsigned *flat; unsigned *leaf; unsigned *root; };   // leaf: 8 counters, 64 B apart

__device__ __forceinline__ unsigned ld_agent(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int kKind>   // 0 flat, 1 tree, 2 elect
__device__ __forceinline__ void grid_barrier(const Bar &b, unsigned phase, unsigned nwg)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        if (kKind == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the tile's stores have left this CU
        else __atomic_thread_fence(__ATOMIC_RELEASE);                 // agent scope by default in HIP: buffer_wbl2 sc1
        if (kKind == 0) {
            __hip_atomic_fetch_add(b.flat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (phase + 1) * nwg;
            while (ld_agent(b.flat) < want) __builtin_amdgcn_s_sleep(1);
        } else {
            const unsigned x = blockIdx.x & 7, per = (nwg + 7 - x) / 8;   // workgroups with blockIdx % 8 == x
            const unsigned old = __hip_atomic_fetch_add(b.leaf + 16 * x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == (phase + 1) * per) {
                if (kKind == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // this XCD's L2, once
                __hip_atomic_fetch_add(b.root, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const unsigned want = (phase + 1) * 8;
            while (ld_agent(b.root) < want) __builtin_amdgcn_s_sleep(1);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);                      // buffer_inv sc1
    }
    __syncthreads();
}

__device__ __forceinline__ void write_tile(float *buf, int wg, unsigned phase)
{
    float4 v = make_float4((float)wg, (float)phase, (float)threadIdx.x, 1.f);
    float4 *dst = (float4 *)(buf + (size_t)wg * kTileFloats);
    for (int i = threadIdx.x; i < kTileFloats / 4; i += kThreads) dst[i] = v;
}
__device__ __forceinline__ int check_tile(const float *buf, int wg, unsigned phase)
{
    const float4 *src = (const float4 *)(buf + (size_t)wg * kTileFloats);
    int bad = 0;
    for (int i = threadIdx.x; i < kTileFloats / 4; i += kThreads) {
        const float4 v = src[i];
        bad += !(v.x == (float)wg && v.y == (float)phase);
    }
    return bad;
}

template <int kKind>
__global__ __launch_bounds__(kThreads) void persistent(float *buf0, float *buf1, Bar b, int phases, int *errors, long long *cycles)
{
    const int wg = blockIdx.x, nwg = gridDim.x;
    const long long t0 = wall_clock64();
    int bad = 0;
    if (threadIdx.x == 0) {   // HW_REG_XCC_ID (id 20), bits 3:0
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        if (xcc != (blockIdx.x & 7u)) atomicAdd(errors + 1, 1);
    }
    for (int p = 0; p < phases; ++p) {
        float *buf = (p & 1) ? buf1 : buf0;
        write_tile(buf, wg, p);
        grid_barrier<kKind>(b, p, nwg);
        bad += check_tile(buf, (wg + 3 + 8 * 5) % nwg, p);            // another XCD's tile (blockIdx % 8 differs)
    }
    if (bad) atomicAdd(errors, bad);
    if (threadIdx.x == 0 && wg == 0) cycles[0] = wall_clock64() - t0;
}

__global__ __launch_bounds__(kThreads) void one_phase(float *buf, const float *prev, int p, int *errors)
{
    const int wg = blockIdx.x, nwg = gridDim.x;
    if (prev) { const int bad = check_tile(prev, (wg + 3 + 8 * 5) % nwg, p - 1); if (bad) atomicAdd(errors, bad); }
    write_tile(buf, wg, p);
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("%s, %d CUs, wall clock %d kHz\n", prop.name, prop.multiProcessorCount, 100000);
    const int phases = 64;
    float *buf0, *buf1;
    unsigned *bar;
    int *errors;
    long long *cycles;
    CHECK(hipMalloc(&buf0, (size_t)2048 * kTileFloats * 4));
    CHECK(hipMalloc(&buf1, (size_t)2048 * kTileFloats * 4));
    CHECK(hipMalloc(&bar, 4096));
    CHECK(hipMalloc(&errors, 8));
    CHECK(hipMalloc(&cycles, 8));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int nwg : {256, 512, 768, 1024}) {
        for (int kind = 0; kind < 4; ++kind) {
            float best = 1e30f;
            int err_h[2] = {0, 0};
            for (int rep = 0; rep < 5; ++rep) {
                CHECK(hipMemsetAsync(bar, 0, 4096, s));
                CHECK(hipMemsetAsync(errors, 0, 8, s));
                CHECK(hipStreamSynchronize(s));
                Bar b = {bar, bar + 64, bar + 512};
                CHECK(hipEventRecord(e0, s));
                if (kind == 0) hipLaunchKernelGGL(persistent<0>, dim3(nwg), dim3(kThreads), 0, s, buf0, buf1, b, phases, errors, cycles);
                else if (kind == 1) hipLaunchKernelGGL(persistent<1>, dim3(nwg), dim3(kThreads), 0, s, buf0, buf1, b, phases, errors, cycles);
                else if (kind == 2) hipLaunchKernelGGL(persistent<2>, dim3(nwg), dim3(kThreads), 0, s, buf0, buf1, b, phases, errors, cycles);
                else
                    for (int p = 0; p < phases; ++p)
                        hipLaunchKernelGGL(one_phase, dim3(nwg), dim3(kThreads), 0, s, (p & 1) ? buf1 : buf0,
                                           p ? ((p & 1) ? buf0 : buf1) : (const float *)nullptr, p, errors);
                CHECK(hipEventRecord(e1, s));
                CHECK(hipStreamSynchronize(s));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
                CHECK(hipMemcpy(err_h, errors, 8, hipMemcpyDeviceToHost));
            }
            printf("workgroups %4d  %-8s  %7.2f us per phase (best of 5, %d phases)  errors %d  xcd mismatches %d\n", nwg,
                   kind == 0 ? "flat" : kind == 1 ? "tree" : kind == 2 ? "elect" : "launches", best * 1e3f / phases, phases, err_h[0], err_h[1]);
        }
    }
    return 0;
}
