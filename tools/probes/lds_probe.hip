// tools/probes/lds_probe.hip -- stand-alone micro-benchmark (not part of libgcc_amd.so): LDS cycles per wave
// instruction for the access patterns of gcc_amd/csrc/gin_wide.hip, as a function of the row stride.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_probe tools/probes/lds_probe.hip && /tmp/lds_probe
//
// Why: profiles/r1_pmc_gin_wide.json shows as many bank-conflict cycles as useful LDS cycles although the strides are
// conflict-free by the ds_read_b128 bank model of MI355X_MICROARCH.md, and padding 32 instead of 16 bytes measured
// the same.  This prints, per pattern and stride, the LDS-bound cycles per wave instruction with all
// eight waves of one workgroup per CU streaming (4 = 256 B/clk for 16-byte reads); pick the layout from the table.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 512, kIters = 256, kUnroll = 8;
constexpr int kLdsBytes = 150 * 1024;

enum Pattern {
    kReadFragment = 0,   // MFMA operand fragment: lane -> row (lane & 15), 16 bytes at column group (lane >> 4); +64 B per k-step
    kReadLinear = 1,     // lane -> 16 consecutive bytes (the conflict-free reference)
    kWriteRows8 = 2,     // epilogue of the swapped products: 8 bytes at row (lane & 15), column (lane >> 4) * 8 bytes
    kWriteCols8 = 3,     // epilogue of the channel-major result: row = lane & 15 (channel), 8 bytes at (lane >> 4) * 8
    kReadFragmentXor = 4 // as 0 with the 16-byte column group XOR-swizzled by (row >> 1) & 3 (stride must be a multiple of 64)
};

__global__ __launch_bounds__(kThreads) void probe(int pattern, int stride, long long *cycles, unsigned *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < kLdsBytes / 4; i += kThreads) ((unsigned *)lds)[i] = i * 2654435761u;
    __syncthreads();
    const int lr = lane & 15, lg = lane >> 4;
    // every wave works on its own 16 rows (as the waves of a product do)
    int base;
    if (pattern == kReadLinear) base = w * 1024 + lane * 16;
    else if (pattern == kReadFragmentXor) base = (w * 16 + lr) * stride + ((lg ^ ((lr >> 1) & 3)) * 16);
    else if (pattern == kReadFragment) base = (w * 16 + lr) * stride + lg * 16;
    else base = (w * 16 + lr) * stride + lg * 8;
    base %= (kLdsBytes - 32 * 1024);
    base &= ~7;
    u32x4 acc = {0u, 0u, 0u, 0u};
    __syncthreads();
    const long long t0 = clock64();
    // (the offset passes through an empty asm every iteration: the optimiser can neither hoist the loads out of the
    // loop nor merge the stores; the accesses stay plain LDS instructions with immediate offsets)
    if (pattern == kWriteRows8 || pattern == kWriteCols8) {
        u32x2 v = {(unsigned)lane, (unsigned)w};
        for (int it = 0; it < kIters; ++it) {
            int off = base;
            asm volatile("" : "+v"(off));
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) *(u32x2 *)(lds + off + u * 4224) = v;   // (too far apart for ds_write2)
            v[0] += 1u;
        }
    } else {
        for (int it = 0; it < kIters; ++it) {
            int off = base;
            asm volatile("" : "+v"(off));
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) acc ^= *(const u32x4 *)(lds + off + u * 64);
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc[0] == 0x12345678u && acc[1] == 1u) sink[0] = acc[2] ^ acc[3];     // keeps the loads alive
}

int main()
{
    int dev = 0, cus = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 1; }
    cus = prop.multiProcessorCount;
    long long *cycles;
    unsigned *sink;
    if (hipMalloc(&cycles, sizeof(long long) * cus) != hipSuccess || hipMalloc(&sink, sizeof(unsigned)) != hipSuccess) return 1;
    (void)hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    const char *names[] = {"read16 fragment", "read16 linear", "write8 rows", "write8 cols", "read16 fragment xor"};
    const int strides[] = {512, 528, 544, 560, 576, 640, 256, 272, 288, 304, 320};
    printf("%s on %d CUs: LDS-bound shader cycles per wave instruction, 8 waves of one 512-thread workgroup per CU\n", prop.name, cus);
    printf("%-22s %8s %10s\n", "pattern", "stride", "cyc/instr");
    std::vector<long long> host(cus);
    for (int p = 0; p < 5; ++p)
        for (int stride : strides) {
            if (p == kReadLinear && stride != 512) continue;
            if (p == kReadFragmentXor && stride % 64) continue;
            for (int rep = 0; rep < 2; ++rep)            // the first launch warms up
                hipLaunchKernelGGL(probe, dim3(cus), dim3(kThreads), kLdsBytes, 0, p, stride, cycles, sink);
            if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "launch failed\n"); return 1; }
            (void)hipMemcpy(host.data(), cycles, sizeof(long long) * cus, hipMemcpyDeviceToHost);
            double mean = 0;
            for (long long c : host) mean += (double)c;
            mean /= cus;
            printf("%-22s %8d %10.2f\n", names[p], stride, mean / (double)((kThreads / 64) * kIters * kUnroll));
        }
    (void)hipFree(cycles);
    (void)hipFree(sink);
    return 0;
}
