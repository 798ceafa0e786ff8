// gcc_amd/csrc/gin_wide.hip -- wide (hidden 256) GIN layers in bf16 on the matrix cores, one subgraph per
// workgroup, resident in LDS across all layers: BASELINE.json configs[4] ("GIN hid=256 layers=8 deg=32 bf16,
// SpMM+MFMA-MLP roofline run on batched subgraphs").  Reference: UnsupervisedGIN.forward gcc/models/gin.py:213-221
// with the modules of gin.py:42-58,107-116 in eval mode (generate.py:71); see include/gcc_amd.h.
//
// A subgraph has at most 128 nodes, so its adjacency (plus the identity: GINConv adds h_v itself, eps = 0) is a
// dense 128x128 block and all three products of a layer run on v_mfma_f32_16x16x32_bf16 with every operand read
// k-contiguous and every result stored 8 bytes at a time:
//     AGG [node][ch]  = sum_u  H^T[ch][u]  * ADJ[node][u]      A = H^T rows (LDS),  B = ADJ rows (16 registers, kept for all layers)
//     Z1  [node][ch]  = sum_k  W0[ch][k]   * AGG[node][k]      A = W0 rows (L2),    B = AGG rows (LDS)      + scale/shift/ReLU
//     H'^T[ch][node]  = sum_k  Z1[node][k] * W1[ch][k]         A = Z1 rows (LDS),   B = W1 rows (L2)        + 2x scale/shift/ReLU
// (D[row][col] = sum_k A[row][k] B[col][k]; a lane ends with 4 consecutive ROWS of one column, so the operand
// order of each product is chosen to make those 4 values contiguous in the layout the next product reads.)
// The two LDS regions swap roles: H^T -> AGG in the other -> Z1 over H^T -> H'^T over AGG.
#include "host_common.h"

#include <stdlib.h>

namespace {

constexpr int kD = GCC_GINW_HIDDEN;
constexpr int kNodes = GCC_GINW_MAX_NODES;
constexpr int kThreads = 512;              // 8 waves, 2 per SIMD: up to 256 VGPRs each for 64x64 register tiles
constexpr int kStrideT = kNodes * 2 + 16;   // bytes per row of the channel-major layout [256 ch][128 nodes] (+16: bank spread)
constexpr int kStrideN = kD * 2 + 16;       // bytes per row of the node-major layout    [128 nodes][256 ch]
constexpr int kRegion = kD * kStrideT;      // 69,632 B >= kNodes * kStrideN
constexpr int kAhead = 1;                   // k-steps the LDS reads of the Linear products run ahead of the matrix instructions
constexpr int kLdsBytes = 2 * kRegion + (kNodes + 1 + 3) / 4 * 16;
static_assert(kNodes * kStrideN <= kRegion, "node-major layout must fit a region");
static_assert(kD == 256 && kNodes == 128, "the wave tilings below are written for 256 channels x 128 nodes");

struct WideArgs {
    const int32_t *node_off, *row_ptr, *col_idx;
    const uint16_t *x_in;
    uint16_t *x_out;
    float *pooled;
    int32_t *status;
    int32_t batch_size, num_layers;
    long long *ticks;                // diagnostics (gcc_ginw_debug_ticks): wall-clock ticks per phase, or NULL
    gcc_ginw_layer layers[GCC_GIN_MAX_LAYERS];
    // subgraphs over kNodes nodes (gin_wide_big_kernel): two [N, 256] bf16 ping-pong buffers and a work list
    // {count, -, (subgraph, row block) pairs}, or NULL: such subgraphs are refused (GCC_STATUS_GINW_TOO_LARGE)
    uint16_t *big0, *big1;
    int32_t *big_work;
    int32_t big_cap;                 // pairs the work list holds
};

long long *g_ticks = nullptr;

__device__ __forceinline__ void phase_tick(long long *row, int ph, long long &tick)
{
    if (row && threadIdx.x == 0) {
        const long long now = device_ticks();
        atomicAdd((unsigned long long *)&row[ph], (unsigned long long)(now - tick));
        tick = now;
    }
}

__device__ __forceinline__ u32x4 lds16(const unsigned char *p) { return *(const u32x4 *)p; }

__device__ __forceinline__ u32x2 pack4_bf16(float a, float b, float c, float d)
{
    u32x2 r;
    r[0] = pack2_bf16(a, b);
    r[1] = pack2_bf16(c, d);
    return r;
}
__device__ __forceinline__ float sum4_bf16(u32x2 v)
{
    return (bf16_bits_to_f32(v[0] & 0xFFFFu) + bf16_bits_to_f32(v[0] >> 16))
         + (bf16_bits_to_f32(v[1] & 0xFFFFu) + bf16_bits_to_f32(v[1] >> 16));
}

// The 16 weight fragments a wave needs for one Linear layer (its 32 output channels x all 256 inputs) come in two
// requests so that L2's latency hides behind arithmetic: k-steps 0..3 are requested before the last pass of the product
// before (into `early`, moved to wf[.][0..3] when that product's own fragments are dead), k-steps 4..7 right after it.
__device__ __forceinline__ void request_early(u32x4 (&early)[2][4], const uint16_t *wmat, int w, int lr, int lg)
{
    const uint16_t *wp = wmat + (int64_t)(w * 32 + lr) * kD + lg * 8;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) early[m][ks] = *(const u32x4 *)(wp + m * 16 * kD + ks * 32);
}
__device__ __forceinline__ void request_late(u32x4 (&wf)[2][8], const uint16_t *wmat, int w, int lr, int lg)
{
    const uint16_t *wp = wmat + (int64_t)(w * 32 + lr) * kD + lg * 8;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ks = 4; ks < 8; ++ks) wf[m][ks] = *(const u32x4 *)(wp + m * 16 * kD + ks * 32);
}
__device__ __forceinline__ void adopt_early(u32x4 (&wf)[2][8], const u32x4 (&early)[2][4])
{
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf[m][ks] = early[m][ks];
}

// Z1[node][ch] = relu(s0 * (AGG W0^T) + t0) for the wave's 32 channels and the first 16 * kNB nodes.  No branches
// inside: the LDS reads of the whole product are scheduled ahead of the matrix instructions that consume them.
// Rows of padding nodes hold finite leftovers or anything at all; every result depends on its own node's row only.
template <int kNB, bool kLastPass>
__device__ __forceinline__ void linear0_tile(const unsigned char *Q, unsigned char *P, u32x4 (&wf)[2][8], const gcc_ginw_layer &ly,
                                             const uint16_t *wnext /* last pass: the next product's matrix */, int f0, int w, int lr, int lg)
{
    u32x4 early[2][4];
    if (kLastPass) request_early(early, wnext, w, lr, lg);
    SCHED_FENCE();
    Q += f0 * 16 * kStrideN;                     // node fragments f0 .. f0 + kNB - 1
    P += f0 * 16 * kStrideN;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[2][kNB];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nb = 0; nb < kNB; ++nb) acc[m][nb] = zero4;
    // software pipeline: the LDS reads of k-step ks + kAhead are issued before the matrix instructions of step ks (the
    // fences keep the scheduler from hoisting every read of the product to the top, which spills, and from sinking them)
    const unsigned char *src = Q + lr * kStrideN + lg * 16;
    u32x4 ring[kAhead + 1][kNB];
#pragma unroll
    for (int j = 0; j < kAhead; ++j)
#pragma unroll
        for (int nb = 0; nb < kNB; ++nb) ring[j][nb] = lds16(src + nb * 16 * kStrideN + j * 64);
#pragma unroll
    for (int ks = 0; ks < kD / 32; ++ks) {
        if (ks + kAhead < kD / 32) {
#pragma unroll
            for (int nb = 0; nb < kNB; ++nb) ring[(ks + kAhead) % (kAhead + 1)][nb] = lds16(src + nb * 16 * kStrideN + (ks + kAhead) * 64);
        }
        SCHED_FENCE();
#pragma unroll
        for (int nb = 0; nb < kNB; ++nb) {
            acc[0][nb] = mfma_16x16x32_bf16(wf[0][ks], ring[ks % (kAhead + 1)][nb], acc[0][nb]);
            acc[1][nb] = mfma_16x16x32_bf16(wf[1][ks], ring[ks % (kAhead + 1)][nb], acc[1][nb]);
        }
        SCHED_FENCE();
    }
    SCHED_FENCE();                               // (not earlier: the fragments of this product are still in use)
    if (kLastPass) {
        adopt_early(wf, early);
        request_late(wf, wnext, w, lr, lg);      // in flight during the epilogue, the barrier and the first four k-steps
    }
    SCHED_FENCE();
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int c = w * 32 + m * 16 + lg * 4;
        const float4 s = *(const float4 *)(ly.s0 + c), t = *(const float4 *)(ly.t0 + c);
#pragma unroll
        for (int nb = 0; nb < kNB; ++nb)
            *(u32x2 *)(P + (nb * 16 + lr) * kStrideN + c * 2) =
                pack4_bf16(fmaxf(fmaf(acc[m][nb][0], s.x, t.x), 0.f), fmaxf(fmaf(acc[m][nb][1], s.y, t.y), 0.f),
                           fmaxf(fmaf(acc[m][nb][2], s.z, t.z), 0.f), fmaxf(fmaf(acc[m][nb][3], s.w, t.w), 0.f));
    }
}

// H'^T[ch][node] = relu(s2 * relu(s1 * (Z1 W1^T) + t1) + t2) for the wave's 32 channels; nodes >= n are written as 0
// (the aggregation multiplies them by ADJ's zeros, which only works for finite values); SumPooling of the result.
template <int kNB, bool kLastPass>
__device__ __forceinline__ void linear1_tile(const unsigned char *P, unsigned char *Q, u32x4 (&wf)[2][8], const gcc_ginw_layer &ly,
                                             const uint16_t *wnext /* last pass: the next layer's first matrix */, float (&pool_part)[2],
                                             int n, int f0, int w, int lr, int lg)
{
    u32x4 early[2][4];
    if (kLastPass) request_early(early, wnext, w, lr, lg);
    SCHED_FENCE();
    P += f0 * 16 * kStrideN;                     // node fragments f0 .. f0 + kNB - 1
    Q += f0 * 32;
    n -= f0 * 16;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[kNB][2];
#pragma unroll
    for (int m = 0; m < kNB; ++m) { acc[m][0] = zero4; acc[m][1] = zero4; }
    const unsigned char *src = P + lr * kStrideN + lg * 16;          // pipelined as in linear0_tile
    u32x4 ring[kAhead + 1][kNB];
#pragma unroll
    for (int j = 0; j < kAhead; ++j)
#pragma unroll
        for (int m = 0; m < kNB; ++m) ring[j][m] = lds16(src + m * 16 * kStrideN + j * 64);
#pragma unroll
    for (int ks = 0; ks < kD / 32; ++ks) {
        if (ks + kAhead < kD / 32) {
#pragma unroll
            for (int m = 0; m < kNB; ++m) ring[(ks + kAhead) % (kAhead + 1)][m] = lds16(src + m * 16 * kStrideN + (ks + kAhead) * 64);
        }
        SCHED_FENCE();
#pragma unroll
        for (int m = 0; m < kNB; ++m) {
            acc[m][0] = mfma_16x16x32_bf16(ring[ks % (kAhead + 1)][m], wf[0][ks], acc[m][0]);
            acc[m][1] = mfma_16x16x32_bf16(ring[ks % (kAhead + 1)][m], wf[1][ks], acc[m][1]);
        }
        SCHED_FENCE();
    }
    SCHED_FENCE();
    if (kLastPass) adopt_early(wf, early);       // (k-steps 4..7 follow after the aggregation, which needs the registers)
    SCHED_FENCE();
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int c = w * 32 + nb * 16 + lr;
        const float s1 = ly.s1[c], t1 = ly.t1[c], s2 = ly.s2[c], t2 = ly.t2[c];
        float psum = 0.f;
#pragma unroll
        for (int m = 0; m < kNB; ++m) {
            const int node = m * 16 + lg * 4;
            float h[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y = fmaxf(fmaf(acc[m][nb][r], s1, t1), 0.f);            // apply_func: relu(bn(mlp))
                h[r] = node + r < n ? fmaxf(fmaf(y, s2, t2), 0.f) : 0.f;            // relu(batch_norms[i](.)); padding nodes stay 0
            }
            const u32x2 hv = pack4_bf16(h[0], h[1], h[2], h[3]);
            psum += sum4_bf16(hv);
            *(u32x2 *)(Q + c * kStrideT + node * 2) = hv;
        }
        psum += wave_shfl_xor(psum, 16);
        psum += wave_shfl_xor(psum, 32);
        pool_part[nb] += psum;
    }
}

__global__ __launch_bounds__(kThreads) void gin_wide_kernel(WideArgs a)
{
    DYN_SMEM(smem);
    unsigned char *P = smem, *Q = smem + kRegion;
    int32_t *rp = (int32_t *)(smem + 2 * kRegion);           // [129] row pointers of the subgraph
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int L = a.num_layers;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    for (int b = blockIdx.x; b < a.batch_size; b += gridDim.x) {
        __syncthreads();                                     // the previous subgraph's output pass is done with P
        long long tick = a.ticks ? device_ticks() : 0;
        const int n0 = a.node_off[b], n = a.node_off[b + 1] - n0;
        if (n <= 0 || n > kNodes) {                          // (uniform over the workgroup)
            if (n > kNodes && !a.big_work) {                 // (with scratch: gin_wide_big_kernel takes it, block by block)
                if (tid == 0) atomicOr(a.status, (int32_t)GCC_STATUS_GINW_TOO_LARGE);
                if (a.x_out)
                    for (int64_t i = tid; i < (int64_t)n * (kD / 2); i += kThreads) ((uint32_t *)(a.x_out + (int64_t)n0 * kD))[i] = 0u;
            }
            if (a.pooled)
                for (int i = tid; i < (L + 1) * kD; i += kThreads) a.pooled[(int64_t)b * (L + 1) * kD + i] = 0.f;
            continue;
        }
        // ---- the subgraph's input rows -> P, channel-major; neighbour counts -> Q (16-bit counters, [node][u])
        for (int i = tid; i < kNodes * kStrideT / 4; i += kThreads) ((uint32_t *)Q)[i] = 0u;
        if (tid <= n) rp[tid] = a.row_ptr[n0 + tid];
        for (int idx = tid; idx < kNodes * (kD / 8); idx += kThreads) {
            const int node = idx & (kNodes - 1), chunk = idx >> 7;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (node < n) v = *(const u32x4 *)(a.x_in + (int64_t)(n0 + node) * kD + chunk * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                *(uint16_t *)(P + (chunk * 8 + e) * kStrideT + node * 2) = (uint16_t)(v[e >> 1] >> ((e & 1) * 16));
        }
        __syncthreads();
        phase_tick(a.ticks, 0, tick);                        // rows in
        {   // every edge of the subgraph by one thread (loads of 8 edges per thread in flight at once); its row by
            // bisection of the row pointers staged by the pass above
            const int e0 = rp[0], e1 = rp[n];
            for (int e = e0 + tid; e < e1; e += kThreads) {
                const int u = a.col_idx[e] - n0;
                int lo = 0, hi = n;                          // largest i with rp[i] <= e
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (rp[mid] <= e) lo = mid; else hi = mid;
                }
                if ((unsigned)u < (unsigned)n) atomicAdd((uint32_t *)(Q + lo * kStrideT + (u >> 1) * 4), (u & 1) ? 0x10000u : 1u);
                else atomicOr(a.status, (int32_t)GCC_STATUS_GINW_BAD_EDGE);
            }
            if (tid < n) atomicAdd((uint32_t *)(Q + tid * kStrideT + (tid >> 1) * 4), (tid & 1) ? 0x10000u : 1u);   // + h_v itself
        }
        __syncthreads();
        phase_tick(a.ticks, 1, tick);                        // neighbour counts
        if (a.pooled && tid < kD) {                          // hidden_rep[0] = the input (gin.py:216)
            float s = 0.f;
            for (int j = 0; j < kNodes / 8; ++j) {
                const u32x4 v = lds16(P + tid * kStrideT + j * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) s += bf16_bits_to_f32(v[q] & 0xFFFFu) + bf16_bits_to_f32(v[q] >> 16);
            }
            a.pooled[((int64_t)b * (L + 1)) * kD + tid] = s;
        }
        // this wave's share of ADJ (32 nodes x all neighbours) as B fragments, kept in registers for every layer
        const int nblk = w & 3, chalf = w >> 2;              // aggregation: 4 node blocks of 32 x 2 channel halves of 128
        u32x4 adj[2][4];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u32x4 c = lds16(Q + (nblk * 32 + nb * 16 + lr) * kStrideT + (ks * 32 + lg * 8) * 2);
                u32x4 f;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    f[q] = pack2_bf16((float)(c[q] & 0xFFFFu), (float)(c[q] >> 16));
                adj[nb][ks] = f;
            }
        __syncthreads();
        phase_tick(a.ticks, 2, tick);                        // input pooling, adjacency fragments
        const int nfrag = (n + 15) >> 4;
        u32x4 wf[2][8];
        {
            u32x4 early[2][4];
            request_early(early, a.layers[0].w0, w, lr, lg);
            adopt_early(wf, early);
        }

        for (int layer = 0; layer < L; ++layer) {
            const gcc_ginw_layer ly = a.layers[layer];
            // ---- AGG[node][ch] -> Q (node-major); wave w: 32 nodes x 128 channels (every LDS fragment feeds two matrix
            // instructions); afterwards the first Linear's weight fragments are requested: they travel during the epilogue
            // and the barrier (an LDS-only barrier: __syncthreads() would wait for them)
            {
                // (no branches on the subgraph's size in here -- data-dependent branches around blocks of matrix instructions
                // make the register allocator spill hundreds of values; rows and columns of padding nodes are zeros anyway)
                f32x4 acc[8][2];
#pragma unroll
                for (int m = 0; m < 8; ++m) { acc[m][0] = zero4; acc[m][1] = zero4; }
                const unsigned char *src = P + (chalf * 128 + lr) * kStrideT + lg * 16;
                u32x4 ring[3][4];                            // half k-steps (4 channel blocks each), two ahead
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int m = 0; m < 4; ++m) ring[j][m] = lds16(src + ((j & 1) * 4 + m) * 16 * kStrideT + (j >> 1) * 64);
#pragma unroll
                for (int step = 0; step < 8; ++step) {       // step = 2 * ks + half
                    if (step + 2 < 8) {
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            ring[(step + 2) % 3][m] = lds16(src + (((step + 2) & 1) * 4 + m) * 16 * kStrideT + ((step + 2) >> 1) * 64);
                    }
                    SCHED_FENCE();
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        acc[(step & 1) * 4 + m][0] = mfma_16x16x32_bf16(ring[step % 3][m], adj[0][step >> 1], acc[(step & 1) * 4 + m][0]);
                        acc[(step & 1) * 4 + m][1] = mfma_16x16x32_bf16(ring[step % 3][m], adj[1][step >> 1], acc[(step & 1) * 4 + m][1]);
                    }
                    SCHED_FENCE();
                }
                request_late(wf, ly.w0, w, lr, lg);
                SCHED_FENCE();
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        *(u32x2 *)(Q + (nblk * 32 + nb * 16 + lr) * kStrideN + (chalf * 128 + m * 16 + lg * 4) * 2) =
                            pack4_bf16(acc[m][nb][0], acc[m][nb][1], acc[m][nb][2], acc[m][nb][3]);
            }
            lds_barrier();
            phase_tick(a.ticks, 3, tick);                    // aggregation
            // ---- Z1[node][ch] = relu(s0 * (AGG W0^T) + t0) -> P (node-major); wave w: channels 32w .. 32w+31, all nodes
            // (two passes over at most 4 node fragments each: 32 accumulator registers at a time, the weights stay)
            if (nfrag > 4) {
                linear0_tile<4, false>(Q, P, wf, ly, nullptr, 0, w, lr, lg);
                if (nfrag > 6) linear0_tile<4, true>(Q, P, wf, ly, ly.w1, 4, w, lr, lg);
                else linear0_tile<2, true>(Q, P, wf, ly, ly.w1, 4, w, lr, lg);
            } else if (nfrag > 2) linear0_tile<4, true>(Q, P, wf, ly, ly.w1, 0, w, lr, lg);
            else linear0_tile<2, true>(Q, P, wf, ly, ly.w1, 0, w, lr, lg);
            lds_barrier();
            phase_tick(a.ticks, 4, tick);                    // first Linear
            // ---- H'^T[ch][node] = relu(s2 * relu(s1 * (Z1 W1^T) + t1) + t2) -> Q (channel-major); wave w: channels 32w .. 32w+31, all nodes
            {
                // (after the last layer the request is a dummy: no branch around it, see the note on branches below)
                const uint16_t *wnext = a.layers[layer + 1 < L ? layer + 1 : layer].w0;
                float pool_part[2] = {0.f, 0.f};
                if (nfrag > 4) {
                    linear1_tile<4, false>(P, Q, wf, ly, nullptr, pool_part, n, 0, w, lr, lg);
                    if (nfrag > 6) linear1_tile<4, true>(P, Q, wf, ly, wnext, pool_part, n, 4, w, lr, lg);
                    else linear1_tile<2, true>(P, Q, wf, ly, wnext, pool_part, n, 4, w, lr, lg);
                } else if (nfrag > 2) linear1_tile<4, true>(P, Q, wf, ly, wnext, pool_part, n, 0, w, lr, lg);
                else linear1_tile<2, true>(P, Q, wf, ly, wnext, pool_part, n, 0, w, lr, lg);
                // node fragments past the last one computed: zeros (read by the next aggregation as multiplicands of 0)
                const int fdone = nfrag > 6 ? 8 : nfrag > 4 ? 6 : nfrag > 2 ? 4 : 2;
                for (int i = lane; i < 32 * (8 - fdone) * 4; i += 64) {
                    const int c = w * 32 + i / ((8 - fdone) * 4), j = i % ((8 - fdone) * 4);
                    *(u32x2 *)(Q + c * kStrideT + fdone * 32 + j * 8) = u32x2{0u, 0u};
                }
                if (lg == 0 && a.pooled) {
                    float *pool = a.pooled + ((int64_t)b * (L + 1) + layer + 1) * kD + w * 32 + lr;
                    pool[0] = pool_part[0];
                    pool[16] = pool_part[1];
                }
            }
            lds_barrier();
            phase_tick(a.ticks, 5, tick);                    // second Linear
            unsigned char *t = P; P = Q; Q = t;
        }
        // ---- the last layer's rows back to node-major global memory
        if (a.x_out)
            for (int idx = tid; idx < kNodes * (kD / 8); idx += kThreads) {
                const int node = idx & (kNodes - 1), chunk = idx >> 7;
                if (node < n) {
                    u32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        v[q] = (uint32_t)*(const uint16_t *)(P + (chunk * 8 + 2 * q) * kStrideT + node * 2)
                             | ((uint32_t)*(const uint16_t *)(P + (chunk * 8 + 2 * q + 1) * kStrideT + node * 2) << 16);
                    *(u32x4 *)(a.x_out + (int64_t)(n0 + node) * kD + chunk * 8) = v;
                }
            }
        phase_tick(a.ticks, 6, tick);                        // rows out
        if (a.ticks && tid == 0) atomicAdd((unsigned long long *)&a.ticks[15], 1ull);
    }
}

// =====================================================================================================================
// Second kernel shape (default): 256 threads = ONE wave per SIMD with up to 512 registers each, so that a wave's register
// tile is 64 channels x 128 nodes (Linear products) or 128 channels x 64 nodes (aggregation): every operand fragment read
// from LDS feeds FOUR matrix instructions (two in the first kernel, whose products were bound by fragment reads), the
// weight fragments stream through a 4-deep register ring straight from L2 (each read once per workgroup, requested four
// k-steps = ~2000 cycles ahead across product boundaries), the adjacency is 16 fragments in registers for all layers, and
// results are stored 16 bytes per lane: the rows of two adjacent A fragments are interleaved (row r of fragment e <->
// index 2 r + e of a 32-block), so a lane's 4 + 4 accumulator rows are 8 consecutive channels (or nodes).  Row strides per
// layout, from the LDS lane-group model of MI355X_MICROARCH.md (a 16-byte fragment read is serviced in 4 groups of 16
// lanes): consecutive rows are conflict-free at 544 bytes, interleaved rows at 272 / 528 (blocks of four rows, the
// first try, are 2-way conflicted at every stride: 53 % of the LDS cycles in profiles/r3_pmc_gin_wide.json's first run).
constexpr int kT2 = 256;
constexpr int kStrT2 = kNodes * 2 + 16;      // H^T, channel-major [256 ch][128 nodes]: read with interleaved rows (aggregation)
constexpr int kStrN2 = kD * 2 + 32;          // AGG, node-major [128 nodes][256 ch]: read with consecutive rows (first Linear)
constexpr int kStrZ2 = kD * 2 + 16;          // Z1,  node-major: read with interleaved rows (second Linear)
constexpr int kReg2 = kD * kStrT2;           // 69,632 B = kNodes * kStrN2
constexpr int kLds2 = 2 * kReg2 + (kNodes + 1 + 3) / 4 * 16;
static_assert(kNodes * kStrN2 <= kReg2 && kNodes * kStrZ2 <= kReg2, "the node-major layouts must fit a region");

__device__ __forceinline__ int perm8(int lr, int e) { return 2 * lr + e; }
__device__ __forceinline__ u32x4 pack8_bf16(const f32x4 &a, const f32x4 &b)
{
    u32x4 r;
    r[0] = pack2_bf16(a[0], b[0]);               // fragment 0 holds the even, fragment 1 the odd indices
    r[1] = pack2_bf16(a[1], b[1]);
    r[2] = pack2_bf16(a[2], b[2]);
    r[3] = pack2_bf16(a[3], b[3]);
    return r;
}
__device__ __forceinline__ float sum8_bf16(u32x4 v, float s = 0.f)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) s = add2_bf16(v[q], s);
    return s;
}
// the 4 weight fragments of one k-step of a wave's 64 output channels.  kRowPerm: the fragments are A operands whose rows
// are interleaved in pairs (first Linear); else B operands, column lr = channel 16 m + lr (second Linear)
// kFrag: wmat is the fragment-major copy made by gcc_ginw_pack_weights (1 KiB contiguous per request)
template <bool kRowPerm, bool kFrag>
__device__ __forceinline__ void request_w(u32x4 (&dst)[4], const uint16_t *wmat, int w, int ks, int lr, int lg)
{
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (kFrag) {
            dst[m] = *(const u32x4 *)(wmat + (int64_t)((w * 4 + m) * 8 + ks) * 512 + (lg * 16 + lr) * 8);
        } else {
            const int ch = w * 64 + (kRowPerm ? (m >> 1) * 32 + perm8(lr, m & 1) : m * 16 + lr);
            dst[m] = *(const u32x4 *)(wmat + (int64_t)ch * kD + ks * 32 + lg * 8);
        }
    }
}

__global__ void ginw_pack_kernel(const uint16_t *w, uint16_t *wf, int which)
{
    const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);      // one 16-byte piece: ((w * 4 + m) * 8 + ks) * 64 + lane
    if (idx >= kD * kD / 8) return;
    const int lane = idx & 63, ks = (idx >> 6) & 7, m = (idx >> 9) & 3, wb = idx >> 11, lr = lane & 15, lg = lane >> 4;
    const int row = wb * 64 + (which == 0 ? (m >> 1) * 32 + perm8(lr, m & 1) : m * 16 + lr);
    *(u32x4 *)(wf + (int64_t)idx * 8) = *(const u32x4 *)(w + (int64_t)row * kD + ks * 32 + lg * 8);
}

// kDbg (timing experiments only, wrong results; builds with -DGCC_GINW_ABLATE select them by GCC_GINW_DBG): 1 no epilogue arithmetic (one store per product keeps the
// accumulators alive), 2 no weight requests inside the products, 4 no operand-fragment reads inside the k loops
// (measured: epilogues 25 % of the launch, requests 15 % row-major / 2 % fragment-major, fragment reads 2 %).
template <int kDbg, bool kFrag>
__global__ __launch_bounds__(kT2) void gin_wide2_kernel(WideArgs a)
{
    DYN_SMEM(smem);
    unsigned char *P = smem, *Q = smem + kReg2;
    int32_t *rp = (int32_t *)(smem + 2 * kReg2);             // [129] row pointers of the subgraph
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int L = a.num_layers;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int nh = w & 1, chh = w >> 1;                      // aggregation: node half x channel half

    for (int b = blockIdx.x; b < a.batch_size; b += gridDim.x) {
        __syncthreads();                                     // the previous subgraph's output pass is done with P
        long long tick = a.ticks ? device_ticks() : 0;
        const int n0 = a.node_off[b], n = a.node_off[b + 1] - n0;
        if (n <= 0 || n > kNodes) {                          // (uniform over the workgroup)
            if (n > kNodes && !a.big_work) {                 // (with scratch: gin_wide_big_kernel takes it, block by block)
                if (tid == 0) atomicOr(a.status, (int32_t)GCC_STATUS_GINW_TOO_LARGE);
                if (a.x_out)
                    for (int64_t i = tid; i < (int64_t)n * (kD / 2); i += kT2) ((uint32_t *)(a.x_out + (int64_t)n0 * kD))[i] = 0u;
            }
            if (a.pooled)
                for (int i = tid; i < (L + 1) * kD; i += kT2) a.pooled[(int64_t)b * (L + 1) * kD + i] = 0.f;
            continue;
        }
        // ---- the subgraph's input rows -> P, channel-major; neighbour counts -> Q (16-bit counters, [node][u])
        for (int i = tid; i < kNodes * kStrT2 / 4; i += kT2) ((uint32_t *)Q)[i] = 0u;
        if (tid <= n) rp[tid] = a.row_ptr[n0 + tid];
        // (4 nodes x 8 channels per work item: four 16-byte loads, eight 8-byte LDS writes; requesting all 16 loads of a
        //  thread's four work items first was measured too: no gain)
        for (int idx = tid; idx < (kNodes / 4) * (kD / 8); idx += kT2) {
            const int chunk = idx & (kD / 8 - 1), node = 4 * (idx >> 5);
            u32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 z = {0u, 0u, 0u, 0u};
                v[j] = node + j < n ? *(const u32x4 *)(a.x_in + (int64_t)(n0 + node + j) * kD + chunk * 8) : z;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int sh = (e & 1) * 16;
                u32x2 o;
                o[0] = ((v[0][e >> 1] >> sh) & 0xFFFFu) | (((v[1][e >> 1] >> sh) & 0xFFFFu) << 16);
                o[1] = ((v[2][e >> 1] >> sh) & 0xFFFFu) | (((v[3][e >> 1] >> sh) & 0xFFFFu) << 16);
                *(u32x2 *)(P + (chunk * 8 + e) * kStrT2 + node * 2) = o;
            }
        }
        __syncthreads();
        phase_tick(a.ticks, 0, tick);                        // rows in
        {
            // runs of 16 consecutive edges per thread: one bisection of the row pointers per run, then the row advances with
            // the edges (a bisection per edge was 9.8 us per subgraph)
            const int e0 = rp[0], e1 = rp[n];
            for (int eb = e0 + 16 * tid; eb < e1; eb += 16 * kT2) {
                int cols[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) cols[j] = eb + j < e1 ? a.col_idx[eb + j] : -1;
                int lo = 0, hi = n;                          // largest i with rp[i] <= eb
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (rp[mid] <= eb) lo = mid; else hi = mid;
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (eb + j >= e1) break;
                    while (rp[lo + 1] <= eb + j) ++lo;       // (empty rows are skipped)
                    const int u = cols[j] - n0;
                    if ((unsigned)u < (unsigned)n) atomicAdd((uint32_t *)(Q + lo * kStrT2 + (u >> 1) * 4), (u & 1) ? 0x10000u : 1u);
                    else atomicOr(a.status, (int32_t)GCC_STATUS_GINW_BAD_EDGE);
                }
            }
            if (tid < n) atomicAdd((uint32_t *)(Q + tid * kStrT2 + (tid >> 1) * 4), (tid & 1) ? 0x10000u : 1u);   // + h_v itself
        }
        __syncthreads();
        phase_tick(a.ticks, 1, tick);                        // neighbour counts
        if (a.pooled) {                                      // hidden_rep[0] = the input (gin.py:216)
            float s = 0.f;
            for (int j = 0; j < kNodes / 8; ++j) s += sum8_bf16(lds16(P + tid * kStrT2 + j * 16));
            a.pooled[((int64_t)b * (L + 1)) * kD + tid] = s;
        }
        // this wave's share of ADJ (64 nodes x all neighbours) as B fragments, kept in registers for every layer
        u32x4 adj[4][4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u32x4 c = lds16(Q + (nh * 64 + nf * 16 + lr) * kStrT2 + (ks * 32 + lg * 8) * 2);
                u32x4 f;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    f[q] = pack2_bf16((float)(c[q] & 0xFFFFu), (float)(c[q] >> 16));
                adj[nf][ks] = f;
            }
        u32x4 wr[4][4];                                      // weight ring: slot = k-step & 3
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) request_w<true, kFrag>(wr[ks], kFrag ? a.layers[0].w0_frag : a.layers[0].w0, w, ks, lr, lg);
        lds_barrier();                                       // (not __syncthreads(): the requests stay in flight)
        phase_tick(a.ticks, 2, tick);                        // input pooling, adjacency fragments

        for (int layer = 0; layer < L; ++layer) {
            const gcc_ginw_layer ly = a.layers[layer];
            // ---- AGG[node][ch] = sum_u H^T[ch][u] ADJ[node][u] -> Q (node-major): 128 channels x 64 nodes per wave
            {
                f32x4 acc[8][4];
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf) acc[m][nf] = zero4;
                const unsigned char *src = P + (chh * 128) * kStrT2 + lg * 16;
                u32x4 buf[2][8];
#pragma unroll
                for (int m = 0; m < 8; ++m) buf[0][m] = lds16(src + ((m >> 1) * 32 + perm8(lr, m & 1)) * kStrT2);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (ks + 1 < 4 && !(kDbg & 4)) {
#pragma unroll
                        for (int m = 0; m < 8; ++m)
                            buf[(ks + 1) & 1][m] = lds16(src + ((m >> 1) * 32 + perm8(lr, m & 1)) * kStrT2 + (ks + 1) * 64);
                    }
                    SCHED_FENCE();
#pragma unroll
                    for (int m = 0; m < 8; ++m)
#pragma unroll
                        for (int nf = 0; nf < 4; ++nf) acc[m][nf] = mfma_16x16x32_bf16(buf[(kDbg & 4) ? 0 : (ks & 1)][m], adj[nf][ks], acc[m][nf]);
                    SCHED_FENCE();
                }
                if (kDbg & 1) {
                    f32x4 t4 = zero4;
#pragma unroll
                    for (int m = 0; m < 8; ++m)
#pragma unroll
                        for (int nf = 0; nf < 4; ++nf) t4 += acc[m][nf];
                    *(u32x4 *)(Q + (nh * 64 + lr) * kStrN2 + (chh * 128 + lg * 8) * 2) = pack8_bf16(t4, t4);
                } else {
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf)
                        *(u32x4 *)(Q + (nh * 64 + nf * 16 + lr) * kStrN2 + (chh * 128 + p * 32 + lg * 8) * 2) =
                            pack8_bf16(acc[2 * p][nf], acc[2 * p + 1][nf]);
                }
            }
            lds_barrier();
            phase_tick(a.ticks, 3, tick);                    // aggregation
            // ---- Z1[node][ch] = relu(s0 * (AGG W0^T) + t0) -> P (node-major): channels 64 w .. 64 w + 63, all nodes
            {
                f32x4 acc[4][8];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int nf = 0; nf < 8; ++nf) acc[m][nf] = zero4;
                float4 sc[2][2], sh[2][2];                   // scale / shift of the lane's 8 channels per fragment pair
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        sc[p][e] = *(const float4 *)(ly.s0 + w * 64 + p * 32 + lg * 8 + e * 4);
                        sh[p][e] = *(const float4 *)(ly.t0 + w * 64 + p * 32 + lg * 8 + e * 4);
                    }
                const unsigned char *src = Q + lr * kStrN2 + lg * 16;
                u32x4 buf[2][8];
#pragma unroll
                for (int nf = 0; nf < 8; ++nf) buf[0][nf] = lds16(src + nf * 16 * kStrN2);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (ks + 1 < 8 && !(kDbg & 4)) {
#pragma unroll
                        for (int nf = 0; nf < 8; ++nf) buf[(ks + 1) & 1][nf] = lds16(src + nf * 16 * kStrN2 + (ks + 1) * 64);
                    }
                    SCHED_FENCE();
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int nf = 0; nf < 8; ++nf) acc[m][nf] = mfma_16x16x32_bf16(wr[ks & 3][m], buf[(kDbg & 4) ? 0 : (ks & 1)][nf], acc[m][nf]);
                    SCHED_FENCE();
                    if (!(kDbg & 2)) {
                        if (ks < 4) request_w<true, kFrag>(wr[ks & 3], kFrag ? ly.w0_frag : ly.w0, w, ks + 4, lr, lg);
                        else request_w<false, kFrag>(wr[ks & 3], kFrag ? ly.w1_frag : ly.w1, w, ks - 4, lr, lg);
                    }
                    SCHED_FENCE();
                }
                if (kDbg & 1) {
                    f32x4 t4 = zero4;
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int nf = 0; nf < 8; ++nf) t4 += acc[m][nf];
                    t4[0] += sc[0][0].x + sh[1][1].w;
                    *(u32x4 *)(P + lr * kStrZ2 + (w * 64 + lg * 8) * 2) = pack8_bf16(t4, t4);
                } else {
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int nf = 0; nf < 8; ++nf) {
                        f32x4 lo = acc[2 * p][nf], hi = acc[2 * p + 1][nf];     // channels 8 lg + {0, 2, 4, 6} and + {1, 3, 5, 7}
                        lo[0] = fmaxf(fmaf(lo[0], sc[p][0].x, sh[p][0].x), 0.f); hi[0] = fmaxf(fmaf(hi[0], sc[p][0].y, sh[p][0].y), 0.f);
                        lo[1] = fmaxf(fmaf(lo[1], sc[p][0].z, sh[p][0].z), 0.f); hi[1] = fmaxf(fmaf(hi[1], sc[p][0].w, sh[p][0].w), 0.f);
                        lo[2] = fmaxf(fmaf(lo[2], sc[p][1].x, sh[p][1].x), 0.f); hi[2] = fmaxf(fmaf(hi[2], sc[p][1].y, sh[p][1].y), 0.f);
                        lo[3] = fmaxf(fmaf(lo[3], sc[p][1].z, sh[p][1].z), 0.f); hi[3] = fmaxf(fmaf(hi[3], sc[p][1].w, sh[p][1].w), 0.f);
                        *(u32x4 *)(P + (nf * 16 + lr) * kStrZ2 + (w * 64 + p * 32 + lg * 8) * 2) = pack8_bf16(lo, hi);
                    }
                }
            }
            lds_barrier();
            phase_tick(a.ticks, 4, tick);                    // first Linear
            // ---- H'^T[ch][node] = relu(s2 * relu(s1 * (Z1 W1^T) + t1) + t2) -> Q (channel-major), nodes >= n as 0; SumPooling
            {
                // (after the last layer the requests are dummies: no branch around them)
                const gcc_ginw_layer &lnext = a.layers[layer + 1 < L ? layer + 1 : layer];
                const uint16_t *wnext = kFrag ? lnext.w0_frag : lnext.w0;
                f32x4 acc[8][4];
#pragma unroll
                for (int nf = 0; nf < 8; ++nf)
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[nf][m] = zero4;
                // relu(s2 * relu(s1 * x + t1) + t2) as ONE multiply-add and ONE clamp per value: with A = s1 s2 and
                // B = s2 t1 + t2 it is max(A x + B, max(t2, 0)) for s2 >= 0 and min(max(A x + B, 0), max(t2, 0)) for s2 < 0
                float ea[4], eb[4], elo[4], ehi[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int c = w * 64 + m * 16 + lr;
                    const float s1 = ly.s1[c], t1 = ly.t1[c], s2 = ly.s2[c], t2 = ly.t2[c];
                    ea[m] = s1 * s2;
                    eb[m] = fmaf(s2, t1, t2);
                    elo[m] = s2 >= 0.f ? fmaxf(t2, 0.f) : 0.f;
                    ehi[m] = s2 >= 0.f ? __uint_as_float(0x7F800000u) : fmaxf(t2, 0.f);
                }
                const unsigned char *src = P + lg * 16;
                u32x4 buf[2][8];
#pragma unroll
                for (int nf = 0; nf < 8; ++nf) buf[0][nf] = lds16(src + ((nf >> 1) * 32 + perm8(lr, nf & 1)) * kStrZ2);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (ks + 1 < 8 && !(kDbg & 4)) {
#pragma unroll
                        for (int nf = 0; nf < 8; ++nf)
                            buf[(ks + 1) & 1][nf] = lds16(src + ((nf >> 1) * 32 + perm8(lr, nf & 1)) * kStrZ2 + (ks + 1) * 64);
                    }
                    SCHED_FENCE();
#pragma unroll
                    for (int nf = 0; nf < 8; ++nf)
#pragma unroll
                        for (int m = 0; m < 4; ++m) acc[nf][m] = mfma_16x16x32_bf16(buf[(kDbg & 4) ? 0 : (ks & 1)][nf], wr[ks & 3][m], acc[nf][m]);
                    SCHED_FENCE();
                    if (!(kDbg & 2)) {
                        if (ks < 4) request_w<false, kFrag>(wr[ks & 3], kFrag ? ly.w1_frag : ly.w1, w, ks + 4, lr, lg);
                        else request_w<true, kFrag>(wr[ks & 3], wnext, w, ks - 4, lr, lg);
                    }
                    SCHED_FENCE();
                }
                if (kDbg & 1) {
                    f32x4 t4 = zero4;
#pragma unroll
                    for (int nf = 0; nf < 8; ++nf)
#pragma unroll
                        for (int m = 0; m < 4; ++m) t4 += acc[nf][m];
                    t4[0] += ea[0] + eb[1] + elo[2] + ehi[3];
                    *(u32x4 *)(Q + (w * 64 + lr) * kStrT2 + lg * 16) = pack8_bf16(t4, t4);
                } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int c = w * 64 + m * 16 + lr;
                    float psum = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int node = q * 32 + lg * 8;
                        f32x4 lo = acc[2 * q][m], hi = acc[2 * q + 1][m];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {        // apply_func: relu(bn(mlp)), then relu(batch_norms[i](.))
                            lo[r] = clamp_f32(fmaf(lo[r], ea[m], eb[m]), elo[m], ehi[m]);
                            hi[r] = clamp_f32(fmaf(hi[r], ea[m], eb[m]), elo[m], ehi[m]);
                        }
                        if (q * 32 + 32 > n) {               // (block-uniform) this 32-block holds padding nodes: they stay 0
#pragma unroll
                            for (int r = 0; r < 4; ++r) {        // (lo: nodes node + 0, 2, 4, 6; hi: + 1, 3, 5, 7)
                                lo[r] = node + 2 * r < n ? lo[r] : 0.f;
                                hi[r] = node + 2 * r + 1 < n ? hi[r] : 0.f;
                            }
                        }
                        const u32x4 hv = pack8_bf16(lo, hi);
                        psum = sum8_bf16(hv, psum);
                        *(u32x4 *)(Q + c * kStrT2 + node * 2) = hv;
                    }
                    psum += wave_shfl_xor(psum, 16);
                    psum += wave_shfl_xor(psum, 32);
                    if (lg == 0 && a.pooled) a.pooled[((int64_t)b * (L + 1) + layer + 1) * kD + c] = psum;
                }
                }
            }
            lds_barrier();
            phase_tick(a.ticks, 5, tick);                    // second Linear
            unsigned char *t = P; P = Q; Q = t;
        }
        // ---- the last layer's rows back to node-major global memory (4 nodes x 8 channels per work item)
        if (a.x_out)
            for (int idx = tid; idx < (kNodes / 4) * (kD / 8); idx += kT2) {
                const int chunk = idx & (kD / 8 - 1), node = 4 * (idx >> 5);
                if (node < n) {
                    u32x2 c[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) c[e] = *(const u32x2 *)(P + (chunk * 8 + e) * kStrT2 + node * 2);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (node + j < n) {
                            const int sh = (j & 1) * 16, h = j >> 1;
                            u32x4 v;
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                v[q] = ((c[2 * q][h] >> sh) & 0xFFFFu) | (((c[2 * q + 1][h] >> sh) & 0xFFFFu) << 16);
                            *(u32x4 *)(a.x_out + (int64_t)(n0 + node + j) * kD + chunk * 8) = v;
                        }
                    }
                }
            }
        phase_tick(a.ticks, 6, tick);                        // rows out
        if (a.ticks && tid == 0) atomicAdd((unsigned long long *)&a.ticks[15], 1ull);
    }
}

// =====================================================================================================================
// Subgraphs over kNodes nodes (ego-nets of the pre-training workload reach ~900 nodes: DESIGN.md section 6), one layer per
// launch, one (subgraph, block of 128 rows) per workgroup at a time.  The layer's structure, LDS layouts and rounding points are
// the fused kernel's (second shape); what changes is the aggregation: the row block's adjacency is a [128 x n] strip, taken
// 128 columns at a time -- the column block's rows H_c^T come from global memory (the previous layer's output), the
// 16-bit neighbour counts of (row block, column block) are rebuilt, and the products ACCUMULATE into the same registers,
// so AGG is rounded to bf16 once, after the whole sum, exactly as for a small subgraph.  Rows travel through two global
// ping-pong buffers between the launches of consecutive layers (a layer needs every row block of the one before).
__global__ void ginw_classify_kernel(WideArgs a)
{
    __shared__ int32_t cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    for (int b = threadIdx.x; b < a.batch_size; b += blockDim.x) {
        const int n = a.node_off[b + 1] - a.node_off[b];
        if (n <= kNodes) continue;
        const int nblk = (n + kNodes - 1) / kNodes;
        const int at = atomicAdd(&cnt, nblk);
        if (at + nblk > a.big_cap) {                         // no room: refused (status); the slots it reserved below the cap are
            atomicOr(a.status, (int32_t)GCC_STATUS_GINW_TOO_LARGE);      // marked so that the block kernel skips them instead of
            for (int r = 0; r < nblk && at + r < a.big_cap; ++r) a.big_work[2 + 2 * (at + r)] = -1;   // reading uninitialised pairs
            continue;
        }
        for (int r = 0; r < nblk; ++r) { a.big_work[2 + 2 * (at + r)] = b; a.big_work[3 + 2 * (at + r)] = r; }
    }
    __syncthreads();
    if (threadIdx.x == 0) a.big_work[0] = cnt < a.big_cap ? cnt : a.big_cap;
}

template <bool kFrag>
__global__ __launch_bounds__(kT2) void gin_wide_big_kernel(WideArgs a, int layer, const uint16_t *hin, uint16_t *hout)
{
    DYN_SMEM(smem);
    unsigned char *P = smem, *Q = smem + kReg2;
    int32_t *rp = (int32_t *)(smem + 2 * kReg2);             // [129] row pointers of the row block
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int L = a.num_layers;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int nh = w & 1, chh = w >> 1;
    const gcc_ginw_layer ly = a.layers[layer];
    const int nitems = a.big_work[0];

    for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
        const int b = a.big_work[2 + 2 * it], r = a.big_work[3 + 2 * it];
        if (b < 0) continue;                                 // (workgroup-uniform) a slot of a refused subgraph
        const int n0 = a.node_off[b], n = a.node_off[b + 1] - n0;
        const int nblk = (n + kNodes - 1) / kNodes;
        const int row0 = n0 + r * kNodes, nr = min(kNodes, n - r * kNodes);
        u32x4 wr[4][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) request_w<true, kFrag>(wr[ks], kFrag ? ly.w0_frag : ly.w0, w, ks, lr, lg);
        f32x4 agg[8][4];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) agg[m][nf] = zero4;
        float pool0 = 0.f;                                   // hidden_rep[0] of the subgraph (layer 0, row block 0: it sees every column block)
        for (int c = 0; c < nblk; ++c) {
            __syncthreads();                                 // the previous block's fragments / this item's predecessor are done with P, Q, rp
            const int col0 = n0 + c * kNodes, nc = min(kNodes, n - c * kNodes);
            for (int i = tid; i < kNodes * kStrT2 / 4; i += kT2) ((uint32_t *)Q)[i] = 0u;
            if (tid <= nr) rp[tid] = a.row_ptr[row0 + tid];
            for (int idx = tid; idx < (kNodes / 4) * (kD / 8); idx += kT2) {       // H_c^T -> P (channel-major), as the fused kernel
                const int chunk = idx & (kD / 8 - 1), node = 4 * (idx >> 5);
                u32x4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    v[j] = node + j < nc ? *(const u32x4 *)(hin + (int64_t)(col0 + node + j) * kD + chunk * 8) : z;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int sh = (e & 1) * 16;
                    u32x2 o;
                    o[0] = ((v[0][e >> 1] >> sh) & 0xFFFFu) | (((v[1][e >> 1] >> sh) & 0xFFFFu) << 16);
                    o[1] = ((v[2][e >> 1] >> sh) & 0xFFFFu) | (((v[3][e >> 1] >> sh) & 0xFFFFu) << 16);
                    *(u32x2 *)(P + (chunk * 8 + e) * kStrT2 + node * 2) = o;
                }
            }
            __syncthreads();
            {   // neighbour counts of (row block r, column block c): targets outside the column block belong to another pass
                const int e0 = rp[0], e1 = rp[nr];
                for (int eb = e0 + 16 * tid; eb < e1; eb += 16 * kT2) {
                    int cols[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) cols[j] = eb + j < e1 ? a.col_idx[eb + j] : -1;
                    int lo = 0, hi = nr;
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (rp[mid] <= eb) lo = mid; else hi = mid;
                    }
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        if (eb + j >= e1) break;
                        while (rp[lo + 1] <= eb + j) ++lo;
                        const int ug = cols[j] - n0, u = ug - c * kNodes;
                        if ((unsigned)ug >= (unsigned)n) { if (c == 0) atomicOr(a.status, (int32_t)GCC_STATUS_GINW_BAD_EDGE); }
                        else if ((unsigned)u < (unsigned)kNodes) atomicAdd((uint32_t *)(Q + lo * kStrT2 + (u >> 1) * 4), (u & 1) ? 0x10000u : 1u);
                    }
                }
                if (c == r && tid < nr) atomicAdd((uint32_t *)(Q + tid * kStrT2 + (tid >> 1) * 4), (tid & 1) ? 0x10000u : 1u);   // + h_v itself
            }
            __syncthreads();
            if (layer == 0 && r == 0 && a.pooled)            // (uniform) hidden_rep[0] = the input (gin.py:216)
                for (int j = 0; j < kNodes / 8; ++j) pool0 += sum8_bf16(lds16(P + tid * kStrT2 + j * 16));
            u32x4 adj[4][4];
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const u32x4 cq = lds16(Q + (nh * 64 + nf * 16 + lr) * kStrT2 + (ks * 32 + lg * 8) * 2);
                    u32x4 f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) f[q] = pack2_bf16((float)(cq[q] & 0xFFFFu), (float)(cq[q] >> 16));
                    adj[nf][ks] = f;
                }
            {   // AGG += H_c^T ADJ(r, c): 128 channels x 64 nodes per wave, as the fused kernel
                const unsigned char *src = P + (chh * 128) * kStrT2 + lg * 16;
                u32x4 buf[2][8];
#pragma unroll
                for (int m = 0; m < 8; ++m) buf[0][m] = lds16(src + ((m >> 1) * 32 + perm8(lr, m & 1)) * kStrT2);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (ks + 1 < 4) {
#pragma unroll
                        for (int m = 0; m < 8; ++m)
                            buf[(ks + 1) & 1][m] = lds16(src + ((m >> 1) * 32 + perm8(lr, m & 1)) * kStrT2 + (ks + 1) * 64);
                    }
                    SCHED_FENCE();
#pragma unroll
                    for (int m = 0; m < 8; ++m)
#pragma unroll
                        for (int nf = 0; nf < 4; ++nf) agg[m][nf] = mfma_16x16x32_bf16(buf[ks & 1][m], adj[nf][ks], agg[m][nf]);
                    SCHED_FENCE();
                }
            }
        }
        if (layer == 0 && r == 0 && a.pooled) a.pooled[((int64_t)b * (L + 1)) * kD + tid] = pool0;
        __syncthreads();                                     // every wave is done with the last column block's P and Q
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
                *(u32x4 *)(Q + (nh * 64 + nf * 16 + lr) * kStrN2 + (chh * 128 + p * 32 + lg * 8) * 2) =
                    pack8_bf16(agg[2 * p][nf], agg[2 * p + 1][nf]);
        lds_barrier();
        // ---- Z1[node][ch] = relu(s0 * (AGG W0^T) + t0) -> P (node-major)
        {
            f32x4 acc[4][8];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nf = 0; nf < 8; ++nf) acc[m][nf] = zero4;
            float4 sc[2][2], sh[2][2];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    sc[p][e] = *(const float4 *)(ly.s0 + w * 64 + p * 32 + lg * 8 + e * 4);
                    sh[p][e] = *(const float4 *)(ly.t0 + w * 64 + p * 32 + lg * 8 + e * 4);
                }
            const unsigned char *src = Q + lr * kStrN2 + lg * 16;
            u32x4 buf[2][8];
#pragma unroll
            for (int nf = 0; nf < 8; ++nf) buf[0][nf] = lds16(src + nf * 16 * kStrN2);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 1 < 8) {
#pragma unroll
                    for (int nf = 0; nf < 8; ++nf) buf[(ks + 1) & 1][nf] = lds16(src + nf * 16 * kStrN2 + (ks + 1) * 64);
                }
                SCHED_FENCE();
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int nf = 0; nf < 8; ++nf) acc[m][nf] = mfma_16x16x32_bf16(wr[ks & 3][m], buf[ks & 1][nf], acc[m][nf]);
                SCHED_FENCE();
                if (ks < 4) request_w<true, kFrag>(wr[ks & 3], kFrag ? ly.w0_frag : ly.w0, w, ks + 4, lr, lg);
                else request_w<false, kFrag>(wr[ks & 3], kFrag ? ly.w1_frag : ly.w1, w, ks - 4, lr, lg);
                SCHED_FENCE();
            }
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int nf = 0; nf < 8; ++nf) {
                    f32x4 lo = acc[2 * p][nf], hi = acc[2 * p + 1][nf];
                    lo[0] = fmaxf(fmaf(lo[0], sc[p][0].x, sh[p][0].x), 0.f); hi[0] = fmaxf(fmaf(hi[0], sc[p][0].y, sh[p][0].y), 0.f);
                    lo[1] = fmaxf(fmaf(lo[1], sc[p][0].z, sh[p][0].z), 0.f); hi[1] = fmaxf(fmaf(hi[1], sc[p][0].w, sh[p][0].w), 0.f);
                    lo[2] = fmaxf(fmaf(lo[2], sc[p][1].x, sh[p][1].x), 0.f); hi[2] = fmaxf(fmaf(hi[2], sc[p][1].y, sh[p][1].y), 0.f);
                    lo[3] = fmaxf(fmaf(lo[3], sc[p][1].z, sh[p][1].z), 0.f); hi[3] = fmaxf(fmaf(hi[3], sc[p][1].w, sh[p][1].w), 0.f);
                    *(u32x4 *)(P + (nf * 16 + lr) * kStrZ2 + (w * 64 + p * 32 + lg * 8) * 2) = pack8_bf16(lo, hi);
                }
        }
        lds_barrier();
        // ---- H'^T[ch][node] = relu(s2 * relu(s1 * (Z1 W1^T) + t1) + t2) -> Q (channel-major), rows >= nr as 0; SumPooling
        {
            f32x4 acc[8][4];
#pragma unroll
            for (int nf = 0; nf < 8; ++nf)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[nf][m] = zero4;
            float ea[4], eb[4], elo[4], ehi[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int ch = w * 64 + m * 16 + lr;
                const float s1 = ly.s1[ch], t1 = ly.t1[ch], s2 = ly.s2[ch], t2 = ly.t2[ch];
                ea[m] = s1 * s2;
                eb[m] = fmaf(s2, t1, t2);
                elo[m] = s2 >= 0.f ? fmaxf(t2, 0.f) : 0.f;
                ehi[m] = s2 >= 0.f ? __uint_as_float(0x7F800000u) : fmaxf(t2, 0.f);
            }
            const unsigned char *src = P + lg * 16;
            u32x4 buf[2][8];
#pragma unroll
            for (int nf = 0; nf < 8; ++nf) buf[0][nf] = lds16(src + ((nf >> 1) * 32 + perm8(lr, nf & 1)) * kStrZ2);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 1 < 8) {
#pragma unroll
                    for (int nf = 0; nf < 8; ++nf)
                        buf[(ks + 1) & 1][nf] = lds16(src + ((nf >> 1) * 32 + perm8(lr, nf & 1)) * kStrZ2 + (ks + 1) * 64);
                }
                SCHED_FENCE();
#pragma unroll
                for (int nf = 0; nf < 8; ++nf)
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[nf][m] = mfma_16x16x32_bf16(buf[ks & 1][nf], wr[ks & 3][m], acc[nf][m]);
                SCHED_FENCE();
                if (ks < 4) request_w<false, kFrag>(wr[ks & 3], kFrag ? ly.w1_frag : ly.w1, w, ks + 4, lr, lg);
                SCHED_FENCE();
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int ch = w * 64 + m * 16 + lr;
                float psum = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int node = q * 32 + lg * 8;
                    f32x4 lo = acc[2 * q][m], hi = acc[2 * q + 1][m];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        lo[rr] = clamp_f32(fmaf(lo[rr], ea[m], eb[m]), elo[m], ehi[m]);
                        hi[rr] = clamp_f32(fmaf(hi[rr], ea[m], eb[m]), elo[m], ehi[m]);
                    }
                    if (q * 32 + 32 > nr) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            lo[rr] = node + 2 * rr < nr ? lo[rr] : 0.f;
                            hi[rr] = node + 2 * rr + 1 < nr ? hi[rr] : 0.f;
                        }
                    }
                    const u32x4 hv = pack8_bf16(lo, hi);
                    psum = sum8_bf16(hv, psum);
                    *(u32x4 *)(Q + ch * kStrT2 + node * 2) = hv;
                }
                psum += wave_shfl_xor(psum, 16);
                psum += wave_shfl_xor(psum, 32);
                // (the row blocks of a subgraph add up in arrival order: fp32 atomics; the fused kernel zeroed the row)
                if (lg == 0 && a.pooled) atomicAdd(&a.pooled[((int64_t)b * (L + 1) + layer + 1) * kD + ch], psum);
            }
        }
        __syncthreads();
        // ---- the row block's rows back to node-major global memory
        for (int idx = tid; idx < (kNodes / 4) * (kD / 8); idx += kT2) {
            const int chunk = idx & (kD / 8 - 1), node = 4 * (idx >> 5);
            if (node < nr) {
                u32x2 c8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) c8[e] = *(const u32x2 *)(Q + (chunk * 8 + e) * kStrT2 + node * 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (node + j < nr) {
                        const int sh = (j & 1) * 16, h = j >> 1;
                        u32x4 v;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            v[q] = ((c8[2 * q][h] >> sh) & 0xFFFFu) | (((c8[2 * q + 1][h] >> sh) & 0xFFFFu) << 16);
                        *(u32x4 *)(hout + (int64_t)(row0 + node + j) * kD + chunk * 8) = v;
                    }
                }
            }
        }
    }
}

}  // namespace

extern "C" void gcc_ginw_debug_ticks(long long *device_ticks64) { g_ticks = device_ticks64; }

extern "C" int32_t gcc_ginw_pack_weights(const uint16_t *w, uint16_t *w_frag, int32_t which, void *stream)
{
    if (!w || !w_frag || which < 0 || which > 1) {
        snprintf(g_err, kErrLen, "gcc_ginw_pack_weights: bad argument");
        return -1;
    }
    hipLaunchKernelGGL(ginw_pack_kernel, dim3(kD * kD / 8 / 256), dim3(256), 0, (hipStream_t)stream, w, w_frag, (int)which);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_err, kErrLen, "gcc_ginw_pack_weights: %s", hipGetErrorString(e)); return -10; }
    return 0;
}

extern "C" int32_t gcc_ginw_forward(const gcc_ginw_args *g, int32_t *status, gcc_prof *prof, void *stream)
{
    if (!g || !status || !g->node_off || !g->row_ptr || !g->col_idx || !g->x_in || g->batch_size < 1 || g->num_layers < 1 ||
        g->num_layers > GCC_GIN_MAX_LAYERS || (!g->x_out && !g->pooled)) {
        snprintf(g_err, kErrLen, "gcc_ginw_forward: bad argument");
        return -1;
    }
    WideArgs a;
    a.node_off = g->node_off; a.row_ptr = g->row_ptr; a.col_idx = g->col_idx;
    a.x_in = g->x_in; a.x_out = g->x_out; a.pooled = g->pooled; a.status = status;
    a.batch_size = g->batch_size; a.num_layers = g->num_layers;
    a.ticks = g_ticks;
    a.big0 = a.big1 = nullptr; a.big_work = nullptr; a.big_cap = 0;
    if (g->scratch) {
        // [2][num_nodes][256] bf16 ping-pong rows | work list {count, -, pairs}: gcc_ginw_scratch_bytes
        const int64_t rows = (int64_t)g->num_nodes * kD * 2;
        const int64_t cap = g->num_nodes / kNodes + g->batch_size;
        if (g->num_nodes < 1 || g->scratch_bytes < 2 * rows + (2 + 2 * cap) * 4 || ((uintptr_t)g->scratch & 15u)) {
            snprintf(g_err, kErrLen, "gcc_ginw_forward: scratch of %lld bytes (16-byte aligned) needed for %lld nodes",
                     (long long)(2 * rows + (2 + 2 * cap) * 4), (long long)g->num_nodes);
            return -3;
        }
        a.big0 = (uint16_t *)g->scratch;
        a.big1 = a.big0 + (int64_t)g->num_nodes * kD;
        a.big_work = (int32_t *)((char *)g->scratch + 2 * rows);
        a.big_cap = (int32_t)cap;
    }
    for (int i = 0; i < GCC_GIN_MAX_LAYERS; ++i) {
        a.layers[i] = g->layers[i];
        const gcc_ginw_layer &l = g->layers[i];
        if (i < g->num_layers && (!l.w0 || !l.w1 || !l.s0 || !l.t0 || !l.s1 || !l.t1 || !l.s2 || !l.t2)) {
            snprintf(g_err, kErrLen, "gcc_ginw_forward: layer %d has a NULL parameter", i);
            return -1;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    static int shape = 0;                                    // GCC_GINW_KERNEL=1: the first kernel shape (A/B runs)
    if (shape == 0) {
        const char *e = getenv("GCC_GINW_KERNEL");
        shape = e && atoi(e) == 1 ? 1 : 2;
#ifndef GCC_AMD_HIPEMU
        // more than 64 KiB of dynamic LDS has to be opted into
        (void)hipFuncSetAttribute((const void *)gin_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)gin_wide2_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds2);
        (void)hipFuncSetAttribute((const void *)gin_wide2_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds2);
        (void)hipFuncSetAttribute((const void *)gin_wide_big_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds2);
        (void)hipFuncSetAttribute((const void *)gin_wide_big_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds2);
#ifdef GCC_GINW_ABLATE                                       // timing-only builds (make EXTRA=-DGCC_GINW_ABLATE): never in the shipped library
        (void)hipFuncSetAttribute((const void *)gin_wide2_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds2);
        (void)hipFuncSetAttribute((const void *)gin_wide2_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds2);
        (void)hipFuncSetAttribute((const void *)gin_wide2_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds2);
        (void)hipFuncSetAttribute((const void *)gin_wide2_kernel<7, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds2);
#endif
#endif
    }
    prof_mark(prof, 0, s);
    // one workgroup per CU (137 / 148 KB of LDS each), walking the subgraphs with a stride of the grid
    if (shape == 1) hipLaunchKernelGGL(gin_wide_kernel, dim3(min(g->batch_size, 256)), dim3(kThreads), kLdsBytes, s, a);
    else {
        bool frag = true;                                    // every layer carries the fragment-major copies?
        for (int i = 0; i < g->num_layers; ++i) frag = frag && g->layers[i].w0_frag && g->layers[i].w1_frag;
        const dim3 grid(min(g->batch_size, 256)), block(kT2);
        if (!frag) hipLaunchKernelGGL((gin_wide2_kernel<0, false>), grid, block, kLds2, s, a);
#ifdef GCC_GINW_ABLATE
        else if (getenv("GCC_GINW_DBG") && atoi(getenv("GCC_GINW_DBG")) == 1) hipLaunchKernelGGL((gin_wide2_kernel<1, true>), grid, block, kLds2, s, a);
        else if (getenv("GCC_GINW_DBG") && atoi(getenv("GCC_GINW_DBG")) == 2) hipLaunchKernelGGL((gin_wide2_kernel<2, true>), grid, block, kLds2, s, a);
        else if (getenv("GCC_GINW_DBG") && atoi(getenv("GCC_GINW_DBG")) == 4) hipLaunchKernelGGL((gin_wide2_kernel<4, true>), grid, block, kLds2, s, a);
        else if (getenv("GCC_GINW_DBG") && atoi(getenv("GCC_GINW_DBG")) == 7) hipLaunchKernelGGL((gin_wide2_kernel<7, true>), grid, block, kLds2, s, a);
#endif
        else hipLaunchKernelGGL((gin_wide2_kernel<0, true>), grid, block, kLds2, s, a);
    }
    if (a.big_work) {
        // subgraphs over 128 nodes: (subgraph, row block) work list, then one launch per layer (the fused launch above left
        // them alone; nothing to do -- a few microseconds per launch -- when the batch has none)
        bool frag = true;
        for (int i = 0; i < g->num_layers; ++i) frag = frag && g->layers[i].w0_frag && g->layers[i].w1_frag;
        hipLaunchKernelGGL(ginw_classify_kernel, dim3(1), dim3(256), 0, s, a);
        const uint16_t *hin = g->x_in;
        for (int l = 0; l < g->num_layers; ++l) {
            uint16_t *hout = (l == g->num_layers - 1 && g->x_out) ? g->x_out : ((l & 1) ? a.big1 : a.big0);
            if (frag) hipLaunchKernelGGL((gin_wide_big_kernel<true>), dim3(256), dim3(kT2), kLds2, s, a, l, hin, hout);
            else hipLaunchKernelGGL((gin_wide_big_kernel<false>), dim3(256), dim3(kT2), kLds2, s, a, l, hin, hout);
            hin = hout;
        }
    }
    prof_mark(prof, 1, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_err, kErrLen, "gcc_ginw_forward: %s", hipGetErrorString(e)); return -10; }
    return 0;
}

extern "C" int64_t gcc_ginw_scratch_bytes(int64_t num_nodes, int32_t batch_size)
{
    if (num_nodes < 1 || batch_size < 1) return -1;
    return 2 * num_nodes * kD * 2 + (2 + 2 * (num_nodes / kNodes + batch_size)) * 4;
}
