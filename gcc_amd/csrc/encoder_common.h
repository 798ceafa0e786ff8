// gcc_amd/csrc/encoder_common.h -- device helpers shared by the GIN encoder's
// forward (encoder.hip) and backward (encoder_bwd.hip) kernels.
#pragma once
#include "host_common.h"

namespace {

constexpr int H = GCC_GIN_HIDDEN;   // 64
constexpr int kTile = 64;           // rows per workgroup tile
constexpr int kLdt = 72;            // LDS row stride in floats (conflict-free ds_read_b128 fragments)
constexpr int kThreads = 256;
constexpr int kGridX = 768;         // tiles are grid-strided
constexpr int kMaxPass = 2;
#ifndef GCC_KREP
#define GCC_KREP GCC_GIN_STAT_REPLICAS   // 16.  Every consumer kernel's workgroups each read ALL replicas of the statistics they normalise
#endif                                   // with (2 x 64 doubles per replica and BatchNorm): at 32 that was 32-64 KiB per workgroup next to a
                                         // 16 KiB tile.  Training stream alone: 0.569 ms (32), 0.558 (16), 0.553 (8, but the producers'
                                         // atomics start to queue: gin_stat 5.0 -> 5.4 us) -- scripts/gpu/r6_call.sh r6c8 / r6c9
constexpr int kRep = GCC_KREP;      // replicas of every atomically accumulated statistics row: a block adds to
                                    // copy (blockIdx.x % kRep), consumers sum the copies -- ~730 workgroups hitting the
                                    // same 8 cache lines with fp64 atomics cost 20-40 us per kernel (rocprof, round 1)

// ---- tile order.  Two properties:
// (1) a workgroup's FIRST tile is a function of blockIdx alone, NOT of the live node count N = node_off[B] (device memory): every
//     tile kernel requests its first tile's rows TOGETHER with N, its statistics and its weights, clamped to the buffers' CAPACITY
//     (gcc_gin_pass.node_cap), and masks rows >= N afterwards.  Rounds 1-5 computed the tile from N (per = ceil(tiles / 8)), so every
//     kernel of the ~40-launch chain ran node count -> wait -> tile rows -> wait: two DEPENDENT memory round trips where one does
//     (profiles/r6_stream_trace_start.txt: the kernels that do nothing else take 6.5 us).
// (2) XCD-aware: the dispatcher places workgroup b on XCD b % 8 (8 XCDs, each with its own 4 MiB L2).  XCD x takes the tiles
//     {32 g + 4 x .. 32 g + 4 x + 3 : g = 0, 1, ...}: runs of four consecutive tiles (256 rows, two or three ego-nets) share an L2, so
//     most of the neighbour rows a tile gathers were fetched by a neighbouring tile of the same XCD, for ANY N.  (The N-dependent
//     contiguous ranges of round 3 measured no different from tile = blockIdx.x at bsz 256: scripts/gpu/r3_call23.sh.)
// A launch with more tiles than workgroups walks on in steps of gridDim.x (a multiple of 32 keeps the pattern).
// no_tiles(N): this workgroup has nothing to walk -- the tile kernels return on it as soon as N is known, before their tables and
// staged weights (the grid is sized for the node CAPACITY: at bsz 256 about half of the 768 workgroups of a launch have no tile).
__device__ __forceinline__ int first_tile()
{
    const int b = (int)blockIdx.x;
    if (((int)gridDim.x & 31) != 0) return b;
    const int x = b & 7, j = b >> 3;
    return ((((j >> 2) << 3) + x) << 2) + (j & 3);
}
struct TileWalk {
    int ti, tend, step;
    __device__ __forceinline__ explicit TileWalk(int N) : ti(first_tile()), tend((N + kTile - 1) / kTile), step((int)gridDim.x) {}
};
__device__ __forceinline__ bool no_tiles(int N) { return first_tile() * kTile >= N; }
// row r of a tile, clamped into the buffers (speculative requests: cap = node capacity; afterwards masked by r < N)
__device__ __forceinline__ int cap_row(int row, int cap) { return min(row, cap - 1); }

struct F4 { float x, y, z, w; };

__device__ __forceinline__ F4 ld4(const float *p)
{
    const float4 v = *reinterpret_cast<const float4 *>(p);
    F4 r = {v.x, v.y, v.z, v.w};
    return r;
}
__device__ __forceinline__ void st4(float *p, F4 v)
{
    *reinterpret_cast<float4 *>(p) = make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ F4 add4(F4 a, F4 b) { F4 r = {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; return r; }
__device__ __forceinline__ float &at(F4 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// y = max(x * scale + shift, 0)
struct Aff4 { F4 scale, shift; };
__device__ __forceinline__ F4 affine_relu(F4 x, const Aff4 &a)
{
    F4 r;
    r.x = fmaxf(fmaf(x.x, a.scale.x, a.shift.x), 0.f);
    r.y = fmaxf(fmaf(x.y, a.scale.y, a.shift.y), 0.f);
    r.z = fmaxf(fmaf(x.z, a.scale.z, a.shift.z), 0.f);
    r.w = fmaxf(fmaf(x.w, a.scale.w, a.shift.w), 0.f);
    return r;
}

struct BnDev {
    const float *weight, *bias;
    float *running_mean, *running_var;
    int64_t *nbt;
    const double *stats;    // [kRep][2][64] column sum / sum of squares of this BN's input (training mode)
    const double *totals;   // [2][64] the replicas added up by the producing kernel's last workgroup, or NULL
};

// ---- column sums over the kRep replicas of an accumulator.  Pair p < 128 = (slot p >> 6, channel p & 63) of replica r
// lives at rep[r * stride + p] (forward statistics: stride 2 * 64; backward sums: stride 3 * 64, slots 0 and 1).
// ALL kThreads threads take part (block-uniform): thread t adds the replicas of half t >> 7 for pair t & 127, requested as
// independent loads.  (The obvious per-channel loop came out of the compiler as load - wait - add, 64 DEPENDENT L2 round
// trips in the prologue of every kernel of the chain: 10-25 us each, profiles/r3_replica_sum_isa.txt.)
// rep_request only REQUESTS (so that a prologue can put every load it needs in flight before the first wait: node count,
// replicas of one or two accumulators, BatchNorm weights); rep_sum adds them in a fixed order.
struct RepReq { double v[kRep / 2]; };
__device__ __forceinline__ RepReq rep_request(const double *rep, int stride)
{
    static_assert(kThreads == 256 && (kRep == 8 || kRep % 16 == 0), "two halves of the replicas, batches of 8 (4 at kRep 8)");
    static_assert(kRep <= GCC_GIN_STAT_REPLICAS, "callers size the statistics buffers by the header's constant");
    const int t = (int)threadIdx.x, p = t & 127, g = t >> 7;
    const double *src = rep + (int64_t)(g * (kRep / 2)) * stride + p;
    RepReq r;
#pragma unroll
    for (int u = 0; u < kRep / 2; ++u) r.v[u] = src[(int64_t)u * stride];
    return r;
}
__device__ __forceinline__ double rep_sum(const RepReq &r)
{
    double acc = 0.0;
    if (kRep / 2 == 4) return (r.v[0] + r.v[1]) + (r.v[2] + r.v[3]);
#pragma unroll
    for (int b = 0; b + 7 < kRep / 2; b += 8)
        acc += ((r.v[b] + r.v[b + 1]) + (r.v[b + 2] + r.v[b + 3])) + ((r.v[b + 4] + r.v[b + 5]) + (r.v[b + 6] + r.v[b + 7]));
    return acc;
}
// After the trailing barrier: total(p) = sums[p] + sums[128 + p].
__device__ __forceinline__ void replica_sums128(const double *rep, int stride, double *sums /* LDS [256] */)
{
    const RepReq r = rep_request(rep, stride);
    sums[threadIdx.x] = rep_sum(r);
    __syncthreads();
}

// BatchNorm1d as y = x * scale + shift for channel c from the column sum s1 and sum of squares s2.  training: biased batch
// variance (gin.py:56,115,219 -> torch.nn.functional.batch_norm); eval: running statistics.
__device__ __forceinline__ void bn_scale_shift_from(const BnDev &bn, int c, double s1, double s2, double n, float eps, int training,
                                                    float &scale, float &shift)
{
    double mean, var;
    if (training) {
        mean = s1 / n;
        var = s2 / n - mean * mean;
        if (var < 0.0) var = 0.0;
    } else {
        mean = (double)bn.running_mean[c];
        var = (double)bn.running_var[c];
    }
    const double rstd = 1.0 / sqrt(var + (double)eps);
    scale = (float)((double)bn.weight[c] * rstd);
    shift = (float)((double)bn.bias[c] - mean * (double)bn.weight[c] * rstd);
}
// one thread, one channel (callers outside the chain of big kernels; with totals this is 2 loads)
__device__ __forceinline__ void bn_scale_shift(const BnDev &bn, int c, double n, float eps, int training,
                                               float &scale, float &shift)
{
    double s1 = 0.0, s2 = 0.0;
    if (training) {
        if (bn.totals) {
            s1 = bn.totals[c];
            s2 = bn.totals[H + c];
        } else {
            for (int r = 0; r < kRep; ++r) { s1 += bn.stats[r * 2 * H + c]; s2 += bn.stats[r * 2 * H + H + c]; }
        }
    }
    bn_scale_shift_from(bn, c, s1, s2, n, eps, training, scale, shift);
}

// scale/shift of all 64 channels into an LDS table tab[2][64] (the forward kernels: statistics replicas, no totals yet).
// bn_request puts every load the table needs in flight -- the replicas, weight, bias and running statistics of channel
// t & 63: all of them exist in both modes -- and bn_table_finish consumes them, so that a kernel's prologue is ONE round
// trip (requests of all its tables, then the node count, then the waits) instead of node count -> replicas -> weights per
// table.  ALL threads call both (block-uniform); scratch = LDS [256] doubles that nothing else uses during the call;
// bn_table_finish ends with a barrier (tab is valid, scratch is free again).
struct BnReq { RepReq rep; float w, b, rm, rv; };
__device__ __forceinline__ BnReq bn_request(const BnDev &bn)
{
    const int c = (int)threadIdx.x & (H - 1);
    BnReq r;
    r.rep = rep_request(bn.stats, 2 * H);
    r.w = bn.weight[c]; r.b = bn.bias[c];
    // (a training-mode caller without running statistics: the values are not used then -- read the weights again instead)
    const float *rmp = bn.running_mean ? bn.running_mean : bn.weight, *rvp = bn.running_var ? bn.running_var : bn.weight;
    r.rm = rmp[c]; r.rv = rvp[c];
    return r;
}
__device__ __forceinline__ void bn_table_finish(float *tab, const BnReq &r, double n, float eps, int training, double *scratch)
{
    const int c = (int)threadIdx.x;
    scratch[c] = rep_sum(r.rep);
    __syncthreads();
    if (c < H) {
        double mean, var;
        if (training) {                              // biased batch variance (gin.py:56,115,219 -> F.batch_norm)
            const double s1 = scratch[c] + scratch[128 + c], s2 = scratch[H + c] + scratch[128 + H + c];
            mean = s1 / n;
            var = s2 / n - mean * mean;
            if (var < 0.0) var = 0.0;
        } else {
            mean = (double)r.rm;
            var = (double)r.rv;
        }
        const double rstd = 1.0 / sqrt(var + (double)eps);
        tab[c] = (float)((double)r.w * rstd);
        tab[H + c] = (float)((double)r.b - mean * (double)r.w * rstd);
    }
    __syncthreads();
}
__device__ __forceinline__ Aff4 aff4_from_table(const float *tab, int c0)
{
    Aff4 a;
    a.scale = ld4(&tab[c0]);
    a.shift = ld4(&tab[H + c0]);
    return a;
}

__device__ __forceinline__ Aff4 bn_aff4(const BnDev &bn, int c0, double n, float eps, int training)
{
    Aff4 a;
    bn_scale_shift(bn, c0 + 0, n, eps, training, a.scale.x, a.shift.x);
    bn_scale_shift(bn, c0 + 1, n, eps, training, a.scale.y, a.shift.y);
    bn_scale_shift(bn, c0 + 2, n, eps, training, a.scale.z, a.shift.z);
    bn_scale_shift(bn, c0 + 3, n, eps, training, a.scale.w, a.shift.w);
    return a;
}

// ---- SumPooling of an LDS tile (rows tile0 .. tile0+nrows of the batched graph) into
// pooled[graph][64] (fp64 atomics; one flush per run of equal graph ids per thread)
// gid_lds (optional): the tile's graph ids already staged in LDS by the caller (fetched together with the rows)
__device__ __forceinline__ void pool_tile(const float *T, int nrows, double *pooled, const int *gid_lds /* LDS [kTile] */)
{
    const int c = (int)threadIdx.x & 63, part = (int)threadIdx.x >> 6;
    // Precondition (every caller stores the tile that way): rows [nrows, kTile) of T are zero.
    // A wave owns 16 consecutive rows; graph ids ascend, so "first == last" means ONE graph -- 85 % of the waves at rw_hops 256
    // (ego-nets of ~100 nodes): 16 independent LDS reads, an add tree, one atomic.  The general loop below (a flush at every
    // change of graph: a branch and a dependent LDS read per row) cost 5.4 us of gin_in_kernel's 38 (profiles/r6_gin_phases_start.txt).
    const int nv = min(16, nrows - part * 16);                   // live rows of this wave (wave-uniform)
    if (nv <= 0) return;
    const int g_first = wave_uniform(gid_lds[part * 16]), g_last = wave_uniform(gid_lds[part * 16 + nv - 1]);
    if (g_first == g_last) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = T[(part * 16 + k) * kLdt + c];
        double d[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) d[k] = (double)v[k];
#pragma unroll
        for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
            for (int k = 0; k < w; ++k) d[k] += d[k + w];
        atomicAdd(&pooled[(int64_t)g_first * H + c], d[0]);
        return;
    }
    int gids[16];                                    // (the tile's graph ids came with the tile's rows: no round trip here)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = part * 16 + k;
        gids[k] = r < nrows ? gid_lds[r] : -1;
    }
    double acc = 0.0;
    int cur = -1;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int g = gids[k];
        if (g < 0) break;
        if (g != cur) {
            if (cur >= 0) atomicAdd(&pooled[(int64_t)cur * H + c], acc);
            cur = g;
            acc = 0.0;
        }
        acc += (double)T[(part * 16 + k) * kLdt + c];
    }
    if (cur >= 0) atomicAdd(&pooled[(int64_t)cur * H + c], acc);
}

// ---- one wave: Z[16 rows][64] = X[16 rows][64] * W^T, X fragments xb[c] = X[row j][16c+4q .. +3]
// (j = lane & 15, q = lane >> 4), wf[cb][c] = W[16cb + j][16c+4q .. +3].
// Result acc[cb][r] = Z[row j][16cb + 4q + r].
__device__ __forceinline__ void mfma_rows16(const F4 xb[4], const F4 wf[4][4], f32x4 acc[4])
{
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            a = mfma_16x16x4_f32(wf[cb][c].x, xb[c].x, a);
            a = mfma_16x16x4_f32(wf[cb][c].y, xb[c].y, a);
            a = mfma_16x16x4_f32(wf[cb][c].z, xb[c].z, a);
            a = mfma_16x16x4_f32(wf[cb][c].w, xb[c].w, a);
        }
        acc[cb] = a;
    }
}

// W [64][kdim] row-major (nn.Linear.weight) -> this lane's fragments; columns >= kdim read as 0
__device__ __forceinline__ void load_w_frags(const float *W, int kdim, F4 wf[4][4])
{
    const int lane = lane_id(), j = lane & 15, q = lane >> 4;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int row = 16 * cb + j, k0 = 16 * c + 4 * q;
            const float *p = W + (int64_t)row * kdim + k0;
            if ((kdim & 3) == 0 && k0 + 3 < kdim) {
                wf[cb][c] = ld4(p);
            } else {
                wf[cb][c].x = k0 + 0 < kdim ? p[0] : 0.f;
                wf[cb][c].y = k0 + 1 < kdim ? p[1] : 0.f;
                wf[cb][c].z = k0 + 2 < kdim ? p[2] : 0.f;
                wf[cb][c].w = k0 + 3 < kdim ? p[3] : 0.f;
            }
        }
}

// transposed fragments for dX = dZ * W (backward): wt[cb][c] = W[16c+4q .. +3][16cb + j] i.e. the
// "weight" seen by the MFMA is W^T [kdim out][64 in]; rows >= kdim read as 0.
__device__ __forceinline__ void load_wt_frags(const float *W, int kdim, F4 wf[4][4])
{
    const int lane = lane_id(), j = lane & 15, q = lane >> 4;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int orow = 16 * cb + j;          // output index = input column of W
            const int k0 = 16 * c + 4 * q;         // reduction index = output row of W
            wf[cb][c].x = orow < kdim ? W[(int64_t)(k0 + 0) * kdim + orow] : 0.f;
            wf[cb][c].y = orow < kdim ? W[(int64_t)(k0 + 1) * kdim + orow] : 0.f;
            wf[cb][c].z = orow < kdim ? W[(int64_t)(k0 + 2) * kdim + orow] : 0.f;
            wf[cb][c].w = orow < kdim ? W[(int64_t)(k0 + 3) * kdim + orow] : 0.f;
        }
}

// bias add, masked store of the 16x64 block and per-channel sum / sum of squares of the
// valid rows into red[wave][2][64]
// one 16-channel block cb of the epilogue: + bias, store, per-wave column sums / sums of squares into red
__device__ __forceinline__ void epilogue_block4(int cb, const f32x4 &acc, F4 bias4, float *Z, int row,
                                                bool valid, float *red /* [128] of this wave */)
{
    const int lane = lane_id(), q = lane >> 4;
    const int ch = 16 * cb + 4 * q;
    F4 z;
    z.x = acc[0] + bias4.x;
    z.y = acc[1] + bias4.y;
    z.z = acc[2] + bias4.z;
    z.w = acc[3] + bias4.w;
    if (valid) st4(Z + (int64_t)row * H + ch, z);
    if (red) {
        float s[4] = {valid ? z.x : 0.f, valid ? z.y : 0.f, valid ? z.z : 0.f, valid ? z.w : 0.f};
        float ss[4] = {s[0] * s[0], s[1] * s[1], s[2] * s[2], s[3] * s[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {                      // the 16 rows of this wave's tile = one DPP row
            s[e] = row16_sum_last(s[e]);
            ss[e] = row16_sum_last(ss[e]);
        }
        if ((lane & 15) == 15) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { red[ch + e] = s[e]; red[H + ch + e] = ss[e]; }
        }
    }
}
__device__ __forceinline__ void epilogue_block(int cb, const f32x4 &acc, const float *bias, float *Z, int row,
                                               bool valid, float *red /* [128] of this wave */)
{
    const int ch = 16 * cb + 4 * (lane_id() >> 4);
    F4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (bias) { b4.x = bias[ch + 0]; b4.y = bias[ch + 1]; b4.z = bias[ch + 2]; b4.w = bias[ch + 3]; }
    epilogue_block4(cb, acc, b4, Z, row, valid, red);
}
__device__ __forceinline__ void epilogue_store_stats(f32x4 acc[4], const float *bias, float *Z, int row,
                                                     bool valid, float *red /* [128] of this wave */)
{
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) epilogue_block(cb, acc[cb], bias, Z, row, valid, red);
}

// Z[16 rows][64] = X W^T + bias with its epilogue, ONE 16-channel block at a time: only 4 weight fragments are live
// (mfma_rows16 + load_w_frags keep 16: 48 more VGPRs, which costs a wave of occupancy in the forward kernels)
__device__ __forceinline__ void linear_rows16_store_stats(const F4 xb[4], const float *W, int kdim, const float *bias,
                                                          float *Z, int row, bool valid, float *red)
{
    const int lane = lane_id(), j = lane & 15, q = lane >> 4;
#pragma unroll 1
    for (int cb = 0; cb < 4; ++cb) {
        F4 wf[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k0 = 16 * c + 4 * q;
            const float *p = W + (int64_t)(16 * cb + j) * kdim + k0;
            if ((kdim & 3) == 0 && k0 + 3 < kdim) {
                wf[c] = ld4(p);
            } else {
                wf[c].x = k0 + 0 < kdim ? p[0] : 0.f;
                wf[c].y = k0 + 1 < kdim ? p[1] : 0.f;
                wf[c].z = k0 + 2 < kdim ? p[2] : 0.f;
                wf[c].w = k0 + 3 < kdim ? p[3] : 0.f;
            }
        }
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            a = mfma_16x16x4_f32(wf[c].x, xb[c].x, a);
            a = mfma_16x16x4_f32(wf[c].y, xb[c].y, a);
            a = mfma_16x16x4_f32(wf[c].z, xb[c].z, a);
            a = mfma_16x16x4_f32(wf[c].w, xb[c].w, a);
        }
        epilogue_block(cb, a, bias, Z, row, valid, red);
    }
}

// ---- the same product with the weight matrix staged in LDS by the whole workgroup (one coalesced request, issued as
// early as the kernel can, instead of every wave fetching all 16 fragments from L2 block by block): Wl[64][kLdt],
// columns >= kdim zero.  stage_weights_request() returns the thread's 4 x 16 bytes, stage_weights_store() parks them.
struct WStage { F4 v[4]; };
__device__ __forceinline__ WStage stage_weights_request(const float *W, int kdim)
{
    WStage st;
    const int tid = (int)threadIdx.x;
    // every load unconditional with a clamped column; columns >= kdim are zeroed when the fragment is STORED, so that the
    // request carries no wait (per-element `c < kdim ? p[c] : 0` came out as one round trip per element: the first layer's
    // 49-column weight took ~8 dependent round trips to arrive)
    if ((kdim & 3) == 0) {                           // block-uniform: a quad is inside the row or outside it
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * kThreads, r = idx >> 4, c4 = 4 * (idx & 15);
            st.v[i] = ld4(W + (int64_t)r * kdim + min(c4, kdim - 4));
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * kThreads, r = idx >> 4, c4 = 4 * (idx & 15);
            const float *p = W + (int64_t)r * kdim;
            st.v[i] = F4{p[min(c4 + 0, kdim - 1)], p[min(c4 + 1, kdim - 1)], p[min(c4 + 2, kdim - 1)], p[min(c4 + 3, kdim - 1)]};
        }
    }
    return st;
}
__device__ __forceinline__ F4 stage_masked(const WStage &st, int i, int kdim)
{
    const int c4 = 4 * (((int)threadIdx.x + i * kThreads) & 15);
    const F4 v = st.v[i];
    return F4{c4 + 0 < kdim ? v.x : 0.f, c4 + 1 < kdim ? v.y : 0.f, c4 + 2 < kdim ? v.z : 0.f, c4 + 3 < kdim ? v.w : 0.f};
}
__device__ __forceinline__ void stage_weights_store(float *Wl, const WStage &st, int kdim)
{
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * kThreads, r = idx >> 4, c4 = 4 * (idx & 15);
        st4(&Wl[r * kLdt + c4], stage_masked(st, i, kdim));
    }
}
// transposed: Wt[c][r] = W[r][c] (the backward product dx = dz W reduces over W's rows)
__device__ __forceinline__ void stage_weights_store_t(float *Wt, const WStage &st, int kdim)
{
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * kThreads, r = idx >> 4, c4 = 4 * (idx & 15);
        const F4 v = stage_masked(st, i, kdim);
        Wt[(c4 + 0) * kLdt + r] = v.x;
        Wt[(c4 + 1) * kLdt + r] = v.y;
        Wt[(c4 + 2) * kLdt + r] = v.z;
        Wt[(c4 + 3) * kLdt + r] = v.w;
    }
}
// bias_lds: the 64 biases in LDS (zeros without a bias), staged with the weights: from global memory the four blocks'
// biases were four dependent round trips per tile
__device__ __forceinline__ void linear_rows16_lds_store_stats(const F4 xb[4], const float *Wl, const float *bias_lds,
                                                              float *Z, int row, bool valid, float *red)
{
    const int lane = lane_id(), j = lane & 15, q = lane >> 4;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const F4 wf = ld4(&Wl[(16 * cb + j) * kLdt + 16 * c + 4 * q]);
            a = mfma_16x16x4_f32(wf.x, xb[c].x, a);
            a = mfma_16x16x4_f32(wf.y, xb[c].y, a);
            a = mfma_16x16x4_f32(wf.z, xb[c].z, a);
            a = mfma_16x16x4_f32(wf.w, xb[c].w, a);
        }
        epilogue_block4(cb, a, ld4(&bias_lds[16 * cb + 4 * q]), Z, row, valid, red);
    }
}

// red[4 waves][128] -> fp64 atomics into stats[2][64]
__device__ __forceinline__ void flush_stats(const float *red, double *stats)
{
    const int tid = (int)threadIdx.x;
    if (tid < 2 * H) {
        const double v = (double)red[tid] + (double)red[128 + tid] + (double)red[256 + tid] + (double)red[384 + tid];
        atomicAdd(&stats[((int)blockIdx.x % kRep) * 2 * H + tid], v);
    }
}


// ---- neighbourhood sum over an LDS tile.  Precondition: T rows [0, nrows) hold the self term, rp_lds[0 .. nrows] the
// tile's row pointers, and a __syncthreads() separates those writes from this call.  Postcondition:
// T[r] = self + nbr_weight * sum_{u in row(tile0 + r)} feat(u); ends with a __syncthreads().
// The tile's EDGES, not its rows, are split evenly over the 16 lane groups (16 lanes x 16 B = one 256 B feature row per
// load): with rows per group the wave ran as many rounds as its longest row (ego-nets have hubs: 12-15 dependent round
// trips per tile for an average degree of 5; the gather was 49 of gin_in_kernel's 79 us).  A group sums its chunk in
// edge order; rows that lie inside the chunk are finished there, the chunk's first and last row go to side slots that
// one wave adds in group order afterwards -- a fixed order, so the result does not depend on timing.
// kGatherJ = feature rows a lane group has in flight (4 VGPRs each): 8 where the registers are there (backward kernels:
// 19 / 34 us against 20.5 / 36.5 with 4), 4 in gin_in_kernel, whose feat() carries two BatchNorm affines (8 spills 23
// VGPRs there: 59.5 against 53.6 us).
#ifndef GATHER_DBG
#define GATHER_DBG 0         // timing experiments only (wrong results): 1 no side-slot pass, 2 no row search, 4 no feature loads
#endif
// load(u) -> the raw 16 bytes of row u, xform(x) -> the feature (the BatchNorm affines of gin_in_kernel): kept apart so that
// the kGatherJ loads of a round are all requested before the first transform -- with one callable doing both, the
// transform's branch made the compiler wait for every load before requesting the next (4 round trips per round: 14 of
// gin_in_kernel's 50 us, profiles/r3_gather_ablations.txt)
// overlap(): block-uniform work that only READS T (gin_in_kernel's SumPooling), run after the first 16 targets of every lane
// group are requested and before anything here writes T: that round trip hides behind it.  Must end with a barrier that orders
// its reads of T before the writes below.
template <int kGatherJ, class Load, class Xform, class Overlap>
__device__ __forceinline__ void gather_tile(float *T, float *part /* [32 * H] */, int *prow /* [32] */, int nrows,
                                            const int32_t *col_idx, Load load, Xform xform, float nbr_weight,
                                            const int *rp_lds /* [nrows + 1] */, Overlap overlap)
{
    const int tid = (int)threadIdx.x, t = tid & 15, gi = tid >> 4, gbase = lane_id() & ~15;
    const int ebeg = rp_lds[0], eend = rp_lds[nrows];
    const int chunk = (eend - ebeg + 15) >> 4;          // edges per group (block-uniform)
    const int e0 = ebeg + gi * chunk, e1 = min(e0 + chunk, eend);
    if (t < 2) prow[gi * 2 + t] = -1;
    F4 acc = {0.f, 0.f, 0.f, 0.f};
    int cur = -1, first = -1;                            // group-uniform: row being summed, the chunk's first row
    auto flush = [&](bool last) {
        if (cur < 0) return;
        if (cur == first || last) {
            const int slot = gi * 2 + (cur == first ? 0 : 1);
            st4(&part[slot * H + 4 * t], acc);
            if (t == 0) prow[slot] = cur;
        } else {                                         // every edge of this row is in this group's chunk
            float *dst = &T[cur * kLdt + 4 * t];
            const F4 self = ld4(dst);
            F4 o;
            o.x = fmaf(nbr_weight, acc.x, self.x); o.y = fmaf(nbr_weight, acc.y, self.y);
            o.z = fmaf(nbr_weight, acc.z, self.z); o.w = fmaf(nbr_weight, acc.w, self.w);
            st4(dst, o);
        }
    };
    int idx_next = e0 + t < e1 ? col_idx[e0 + t] : -1;
    overlap();
    for (int c = 0; c < chunk; c += 16) {                // block-uniform trip counts (the shuffles need every lane)
        const int e = e0 + c + t;
        const bool valid = e < e1;
        const int idx = idx_next;
        idx_next = e + 16 < e1 ? col_idx[e + 16] : -1;   // the next 16 targets travel with this round's rows (a round trip per round before)
        int rid = 0;                                     // last r with rp_lds[r] <= e
        if (valid && !(GATHER_DBG & 2)) {
            int hi = nrows;
            while (hi - rid > 1) {
                const int mid = (rid + hi) >> 1;
                if (rp_lds[mid] <= e) rid = mid; else hi = mid;
            }
        }
        if (c == 0) first = wave_shfl(rid, gbase);       // (an empty chunk never flushes)
        for (int eb = 0; eb < 16 && c + eb < chunk; eb += kGatherJ) {
            F4 f[kGatherJ];
            int u[kGatherJ], rj[kGatherJ];
#pragma unroll
            for (int j = 0; j < kGatherJ; ++j) {
                u[j] = wave_shfl(idx, gbase + eb + j);
                rj[j] = wave_shfl(rid, gbase + eb + j);
            }
#pragma unroll
            for (int j = 0; j < kGatherJ; ++j) f[j] = (GATHER_DBG & 4) ? F4{1.f, 1.f, 1.f, 1.f} : load(u[j] < 0 ? 0 : u[j]);   // branch free: all loads in flight
#pragma unroll
            for (int j = 0; j < kGatherJ; ++j) f[j] = xform(f[j]);
#pragma unroll
            for (int j = 0; j < kGatherJ; ++j) {
                if (u[j] >= 0) {                         // group-uniform
                    if (rj[j] != cur) {
                        flush(false);
                        acc = F4{0.f, 0.f, 0.f, 0.f};
                        cur = rj[j];
                    }
                    acc = add4(acc, f[j]);
                }
            }
        }
    }
    flush(true);
    __syncthreads();
    if (tid < H && !(GATHER_DBG & 1)) {                  // one wave, channel per lane: the side slots in group order
        // (all 64 LDS reads requested first; a row that several consecutive slots add to -- the end of one group's chunk
        //  and the start of the next -- stays in a register: the same additions in the same order, 17 instead of 32
        //  dependent read-modify-writes: 3.7 -> ~1.2 us per tile)
        int rows[32];
        float pv[32];
#pragma unroll
        for (int slot = 0; slot < 32; ++slot) { rows[slot] = prow[slot]; pv[slot] = part[slot * H + tid]; }
        int cur = -1;
        float tv = 0.f;
#pragma unroll
        for (int slot = 0; slot < 32; ++slot) {
            const int r = rows[slot];                    // wave-uniform
            if (r >= 0) {
                if (r != cur) {
                    if (cur >= 0) T[cur * kLdt + tid] = tv;
                    cur = r;
                    tv = T[r * kLdt + tid];
                }
                tv = fmaf(nbr_weight, pv[slot], tv);
            }
        }
        if (cur >= 0) T[cur * kLdt + tid] = tv;
    }
    __syncthreads();
}

template <int kGatherJ, class Load, class Xform>
__device__ __forceinline__ void gather_tile(float *T, float *part, int *prow, int nrows, const int32_t *col_idx, Load load, Xform xform,
                                            float nbr_weight, const int *rp_lds)
{
    gather_tile<kGatherJ>(T, part, prow, nrows, col_idx, load, xform, nbr_weight, rp_lds, [] {});
}

// workgroups per pass of a tile kernel: enough for the pass's row capacity, at most kGridX, a multiple of 32 (first_tile's
// pattern); GCC_GIN_GRID overrides (timing experiments)
inline int tile_grid(int64_t node_cap)
{
    static int forced = -1;
    if (forced < 0) { const char *e = getenv("GCC_GIN_GRID"); forced = e ? atoi(e) : 0; }
    int64_t g = forced > 0 ? forced : (node_cap + kTile - 1) / kTile;
    g = (g + 31) / 32 * 32;
    return (int)(g < 32 ? 32 : g > kGridX ? kGridX : g);
}

// ---- dropout multiplier of linears_prediction[layer](pooled)[b][ch .. ch+3] (gin.py:230):
// explicit keep masks if given, else Philox (one call per 4 consecutive channels), else 1
struct DropCfg {
    const float *keep;
    uint64_t seed;
    int32_t philox, B;
    float p, inv_keep;
    const gcc_step_scalars *sc;      // replayed step (hipGraph): the Philox key is sc->dropout_seed
};
// the readout kernels call this once at their top: one uniform load, in flight with their first requests
__device__ __forceinline__ DropCfg drop_resolve(DropCfg d)
{
    if (d.sc && d.philox) d.seed += d.sc->dropout_seed;     // by-value seed = the pass's addend (E2E: the k pass's offset)
    return d;
}
__device__ __forceinline__ F4 drop_mul4(const DropCfg &d, int layer, int b, int ch)
{
    F4 m = {1.f, 1.f, 1.f, 1.f};
    if (d.keep) {
        m = ld4(d.keep + ((int64_t)layer * d.B + b) * H + ch);
        m.x *= d.inv_keep; m.y *= d.inv_keep; m.z *= d.inv_keep; m.w *= d.inv_keep;
    } else if (d.philox) {
        uint32_t x[4];
        philox4x32_10((uint32_t)(b * (H / 4) + (ch >> 2)), (uint32_t)layer, 0xD50Fu, 0u, (uint32_t)d.seed,
                      (uint32_t)(d.seed >> 32), x);
        const float thr = d.p * 16777216.0f;
        m.x = (float)(x[0] >> 8) >= thr ? d.inv_keep : 0.f;
        m.y = (float)(x[1] >> 8) >= thr ? d.inv_keep : 0.f;
        m.z = (float)(x[2] >> 8) >= thr ? d.inv_keep : 0.f;
        m.w = (float)(x[3] >> 8) >= thr ? d.inv_keep : 0.f;
    }
    return m;
}
inline DropCfg drop_cfg(const gcc_gin_pass &p)
{
    DropCfg d = {p.dropout_keep, p.dropout_seed, p.dropout_keep ? 0 : p.dropout_philox, p.batch_size,
                 p.w.dropout_p, 1.0f / (1.0f - p.w.dropout_p), p.scalars};
    return d;
}

// true hidden width of the model (<= 64; 0 = 64): the column count of every Linear that reads a hidden representation.
// Every per-channel array and every weight's ROW count stays 64: rows / channels >= hidden are zero padding kept by the caller.
inline int hidden_of(const gcc_gin_weights &w) { return w.hidden > 0 ? w.hidden : H; }

inline BnDev bn_dev(const gcc_bn &b, const double *stats, const double *totals = nullptr)
{
    BnDev d = {b.weight, b.bias, b.running_mean, b.running_var, b.num_batches_tracked, stats, totals};
    return d;
}
inline double *totals_of(const gcc_gin_pass &p, int layer, int which)   // [layer][bn a|b|c][2][64] or NULL
{
    return p.bn_totals ? p.bn_totals + ((int64_t)layer * 3 + which) * 2 * H : nullptr;
}
// BnDev of BatchNorm `which` (0 = mlp bn, 1 = apply_func bn, 2 = outer bn) of GIN layer `layer`.  The totals exist once
// the forward pass's readout kernel has run: the forward kernels add the replicas up themselves (with_totals = false)
inline BnDev bn_of(const gcc_gin_pass &p, const gcc_bn &b, int layer, int which, bool with_totals = true)
{
    return bn_dev(b, p.stats + ((int64_t)layer * 3 + which) * kRep * 2 * H, with_totals ? totals_of(p, layer, which) : nullptr);
}

inline double *stats_of(const gcc_gin_pass &p, int layer, int which)   // [layer][bn a|b|c][kRep][2][64]
{
    return p.stats + ((int64_t)layer * 3 + which) * kRep * 2 * H;
}

}  // namespace
