// gcc_amd/csrc/encoder.hip -- GIN encoder forward (gfx950).
//
// Replaces GraphEncoder.forward (gcc/models/graph_encoder.py:132-200) ->
// UnsupervisedGIN.forward (gcc/models/gin.py:213-232) and the DGL/torch kernels
// behind it (GINConv copy_u/sum SpMM, nn.Linear GEMMs, BatchNorm1d, SumPooling,
// linears_prediction, Dropout, F.normalize).
//
// The batches are small (N ~ 25k nodes, nnz ~ 1e5 at bsz 256), so a pass is
// launch/latency bound, not FLOP bound.  Design:
//   * BatchNorm in training mode needs statistics over all N nodes: every BN is
//     a kernel boundary.  Statistics are column sums accumulated with fp64
//     atomics (one per channel per workgroup); consumers turn them into
//     scale/shift on the fly, so normalised activations are never materialised:
//     only the Linear outputs z1, z2 (and agg for backward) are stored.
//   * per GIN layer:  in  : gather (h + sum_{u->v} h_u, h = relu(bn_c(relu(bn_b(z2')))))
//                           -> MFMA X W0^T + b0 -> z1, stats_a, SumPooling(h)
//                     mid : relu(bn_a(z1)) -> MFMA X W1^T + b1 -> z2, stats_b
//                     stat: relu(bn_b(z2)) -> stats_c
//   * GEMMs use the exact-f32 MFMA (v_mfma_f32_16x16x4_f32) with swapped
//     operands so that each lane ends up with 4 consecutive output channels of
//     one node (16-byte stores), and with a K permutation so that A/B fragments
//     are plain 16-byte row loads.
//   * node tiles (64 rows per workgroup, grid-strided because N lives on the
//     device); the gather splits a tile's EDGES (not its rows) evenly over the
//     workgroup's 16 lane groups, so hub rows do not serialise one wave
//     (encoder_common.h: gather_tile).
//   * several passes (query with model, key with model_ema) share each launch
//     (blockIdx.y = pass).
#include "encoder_common.h"

namespace {

// =========================================================================
// F0: assemble input features (graph_encoder.py:158-165); also zeroes the pass's accumulators (BatchNorm statistics
// replicas, pooled sums), which the kernels after it add to -- four memset launches per step otherwise
struct FeatArgs {
    const int32_t *node_off, *row_ptr, *graph_id, *seed_local;
    const float *pos, *emb;
    float *x0;
    float4 *zero_a, *zero_b;     // two regions to clear, in 16-byte units
    int64_t zero_a16, zero_b16;
    int32_t B, pos_dim, emb_dim, max_degree, mult;
    int32_t cap;                 // gcc_gin_pass.node_cap
};
struct FeatLaunch { FeatArgs p[kMaxPass]; };

__global__ __launch_bounds__(kThreads) void gin_feat_kernel(FeatLaunch L)
{
    TRAIN_STEP_WAVE_PRIORITY();
    const FeatArgs &a = L.p[blockIdx.y];
    const int tid = (int)threadIdx.x, t = tid & 15, gi = tid >> 4;
    constexpr int kR = kTile / 16;
    // the lane group's 4 rows in two round trips: (1) row extents, graph ids and the positional columns -- requested for the
    // workgroup's first tile TOGETHER with the node count (addresses clamped to the capacity, rows >= N dropped below) --, then
    // (2) everything that depends on them; every load unconditional: one row at a time this was 3 dependent round trips per row
    int vv[kR], r0[kR], r1[kR], gv[kR];
    float pv[kR][4];
    auto request = [&](int tile0) {
#pragma unroll
        for (int i = 0; i < kR; ++i) {
            vv[i] = cap_row(tile0 + gi + 16 * i, a.cap);
            r0[i] = a.row_ptr[vv[i]];
            r1[i] = a.row_ptr[vv[i] + 1];
            gv[i] = a.graph_id[vv[i]];
#pragma unroll
            for (int e = 0; e < 4; ++e)                                       // (block-uniform condition)
                pv[i][e] = a.pos_dim > 0 ? a.pos[(int64_t)vv[i] * a.pos_dim + min(4 * t + e, a.pos_dim - 1)] : 0.f;
        }
    };
    const int tf = first_tile();
    request(tf * kTile);
    const int N = a.node_off[a.B];
    SCHED_FENCE();
    {
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const int64_t stride = (int64_t)gridDim.x * kThreads;
        for (int64_t i = (int64_t)blockIdx.x * kThreads + tid; i < a.zero_a16; i += stride) a.zero_a[i] = z4;
        for (int64_t i = (int64_t)blockIdx.x * kThreads + tid; i < a.zero_b16; i += stride) a.zero_b[i] = z4;
    }
    for (TileWalk tw(N); tw.ti < tw.tend; tw.ti += tw.step) {
        const int tile0 = tw.ti * kTile;
        const int nrows = min(kTile, N - tile0);
        if (tw.ti != tf) request(tile0);
        int first[kR], sl[kR];
        float ev[kR][4];
        const int dtot = a.pos_dim + a.emb_dim;
#pragma unroll
        for (int i = 0; i < kR; ++i) {
            const int deg = (r1[i] - r0[i]) * a.mult;                         // g.in_degrees(), :154
            const int dcl = max(0, deg < a.max_degree ? deg : a.max_degree);  // clamp(0, max_degree), :161 (rows >= N hold anything)
            const int g = min(max(gv[i], 0), a.B - 1);
            first[i] = a.node_off[g];
            sl[i] = a.seed_local ? a.seed_local[g] : 0;                       // (block-uniform branch)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * t + e;
                ev[i][e] = a.emb[(int64_t)dcl * a.emb_dim + (c >= a.pos_dim && c < dtot ? c - a.pos_dim : 0)];
            }
        }
#pragma unroll
        for (int i = 0; i < kR; ++i) {
            if (gi + 16 * i >= nrows) continue;
            const bool is_seed = vv[i] == first[i] + sl[i];                   // ndata["seed"], data_util.py:234-238
            F4 x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * t + e;
                at(x, e) = c < a.pos_dim ? pv[i][e] : c < dtot ? ev[i][e] : (c == dtot && is_seed ? 1.f : 0.f);
            }
            st4(a.x0 + (int64_t)vv[i] * H + 4 * t, x);
        }
    }
}

// =========================================================================
// F1: gather + Linear0
struct InArgs {
    const int32_t *node_off, *row_ptr, *col_idx, *graph_id;
    const float *src;         // layer 0: x0; layer > 0: z2 of the previous layer
    BnDev bnb, bnc;           // previous layer's apply_func.bn and gnn.batch_norms (layer > 0)
    const float *w0, *b0;
    float *agg;               // or NULL
    float *z1;
    double *stats_a;
    double *pooled;           // SumPooling of this layer's input h (hidden_rep[layer], gin.py:216,228)
    int32_t B, first, kdim, training;
    float eps, nbr_weight;    // nbr_weight: edge multiplicity
    int32_t cap;              // gcc_gin_pass.node_cap
};
struct InLaunch { InArgs p[kMaxPass]; long long *ticks; };
#ifndef GIN_IN_LDS_W
#define GIN_IN_LDS_W 1       // linears.0's weight through LDS (+18 KiB per workgroup)
#endif
#ifndef GIN_DBG_SKIP
#define GIN_DBG_SKIP 0       // timing experiments only (wrong results): 1 no pooling, 2 no statistics flush, 4 no gather
#endif
static long long *g_gin_ticks = nullptr;   // diagnostics (gcc_gin_debug_ticks)
// phase ticks (diagnostics, tools/gin_phases.py): ticks[kind][phase 0..15][workgroup 0..2047] (workgroup = blockIdx.y * 1024 + blockIdx.x),
// kind 0 = gin_in first layer, 1 = gin_in other layers, 2 = gin_mid.  Every workgroup ADDS its own durations to its own slots with
// non-returning atomics of thread 0 on slots nobody else touches: no contention, and no load in the listing (tests/test_isa_chains.py).  (Rounds 1-5 used one atomic per
// phase on a shared slot: 780 workgroups queuing on one address cost more than the phases they timed.)  Phase 15 counts tiles.
constexpr int kTickWgs = 2048, kTickPhases = 16;
#define TICK_ADD(kind, ph, v) atomicAdd((unsigned long long *)&L.ticks[((kind) * kTickPhases + (ph)) * kTickWgs + (int)blockIdx.y * 1024 + (int)blockIdx.x], (unsigned long long)(v))
#define GIN_TICK(ph) do { if (L.ticks && tid == 0) { const long long now_ = device_ticks(); TICK_ADD(a.first ? 0 : 1, ph, now_ - tick_); tick_ = now_; } } while (0)

// (3 workgroups per CU by LDS -- 48.8 KiB with the staged weight -- so up to 168 registers are free: 8 gathered rows in flight)
#ifndef GIN_IN_PER_CU
#define GIN_IN_PER_CU 3
#endif
#ifndef GIN_GATHER_J
#define GIN_GATHER_J 8
#endif
__global__ __launch_bounds__(kThreads, GIN_IN_PER_CU) void gin_in_kernel(InLaunch L)
{
    TRAIN_STEP_WAVE_PRIORITY();
    __shared__ float T[kTile * kLdt];
    __shared__ __attribute__((aligned(16))) float part[32 * H];      // (also the fp64 scratch of bn_table)
    __shared__ float red[4 * 2 * H];
    __shared__ int prow[32];
    __shared__ int rpl[kTile + 1], gidl[kTile];
    const InArgs &a = L.p[blockIdx.y];
    const int tid = (int)threadIdx.x, t = tid & 15, gi = tid >> 4, lane = lane_id(), wv = tid >> 6;
    int N;
    __shared__ float tabb[2 * H], tabc[2 * H];
#if GIN_IN_LDS_W
    __shared__ float Wl[H * kLdt];                 // linears.0 weight, staged once per workgroup
    __shared__ __attribute__((aligned(16))) float bl[H];     // and its bias
#endif
    Aff4 ab, ac;
    long long tick_ = L.ticks ? device_ticks() : 0;
    const float *src = a.src;
    auto load = [&](int u) -> F4 { return ld4(src + (int64_t)u * H + 4 * t); };
    // The first tile's own rows, row pointers and graph ids are requested HERE, with the weights, the statistics and the node
    // count: one round trip (they were a second one, after the node count; addresses clamped to the capacity, rows >= N are
    // dropped when the tile is stored).  (all requested together and stored afterwards: a load under `if (tid < ...)` next to
    // its LDS store is a round trip of its own)
    const int tf = first_tile();
    int rp_own, gid_own;
    F4 own[kTile / 16];                           // the 4 rows of this lane group
    auto request = [&](int tile0) {
        rp_own = a.row_ptr[min(tile0 + min(tid, kTile), a.cap)];
        gid_own = a.graph_id[cap_row(tile0 + (tid & (kTile - 1)), a.cap)];
#pragma unroll
        for (int i = 0; i < kTile / 16; ++i) own[i] = load(cap_row(tile0 + gi + 16 * i, a.cap));
    };
    {
#if GIN_IN_LDS_W
        const WStage wst = stage_weights_request(a.w0, a.kdim);     // in flight with N and the statistics
        const float b_own = a.b0 ? a.b0[tid & (H - 1)] : 0.f;
#endif
        if (!a.first) {                            // block-uniform
            const BnReq rb = bn_request(a.bnb), rc = bn_request(a.bnc);
            request(tf * kTile);
            N = a.node_off[a.B];                   // (requested after the statistics: the wait for it is the wait for all)
            SCHED_FENCE();
            if (no_tiles(N)) return;
            bn_table_finish(tabb, rb, (double)N, a.eps, a.training, (double *)part);
            bn_table_finish(tabc, rc, (double)N, a.eps, a.training, (double *)part);
        } else {
            request(tf * kTile);
            N = ((const volatile int32_t *)a.node_off)[a.B];       // (volatile: an ordinary load is hoisted above the branch, ahead of the other arm's requests)
            if (no_tiles(N)) return;
        }
#if GIN_IN_LDS_W
        stage_weights_store(Wl, wst, a.kdim);
        if (tid < H) bl[tid] = b_own;
#endif
    }
    __syncthreads();
    if (!a.first) {
        ab = aff4_from_table(tabb, 4 * t);
        ac = aff4_from_table(tabc, 4 * t);
    }
    const bool first = a.first != 0;              // (block-uniform; the launch has one layer)
    auto xform = [&](F4 x) -> F4 { return first ? x : affine_relu(affine_relu(x, ab), ac); };   // h = relu(bn_c(relu(bn_b(z2))))  gin.py:56-57,219-220
    GIN_TICK(0);                                  // BatchNorm tables
    for (TileWalk tw(N); tw.ti < tw.tend; tw.ti += tw.step) {
        const int tile0 = tw.ti * kTile;
        const int nrows = min(kTile, N - tile0);
        if (L.ticks && tid == 0) TICK_ADD(a.first ? 0 : 1, 15, 1);
        // 1. own rows; the tile's row pointers and graph ids ride in the same round trip (the pooling and the gather
        //    would otherwise each start with one of their own)
        {
            if (tw.ti != tf) request(tile0);      // (a launch with more tiles than workgroups)
            if (tid <= nrows) rpl[tid] = rp_own;
            if (tid >= 128 && tid - 128 < nrows) gidl[tid - 128] = gid_own;
#pragma unroll
            for (int i = 0; i < kTile / 16; ++i) {
                const int r = gi + 16 * i;
                const F4 z = {0.f, 0.f, 0.f, 0.f};
                st4(&T[r * kLdt + 4 * t], r < nrows ? xform(own[i]) : z);
            }
        }
        __syncthreads();
        GIN_TICK(1);
        // 2. SumPooling of hidden_rep[layer] (gin.py:228): inside the gather, behind its first request of neighbour ids
        // 3. GINConv aggregate: (1 + eps) * h_v + sum_{u -> v} h_u, eps = 0 (gin.py:179-185,218)
        auto pooling = [&] {
#if !(GIN_DBG_SKIP & 1)
            if (a.pooled) pool_tile(T, nrows, a.pooled, gidl);
#endif
            lds_barrier();                         // (the pooling atomics stay in flight)
            GIN_TICK(2);
        };
#if !(GIN_DBG_SKIP & 4)
        gather_tile<GIN_GATHER_J>(T, part, prow, nrows, a.col_idx, load, xform, a.nbr_weight, rpl, pooling);
#else
        pooling();
#endif
        GIN_TICK(3);
        // 4. keep agg for the weight gradient of linears.0
        if (a.agg)
            for (int r = gi; r < nrows; r += 16) st4(a.agg + (int64_t)(tile0 + r) * H + 4 * t, ld4(&T[r * kLdt + 4 * t]));
        GIN_TICK(4);
        // 5. z1 = agg W0^T + b0 (gin.py:115: linears[0])
        {
            const int j = lane & 15, q = lane >> 4, rl = 16 * wv + j;
            F4 xb[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) xb[c] = ld4(&T[rl * kLdt + 16 * c + 4 * q]);
#if GIN_IN_LDS_W
            linear_rows16_lds_store_stats(xb, Wl, bl, a.z1, tile0 + rl, rl < nrows, &red[wv * 2 * H]);
#else
            linear_rows16_store_stats(xb, a.w0, a.kdim, a.b0, a.z1, tile0 + rl, rl < nrows, &red[wv * 2 * H]);
#endif
        }
        __syncthreads();
        GIN_TICK(5);
#if !(GIN_DBG_SKIP & 2)
        flush_stats(red, a.stats_a);
#endif
        lds_barrier();
        GIN_TICK(6);
    }
}

// =========================================================================
// F2: relu(bn_a(z1)) -> Linear1
struct MidArgs {
    const int32_t *node_off;
    const float *z1;
    BnDev bna;
    const float *w1, *b1;
    float *z2;
    double *stats_b;
    int32_t B, training;
    float eps;
    int32_t hid;              // columns of w1 (gcc_gin_weights.hidden: the true hidden width, <= 64)
    int32_t cap;              // gcc_gin_pass.node_cap
};
struct MidLaunch { MidArgs p[kMaxPass]; long long *ticks; };   // ticks: diagnostics (kind 2)
#define MID_TICK(ph) do { if (L.ticks && tid == 0) { const long long now_ = device_ticks(); TICK_ADD(2, ph, now_ - tick_); tick_ = now_; } } while (0)

__global__ __launch_bounds__(kThreads) void gin_mid_kernel(MidLaunch L)
{
    TRAIN_STEP_WAVE_PRIORITY();
    __shared__ __attribute__((aligned(16))) float red[4 * 2 * H];    // (also the fp64 scratch of bn_table)
    const MidArgs &a = L.p[blockIdx.y];
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = tid >> 6, j = lane & 15, q = lane >> 4;
    long long tick_ = L.ticks ? device_ticks() : 0;
    __shared__ float Wl[H * kLdt];
    const WStage wst = stage_weights_request(a.w1, a.hid);   // in flight with N and the statistics
    __shared__ float taba[2 * H];
    __shared__ __attribute__((aligned(16))) float bl[H];     // linears.1's bias
    const float b_own = a.b1 ? a.b1[tid & (H - 1)] : 0.f;
    const BnReq ra = bn_request(a.bna);
    // the first tile's row of z1 rides in the same round trip (clamped to the capacity; rows >= N are zeroed below)
    const int tf = first_tile();
    F4 xb[4];
    auto request = [&](int tile0) {
        const float *zrow = a.z1 + (int64_t)cap_row(tile0 + 16 * wv + j, a.cap) * H;
#pragma unroll
        for (int c = 0; c < 4; ++c) xb[c] = ld4(zrow + 16 * c + 4 * q);
    };
    request(tf * kTile);
    const int N = a.node_off[a.B];                           // (requested last: the wait for it is the wait for all)
    SCHED_FENCE();
    if (no_tiles(N)) return;
    MID_TICK(0);                                              // node count (scalar) arrived
    bn_table_finish(taba, ra, (double)N, a.eps, a.training, (double *)red);
    MID_TICK(1);                                              // statistics arrived, table computed
    stage_weights_store(Wl, wst, a.hid);
    if (tid < H) bl[tid] = b_own;
    __syncthreads();
    MID_TICK(2);                                              // weights in LDS
    Aff4 aa[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) aa[c] = aff4_from_table(taba, 16 * c + 4 * q);
    for (TileWalk tw(N); tw.ti < tw.tend; tw.ti += tw.step) {
        const int tile0 = tw.ti * kTile;
        const int row = tile0 + 16 * wv + j;
        const bool valid = row < N;
        if (tw.ti != tf) request(tile0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const F4 z = {0.f, 0.f, 0.f, 0.f};
            xb[c] = valid ? affine_relu(xb[c], aa[c]) : z;                                      // gin.py:115
        }
        if (L.ticks && tid == 0) TICK_ADD(2, 15, 1);
        MID_TICK(3);                                          // the tile's rows arrived, normalised
        linear_rows16_lds_store_stats(xb, Wl, bl, a.z2, row, valid, &red[wv * 2 * H]);      // gin.py:116
        MID_TICK(4);                                          // products, stores issued
        __syncthreads();
        MID_TICK(5);                                          // barrier (waits for the stores' acknowledgement)
        flush_stats(red, a.stats_b);
        lds_barrier();
        MID_TICK(6);
    }
}

// =========================================================================
// F3: statistics of y2 = relu(bn_b(z2)) for gnn.batch_norms[i] (gin.py:56-57,219)
struct StatArgs {
    const int32_t *node_off;
    const float *z2;
    BnDev bnb;
    double *stats_c;
    int32_t B, training;
    float eps;
    int32_t cap;              // gcc_gin_pass.node_cap
};
struct StatLaunch { StatArgs p[kMaxPass]; };

__global__ __launch_bounds__(kThreads) void gin_stat_kernel(StatLaunch L)
{
    TRAIN_STEP_WAVE_PRIORITY();
    __shared__ __attribute__((aligned(16))) float part[16 * 2 * H];  // (also the fp64 scratch of bn_table)
    const StatArgs &a = L.p[blockIdx.y];
    const int tid = (int)threadIdx.x, t = tid & 15, gi = tid >> 4;
    __shared__ float tabb[2 * H];
    const BnReq rb = bn_request(a.bnb);
    // the first tile's rows of z2 ride in the same round trip (clamped to the capacity; rows >= N are skipped below)
    const int tf = first_tile();
    F4 z4[kTile / 16];                               // the lane group's 4 rows, requested together
    auto request = [&](int tile0) {
#pragma unroll
        for (int i = 0; i < kTile / 16; ++i) z4[i] = ld4(a.z2 + (int64_t)cap_row(tile0 + gi + 16 * i, a.cap) * H + 4 * t);
    };
    request(tf * kTile);
    const int N = a.node_off[a.B];                           // (requested last: the wait for it is the wait for all)
    SCHED_FENCE();
    if (no_tiles(N)) return;
    bn_table_finish(tabb, rb, (double)N, a.eps, a.training, (double *)part);
    const Aff4 ab = aff4_from_table(tabb, 4 * t);
    F4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
    bool any = false;
    for (TileWalk tw(N); tw.ti < tw.tend; tw.ti += tw.step) {
        const int tile0 = tw.ti * kTile;
        any = true;
        if (tw.ti != tf) request(tile0);
#pragma unroll
        for (int i = 0; i < kTile / 16; ++i) {
            if (tile0 + gi + 16 * i < N) {
                const F4 y = affine_relu(z4[i], ab);
                s = add4(s, y);
                ss.x = fmaf(y.x, y.x, ss.x); ss.y = fmaf(y.y, y.y, ss.y);
                ss.z = fmaf(y.z, y.z, ss.z); ss.w = fmaf(y.w, y.w, ss.w);
            }
        }
    }
    if (any) {            // block-uniform
        st4(&part[gi * 2 * H + 4 * t], s);
        st4(&part[gi * 2 * H + H + 4 * t], ss);
        __syncthreads();
        if (tid < 2 * H) {
            double v = 0.0;
            for (int k = 0; k < 16; ++k) v += (double)part[k * 2 * H + tid];
            atomicAdd(&a.stats_c[((int)blockIdx.x % kRep) * 2 * H + tid], v);
        }
    }
}

// =========================================================================
// F4: SumPooling of the last hidden representation (gin.py:228, i = num_layers - 1); the earlier ones are pooled by
// gin_in_kernel, where the atomics' latency hides behind the gather (as launches of their own they take 22-33 us each,
// and on a second stream the event hand-offs cost more than they hide: profiles/r3_side_stream_probe.txt)
struct PoolArgs {
    const int32_t *node_off, *graph_id;
    const float *z2;
    BnDev bnb, bnc;
    double *pooled;
    int32_t B, training;
    float eps;
    int32_t cap;              // gcc_gin_pass.node_cap
};
struct PoolLaunch { PoolArgs p[kMaxPass]; };

__global__ __launch_bounds__(kThreads) void gin_pool_kernel(PoolLaunch L)
{
    TRAIN_STEP_WAVE_PRIORITY();
    __shared__ __attribute__((aligned(16))) float T[kTile * kLdt];   // (also the fp64 scratch of bn_table)
    __shared__ int gidl[kTile];
    const PoolArgs &a = L.p[blockIdx.y];
    const int tid = (int)threadIdx.x, t = tid & 15, gi = tid >> 4;
    __shared__ float tabb[2 * H], tabc[2 * H];
    const BnReq rb = bn_request(a.bnb), rc = bn_request(a.bnc);
    // the first tile's rows and graph ids ride in the same round trip (clamped to the capacity; rows >= N are dropped below)
    const int tf = first_tile();
    F4 z4[kTile / 16];
    int gid_own;
    auto request = [&](int tile0) {
#pragma unroll
        for (int i = 0; i < kTile / 16; ++i) z4[i] = ld4(a.z2 + (int64_t)cap_row(tile0 + gi + 16 * i, a.cap) * H + 4 * t);
        gid_own = a.graph_id[cap_row(tile0 + (tid & (kTile - 1)), a.cap)];
    };
    request(tf * kTile);
    const int N = a.node_off[a.B];                           // (requested last: the wait for it is the wait for all)
    SCHED_FENCE();
    if (no_tiles(N)) return;
    bn_table_finish(tabb, rb, (double)N, a.eps, a.training, (double *)T);
    bn_table_finish(tabc, rc, (double)N, a.eps, a.training, (double *)T);
    const Aff4 ab = aff4_from_table(tabb, 4 * t);
    const Aff4 ac = aff4_from_table(tabc, 4 * t);
    for (TileWalk tw(N); tw.ti < tw.tend; tw.ti += tw.step) {
        const int tile0 = tw.ti * kTile;
        const int nrows = min(kTile, N - tile0);
        if (tw.ti != tf) request(tile0);
        if (tid < nrows) gidl[tid] = gid_own;
#pragma unroll
        for (int i = 0; i < kTile / 16; ++i) {
            const F4 z = {0.f, 0.f, 0.f, 0.f};        // (rows past the tile's end are zero: pool_tile's one-graph path sums all 16 rows of a wave)
            st4(&T[(gi + 16 * i) * kLdt + 4 * t], gi + 16 * i < nrows ? affine_relu(affine_relu(z4[i], ab), ac) : z);
        }
        __syncthreads();
        pool_tile(T, nrows, a.pooled, gidl);
        __syncthreads();
    }
}

// =========================================================================
// F5: readout (gin.py:223-232) + F.normalize (graph_encoder.py:195-196) + running statistics.
// score[B, 64] = sum_i drop(pooled_i W_i^T + b_i) is a [B x d] x [d x 64] GEMM per hidden_rep: one wave
// = 16 graphs on the exact-f32 MFMA (the per-lane GEMV it replaces read the weights uncoalesced: 69 us).
struct ReadArgs {
    const int32_t *node_off;
    const double *pooled;       // [L+1][B][64]
    const float *pred_w[GCC_GIN_MAX_LAYERS + 1], *pred_b[GCC_GIN_MAX_LAYERS + 1];
    DropCfg drop;
    float *score, *feat;
    BnDev bn[3 * GCC_GIN_MAX_LAYERS];
    int32_t B, nlayers, kdim0, normalize, update_running;
    int32_t hid;                // columns of pred_w[i > 0] (the true hidden width)
    float norm_eps, momentum;
    double *totals;             // [3 * nlayers][2][64] or NULL: the statistics replicas added up, for the backward pass
};
struct ReadLaunch { ReadArgs p[kMaxPass]; };

__global__ __launch_bounds__(kThreads) void gin_readout_kernel(ReadLaunch L)
{
    TRAIN_STEP_WAVE_PRIORITY();
    // 16 graphs per workgroup; the prediction layers are split over the four waves (wave w: layers w, w + 4, ...) and
    // the partial scores meet in LDS, so the chain of dependent loads is 2 layers long instead of 5
    __shared__ __attribute__((aligned(16))) float part[4][16][H + 4];
    const ReadArgs &a = L.p[blockIdx.y];
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = tid >> 6, j = lane & 15, q = lane >> 4;
    const int b = (int)blockIdx.x * 16 + j;
    const bool valid = b < a.B;
    const DropCfg drop = drop_resolve(a.drop);
    // the statistics of BatchNorm blockIdx.x (added up at the end of this kernel) are requested now: in flight with the scores
    const bool bn_work = (a.update_running || a.totals) && (int)blockIdx.x < 3 * a.nlayers;     // block-uniform
    RepReq rq = {};
    float rm0 = 0.f, rv0 = 0.f;
    if (bn_work) {       // (not the node count: the compiler moves it to a scalar register at once, a wait for everything here)
        const BnDev &bn0 = a.bn[blockIdx.x];
        rq = rep_request(bn0.stats, 2 * H);
        rm0 = bn0.running_mean[tid & (H - 1)];
        rv0 = bn0.running_var[tid & (H - 1)];
    }
    F4 score[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) { F4 z = {0.f, 0.f, 0.f, 0.f}; score[cb] = z; }
    for (int i = wave_uniform(wv); i <= a.nlayers; i += 4) {       // (scalar: pred_w[i] / pred_b[i] are scalar loads, not a round trip)
        const int kd = i == 0 ? a.kdim0 : a.hid;
        F4 wf[4][4];
        load_w_frags(a.pred_w[i], kd, wf);                         // weights and biases first: the fp64 -> f32 conversions
        F4 bias4[4];                                               // below wait for whatever was requested before them
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) bias4[cb] = ld4(a.pred_b[i] + 16 * cb + 4 * q);
        F4 xb[4];
        const double *prow = a.pooled + ((int64_t)i * a.B + (valid ? b : 0)) * H + 4 * q;   // (unconditional loads, masked afterwards)
        double pd[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) pd[c][e] = prow[16 * c + e];
        SCHED_FENCE();                                             // (all requested before the first conversion waits)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const F4 x = {(float)pd[c][0], (float)pd[c][1], (float)pd[c][2], (float)pd[c][3]}, z = {0.f, 0.f, 0.f, 0.f};
            xb[c] = valid ? x : z;
        }
        f32x4 acc[4];
        mfma_rows16(xb, wf, acc);                                  // linears_prediction[i](pooled_h)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int ch = 16 * cb + 4 * q;
            const F4 bias = bias4[cb];
            const F4 m = valid ? drop_mul4(drop, i, b, ch) : bias;   // self.drop, gin.py:230
            score[cb].x += (acc[cb][0] + bias.x) * m.x;
            score[cb].y += (acc[cb][1] + bias.y) * m.y;
            score[cb].z += (acc[cb][2] + bias.z) * m.z;
            score[cb].w += (acc[cb][3] + bias.w) * m.w;
        }
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) st4(&part[wv][j][16 * cb + 4 * q], score[cb]);
    __syncthreads();
    if (wv == 0) {                                   // layer order 0, 1, 2, ... inside each wave, then waves 0..3: fixed
        float ss = 0.f;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            F4 t = ld4(&part[0][j][16 * cb + 4 * q]);
            for (int w2 = 1; w2 < 4; ++w2) t = add4(t, ld4(&part[w2][j][16 * cb + 4 * q]));
            score[cb] = t;
            ss += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
        }
        ss += wave_shfl_xor(ss, 16);
        ss += wave_shfl_xor(ss, 32);
        const float inv = a.normalize ? 1.0f / fmaxf(sqrtf(ss), a.norm_eps) : 1.0f;   // F.normalize(p=2, eps=1e-5)
        if (valid) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const int ch = 16 * cb + 4 * q;
                st4(a.score + (int64_t)b * H + ch, score[cb]);
                F4 o = {score[cb].x * inv, score[cb].y * inv, score[cb].z * inv, score[cb].w * inv};
                st4(a.feat + (int64_t)b * H + ch, o);
            }
        }
    }
    // BatchNorm k by workgroup k % gridDim.x: the replicas of its batch statistics added up once for the ~16 kernels of
    // the backward pass that need them, and the running statistics (torch: momentum 0.1, unbiased variance)
    if (bn_work) {                                            // block-uniform
        const int n_nodes = a.node_off[a.B];
        double *scratch = (double *)&part[0][0][0];
        for (int k = (int)blockIdx.x; k < 3 * a.nlayers; k += (int)gridDim.x) {
            __syncthreads();                                   // the partial scores / the previous BatchNorm's sums are done with
            const BnDev &bn = a.bn[k];
            float rm = rm0, rv = rv0;
            if (k == (int)blockIdx.x) {                        // requested at the top of the kernel
                scratch[tid] = rep_sum(rq);
                __syncthreads();
            } else {                                           // (batches of fewer than 16 * 3 * nlayers graphs)
                replica_sums128(bn.stats, 2 * H, scratch);
                rm = bn.running_mean[tid & (H - 1)];
                rv = bn.running_var[tid & (H - 1)];
            }
            if (tid < H) {
                const int c = tid;
                const double n = (double)n_nodes;
                const double s1 = scratch[c] + scratch[128 + c], s2 = scratch[H + c] + scratch[128 + H + c];
                if (a.totals) { a.totals[(int64_t)k * 2 * H + c] = s1; a.totals[(int64_t)k * 2 * H + H + c] = s2; }
                if (a.update_running) {
                    const double mean = s1 / n;
                    double var = s2 / n - mean * mean;
                    if (var < 0.0) var = 0.0;
                    const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
                    bn.running_mean[c] = (float)((1.0 - a.momentum) * (double)rm + a.momentum * mean);
                    bn.running_var[c] = (float)((1.0 - a.momentum) * (double)rv + a.momentum * unb);
                    if (c == 0 && bn.nbt) bn.nbt[0] += 1;
                }
            }
        }
    }
}

}  // namespace

extern "C" void gcc_gin_debug_ticks(long long *device_ticks64) { g_gin_ticks = device_ticks64; }   /* diagnostics only */

extern "C" int32_t gcc_gin_forward(const gcc_gin_pass *passes, int32_t npass, gcc_prof *prof, void *stream)
{
    if (!passes || npass < 1 || npass > kMaxPass) {
        snprintf(g_err, kErrLen, "gcc_gin_forward: npass must be 1..%d", kMaxPass);
        return -1;
    }
    const int Lg = passes[0].w.num_gin_layers;
    int maxB = 0;
    for (int i = 0; i < npass; ++i) {
        const gcc_gin_pass &p = passes[i];
        const int din = p.w.pos_dim + p.w.deg_emb_dim + 1;
        if (p.w.num_gin_layers != Lg || Lg < 1 || Lg > GCC_GIN_MAX_LAYERS || din > H || p.batch_size < 1) {
            snprintf(g_err, kErrLen, "gcc_gin_forward: unsupported shape (layers=%d d_in=%d B=%d)",
                     p.w.num_gin_layers, din, p.batch_size);
            return -2;
        }
        if (p.node_cap < 1 || p.node_cap > 0x7fffffff) {        // (ABI 3: the kernels clamp their speculative first-tile requests to it)
            snprintf(g_err, kErrLen, "gcc_gin_forward: gcc_gin_pass.node_cap = %lld (the row capacity of the pass's buffers is required)",
                     (long long)p.node_cap);
            return -2;
        }
        maxB = p.batch_size > maxB ? p.batch_size : maxB;
    }
    hipStream_t s = (hipStream_t)stream;
    prof_mark(prof, 0, s);
    int64_t rows_max = 0;                            // workgroups per pass: for the rows expected (rows_hint), else for the capacity
    for (int i = 0; i < npass; ++i) {
        const gcc_gin_pass &p = passes[i];
        const int64_t r = p.rows_hint > 0 && p.rows_hint < p.node_cap ? p.rows_hint : p.node_cap;
        rows_max = r > rows_max ? r : rows_max;
    }
    const dim3 grid(tile_grid(rows_max), npass), block(kThreads);
    {
        FeatLaunch L;
        for (int i = 0; i < npass; ++i) {
            const gcc_gin_pass &p = passes[i];
            L.p[i] = {p.node_off, p.row_ptr, p.graph_id, p.seed_local, p.pos, p.w.degree_embedding, p.x0,
                      (float4 *)p.stats, (float4 *)p.pooled, (int64_t)Lg * 3 * kRep * 2 * H * (int64_t)sizeof(double) / 16,
                      (int64_t)(Lg + 1) * p.batch_size * H * (int64_t)sizeof(double) / 16,
                      p.batch_size, p.w.pos_dim, p.w.deg_emb_dim, p.w.max_degree, p.edge_multiplicity > 1 ? p.edge_multiplicity : 1,
                      (int32_t)p.node_cap};
        }
        hipLaunchKernelGGL(gin_feat_kernel, grid, block, 0, s, L);
    }
    for (int l = 0; l < Lg; ++l) {
        {
            InLaunch L;
            for (int i = 0; i < npass; ++i) {
                const gcc_gin_pass &p = passes[i];
                InArgs a;
                a.node_off = p.node_off; a.row_ptr = p.row_ptr; a.col_idx = p.col_idx; a.graph_id = p.graph_id;
                a.src = l == 0 ? p.x0 : p.z2[l - 1];
                a.bnb = l == 0 ? BnDev() : bn_of(p, p.w.bn_b[l - 1], l - 1, 1, false);
                a.bnc = l == 0 ? BnDev() : bn_of(p, p.w.bn_c[l - 1], l - 1, 2, false);
                a.w0 = p.w.lin0_w[l]; a.b0 = p.w.lin0_b[l];
                a.agg = p.agg[l]; a.z1 = p.z1[l];
                a.stats_a = stats_of(p, l, 0);
                a.pooled = p.pooled + (int64_t)l * p.batch_size * H;
                a.B = p.batch_size; a.first = l == 0;
                a.kdim = l == 0 ? p.w.pos_dim + p.w.deg_emb_dim + 1 : hidden_of(p.w);
                a.training = p.training; a.eps = p.w.bn_eps;
                a.nbr_weight = p.edge_multiplicity > 1 ? (float)p.edge_multiplicity : 1.0f;
                a.cap = (int32_t)p.node_cap;
                L.p[i] = a;
            }
            L.ticks = g_gin_ticks;
            hipLaunchKernelGGL(gin_in_kernel, grid, block, 0, s, L);
        }
        {
            MidLaunch L;
            for (int i = 0; i < npass; ++i) {
                const gcc_gin_pass &p = passes[i];
                L.p[i] = {p.node_off, p.z1[l], bn_of(p, p.w.bn_a[l], l, 0, false), p.w.lin1_w[l], p.w.lin1_b[l],
                          p.z2[l], stats_of(p, l, 1), p.batch_size, p.training,
                          p.w.bn_eps, hidden_of(p.w), (int32_t)p.node_cap};
            }
            L.ticks = g_gin_ticks;
            hipLaunchKernelGGL(gin_mid_kernel, grid, block, 0, s, L);
        }
        {
            StatLaunch L;
            bool need = false;
            for (int i = 0; i < npass; ++i) {
                const gcc_gin_pass &p = passes[i];
                L.p[i] = {p.node_off, p.z2[l], bn_of(p, p.w.bn_b[l], l, 1, false), stats_of(p, l, 2), p.batch_size, p.training, p.w.bn_eps,
                          (int32_t)p.node_cap};
                need = need || p.training;
            }
            if (need) hipLaunchKernelGGL(gin_stat_kernel, grid, block, 0, s, L);
        }
    }
    {
        PoolLaunch L;
        for (int i = 0; i < npass; ++i) {
            const gcc_gin_pass &p = passes[i];
            L.p[i] = {p.node_off, p.graph_id, p.z2[Lg - 1], bn_of(p, p.w.bn_b[Lg - 1], Lg - 1, 1, false),
                      bn_of(p, p.w.bn_c[Lg - 1], Lg - 1, 2, false), p.pooled + (int64_t)Lg * p.batch_size * H,
                      p.batch_size, p.training, p.w.bn_eps, (int32_t)p.node_cap};
        }
        hipLaunchKernelGGL(gin_pool_kernel, grid, block, 0, s, L);
    }
    {
        ReadLaunch L;
        for (int i = 0; i < npass; ++i) {
            const gcc_gin_pass &p = passes[i];
            ReadArgs a;
            a.node_off = p.node_off; a.pooled = p.pooled;
            for (int k = 0; k <= Lg; ++k) { a.pred_w[k] = p.w.pred_w[k]; a.pred_b[k] = p.w.pred_b[k]; }
            a.drop = drop_cfg(p); a.score = p.score; a.feat = p.feat;
            for (int l = 0; l < Lg; ++l) {
                a.bn[3 * l + 0] = bn_of(p, p.w.bn_a[l], l, 0, false);
                a.bn[3 * l + 1] = bn_of(p, p.w.bn_b[l], l, 1, false);
                a.bn[3 * l + 2] = bn_of(p, p.w.bn_c[l], l, 2, false);
            }
            a.totals = p.training ? p.bn_totals : nullptr;
            a.B = p.batch_size; a.nlayers = Lg; a.kdim0 = p.w.pos_dim + p.w.deg_emb_dim + 1; a.hid = hidden_of(p.w);
            a.normalize = p.normalize; a.update_running = p.training && p.update_running_stats;
            a.norm_eps = p.w.norm_eps; a.momentum = p.w.bn_momentum;
            L.p[i] = a;
        }
        hipLaunchKernelGGL(gin_readout_kernel, dim3((maxB + 15) / 16, npass), block, 0, s, L);
    }
    prof_mark(prof, 1, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, kErrLen, "gcc_gin_forward: launch failed: %s", hipGetErrorString(e));
        return -10;
    }
    return 0;
}
