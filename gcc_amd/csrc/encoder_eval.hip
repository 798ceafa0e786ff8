// gcc_amd/csrc/encoder_eval.hip -- the eval-mode GIN encoder of generate.py as ONE launch: SURVEY.md 8(f)#2's
// "per-subgraph LDS-resident megakernel".  Reference: generate.py:33-53 (test_moco: model.eval(), feat_q = model(graph_q),
// feat_k = model(graph_k), emb = (feat_q + feat_k) / 2) -> GraphEncoder.forward gcc/models/graph_encoder.py:132-200 ->
// UnsupervisedGIN.forward gcc/models/gin.py:213-232 with every BatchNorm on its running statistics (gin.py:54-58,113-116).
//
// In eval mode nothing couples two subgraphs of a batch (training-mode BatchNorm did: 12 batch-wide statistics per pass
// are why gcc_gin_forward is a chain of 15 launches), so one workgroup takes ONE subgraph through feature assembly, all
// GIN layers, the pooled readout and F.normalize:
//   * the subgraph's hidden representation [n, 64] f32 lives in LDS (n <= kEvalCap rows; larger ego-nets -- ~2 % at
//     rw_hops 256 -- gather from the L2-resident global copy instead, same code);
//   * a layer walks the subgraph's 64-row tiles: own rows + neighbour sum (gather_tile of encoder_common.h: the tile's
//     EDGES split over the 16 lane groups, fixed summation order) -> linears.0 on the exact-f32 MFMA -> BatchNorm affine +
//     ReLU in registers (the MFMA's output layout IS the next product's input layout) -> linears.1 -> two more affines ->
//     the new rows go to a global ping-pong buffer and are mirrored into LDS after the layer's last tile;
//   * SumPooling of every hidden_rep in fp64, the five prediction layers, normalisation, and -- with two passes in the
//     launch -- mean_out[b] += feat / 2, i.e. generate.py:52's (feat_q + feat_k) / 2.
// Same arithmetic as gcc_gin_forward with training = 0 (same affine tables, same MFMA sequences; the gather's summation
// order differs because tiles start at the subgraph, not at multiples of 64 of the batch): agreement ~1e-6.
#include "encoder_common.h"

namespace {

constexpr int kEvalCap = 256;            // rows of a subgraph kept in LDS
constexpr int kEvalLd = 68;              // floats per LDS row (272 B: 16-byte aligned, rows 8 apart share a bank group)

struct EvalLayer {
    const float *w0, *b0, *w1, *b1;
    const float *bn_w[3], *bn_b[3], *bn_rm[3], *bn_rv[3];     // mlp.batch_norms.0, apply_func.bn, gnn.batch_norms.i
};
struct EvalArgs {
    const int32_t *node_off, *row_ptr, *col_idx, *seed_local;
    const float *pos, *emb;
    float *g0, *g1;                      // global ping-pong [node_cap][64] (the pass's z1[0] / z2[0] buffers)
    double *pooled;                      // [L + 1][B][64] or NULL
    float *score, *feat;                 // [B][64]
    float *mean_out;                     // [B][64] or NULL: += mean_w * feat (zeroed by the host side of the call)
    float mean_w;
    EvalLayer layer[GCC_GIN_MAX_LAYERS];
    const float *pred_w[GCC_GIN_MAX_LAYERS + 1], *pred_b[GCC_GIN_MAX_LAYERS + 1];
    int32_t B, L, pos_dim, emb_dim, max_degree, mult, normalize, hid, kdim0;
    float eps, norm_eps;
};
struct EvalLaunch { EvalArgs p[kMaxPass]; long long *ticks; int32_t split; };   // split: small subgraphs are gin_eval_small_kernel's
static long long *g_eval_ticks = nullptr;    // diagnostics (gcc_gin_eval_debug_ticks): device int64[16]
#define EV_TICK(ph) do { if (Ln.ticks && tid == 0) { const long long now_ = device_ticks(); atomicAdd((unsigned long long *)&Ln.ticks[(ph)], (unsigned long long)(now_ - tick_)); tick_ = now_; } } while (0)

// ---- pieces shared by the two kernel shapes ------------------------------------------------------------------------------
// a layer's weights / BatchNorm numbers are requested one layer ahead (registers) and stored when the LDS buffers are free
struct LayerRegs { WStage s0, s1; float v0, v1, v2, v3; };
__device__ __forceinline__ LayerRegs eval_request_layer(const EvalArgs &a, int l)
{
    const EvalLayer &ly = a.layer[l];
    const int tid = (int)threadIdx.x;
    LayerRegs r;
    r.s0 = stage_weights_request(ly.w0, l == 0 ? a.kdim0 : a.hid);
    r.s1 = stage_weights_request(ly.w1, a.hid);
    const int c = tid & 63, which = tid >> 6;                    // which < 3: a BatchNorm; 3: the two biases
    r.v2 = r.v3 = 0.f;
    if (which < 3) { r.v0 = ly.bn_w[which][c]; r.v1 = ly.bn_b[which][c]; r.v2 = ly.bn_rm[which][c]; r.v3 = ly.bn_rv[which][c]; }
    else { r.v0 = ly.b0 ? ly.b0[c] : 0.f; r.v1 = ly.b1 ? ly.b1[c] : 0.f; }
    return r;
}
__device__ __forceinline__ void eval_store_layer(const EvalArgs &a, int l, const LayerRegs &r, float *Wl0, float *Wl1, float *tab,
                                                 float *bias)
{
    const int tid = (int)threadIdx.x;
    stage_weights_store(Wl0, r.s0, l == 0 ? a.kdim0 : a.hid);
    stage_weights_store(Wl1, r.s1, a.hid);
    const int c = tid & 63, which = tid >> 6;
    if (which < 3) {                                             // bn_scale_shift_from(training = 0), encoder_common.h
        const double rstd = 1.0 / sqrt((double)r.v3 + (double)a.eps);
        tab[which * 2 * H + c] = (float)((double)r.v0 * rstd);
        tab[which * 2 * H + H + c] = (float)((double)r.v1 - (double)r.v2 * (double)r.v0 * rstd);
    } else {
        bias[c] = r.v0;
        bias[H + c] = r.v1;
    }
}
// input features of local row r (graph_encoder.py:158-165), channels 4 t .. 4 t + 3
__device__ __forceinline__ F4 eval_feature4(const EvalArgs &a, int n0, int r, int sl, int t)
{
    const int dtot = a.pos_dim + a.emb_dim;
    const int v = n0 + r;
    const int deg = (a.row_ptr[v + 1] - a.row_ptr[v]) * a.mult;                  // g.in_degrees(), :154
    const int dcl = deg < a.max_degree ? deg : a.max_degree;                     // clamp(0, max_degree), :161
    F4 x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = 4 * t + e;
        const float *src = c < a.pos_dim ? a.pos + (int64_t)v * a.pos_dim + c
                                         : a.emb + (int64_t)dcl * a.emb_dim + (c < dtot ? c - a.pos_dim : 0);
        const float val = *src;
        at(x, e) = c < dtot ? val : (c == dtot && r == sl ? 1.f : 0.f);           // ndata["seed"], data_util.py:234-238
    }
    return x;
}
// the MLP and the three BatchNorm / ReLU stages on a wave's 16 rows, in registers: xb[c] = agg[row j][16 c + 4 q ..] in,
// h[cb] = the new representation's channels 16 cb + 4 q .. out (the MFMA's output layout is the next product's input layout)
__device__ __forceinline__ void eval_mlp_rows16(const F4 xb[4], const float *Wl0, const float *Wl1, const float *tab,
                                                const float *bias, int j, int q, F4 h[4])
{
    F4 y[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {                             // z1 = agg W0^T + b0; relu(bn_a(z1))  (gin.py:113-116)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const F4 wf = ld4(&Wl0[(16 * cb + j) * kLdt + 16 * c + 4 * q]);
            acc = mfma_16x16x4_f32(wf.x, xb[c].x, acc);
            acc = mfma_16x16x4_f32(wf.y, xb[c].y, acc);
            acc = mfma_16x16x4_f32(wf.z, xb[c].z, acc);
            acc = mfma_16x16x4_f32(wf.w, xb[c].w, acc);
        }
        const int ch = 16 * cb + 4 * q;
        const F4 b4 = ld4(&bias[ch]);
        const F4 z = {acc[0] + b4.x, acc[1] + b4.y, acc[2] + b4.z, acc[3] + b4.w};
        y[cb] = affine_relu(z, aff4_from_table(tab, ch));
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {                             // z2 = y W1^T + b1; relu(bn_b(z2)); relu(bn_c(.))  (gin.py:55-57,219-220)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const F4 wf = ld4(&Wl1[(16 * cb + j) * kLdt + 16 * c + 4 * q]);
            acc = mfma_16x16x4_f32(wf.x, y[c].x, acc);
            acc = mfma_16x16x4_f32(wf.y, y[c].y, acc);
            acc = mfma_16x16x4_f32(wf.z, y[c].z, acc);
            acc = mfma_16x16x4_f32(wf.w, y[c].w, acc);
        }
        const int ch = 16 * cb + 4 * q;
        const F4 b4 = ld4(&bias[H + ch]);
        const F4 z = {acc[0] + b4.x, acc[1] + b4.y, acc[2] + b4.z, acc[3] + b4.w};
        h[cb] = affine_relu(affine_relu(z, aff4_from_table(tab + 2 * H, ch)), aff4_from_table(tab + 4 * H, ch));
    }
}
// readout: score = sum_i linears_prediction[i](pooled_i) (gin.py:227-230; eval: dropout is the identity), F.normalize
// (graph_encoder.py:195-196), the mean over the passes, the pooled sums.  pool: LDS [L + 1][64] fp64; ppart: LDS scratch
__device__ __forceinline__ void eval_readout(const EvalArgs &a, int b, const double *pool, double *ppart)
{
    const int tid = (int)threadIdx.x, L = a.L;
    const int o = tid & 63, pt = tid >> 6;
    float s = 0.f;
    for (int i = pt; i <= L; i += 4) {
        const int kd = i == 0 ? a.kdim0 : a.hid;
        const float *w = a.pred_w[i] + (int64_t)o * kd;
        float acc = a.pred_b[i] ? a.pred_b[i][o] : 0.f;
        for (int k = 0; k < kd; ++k) acc = fmaf(w[k], (float)pool[i * H + k], acc);
        s += acc;
    }
    float *sp = (float *)ppart;                                  // [4][64] partial scores
    sp[pt * H + o] = s;
    __syncthreads();
    if (tid < H) {
        const float sc = (sp[tid] + sp[H + tid]) + (sp[2 * H + tid] + sp[3 * H + tid]);
        float ss = sc * sc;
        ss = wave_sum(ss);                                       // (the first wave holds all 64 channels)
        float f = sc;
        if (a.normalize) {
            const float nrm = sqrtf(ss);
            f = sc / (nrm > a.norm_eps ? nrm : a.norm_eps);
        }
        a.score[(int64_t)b * H + tid] = sc;
        a.feat[(int64_t)b * H + tid] = f;
        if (a.mean_out) atomicAdd(&a.mean_out[(int64_t)b * H + tid], a.mean_w * f);
        if (a.pooled) for (int i = 0; i <= L; ++i) a.pooled[((int64_t)i * a.B + b) * H + tid] = pool[i * H + tid];
    }
}

// ---- small subgraphs (n <= 64 rows, at most kSmallEdges CSR entries): the common case (median ego-net 23 .. 55 nodes) ----------
// One tile, nothing but the weights comes from global memory after the set-up: the subgraph's local column ids sit in LDS as
// bytes, every lane sums ITS row's neighbours straight into the registers the first matrix product reads (lane (j, q) of wave w
// owns row 16 w + j, channels 16 c + 4 q ..: the four lanes of a row read different quads of a neighbour's row, so nothing is
// read twice and there is no staging tile, no side slots, no barrier inside the aggregation), the layer's result goes back into
// the same rows after one barrier.  69 KB of LDS: two workgroups per CU (the general kernel below: 142 KB, one).
constexpr int kSmallCap = kTile;
constexpr int kSmallEdges = 6144;
constexpr int kSmallLds = (kSmallCap * kEvalLd + 2 * H * kLdt + 6 * H + 2 * H) * 4 + (GCC_GIN_MAX_LAYERS + 1) * H * 8 + 4 * H * 8
                          + 68 * 4 + kSmallEdges;
__device__ __forceinline__ bool eval_is_small(int n, int nnz) { return n <= kSmallCap && nnz <= kSmallEdges; }

__global__ __launch_bounds__(kThreads, 2) void gin_eval_small_kernel(EvalLaunch Ln)
{
    DYN_SMEM(smem);
    const EvalArgs &a = Ln.p[blockIdx.y];
    float *A = (float *)smem;                                   // [64][kEvalLd]
    float *Wl0 = A + kSmallCap * kEvalLd, *Wl1 = Wl0 + H * kLdt;
    float *tab = Wl1 + H * kLdt;
    float *bias = tab + 6 * H;
    double *pool = (double *)(bias + 2 * H);
    double *ppart = pool + (GCC_GIN_MAX_LAYERS + 1) * H;
    int *rp = (int *)(ppart + 4 * H);                           // [65] local row pointers
    uint8_t *cols = (uint8_t *)(rp + 68);                       // [nnz] local column ids
    const int tid = (int)threadIdx.x, t = tid & 15, gi = tid >> 4, lane = lane_id(), wv = tid >> 6;
    long long tick_ = Ln.ticks ? device_ticks() : 0;
    const int b = (int)blockIdx.x;
    const int n0 = a.node_off[b], n = a.node_off[b + 1] - n0;
    const int e0 = a.row_ptr[n0], nnz = a.row_ptr[n0 + n] - e0;
    if (!eval_is_small(n, nnz)) return;                          // (workgroup-uniform) the general kernel's
    const int L = a.L;
    LayerRegs regs = eval_request_layer(a, 0);
    {
        const int sl = a.seed_local ? a.seed_local[b] : 0;
        for (int r = gi; r < n; r += 16) st4(&A[r * kEvalLd + 4 * t], eval_feature4(a, n0, r, sl, t));
        if (tid <= n) rp[tid] = a.row_ptr[n0 + tid] - e0;
        for (int e = tid; e < nnz; e += kThreads) cols[e] = (uint8_t)(a.col_idx[e0 + e] - n0);
    }
    __syncthreads();
    auto pool_rows = [&](int i) {
        const int c = tid & 63, pt = tid >> 6;
        double acc = 0.0;
        for (int r = pt; r < n; r += 4) acc += (double)A[r * kEvalLd + c];
        ppart[pt * H + c] = acc;
        __syncthreads();
        if (tid < H) pool[i * H + tid] = (ppart[tid] + ppart[H + tid]) + (ppart[2 * H + tid] + ppart[3 * H + tid]);
        __syncthreads();
    };
    EV_TICK(0);
    pool_rows(0);
    EV_TICK(1);
    if (n <= 0)
        for (int i = tid; i < L * H; i += kThreads) pool[H + i] = 0.0;
    __syncthreads();
    const int j = lane & 15, q = lane >> 4, row = 16 * wv + j;
    const float mult = (float)a.mult;
    for (int l = 0; l < (n > 0 ? L : 0); ++l) {
        eval_store_layer(a, l, regs, Wl0, Wl1, tab, bias);
        if (l + 1 < L) regs = eval_request_layer(a, l + 1);       // in flight during this layer
        __syncthreads();
        EV_TICK(2);
        // GINConv aggregate (eps = 0; gin.py:179-185,218) of this lane's row and channel quads, neighbours in CSR order
        F4 xb[4];
        {
            const bool live = row < n;
            const int rb = live ? rp[row] : 0, re = live ? rp[row + 1] : 0;
            F4 acc[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = F4{0.f, 0.f, 0.f, 0.f};
            for (int e = rb; e < re; ++e) {                      // (lanes of shorter rows wait: a wave runs its longest row)
                const float *src = &A[(int)cols[e] * kEvalLd + 4 * q];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = add4(acc[c], ld4(src + 16 * c));
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const F4 self = live ? ld4(&A[row * kEvalLd + 16 * c + 4 * q]) : F4{0.f, 0.f, 0.f, 0.f};
                xb[c].x = fmaf(mult, acc[c].x, self.x); xb[c].y = fmaf(mult, acc[c].y, self.y);
                xb[c].z = fmaf(mult, acc[c].z, self.z); xb[c].w = fmaf(mult, acc[c].w, self.w);
            }
        }
        EV_TICK(4);
        F4 h[4];
        eval_mlp_rows16(xb, Wl0, Wl1, tab, bias, j, q, h);
        __syncthreads();                                         // every lane has read what it needs of the old rows
        if (row < n) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) st4(&A[row * kEvalLd + 16 * cb + 4 * q], h[cb]);
        }
        __syncthreads();
        EV_TICK(5);
        pool_rows(l + 1);
        EV_TICK(1);
    }
    eval_readout(a, b, pool, ppart);
    EV_TICK(7);
    if (Ln.ticks && tid == 0) atomicAdd((unsigned long long *)&Ln.ticks[15], 1ull);
}

constexpr int kEvalLds = (kEvalCap * kEvalLd + kTile * kLdt + 2 * H * kLdt + 32 * H + 6 * H + 2 * H) * 4   // A, T, Wl0, Wl1, part, tables, biases
                         + (GCC_GIN_MAX_LAYERS + 1) * H * 8 + 4 * H * 8                                      // pooled sums (fp64) + their partials
                         + (kTile + 1 + 32 + 3) / 4 * 16;                                                    // rpl, prow

__global__ __launch_bounds__(kThreads) void gin_eval_fused_kernel(EvalLaunch Ln)
{
    DYN_SMEM(smem);
    const EvalArgs &a = Ln.p[blockIdx.y];
    float *A = (float *)smem;                                   // [kEvalCap][kEvalLd]
    float *T = A + kEvalCap * kEvalLd;                          // [kTile][kLdt]
    float *Wl0 = T + kTile * kLdt, *Wl1 = Wl0 + H * kLdt;       // staged Linear weights of the current layer
    float *part = Wl1 + H * kLdt;                               // [32 * H] side slots of gather_tile
    float *tab = part + 32 * H;                                 // [3][2][64] scale / shift of the layer's three BatchNorms
    float *bias = tab + 6 * H;                                  // [2][64]
    double *pool = (double *)(bias + 2 * H);                    // [L + 1][64]
    double *ppart = pool + (GCC_GIN_MAX_LAYERS + 1) * H;        // [4][64]
    int *rpl = (int *)(ppart + 4 * H);                          // [kTile + 1]
    int *prow = rpl + kTile + 1;                                // [32]

    const int tid = (int)threadIdx.x, t = tid & 15, gi = tid >> 4, lane = lane_id(), wv = tid >> 6;
    long long tick_ = Ln.ticks ? device_ticks() : 0;
    const int b = (int)blockIdx.x;
    const int n0 = a.node_off[b], n = a.node_off[b + 1] - n0;
    const bool in_lds = n <= kEvalCap;                           // (workgroup-uniform)
    const int L = a.L;
    // (an empty padding graph has no rows: its pooled sums are zero and its score is the sum of the prediction biases, as in
    //  the reference and in gcc_gin_forward)
    float *cur = a.g0, *nxt = a.g1;
    // Visibility of the workgroup's own global writes (big subgraphs; rows in transit between two layers): a workgroup runs on
    // one CU and __syncthreads() orders its stores before its later loads at workgroup scope -- no agent-scope fence (on this
    // multi-die part an agent-scope release writes L2 back: 40 us per layer with 256 workgroups doing it, measured)
    const bool single = in_lds && n <= kTile;                    // one tile: the layer updates A in place, nothing leaves LDS

    if (Ln.split && eval_is_small(n, a.row_ptr[n0 + n] - a.row_ptr[n0])) return;     // (workgroup-uniform) gin_eval_small_kernel's
    LayerRegs regs = eval_request_layer(a, 0);                   // (n == 0: harmless)

    // ---- hidden_rep[0]: input features (graph_encoder.py:158-165) -> A (LDS) or, for a big subgraph, cur (global)
    {
        const int sl = a.seed_local ? a.seed_local[b] : 0;
        for (int r = gi; r < n; r += 16) {
            const F4 x = eval_feature4(a, n0, r, sl, t);
            if (in_lds) st4(&A[r * kEvalLd + 4 * t], x);
            else st4(cur + (int64_t)(n0 + r) * H + 4 * t, x);
        }
    }
    __syncthreads();

    // SumPooling of the current representation (gin.py:228) into pool[i] (fp64, fixed order)
    auto pool_rows = [&](int i) {
        const int c = tid & 63, pt = tid >> 6;
        double acc = 0.0;
        if (in_lds) for (int r = pt; r < n; r += 4) acc += (double)A[r * kEvalLd + c];
        else for (int r = pt; r < n; r += 4) acc += (double)cur[(int64_t)(n0 + r) * H + c];
        ppart[pt * H + c] = acc;
        __syncthreads();
        if (tid < H) pool[i * H + tid] = (ppart[tid] + ppart[H + tid]) + (ppart[2 * H + tid] + ppart[3 * H + tid]);
        __syncthreads();
    };
    EV_TICK(0);                                                  // features
    pool_rows(0);
    EV_TICK(1);                                                  // pooling
    if (n <= 0)                                                   // (workgroup-uniform)
        for (int i = tid; i < L * H; i += kThreads) pool[H + i] = 0.0;
    __syncthreads();

    for (int l = 0; l < (n > 0 ? L : 0); ++l) {
        eval_store_layer(a, l, regs, Wl0, Wl1, tab, bias);
        if (l + 1 < L) regs = eval_request_layer(a, l + 1);       // in flight during this layer
        __syncthreads();
        EV_TICK(2);                                              // weights -> LDS
        const float *src = cur;
        auto load = [&](int u) -> F4 {                           // row u (batched id) of the current representation
            // (gather_tile asks for row 0 in lanes without an edge: clamped into the subgraph, the value is not used)
            return in_lds ? ld4(&A[max(u - n0, 0) * kEvalLd + 4 * t]) : ld4(src + (int64_t)u * H + 4 * t);
        };
        auto ident = [&](F4 x) -> F4 { return x; };
        for (int tile0 = 0; tile0 < n; tile0 += kTile) {
            const int nrows = min(kTile, n - tile0);
            // 1. own rows + the tile's row pointers
            {
                const int rp_own = a.row_ptr[n0 + tile0 + min(tid, nrows)];
                F4 own[kTile / 16];
#pragma unroll
                for (int i = 0; i < kTile / 16; ++i) own[i] = load(n0 + min(tile0 + gi + 16 * i, n - 1));
                if (tid <= nrows) rpl[tid] = rp_own;
#pragma unroll
                for (int i = 0; i < kTile / 16; ++i) {
                    const int r = gi + 16 * i;
                    const F4 z = {0.f, 0.f, 0.f, 0.f};
                    st4(&T[r * kLdt + 4 * t], r < nrows ? own[i] : z);
                }
            }
            __syncthreads();
            EV_TICK(3);                                          // own rows
            // 2. GINConv aggregate: h_v + sum_{u -> v} h_u (eps = 0; gin.py:179-185,218); every CSR edge counts `mult` times
            gather_tile<8>(T, part, prow, nrows, a.col_idx, load, ident, (float)a.mult, rpl);
            EV_TICK(4);                                          // gather
            // 3. the MLP and the three BatchNorm / ReLU stages on the wave's 16 rows, in registers
            {
                const int j = lane & 15, q = lane >> 4, rl = 16 * wv + j;
                F4 xb[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) xb[c] = ld4(&T[rl * kLdt + 16 * c + 4 * q]);
                F4 h[4];
                eval_mlp_rows16(xb, Wl0, Wl1, tab, bias, j, q, h);
                if (rl < nrows) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        // (single tile: every read of A by this layer's gather is behind gather_tile's closing barrier)
                        if (single) st4(&A[rl * kEvalLd + 16 * cb + 4 * q], h[cb]);
                        else st4(nxt + (int64_t)(n0 + tile0 + rl) * H + 16 * cb + 4 * q, h[cb]);
                    }
                }
            }
            __syncthreads();                                     // T, rpl and the side slots are reused by the next tile
            EV_TICK(5);                                          // the two Linears + affines
        }
        // ---- the layer's output becomes the current representation
        if (!single) {
            { float *sw = cur; cur = nxt; nxt = sw; }
            if (in_lds) {                                        // several tiles: the new rows come back from their transit buffer
                for (int idx = tid; idx < n * 16; idx += kThreads) {
                    const int r = idx >> 4, c4 = (idx & 15) * 4;
                    st4(&A[r * kEvalLd + c4], ld4(cur + (int64_t)(n0 + r) * H + c4));
                }
                __syncthreads();
            }
        }
        EV_TICK(6);                                              // mirror
        pool_rows(l + 1);
        EV_TICK(1);
    }

    eval_readout(a, b, pool, ppart);
    EV_TICK(7);                                                  // readout
    if (Ln.ticks && tid == 0) atomicAdd((unsigned long long *)&Ln.ticks[15], 1ull);
}

}  // namespace

extern "C" {

/* diagnostics, as gcc_gin_debug_ticks: device int64[16] of wall-clock ticks per phase of gcc_gin_eval_fused (features,
 * pooling, weights, own rows, gather, Linears, mirror, readout; [15] = workgroups); NULL switches it off */
void gcc_gin_eval_debug_ticks(long long *device_ticks64) { g_eval_ticks = device_ticks64; }

int32_t gcc_gin_eval_fused(const gcc_gin_pass *passes, int32_t npass, float *mean_out, void *stream)
{
    if (!passes || npass < 1 || npass > kMaxPass) {
        snprintf(g_err, kErrLen, "gcc_gin_eval_fused: npass must be 1..%d", kMaxPass);
        return -1;
    }
    EvalLaunch Ln;
    memset(&Ln, 0, sizeof(Ln));
    Ln.ticks = g_eval_ticks;
    const int B = passes[0].batch_size;
    for (int i = 0; i < npass; ++i) {
        const gcc_gin_pass &p = passes[i];
        const int L = p.w.num_gin_layers, kdim0 = p.w.pos_dim + p.w.deg_emb_dim + 1;
        if (p.training || L < 1 || L > GCC_GIN_MAX_LAYERS || kdim0 > H || p.batch_size != B || B < 1 || !p.z1[0] || !p.z2[0] ||
            !p.score || !p.feat || !p.node_off || !p.row_ptr || !p.col_idx || !p.pos) {
            snprintf(g_err, kErrLen, "gcc_gin_eval_fused: an eval-mode pass with z1[0] / z2[0] scratch, score and feat is needed "
                                     "(training=%d layers=%d d_in=%d B=%d)", p.training, L, kdim0, p.batch_size);
            return -2;
        }
        EvalArgs &a = Ln.p[i];
        a.node_off = p.node_off; a.row_ptr = p.row_ptr; a.col_idx = p.col_idx; a.seed_local = p.seed_local;
        a.pos = p.pos; a.emb = p.w.degree_embedding;
        a.g0 = p.z1[0]; a.g1 = p.z2[0];
        a.pooled = p.pooled; a.score = p.score; a.feat = p.feat;
        a.mean_out = mean_out; a.mean_w = 1.0f / (float)npass;
        for (int l = 0; l < L; ++l) {
            EvalLayer &ly = a.layer[l];
            ly.w0 = p.w.lin0_w[l]; ly.b0 = p.w.lin0_b[l]; ly.w1 = p.w.lin1_w[l]; ly.b1 = p.w.lin1_b[l];
            const gcc_bn *bn[3] = {&p.w.bn_a[l], &p.w.bn_b[l], &p.w.bn_c[l]};
            for (int k = 0; k < 3; ++k) {
                ly.bn_w[k] = bn[k]->weight; ly.bn_b[k] = bn[k]->bias; ly.bn_rm[k] = bn[k]->running_mean; ly.bn_rv[k] = bn[k]->running_var;
                if (!ly.bn_w[k] || !ly.bn_b[k] || !ly.bn_rm[k] || !ly.bn_rv[k]) {
                    snprintf(g_err, kErrLen, "gcc_gin_eval_fused: BatchNorm %d of layer %d has no running statistics", k, l);
                    return -2;
                }
            }
        }
        for (int l = 0; l <= L; ++l) { a.pred_w[l] = p.w.pred_w[l]; a.pred_b[l] = p.w.pred_b[l]; }
        a.B = B; a.L = L; a.pos_dim = p.w.pos_dim; a.emb_dim = p.w.deg_emb_dim; a.max_degree = p.w.max_degree;
        a.mult = p.edge_multiplicity > 1 ? p.edge_multiplicity : 1;
        a.normalize = p.normalize; a.hid = hidden_of(p.w); a.kdim0 = kdim0;
        a.eps = p.w.bn_eps; a.norm_eps = p.w.norm_eps;
    }
    hipStream_t s = (hipStream_t)stream;
#ifndef GCC_AMD_HIPEMU
    static bool opted = false;                               // more than 64 KiB of dynamic LDS is opted into once
    if (!opted) {
        (void)hipFuncSetAttribute((const void *)gin_eval_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kEvalLds);
        (void)hipFuncSetAttribute((const void *)gin_eval_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kSmallLds);
        opted = true;
    }
    if (mean_out) (void)hipMemsetAsync(mean_out, 0, (size_t)B * H * sizeof(float), s);
#else
    if (mean_out) memset(mean_out, 0, (size_t)B * H * sizeof(float));
#endif
    // two launches over the same grid: subgraphs of at most 64 nodes in the two-per-CU kernel, the rest in the general one
    Ln.split = 1;
    hipLaunchKernelGGL(gin_eval_small_kernel, dim3(B, npass), dim3(kThreads), kSmallLds, s, Ln);
    hipLaunchKernelGGL(gin_eval_fused_kernel, dim3(B, npass), dim3(kThreads), kEvalLds, s, Ln);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, kErrLen, "gcc_gin_eval_fused: launch failed: %s", hipGetErrorString(e));
        return -10;
    }
    return 0;
}

}  // extern "C"
