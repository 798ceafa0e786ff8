// gcc_amd/csrc/encoder_eval.hip -- the eval-mode GIN encoder of generate.py as ONE call: SURVEY.md 8(f)#2's
// "per-subgraph LDS-resident megakernel".  Reference: generate.py:33-53 (test_moco: model.eval(), feat_q = model(graph_q),
// feat_k = model(graph_k), emb = (feat_q + feat_k) / 2) -> GraphEncoder.forward gcc/models/graph_encoder.py:132-200 ->
// UnsupervisedGIN.forward gcc/models/gin.py:213-232 with every BatchNorm on its running statistics (gin.py:54-58,113-116).
//
// In eval mode nothing couples two subgraphs of a batch (training-mode BatchNorm did: 12 batch-wide statistics per pass
// are why gcc_gin_forward is a chain of 15 launches), so a workgroup takes a subgraph -- or a run of small ones -- through
// feature assembly, all GIN layers, the pooled readout and F.normalize.  Three launches per call (both views of generate.py in
// each):
//   1. gin_eval_plan_kernel: the work list of (2), largest first, and mean_out = 0;
//   2. gin_eval_lds_kernel: subgraphs of up to 320 nodes and runs of up to four subgraphs of at most 64 -- rows, local column
//      ids and the layer's weights in LDS, 8 waves, per-lane neighbour sums straight into the registers the exact-f32 MFMA
//      products read (eval_gather_row, eval_mlp_rows16), results back in place;
//   3. gin_eval_fused_kernel: the rest (hub ego-nets of more than 320 nodes, subgraphs whose entries do not fit the column-id
//      space) -- 64-row tiles through gather_tile of encoder_common.h, rows in LDS up to 256 nodes, in L2 above.
// SumPooling of every hidden_rep in fp64, the prediction layers, normalisation, and -- with two passes in the call --
// mean_out[b] += feat / 2, i.e. generate.py:52's (feat_q + feat_k) / 2.  Same arithmetic as gcc_gin_forward with training = 0
// (same affine tables, same MFMA sequences; the neighbour sums run in another order): agreement ~1e-6.
#include "encoder_common.h"

namespace {

constexpr int kEvalCap = 256;            // rows of a subgraph the general kernel keeps in LDS
constexpr int kEvalLd = 68;              // floats per LDS row (272 B: 16-byte aligned, rows 8 apart share a bank group)

struct EvalLayer {
    const float *w0, *b0, *w1, *b1;
    const float *bn_w[3], *bn_b[3], *bn_rm[3], *bn_rv[3];     // mlp.batch_norms.0, apply_func.bn, gnn.batch_norms.i
};
struct EvalArgs {
    const int32_t *node_off, *row_ptr, *col_idx, *seed_local;
    const float *pos, *emb;
    float *g0, *g1;                      // global ping-pong [node_cap][64] (the pass's z1[0] / z2[0] buffers)
    int32_t *plan;                       // [1 + B] work list of the LDS-resident kernel (the pass's x0 buffer, unused in eval mode)
    double *pooled;                      // [L + 1][B][64] or NULL
    float *score, *feat;                 // [B][64]
    float *mean_out;                     // [B][64] or NULL: += mean_w * feat (zeroed by the host side of the call)
    float mean_w;
    EvalLayer layer[GCC_GIN_MAX_LAYERS];
    const float *pred_w[GCC_GIN_MAX_LAYERS + 1], *pred_b[GCC_GIN_MAX_LAYERS + 1];
    int32_t B, L, pos_dim, emb_dim, max_degree, mult, normalize, hid, kdim0;
    float eps, norm_eps;
};
struct EvalLaunch { EvalArgs p[kMaxPass]; long long *ticks; int32_t split, npass; };   // split: the general kernel leaves the LDS-resident kernel's subgraphs alone
static long long *g_eval_ticks = nullptr;    // diagnostics (gcc_gin_eval_debug_ticks): device int64[2][16] (LDS-resident kernel, general kernel);
                                             // every 8th workgroup reports (all of them adding to the same counters waited on their own atomics)
#define EV_TICK(ph) do { if (tick_on) { const long long now_ = device_ticks(); atomicAdd((unsigned long long *)&Ln.ticks[kTickBase + (ph)], (unsigned long long)(now_ - tick_)); tick_ = now_; } } while (0)

// ---- pieces shared by the two kernel shapes ------------------------------------------------------------------------------
// a layer's weights / BatchNorm numbers are requested one layer ahead (registers) and stored when the LDS buffers are free
struct LayerRegs { WStage s0, s1; float v0, v1, v2, v3; };
// kQ: the matrix's rows are whole 16-byte quads (width % 4 == 0, base aligned: the host checks) -> four global_load_dwordx4 per
// thread.  A compile-time choice per kernel: with both forms behind a run-time branch (stage_weights_request) the compiler merged
// them into sixteen global_load_dword -- four times the requests for the same bytes.
template <bool kQ> __device__ __forceinline__ WStage eval_weights_request(const float *W, int kdim)
{
    WStage st;
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                // (unconditional, clamped; columns >= kdim are zeroed when stored)
        const int idx = tid + i * kThreads, r = idx >> 4, c4 = 4 * (idx & 15);
        const float *p = W + (int64_t)r * kdim;
        if constexpr (kQ) st.v[i] = ld4(p + min(c4, kdim - 4));
        else st.v[i] = F4{p[min(c4 + 0, kdim - 1)], p[min(c4 + 1, kdim - 1)], p[min(c4 + 2, kdim - 1)], p[min(c4 + 3, kdim - 1)]};
    }
    return st;
}
// kFirst: layer 0, whose first matrix is d_in = 49 wide (element loads, once per kernel)
template <bool kQ, bool kFirst> __device__ __forceinline__ LayerRegs eval_request_layer(const EvalArgs &a, int l)
{
    const EvalLayer &ly = a.layer[l];
    const int tid = (int)threadIdx.x;
    LayerRegs r;
    if constexpr (kFirst) r.s0 = eval_weights_request<false>(ly.w0, a.kdim0);
    else r.s0 = eval_weights_request<kQ>(ly.w0, a.hid);
    r.s1 = eval_weights_request<kQ>(ly.w1, a.hid);
    // which < 3: a BatchNorm; 3: the two biases.  wave_uniform: the index into the descriptor's pointer arrays is then a scalar
    // (s_load); as a vector index the POINTERS came through global_load + s_waitcnt vmcnt(0) -- behind the weight requests just
    // issued, i.e. every layer waited for its successor's weights (the "weights" phase: 46 of a workgroup's 95 us)
    const int c = tid & 63, which = wave_uniform(tid >> 6);
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
// The LDS-resident kernels' aggregation: lane (j, q) of a wave owns row j of the wave's 16 and the channel quads 16 c + 4 q; its
// result is the sum of rows cols[rb .. re) of A.
//   * rows of at most kEvalHub neighbours: every lane walks ITS row in CSR order, four neighbours per round trip -- their ids
//     first, then the sixteen quad reads, then the adds (one id and four reads at a time was two dependent LDS latencies per
//     neighbour: ~300 clocks each).  A lane past its row's end reads the zero row `zrow` of A (x + 0 = x: the sums are those of
//     the one-at-a-time loop, bit for bit).
//   * hub rows (the seed of an ego-net is adjacent to a large part of it: 100-300 neighbours): one at a time by the whole wave --
//     lane (j, q) sums the neighbours hb + j, hb + j + 16, ... (again four per round trip), the 16 partial sums of a quad are
//     added over the DPP row (fixed tree) and handed to the owning lane.  A wave used to run as long as its longest row: 75
//     rounds for a 300-neighbour hub while the other 15 rows had finished after 2.
#ifndef GCC_EVAL_HUB
#define GCC_EVAL_HUB 32
#endif
constexpr int kEvalHub = GCC_EVAL_HUB;      // (GPU time per call at rw_hops 64 / 256 by this threshold: 20: 83.7 / 127.2 us, 32: 83.9 / 127.8, 48: 88.0 / 128.5, 64: 91.0 / 129.5)
template <class ColT>
__device__ __forceinline__ void eval_gather4(const float *A, const ColT *cols, int e, int stride, int end, int zrow, int q, F4 acc[4])
{
    int id[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) id[u] = (int)cols[min(e + u * stride, end - 1)];
    F4 v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float *src = &A[(e + u * stride < end ? id[u] : zrow) * kEvalLd + 4 * q];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[u][c] = ld4(src + 16 * c);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = add4(acc[c], v[u][c]);
}
template <class ColT>
__device__ __forceinline__ void eval_gather_row(const float *A, const ColT *cols, int rb, int re, int zrow, int j, int q, F4 acc[4])
{
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = F4{0.f, 0.f, 0.f, 0.f};
    const bool hub = re - rb > kEvalHub;
    unsigned long long hubs = wave_ballot(hub) & 0xFFFFull;      // (lanes 0 .. 15 are q = 0 of the 16 rows)
    for (int e = rb; e < (hub ? rb : re); e += 4) eval_gather4(A, cols, e, 1, re, zrow, q, acc);
    while (hubs) {                                               // (wave-uniform)
        const int jh = __builtin_ctzll(hubs);
        hubs &= hubs - 1;
        const int hb = wave_readlane(rb, jh), he = wave_readlane(re, jh);
        F4 s[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] = F4{0.f, 0.f, 0.f, 0.f};
        for (int e = hb + j; e < he; e += 64) eval_gather4(A, cols, e, 16, he, zrow, q, s);
        const int last = (lane_id() & 48) | 15;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const F4 tot = {wave_shfl(row16_sum_last(s[c].x), last), wave_shfl(row16_sum_last(s[c].y), last),
                            wave_shfl(row16_sum_last(s[c].z), last), wave_shfl(row16_sum_last(s[c].w), last)};
            if (j == jh) acc[c] = tot;
        }
    }
}
// one output of a prediction layer: b + sum_k w[k] pooled[k], k ascending (gin.py:229); the weight row in quads when it can be
__device__ __forceinline__ float eval_pred_dot(const float *w, const float *bias_o, const double *pool, int kd, bool quads)
{
    float acc = bias_o ? *bias_o : 0.f;
    if (quads) {
        for (int k = 0; k < kd; k += 4) {
            const F4 w4 = ld4(w + k);
            acc = fmaf(w4.x, (float)pool[k], acc);
            acc = fmaf(w4.y, (float)pool[k + 1], acc);
            acc = fmaf(w4.z, (float)pool[k + 2], acc);
            acc = fmaf(w4.w, (float)pool[k + 3], acc);
        }
    } else {
        for (int k = 0; k < kd; ++k) acc = fmaf(w[k], (float)pool[k], acc);
    }
    return acc;
}
__device__ __forceinline__ bool eval_pred_quads(const float *W, int kd) { return (kd & 3) == 0 && ((uintptr_t)W & 15) == 0; }   // (block-uniform)
// readout: score = sum_i linears_prediction[i](pooled_i) (gin.py:227-230; eval: dropout is the identity), F.normalize
// (graph_encoder.py:195-196), the mean over the passes, the pooled sums.  pool: LDS [L + 1][64] fp64; ppart: LDS scratch
__device__ __forceinline__ void eval_readout(const EvalArgs &a, int b, const double *pool, double *ppart)
{
    const int tid = (int)threadIdx.x, L = a.L;
    const int o = tid & 63, pt = wave_uniform(tid >> 6);         // (scalar: the layer's pointers come by s_load)
    float s = 0.f;
    for (int i = pt; i <= L; i += 4) {
        const int kd = i == 0 ? a.kdim0 : a.hid;
        s += eval_pred_dot(a.pred_w[i] + (int64_t)o * kd, a.pred_b[i] ? a.pred_b[i] + o : nullptr, pool + i * H, kd,
                           eval_pred_quads(a.pred_w[i], kd));
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

// ---- the LDS-resident kernel: subgraphs of up to kMedCap nodes, and runs of small ones ------------------------------------------
// History: round 4 gave subgraphs of <= 64 nodes a 4-wave kernel of their own (two per CU) and sent the rest through the general
// kernel below, which walked their 64-row tiles one after the other (own rows -> gather_tile -> MLP -> transit through global
// memory, ~12-16 us per tile and layer: 147 us for a 145-node subgraph, 321 us for a 298-node one, profiles/r6_eval_trace_before.txt).
// Now ONE kernel of 8 waves keeps a subgraph's -- or a run's, eval_lds_run -- rows and 16-bit local column ids in LDS; a wave owns
// 16 rows per pass of 128 and sums ITS rows' neighbours into the registers the first product reads (eval_gather_row: no staging
// tile, no side slots, no global traffic inside a layer); the passes' results wait in registers for the one barrier after which
// the rows are overwritten in place.  The products are eval_mlp_rows16, the prediction layers keep eval_readout's summation
// order: a subgraph gives the same result alone, in a run, or in the general kernel, up to the order of its hub row's sum and
// of the fp64 pooled sums.
constexpr int kSmallCap = kTile;         // a "small" subgraph: at most 64 nodes (the median ego-net has 23 .. 55)
constexpr int kMedThreads = 512;
constexpr int kMedCap = 320;
constexpr int kMedPasses = (kMedCap + 127) / 128;
constexpr int kMedEdges = 13664;
constexpr int kMedRp = 328;              // ints: kMedCap + 1 row pointers, padded to 16 bytes
constexpr int kMedLds = ((kMedCap + 1) * kEvalLd + 2 * H * kLdt + 6 * H + 2 * H) * 4 + (GCC_GIN_MAX_LAYERS + 1) * H * 8 + 8 * H * 8
                        + kMedRp * 4 + kMedEdges * 2;
static_assert(kMedLds <= 160 * 1024, "one workgroup per CU");
static_assert((GCC_GIN_MAX_LAYERS + 1) * H * 4 <= 8 * H * 8, "the per-layer scores share the pooling partials' space");
// the LDS-resident kernel's subgraphs (alone or in a run); the general kernel takes the others
__device__ __forceinline__ bool eval_in_lds(int n, int nnz) { return n <= kMedCap && nnz <= kMedEdges; }

struct MedRegs { F4 w0[2], w1[2]; float v0, v1, v2, v3; };
template <bool kQ> __device__ __forceinline__ void med_weights_request(const float *W, int kdim, F4 (&v)[2])
{
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {                                // (as eval_weights_request, 512 threads)
        const int idx = tid + i * kMedThreads, r = idx >> 4, c4 = 4 * (idx & 15);
        const float *p = W + (int64_t)r * kdim;
        if constexpr (kQ) v[i] = ld4(p + min(c4, kdim - 4));
        else v[i] = F4{p[min(c4 + 0, kdim - 1)], p[min(c4 + 1, kdim - 1)], p[min(c4 + 2, kdim - 1)], p[min(c4 + 3, kdim - 1)]};
    }
}
__device__ __forceinline__ void med_weights_store(float *Wl, const F4 (&v)[2], int kdim)
{
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * kMedThreads, r = idx >> 4, c4 = 4 * (idx & 15);
        const F4 x = v[i];
        st4(&Wl[r * kLdt + c4], F4{c4 + 0 < kdim ? x.x : 0.f, c4 + 1 < kdim ? x.y : 0.f, c4 + 2 < kdim ? x.z : 0.f, c4 + 3 < kdim ? x.w : 0.f});
    }
}
template <bool kQ, bool kFirst> __device__ __forceinline__ MedRegs med_request_layer(const EvalArgs &a, int l)
{
    const EvalLayer &ly = a.layer[l];
    const int tid = (int)threadIdx.x;
    MedRegs r;
    if constexpr (kFirst) med_weights_request<false>(ly.w0, a.kdim0, r.w0);
    else med_weights_request<kQ>(ly.w0, a.hid, r.w0);
    med_weights_request<kQ>(ly.w1, a.hid, r.w1);
    const int c = tid & 63, which = wave_uniform(tid >> 6);      // which < 3: a BatchNorm; 3: the two biases; 4 .. 7: nothing (scalar: see eval_request_layer)
    r.v0 = r.v1 = r.v2 = r.v3 = 0.f;
    if (which < 3) { r.v0 = ly.bn_w[which][c]; r.v1 = ly.bn_b[which][c]; r.v2 = ly.bn_rm[which][c]; r.v3 = ly.bn_rv[which][c]; }
    else if (which == 3) { r.v0 = ly.b0 ? ly.b0[c] : 0.f; r.v1 = ly.b1 ? ly.b1[c] : 0.f; }
    return r;
}
__device__ __forceinline__ void med_store_layer(const EvalArgs &a, int l, const MedRegs &r, float *Wl0, float *Wl1, float *tab, float *bias)
{
    const int tid = (int)threadIdx.x;
    med_weights_store(Wl0, r.w0, l == 0 ? a.kdim0 : a.hid);
    med_weights_store(Wl1, r.w1, a.hid);
    const int c = tid & 63, which = tid >> 6;
    if (which < 3) {                                             // (eval_store_layer's table)
        const double rstd = 1.0 / sqrt((double)r.v3 + (double)a.eps);
        tab[which * 2 * H + c] = (float)((double)r.v0 * rstd);
        tab[which * 2 * H + H + c] = (float)((double)r.v1 - (double)r.v2 * (double)r.v0 * rstd);
    } else if (which == 3) {
        bias[c] = r.v0;
        bias[H + c] = r.v1;
    }
}

// Which subgraphs workgroup b takes, from the node offsets alone (one round trip: three workgroups in four leave after it):
//   * a subgraph of 65 .. kMedCap nodes: itself;
//   * a subgraph of at most 64 nodes: the RUN of consecutive such subgraphs around it inside its aligned group of four -- rows
//     [n0, n0 + n) of the batched CSR are then ONE block-diagonal graph of at most 256 rows, and the run's first workgroup takes
//     it whole: the layer's weights are staged once for up to four subgraphs, 7 of the 8 waves have rows at the median size
//     (23 nodes: a workgroup of its own used 2 of 4 waves), and the small subgraphs do not wait in a launch of their own.
// (A run with more entries than the column-id space holds -- four complete 64-node graphs, multi-edges -- is taken member by
// member by the same workgroup.)
constexpr int kRunMax = 4;
struct EvalRun { int t, g, n0, n, o1, o2, o3; };     // subgraphs [t, t + g), rows [n0, n0 + n); o1..o3: local first rows of members 1..3 (n past the run)
                                                      // (named scalars: an offset ARRAY picked by wave index became a table in scratch memory)
// kUniform: b is the same in every lane of the wave (the results go to scalar registers); the work-list kernel asks per lane
template <bool kUniform = true> __device__ __forceinline__ EvalRun eval_lds_run(const EvalArgs &a, int b, int &nb)
{
    auto uni = [](int v) { return kUniform ? wave_uniform(v) : v; };
    const int q0 = b & ~(kRunMax - 1), i0 = b - q0;
    int off[kRunMax + 1];
#pragma unroll
    for (int i = 0; i <= kRunMax; ++i) off[i] = a.node_off[min(q0 + i, a.B)];
    bool small[kRunMax];
#pragma unroll
    for (int i = 0; i < kRunMax; ++i) small[i] = q0 + i < a.B && off[i + 1] - off[i] <= kSmallCap;
    nb = 0;
#pragma unroll
    for (int i = 0; i < kRunMax; ++i) nb = i == i0 ? off[i + 1] - off[i] : nb;
    nb = uni(nb);
    int t = i0, e = i0 + 1;
    if (nb <= kSmallCap) {
#pragma unroll
        for (int i = kRunMax - 1; i >= 1; --i) t = (i == t && small[i - 1]) ? i - 1 : t;     // (descending: each step may extend by one)
#pragma unroll
        for (int i = 1; i < kRunMax; ++i) e = (i == e && small[i]) ? i + 1 : e;
    }
    auto at = [&](int k) {                                       // node offset k of the group of four (k <= 4), past the run: the run's end
        int v = 0;
#pragma unroll
        for (int i = 0; i <= kRunMax; ++i) v = i == min(k, e) ? off[i] : v;
        return v;
    };
    EvalRun r;
    r.t = uni(q0 + t); r.g = uni(e - t);      // (all of it is workgroup-uniform: scalar registers)
    r.n0 = uni(at(t)); r.n = uni(at(e)) - r.n0;
    r.o1 = uni(at(t + 1)) - r.n0; r.o2 = uni(at(t + 2)) - r.n0; r.o3 = uni(at(t + 3)) - r.n0;
    return r;
}
// member g of a run as a run of its own
__device__ __forceinline__ EvalRun eval_run_member(const EvalRun &run, int g)
{
    const int a1 = run.o1, a2 = run.o2, a3 = run.o3, an = run.n;   // (values first: selects over the fields become an indexed load)
    const int lo = g == 0 ? 0 : g == 1 ? a1 : g == 2 ? a2 : a3;
    const int hi = g == 0 ? a1 : g == 1 ? a2 : g == 2 ? a3 : an;
    EvalRun r;
    r.t = run.t + g; r.g = 1; r.n0 = run.n0 + lo; r.n = hi - lo; r.o1 = r.o2 = r.o3 = r.n;
    return r;
}

// The work list of the LDS-resident kernel, one workgroup per pass: plan[0] = number of runs / subgraphs it takes, plan[1 ..] = their
// first subgraphs, LARGEST FIRST (counting sort by rows).  Why a list: the kernel holds one workgroup per CU, and with a workgroup
// per subgraph three in four only looked at the node offsets and left -- but each of them held a CU's LDS for the microseconds
// that takes, in dispatch order, so the last workgroups WITH work started 46 us (rw_hops 64: 176 of 512 have work) to 76 us
// (rw_hops 256) after the first.  With the list the first plan[0] workgroups all have work, the longest go first and round-robin
// over the XCDs, and the rest leave after one scalar load.  Also zeroes mean_out (the memset this launch replaces).
constexpr int kPlanThreads = 256;
__global__ __launch_bounds__(kPlanThreads) void gin_eval_plan_kernel(EvalLaunch Ln)
{
    constexpr int kPlanKeys = 2048;
    __shared__ int hist[kMedCap + 1 + 63], start[kMedCap + 1 + 63];
    __shared__ short key[kPlanKeys];
    const EvalArgs &a = Ln.p[blockIdx.x];
    const int tid = (int)threadIdx.x;
    if (blockIdx.x == 0 && a.mean_out)
        for (int i = tid; i < a.B * (H / 4); i += kPlanThreads) st4(a.mean_out + 4 * (int64_t)i, F4{0.f, 0.f, 0.f, 0.f});
    for (int i = tid; i < kMedCap + 1 + 63; i += kPlanThreads) hist[i] = 0;
    __syncthreads();
    auto item = [&](int b, int &rows) {                          // does a workgroup of the LDS-resident kernel start at subgraph b?
        int nb;
        const EvalRun run = eval_lds_run<false>(a, b, nb);
        rows = run.n;
        return nb <= kMedCap && run.t == b;
    };
    for (int b = tid; b < a.B; b += kPlanThreads) {
        int rows;
        const bool it = item(b, rows);
        if (it) atomicAdd(&hist[rows], 1);
        if (b < kPlanKeys) key[b] = it ? (short)rows : (short)-1;                // (the second pass reads these instead of the offsets again)
    }
    __syncthreads();
    if (tid < 64) {                                              // start[r] = number of items with more rows than r (6 bins per lane, descending)
        int mine = 0;
        for (int k = 0; k < 6; ++k) mine += hist[kMedCap + 63 - (6 * tid + k)];
        int before = wave_scan_incl(mine) - mine;
        for (int k = 0; k < 6; ++k) {
            const int r = kMedCap + 63 - (6 * tid + k);
            start[r] = before;
            before += hist[r];
        }
        if (tid == 63) a.plan[0] = before;
    }
    __syncthreads();
    for (int i = tid; i < kMedCap + 1 + 63; i += kPlanThreads) hist[i] = 0;
    __syncthreads();
    for (int b = tid; b < a.B; b += kPlanThreads) {
        int rows;
        bool it;
        if (b < kPlanKeys) { rows = key[b]; it = rows >= 0; }
        else it = item(b, rows);
        if (it) a.plan[1 + start[rows] + atomicAdd(&hist[rows], 1)] = b;
    }
}

template <bool kQ> __global__ __launch_bounds__(kMedThreads) void gin_eval_lds_kernel(EvalLaunch Ln)
{
    DYN_SMEM(smem);
    // grid (B, passes): workgroup x takes item x of the pass's work list (gin_eval_plan_kernel), if there is one
    const int pass = (int)blockIdx.y;
    const EvalArgs &a = Ln.p[pass];
    float *A = (float *)smem;                                   // [kMedCap + the zero row][kEvalLd]
    float *Wl0 = A + (kMedCap + 1) * kEvalLd, *Wl1 = Wl0 + H * kLdt;
    float *tab = Wl1 + H * kLdt;
    float *bias = tab + 6 * H;
    double *pool = (double *)(bias + 2 * H);                    // [L + 1][64]
    double *ppart = pool + (GCC_GIN_MAX_LAYERS + 1) * H;        // [8][64]; the readout's per-layer scores afterwards
    int *rp = (int *)(ppart + 8 * H);                           // [n + 1] local row pointers
    uint16_t *cols = (uint16_t *)(rp + kMedRp);                 // [nnz] local column ids
    const int tid = (int)threadIdx.x, t = tid & 15, gi = tid >> 4, lane = lane_id(), wv = tid >> 6;
    constexpr int kTickBase = 0;
    long long tick_ = Ln.ticks ? device_ticks() : 0;
    const long long wg_start_ = tick_;
    if ((int)blockIdx.x >= wave_uniform(a.plan[0])) return;      // (workgroup-uniform) nothing left
    const int b = wave_uniform(a.plan[1 + blockIdx.x]);
    int nb;
    const EvalRun whole = eval_lds_run(a, b, nb);
    const int whole_e0 = wave_uniform(a.row_ptr[whole.n0]);
    const int whole_nnz = wave_uniform(a.row_ptr[whole.n0 + whole.n]) - whole_e0;
    const int L = a.L;
    // a run as a whole or -- more entries than column ids fit (workgroup-uniform) -- member by member.  (A loop around the body, not
    // a lambda called in a loop: through the lambda's captures the LDS pointers became generic ones, 180 flat_load / flat_store.)
    if (whole_nnz > kMedEdges && whole.g == 1) return;           // (workgroup-uniform) one subgraph with too many entries: the general kernel's
    const int parts = whole_nnz <= kMedEdges ? 1 : whole.g;
    for (int part = 0; part < parts; ++part) {
    EvalRun one = whole;
    int e0 = whole_e0, nnz = whole_nnz;
    if (parts > 1) {
        __syncthreads();                                         // the next member overwrites everything
        one = eval_run_member(whole, part);
        e0 = wave_uniform(a.row_ptr[one.n0]);
        nnz = wave_uniform(a.row_ptr[one.n0 + one.n]) - e0;
        if (nnz > kMedEdges) continue;                           // (the general kernel's)
    }
    const int first = one.t, G = one.g, n0 = one.n0, n = one.n, o0 = 0, o1 = one.o1, o2 = one.o2, o3 = one.o3, o4 = one.n;
#ifdef GCC_EVAL_TICK_BIG                                         // (diagnostic build: the phases of the largest subgraphs only)
    const bool tick_on = Ln.ticks && threadIdx.x == 0 && n >= GCC_EVAL_TICK_BIG;
#else
    const bool tick_on = Ln.ticks && threadIdx.x == 0 && (blockIdx.x & 3) == 0;
#endif
    MedRegs regs = med_request_layer<kQ, true>(a, 0);
    {
        int seedrow[kRunMax];                                    // local row of each subgraph's seed (ndata["seed"], data_util.py:234-238)
#pragma unroll
        for (int g = 0; g < kRunMax; ++g)
            seedrow[g] = (g == 0 ? o0 : g == 1 ? o1 : g == 2 ? o2 : o3) + ((a.seed_local && g < G) ? a.seed_local[first + g] : 0);
        for (int r = gi; r < n; r += kMedThreads / 16) {
            const int sl = r >= o3 ? seedrow[3] : r >= o2 ? seedrow[2] : r >= o1 ? seedrow[1] : seedrow[0];
            st4(&A[r * kEvalLd + 4 * t], eval_feature4(a, n0, r, sl, t));
        }
        if (tid < 16) st4(&A[kMedCap * kEvalLd + 4 * tid], F4{0.f, 0.f, 0.f, 0.f});
        if (tid <= n) rp[tid] = a.row_ptr[n0 + tid] - e0;
        for (int e = tid; e < nnz; e += kMedThreads) cols[e] = (uint16_t)(a.col_idx[e0 + e] - n0);
    }
    __syncthreads();
    // pooled sums of subgraph g of the run: the first one's behind the tables, the others' in rows 257 .. of A (a run has at most
    // 256 rows; the zero row is row kMedCap)
    static_assert(kRunMax * kSmallCap + 1 + ((kRunMax - 1) * (GCC_GIN_MAX_LAYERS + 1) * H * 8 + kEvalLd * 4 - 1) / (kEvalLd * 4) <= kMedCap,
                  "the other subgraphs' pooled sums fit between a run's rows and the zero row");
    auto pool_of = [&](int g) __attribute__((always_inline)) -> double * {
        return g == 0 ? pool : (double *)(A + (kRunMax * kSmallCap + 1) * kEvalLd) + (g - 1) * (GCC_GIN_MAX_LAYERS + 1) * H;
    };
    // rows of the subgraph this wave pools in a run (waves g and g + 4: subgraph g).  Picked here, from values: inside the lambda the
    // selects over captured variables became an indexed load from the closure object, which then had to live in scratch memory
    // -- and every LDS pointer it held turned into a generic one (180 flat_load / flat_store instead of ds_read / ds_write)
    const int seg_g = wave_uniform(tid >> 6) & 3;
    const int seg_r0 = seg_g == 0 ? o0 : seg_g == 1 ? o1 : seg_g == 2 ? o2 : o3;
    const int seg_r1 = seg_g == 0 ? o1 : seg_g == 1 ? o2 : seg_g == 2 ? o3 : o4;
    auto pool_rows = [&](int i) __attribute__((always_inline)) {   // SumPooling (gin.py:228), fp64, fixed order
        const int c = tid & 63, pt = wave_uniform(tid >> 6);    // (scalar: the run's offsets are picked by scalar selects, not from a copy in scratch)
        if (G == 1) {                                            // (workgroup-uniform) one subgraph: 8 strided partials
            double acc = 0.0;
            for (int r = pt; r < n; r += 8) acc += (double)A[r * kEvalLd + c];
            ppart[pt * H + c] = acc;
            __syncthreads();
            if (tid < H)
                pool[i * H + tid] = ((ppart[tid] + ppart[H + tid]) + (ppart[2 * H + tid] + ppart[3 * H + tid]))
                                    + ((ppart[4 * H + tid] + ppart[5 * H + tid]) + (ppart[6 * H + tid] + ppart[7 * H + tid]));
        } else {                                                 // a run: waves g and g + 4 take the even / odd rows of subgraph g
            double acc = 0.0;
            for (int r = seg_r0 + (pt >> 2); r < seg_r1; r += 2) acc += (double)A[r * kEvalLd + c];
            ppart[pt * H + c] = acc;
            __syncthreads();
            if (pt < G) pool_of(pt)[i * H + c] = ppart[pt * H + c] + ppart[(pt + 4) * H + c];
        }
        // (no barrier here: the partials are next written behind the following layer's barriers, the sums are read by the readout)
    };
    EV_TICK(0);
    pool_rows(0);
    EV_TICK(1);
    const int j = lane & 15, q = lane >> 4;
    const float mult = (float)a.mult;
    for (int l = 0; l < L; ++l) {
        med_store_layer(a, l, regs, Wl0, Wl1, tab, bias);
        if (l + 1 < L) regs = med_request_layer<kQ, false>(a, l + 1);        // in flight during this layer
        __syncthreads();
        EV_TICK(2);
        F4 hh[kMedPasses][4];
#pragma unroll
        for (int p = 0; p < kMedPasses; ++p) {
            const int row = 128 * p + 16 * wv + j;
            if (128 * p + 16 * wv < n) {                         // (wave-uniform) the wave has rows in this pass
                // GINConv aggregate (eps = 0; gin.py:179-185,218) of this lane's row and channel quads, neighbours in CSR order
                F4 xb[4];
                const bool live = row < n;
                const int rb = live ? rp[row] : 0, re = live ? rp[row + 1] : 0;
                F4 acc[4];
                eval_gather_row(A, cols, rb, re, kMedCap, j, q, acc);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const F4 self = live ? ld4(&A[row * kEvalLd + 16 * c + 4 * q]) : F4{0.f, 0.f, 0.f, 0.f};
                    xb[c].x = fmaf(mult, acc[c].x, self.x); xb[c].y = fmaf(mult, acc[c].y, self.y);
                    xb[c].z = fmaf(mult, acc[c].z, self.z); xb[c].w = fmaf(mult, acc[c].w, self.w);
                }
                EV_TICK(3);                                      // (wave 0's aggregation)
                eval_mlp_rows16(xb, Wl0, Wl1, tab, bias, j, q, hh[p]);
                EV_TICK(4);                                      // (wave 0's products)
            }
        }
        __syncthreads();                                         // every lane has read what it needs of the old rows
        EV_TICK(6);                                              // (wave 0 waiting for the other waves)
#pragma unroll
        for (int p = 0; p < kMedPasses; ++p) {
            const int row = 128 * p + 16 * wv + j;
            if (row < n) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) st4(&A[row * kEvalLd + 16 * cb + 4 * q], hh[p][cb]);
            }
        }
        __syncthreads();
        EV_TICK(5);
        pool_rows(l + 1);
        EV_TICK(1);
    }
    // readout (eval_readout's sums in eval_readout's order)
    __syncthreads();                                             // the pooled sums are complete, their partials are free
    auto emit = [&](int sub, const double *pl, float sc) __attribute__((always_inline)) {      // one wave, lane = channel: F.normalize, outputs of subgraph `sub`
        float ss = sc * sc;
        ss = wave_sum(ss);
        float f = sc;
        if (a.normalize) {
            const float nrm = sqrtf(ss);
            f = sc / (nrm > a.norm_eps ? nrm : a.norm_eps);
        }
        const int o = tid & 63;
        a.score[(int64_t)sub * H + o] = sc;
        a.feat[(int64_t)sub * H + o] = f;
        if (a.mean_out) atomicAdd(&a.mean_out[(int64_t)sub * H + o], a.mean_w * f);
        if (a.pooled) for (int i = 0; i <= L; ++i) a.pooled[((int64_t)i * a.B + sub) * H + o] = pl[i * H + o];
    };
    if (G == 1) {                                                // (workgroup-uniform) one prediction layer per wave
        const int o = tid & 63, pt = wave_uniform(tid >> 6);
        float *sl = (float *)ppart;                              // [L + 1][64] per-layer scores
        for (int i = pt; i <= L; i += 8) {
            const int kd = i == 0 ? a.kdim0 : a.hid;
            sl[i * H + o] = eval_pred_dot(a.pred_w[i] + (int64_t)o * kd, a.pred_b[i] ? a.pred_b[i] + o : nullptr, pool + i * H, kd,
                                          eval_pred_quads(a.pred_w[i], kd));
        }
        __syncthreads();
        if (tid < H) {
            float sp[4] = {0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i <= L; ++i) sp[i & 3] += sl[i * H + tid];
            emit(first, pool, (sp[0] + sp[1]) + (sp[2] + sp[3]));
        }
    } else {                                                     // a run: waves g and g + 4 take the even / odd layers of subgraph g
        const int o = tid & 63, pt = wave_uniform(tid >> 6), g = pt & 3, h = pt >> 2;
        float *spx = (float *)ppart;                             // [4 subgraphs][4 partial sums][64]
        float sa = 0.f, sb = 0.f;                                // eval_readout's partial sums h and h + 2 (layers i = h, h + 2, ... ascending)
        if (g < G) {
            const double *pl = pool_of(g);
            for (int i = h; i <= L; i += 2) {
                const int kd = i == 0 ? a.kdim0 : a.hid;
                const float v = eval_pred_dot(a.pred_w[i] + (int64_t)o * kd, a.pred_b[i] ? a.pred_b[i] + o : nullptr, pl + i * H, kd,
                                              eval_pred_quads(a.pred_w[i], kd));
                if ((i & 3) == h) sa += v;
                else sb += v;
            }
        }
        spx[(g * 4 + h) * H + o] = sa;
        spx[(g * 4 + h + 2) * H + o] = sb;
        __syncthreads();
        if (pt < G) {
            const float *sp = spx + pt * 4 * H + o;
            emit(first + pt, pool_of(pt), (sp[0] + sp[H]) + (sp[2 * H] + sp[3 * H]));
        }
    }
    EV_TICK(7);
    if (tick_on) atomicAdd((unsigned long long *)&Ln.ticks[kTickBase + 15], 1ull);
    }  // parts
    if (Ln.ticks && tid == 0) {                                  // every workgroup with work: first / last start, last end, longest stay (100 MHz clock)
        const unsigned long long now = (unsigned long long)device_ticks();
        atomicMin((unsigned long long *)&Ln.ticks[kTickBase + 8], (unsigned long long)wg_start_);
        atomicMax((unsigned long long *)&Ln.ticks[kTickBase + 9], (unsigned long long)wg_start_);
        atomicMax((unsigned long long *)&Ln.ticks[kTickBase + 10], now);
        atomicMax((unsigned long long *)&Ln.ticks[kTickBase + 11], now - (unsigned long long)wg_start_);
    }
}

constexpr int kEvalLds = (kEvalCap * kEvalLd + kTile * kLdt + 2 * H * kLdt + 32 * H + 6 * H + 2 * H) * 4   // A, T, Wl0, Wl1, part, tables, biases
                         + (GCC_GIN_MAX_LAYERS + 1) * H * 8 + 4 * H * 8                                      // pooled sums (fp64) + their partials
                         + (kTile + 1 + 32 + 3) / 4 * 16;                                                    // rpl, prow

template <bool kQ> __global__ __launch_bounds__(kThreads) void gin_eval_fused_kernel(EvalLaunch Ln)
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
    constexpr int kTickBase = 16;
    const bool tick_on = Ln.ticks && threadIdx.x == 0 && ((int)blockIdx.x & 7) == 0;
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

    if (Ln.split && eval_in_lds(n, a.row_ptr[n0 + n] - a.row_ptr[n0])) return;      // (workgroup-uniform) the LDS-resident kernel's
    LayerRegs regs = eval_request_layer<kQ, true>(a, 0);                   // (n == 0: harmless)

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
        if (l + 1 < L) regs = eval_request_layer<kQ, false>(a, l + 1);       // in flight during this layer
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
    if (tick_on) atomicAdd((unsigned long long *)&Ln.ticks[kTickBase + 15], 1ull);
}

// Three launches: the work list (and mean_out = 0), the LDS-resident kernel, then the general one (hub ego-nets of more than kMedCap nodes; 5 us when there are none).
// (Side by side on forked streams they gained 8 us as launches and lost 30 replayed from a hipGraph, plus 120 us of host time
// per call for the events: profiles/r6_eval_probe.txt's header.)
template <bool kQ> void eval_launch(const EvalLaunch &Ln, int B, int npass, hipStream_t s)
{
#ifndef GCC_AMD_HIPEMU
    static bool opted = false;                               // more than 64 KiB of dynamic LDS is opted into once
    if (!opted) {
        (void)hipFuncSetAttribute((const void *)gin_eval_fused_kernel<kQ>, hipFuncAttributeMaxDynamicSharedMemorySize, kEvalLds);
        (void)hipFuncSetAttribute((const void *)gin_eval_lds_kernel<kQ>, hipFuncAttributeMaxDynamicSharedMemorySize, kMedLds);
        opted = true;
    }
#endif
    hipLaunchKernelGGL(gin_eval_plan_kernel, dim3(npass), dim3(kPlanThreads), 0, s, Ln);
    hipLaunchKernelGGL(gin_eval_lds_kernel<kQ>, dim3(B, npass), dim3(kMedThreads), kMedLds, s, Ln);
    hipLaunchKernelGGL(gin_eval_fused_kernel<kQ>, dim3(B, npass), dim3(kThreads), kEvalLds, s, Ln);
}

}  // namespace

extern "C" {

/* diagnostics, as gcc_gin_debug_ticks: device int64[2][16] (per kernel) of wall-clock ticks per phase of gcc_gin_eval_fused (features,
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
        if (p.training || L < 1 || L > GCC_GIN_MAX_LAYERS || kdim0 > H || p.batch_size != B || B < 1 || !p.z1[0] || !p.z2[0] || !p.x0 ||
            !p.score || !p.feat || !p.node_off || !p.row_ptr || !p.pos) {               // (col_idx may be NULL: a batch without edges has none to read)
            snprintf(g_err, kErrLen, "gcc_gin_eval_fused: an eval-mode pass with x0 / z1[0] / z2[0] scratch, score and feat is needed "
                                     "(training=%d layers=%d d_in=%d B=%d)", p.training, L, kdim0, p.batch_size);
            return -2;
        }
        EvalArgs &a = Ln.p[i];
        a.node_off = p.node_off; a.row_ptr = p.row_ptr; a.col_idx = p.col_idx; a.seed_local = p.seed_local;
        a.pos = p.pos; a.emb = p.w.degree_embedding;
        a.g0 = p.z1[0]; a.g1 = p.z2[0]; a.plan = (int32_t *)p.x0;
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
    // the hidden-wide matrices come by 16-byte quads when their rows are whole quads (eval_weights_request)
    bool quads = true;
    for (int i = 0; i < npass; ++i) {
        const EvalArgs &a = Ln.p[i];
        quads = quads && (a.hid & 3) == 0;
        for (int l = 0; l < a.L; ++l)
            quads = quads && ((uintptr_t)a.layer[l].w1 & 15) == 0 && (l == 0 || ((uintptr_t)a.layer[l].w0 & 15) == 0);
    }
    hipStream_t s = (hipStream_t)stream;
    Ln.split = 1;
    Ln.npass = npass;
    if (quads) eval_launch<true>(Ln, B, npass, s);
    else eval_launch<false>(Ln, B, npass, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, kErrLen, "gcc_gin_eval_fused: launch failed: %s", hipGetErrorString(e));
        return -10;
    }
    return 0;
}

}  // extern "C"
