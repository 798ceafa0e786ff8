// gcc_amd/csrc/encoder_bwd.hip -- GIN encoder backward (gfx950).
//
// Backward of encoder.hip's forward, i.e. of loss.backward() through
// GraphEncoder (train.py:407-408; gin.py:213-232; graph_encoder.py:152-200).
// Per GIN layer l (reverse order), with g = dL/dh' (h' = layer output):
//   bwd_c   : g = dpooled_{l+1}[graph] + (I + A) dagg_{l+1}      (SumPooling + GINConv backward)
//             u = g * [h' > 0];             sums S(u), S(u*yhat2)  -> d gamma_c, d beta_c
//   bwd_b   : dy2 = BN_c'(u); v = dy2 * [y2 > 0]; S(v), S(v*zhat2) -> d gamma_b, d beta_b
//   bwd_lin1: dz2 = BN_b'(v); da1 = dz2 W1 (MFMA); w = da1 * [a1 > 0]; S(w), S(w*zhat1); S(dz2) = db1
//   bwd_lin0: dz1 = BN_a'(w); dagg = dz1 W0 (MFMA); S(dz1) = db0
// BN'(g) = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)) needs two column sums over all N nodes, so
// (like the forward) every BatchNorm is a kernel boundary; the sums are fp64 atomics.
// Normalised activations are recomputed from the stored Linear outputs z1/z2.
// Weight gradients: one MFMA kernel over all 2L (dZ, X) pairs writes per-chunk 64x64 slabs that
// a final kernel reduces in a fixed order (deterministic), together with the prediction-layer,
// BatchNorm, bias and degree-embedding gradients.
#include "encoder_common.h"

namespace {

constexpr int kWgChunks = 64;     // row chunks (slabs) per weight gradient

// per-column training-mode BatchNorm constants from the column sum s1 and sum of squares s2 of its input
struct BnCol { float mean, rstd, gamma, beta; };
__device__ __forceinline__ BnCol bn_col(const BnDev &bn, int c, double s1, double s2, double n, float eps)
{
    const double mean = s1 / n;
    double var = s2 / n - mean * mean;
    if (var < 0.0) var = 0.0;
    BnCol r;
    r.mean = (float)mean;
    r.rstd = (float)(1.0 / sqrt(var + (double)eps));
    r.gamma = bn.weight[c];
    r.beta = bn.bias[c];
    return r;
}

// LDS coefficient table rows (each [64] floats)
//   xhat = x * RS + RM            (RS = rstd, RM = -mean * rstd)
//   pre  = x * PS + PH            (gamma * xhat + beta)
//   dx   = g * K1 + x * K2 + K3   (BatchNorm backward as an affine map of (g, x))
enum { RS = 0, RM = 1, PS = 2, PH = 3, K1 = 4, K2 = 5, K3 = 6, kCoefRows = 7 };

// ALL threads call it (block-uniform); scratch = LDS [256] doubles that nothing else uses during the call (replica_sums128);
// ends with a barrier
__device__ __forceinline__ void fill_coefs(float *C /* [7][64] */, const BnDev &bn, const double *bst, double n,
                                           float eps, double *scratch)
{
    const int c = (int)threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    if (bn.totals) {                                   // (the forward pass's readout kernel added the replicas up)
        if (c < H) { s1 = bn.totals[c]; s2 = bn.totals[H + c]; }
    } else {
        replica_sums128(bn.stats, 2 * H, scratch);
        if (c < H) { s1 = scratch[c] + scratch[128 + c]; s2 = scratch[H + c] + scratch[128 + H + c]; }
        __syncthreads();
    }
    if (bst) replica_sums128(bst, 3 * H, scratch);     // slots 0 and 1 of the backward sums
    if (c < H) {
        const BnCol b = bn_col(bn, c, s1, s2, n, eps);
        C[RS * H + c] = b.rstd;
        C[RM * H + c] = -b.mean * b.rstd;
        C[PS * H + c] = b.gamma * b.rstd;
        C[PH * H + c] = b.beta - b.gamma * b.rstd * b.mean;
        if (bst) {
            const double b1 = scratch[c] + scratch[128 + c], b2 = scratch[H + c] + scratch[128 + H + c];
            const float m1 = (float)(b1 / n), m2 = (float)(b2 / n);
            const float k1 = b.gamma * b.rstd;
            C[K1 * H + c] = k1;
            C[K2 * H + c] = -k1 * m2 * b.rstd;
            C[K3 * H + c] = k1 * (m2 * b.mean * b.rstd - m1);
        }
    }
    __syncthreads();
}

// The same tables when the forward pass left the statistics' totals (every training pass of gcc_gin_forward does): coef_request
// / rep_request put every load in flight, fill_coefs_from consumes them -- a kernel requests all its tables, then the node
// count, and waits once (the chain was node count -> totals -> backward sums -> weights, per table).  ALL threads call
// them (block-uniform); fill_coefs_from does NOT end with a barrier: the caller synchronises once after its last table.
struct CoefReq { double s1, s2; float gamma, beta; };
__device__ __forceinline__ CoefReq coef_request(const BnDev &bn)
{
    const int c = (int)threadIdx.x & (H - 1);
    CoefReq r = {bn.totals[c], bn.totals[H + c], bn.weight[c], bn.bias[c]};
    return r;
}
template <bool kBst>
__device__ __forceinline__ void fill_coefs_from(float *C /* [7][64] */, const CoefReq &r, const RepReq &bst, double n,
                                                float eps, double *scratch)
{
    if (kBst) {
        scratch[threadIdx.x] = rep_sum(bst);         // slots 0 and 1 of the backward sums
        __syncthreads();
    }
    // every thread computes channel t & 63 and stores it (four threads store the same value): under `if (t < 64)` the
    // compiler sinks the requests into the branch, behind the wait for the node count -- a second round trip
    const int c = (int)threadIdx.x & (H - 1);
    {
        const double mean = r.s1 / n;
        double var = r.s2 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float meanf = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
        C[RS * H + c] = rstd;
        C[RM * H + c] = -meanf * rstd;
        C[PS * H + c] = r.gamma * rstd;
        C[PH * H + c] = r.beta - r.gamma * rstd * meanf;
        if (kBst) {
            const double b1 = scratch[c] + scratch[128 + c], b2 = scratch[H + c] + scratch[128 + H + c];
            const float m1 = (float)(b1 / n), m2 = (float)(b2 / n);
            const float k1 = r.gamma * rstd;
            C[K1 * H + c] = k1;
            C[K2 * H + c] = -k1 * m2 * rstd;
            C[K3 * H + c] = k1 * (m2 * meanf * rstd - m1);
        }
    }
}

__device__ __forceinline__ F4 fma4(F4 a, F4 b, F4 c)
{
    F4 r = {fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w)};
    return r;
}
__device__ __forceinline__ F4 relu4(F4 a) { F4 r = {fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)}; return r; }
__device__ __forceinline__ F4 mask4(F4 g, F4 pre)
{
    F4 r = {pre.x > 0.f ? g.x : 0.f, pre.y > 0.f ? g.y : 0.f, pre.z > 0.f ? g.z : 0.f, pre.w > 0.f ? g.w : 0.f};
    return r;
}
__device__ __forceinline__ F4 zero4() { F4 r = {0.f, 0.f, 0.f, 0.f}; return r; }

// block reduction of per-thread column sums held in the 16-lane-group layout
// (thread -> columns 4t .. 4t+3, 16 groups) into fp64 atomics on dst0 / dst1
__device__ __forceinline__ void reduce_groups(float *part /* [16][128] */, F4 s1, F4 s2, double *dst0, double *dst1)
{
    const int tid = (int)threadIdx.x, t = tid & 15, gi = tid >> 4;
    st4(&part[gi * 2 * H + 4 * t], s1);
    st4(&part[gi * 2 * H + H + 4 * t], s2);
    __syncthreads();
    if (tid < 2 * H) {
        double v = 0.0;
        for (int k = 0; k < 16; ++k) v += (double)part[k * 2 * H + tid];
        const int rep = ((int)blockIdx.x % kRep) * 3 * H;
        if (tid < H) atomicAdd(&dst0[rep + tid], v);
        else atomicAdd(&dst1[rep + tid - H], v);
    }
    __syncthreads();
}

// sum over the 16 lanes that share q (= the 16 rows of a wave's block)
// sum over the 16 rows of a wave's tile (one DPP row); valid in the lane with (lane & 15) == 15
__device__ __forceinline__ float sum_rows16(float v) { return row16_sum_last(v); }

// =========================================================================
// R1: d score, G_i = dscore * keep_i / (1 - p), dpooled_i = G_i W_i      (gin.py:227-230, graph_encoder.py:196)
// one wave = 16 graphs; G_i in the MFMA output layout is already the input layout of the next MFMA
struct ReadBwdArgs {
    const float *dfeat, *score, *feat;
    DropCfg drop;
    const float *pred_w[GCC_GIN_MAX_LAYERS + 1];
    float *G, *dpooled;       // [L+1][B][64]
    int32_t B, nlayers, kdim0, normalize;
    float norm_eps;
    int32_t hid;              // columns of pred_w[i > 0]
    float4 *zero;             // workgroups past the first `nread` clear this region (the accumulators of the kernels that follow)
    int64_t zero16;
    int32_t nread;
};

__global__ __launch_bounds__(kThreads) void gin_bwd_readout_kernel(ReadBwdArgs a)
{
    TRAIN_STEP_WAVE_PRIORITY();
    if ((int)blockIdx.x >= a.nread) {              // block-uniform
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const int64_t stride = (int64_t)((int)gridDim.x - a.nread) * kThreads;
        for (int64_t i = (int64_t)((int)blockIdx.x - a.nread) * kThreads + threadIdx.x; i < a.zero16; i += stride) a.zero[i] = z4;
        return;
    }
    // one wave = (16 graphs, one prediction layer): the layers of a block of graphs run side by side instead of as a
    // chain of 5 dependent (weights round trip -> product -> store) steps in one wave; d score is recomputed per layer
    const int lane = lane_id(), wv = (int)threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const int nl1 = a.nlayers + 1, job = wave_uniform((int)blockIdx.x * 4 + wv);   // (scalar: pred_w[i] is a scalar load)
    const int i = job % nl1, blk = job / nl1;
    if (blk * 16 >= a.B) return;                   // wave-uniform (the kernel has no barriers)
    const int b = blk * 16 + j;
    const bool valid = b < a.B;
    const DropCfg drop = drop_resolve(a.drop);
    F4 wf[4][4];
    load_wt_frags(a.pred_w[i], i == 0 ? a.kdim0 : a.hid, wf);          // requested first: in flight with d feat / score / feat
    F4 ds[4];
    float ss = 0.f, dot = 0.f;
    F4 fv[4], sv[4];
    {   // d feat, score and feat of the graph: all 12 loads requested together (unconditional, masked afterwards)
        const int64_t rb = (int64_t)(valid ? b : 0) * H + 4 * q;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            ds[cb] = ld4(a.dfeat + rb + 16 * cb);
            sv[cb] = ld4(a.score + rb + 16 * cb);
            fv[cb] = ld4(a.feat + rb + 16 * cb);
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const F4 z = {0.f, 0.f, 0.f, 0.f};
            if (!valid) ds[cb] = z;
            if (a.normalize && valid) {
                const F4 s = sv[cb];
                ss += s.x * s.x + s.y * s.y + s.z * s.z + s.w * s.w;
                dot += fv[cb].x * ds[cb].x + fv[cb].y * ds[cb].y + fv[cb].z * ds[cb].z + fv[cb].w * ds[cb].w;
            } else {
                fv[cb] = z;
            }
        }
    }
    if (a.normalize) {
        ss += wave_shfl_xor(ss, 16); ss += wave_shfl_xor(ss, 32);
        dot += wave_shfl_xor(dot, 16); dot += wave_shfl_xor(dot, 32);
        const float nrm = sqrtf(ss);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            if (nrm > a.norm_eps) {
                ds[cb].x = (ds[cb].x - fv[cb].x * dot) / nrm; ds[cb].y = (ds[cb].y - fv[cb].y * dot) / nrm;
                ds[cb].z = (ds[cb].z - fv[cb].z * dot) / nrm; ds[cb].w = (ds[cb].w - fv[cb].w * dot) / nrm;
            } else {
                ds[cb].x /= a.norm_eps; ds[cb].y /= a.norm_eps; ds[cb].z /= a.norm_eps; ds[cb].w /= a.norm_eps;
            }
        }
    }
    {
        F4 g[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int ch = 16 * cb + 4 * q;
            F4 m = {0.f, 0.f, 0.f, 0.f};
            if (valid) m = drop_mul4(drop, i, b, ch);
            F4 gg = {ds[cb].x * m.x, ds[cb].y * m.y, ds[cb].z * m.z, ds[cb].w * m.w};
            g[cb] = gg;
            if (valid) st4(a.G + ((int64_t)i * a.B + b) * H + ch, gg);
        }
        f32x4 acc[4];
        mfma_rows16(g, wf, acc);                                   // dpooled_i = G_i W_i
        if (valid) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                F4 o = {acc[cb][0], acc[cb][1], acc[cb][2], acc[cb][3]};
                st4(a.dpooled + ((int64_t)i * a.B + b) * H + 16 * cb + 4 * q, o);
            }
        }
    }
}

// =========================================================================
// Bk1
struct BwdCArgs {
    const int32_t *node_off, *row_ptr, *col_idx, *graph_id;
    const float *D;           // dagg of layer l+1, or NULL for the last layer
    const float *dpooled;     // [B][64] of hidden_rep[l+1]
    const float *z2;
    BnDev bnb, bnc;
    float *U;
    double *bst_c;            // [3][64]
    int32_t B;
    float eps;
    int32_t cap;              // node capacity of the per-node buffers
};

// (the grid is 3 workgroups per CU: 168 registers each)
#ifndef BWD_C_PER_CU
#define BWD_C_PER_CU 3
#endif
#ifndef BWD_GATHER_J
#define BWD_GATHER_J 8
#endif
__global__ __launch_bounds__(kThreads, BWD_C_PER_CU) void gin_bwd_c_kernel(BwdCArgs a)
{
    TRAIN_STEP_WAVE_PRIORITY();
    __shared__ float T[kTile * kLdt];
    __shared__ __attribute__((aligned(16))) float part[16 * 2 * H];      // (also the fp64 scratch of fill_coefs)
    __shared__ float Cb[kCoefRows * H], Cc[kCoefRows * H];
    __shared__ int prow[32];
    __shared__ int rpl[kTile + 1];
    const int tid = (int)threadIdx.x, t = tid & 15, gi = tid >> 4;
    int N;
    const float *Dp = a.D ? a.D : a.z2;              // (last layer: no D -- the requests below stay unconditional and read z2 twice)
    auto ident = [&](int u) -> F4 { return ld4(Dp + (int64_t)u * H + 4 * t); };
    // the lane group's 4 rows of the FIRST tile: graph ids, z2 rows, the tile's own rows of D and its row pointers are requested
    // together with the node count and the statistics (addresses clamped to the capacity, rows >= N dropped below); the
    // pooled-path gradients (which need the graph ids) go out with the gather's first request.  Every load unconditional (a loop
    // over r with its loads inside ran 8 dependent round trips, after the gather)
    const int tf = first_tile();
    int gid4[kTile / 16], rp_own = 0;
    F4 own[kTile / 16], z4[kTile / 16];
    auto request = [&](int tile0) {
#pragma unroll
        for (int i = 0; i < kTile / 16; ++i) gid4[i] = a.graph_id[cap_row(tile0 + gi + 16 * i, a.cap)];
        rp_own = a.row_ptr[min(tile0 + min(tid, kTile), a.cap)];       // (no branch on a.D here: loads behind a branch are a batch of their own)
#pragma unroll
        for (int i = 0; i < kTile / 16; ++i) own[i] = ident(cap_row(tile0 + gi + 16 * i, a.cap));
#pragma unroll
        for (int i = 0; i < kTile / 16; ++i) z4[i] = ld4(a.z2 + (int64_t)cap_row(tile0 + gi + 16 * i, a.cap) * H + 4 * t);
    };
    if (a.bnb.totals && a.bnc.totals) {            // block-uniform; the usual case
        const CoefReq rb = coef_request(a.bnb), rc = coef_request(a.bnc);
        const RepReq none = {};
        request(tf * kTile);
        N = a.node_off[a.B];                       // (requested last: the wait for it is the wait for all)
        SCHED_FENCE();
        if (no_tiles(N)) return;
        fill_coefs_from<false>(Cb, rb, none, (double)N, a.eps, (double *)part);
        fill_coefs_from<false>(Cc, rc, none, (double)N, a.eps, (double *)part);
        __syncthreads();
    } else {
        request(tf * kTile);
        N = ((const volatile int32_t *)a.node_off)[a.B];      // (volatile: an ordinary load is hoisted above the branch, ahead of the other arm's requests)
        fill_coefs(Cb, a.bnb, nullptr, (double)N, a.eps, (double *)part);
        fill_coefs(Cc, a.bnc, nullptr, (double)N, a.eps, (double *)part);
    }
    F4 s1 = zero4(), s2 = zero4();
    bool any = false;
    for (TileWalk tw(N); tw.ti < tw.tend; tw.ti += tw.step) {
        const int tile0 = tw.ti * kTile;
        any = true;
        const int nrows = min(kTile, N - tile0);
        if (tw.ti != tf) request(tile0);
        F4 g4[kTile / 16];
        auto pooled_path = [&] {                     // d pooled of the rows' graphs (ids of rows >= N clamped: not used)
#pragma unroll
            for (int i = 0; i < kTile / 16; ++i) g4[i] = ld4(a.dpooled + (int64_t)min(max(gid4[i], 0), a.B - 1) * H + 4 * t);
        };
        if (a.D) {                                   // block-uniform
            if (tid <= nrows) rpl[tid] = rp_own;
#pragma unroll
            for (int i = 0; i < kTile / 16; ++i) st4(&T[(gi + 16 * i) * kLdt + 4 * t], gi + 16 * i < nrows ? own[i] : zero4());
            __syncthreads();
            gather_tile<BWD_GATHER_J>(T, part, prow, nrows, a.col_idx, ident, [](F4 x) { return x; }, 1.0f, rpl, pooled_path);
        } else {
            pooled_path();
        }
        {
            // (the coefficient rows are read from LDS here, not held in 24 registers across the gather)
            const F4 pbs = ld4(&Cb[PS * H + 4 * t]), pbh = ld4(&Cb[PH * H + 4 * t]);
            const F4 rcs = ld4(&Cc[RS * H + 4 * t]), rcm = ld4(&Cc[RM * H + 4 * t]);
            const F4 pcs = ld4(&Cc[PS * H + 4 * t]), pch = ld4(&Cc[PH * H + 4 * t]);
#pragma unroll
            for (int i = 0; i < kTile / 16; ++i) {
                const int r = gi + 16 * i, v = tile0 + r;
                if (r < nrows) {
                    F4 g = g4[i];
                    if (a.D) g = add4(g, ld4(&T[r * kLdt + 4 * t]));
                    const F4 y2 = relu4(fma4(z4[i], pbs, pbh));
                    const F4 yh = fma4(y2, rcs, rcm);
                    const F4 u = mask4(g, fma4(y2, pcs, pch));
                    st4(a.U + (int64_t)v * H + 4 * t, u);
                    s1 = add4(s1, u);
                    s2 = fma4(u, yh, s2);
                }
            }
        }
        __syncthreads();
    }
    if (!any) return;
    reduce_groups(part, s1, s2, a.bst_c, a.bst_c + H);
}

// =========================================================================
// Bk2
struct BwdBArgs {
    const int32_t *node_off;
    const float *U, *z2;
    BnDev bnb, bnc;
    const double *bst_c;
    float *V;
    double *bst_b;
    int32_t B;
    float eps;
    int32_t cap;              // node capacity of the per-node buffers
};

__global__ __launch_bounds__(kThreads) void gin_bwd_b_kernel(BwdBArgs a)
{
    TRAIN_STEP_WAVE_PRIORITY();
    __shared__ __attribute__((aligned(16))) float part[16 * 2 * H];      // (also the fp64 scratch of fill_coefs)
    __shared__ float Cb[kCoefRows * H], Cc[kCoefRows * H];
    const int tid = (int)threadIdx.x, t = tid & 15, gi = tid >> 4;
    int N;
    // the first tile's rows of z2 and U ride in the round trip of the node count and the statistics (clamped to the capacity)
    const int tf = first_tile();
    F4 z4[kTile / 16], u4[kTile / 16];              // the lane group's 4 rows of z2 and U, requested together
    auto request = [&](int tile0) {
#pragma unroll
        for (int i = 0; i < kTile / 16; ++i) {
            const int64_t off = (int64_t)cap_row(tile0 + gi + 16 * i, a.cap) * H + 4 * t;
            z4[i] = ld4(a.z2 + off);
            u4[i] = ld4(a.U + off);
        }
    };
    if (a.bnb.totals && a.bnc.totals) {            // block-uniform; the usual case
        const CoefReq rb = coef_request(a.bnb), rc = coef_request(a.bnc);
        const RepReq sc = rep_request(a.bst_c, 3 * H);
        request(tf * kTile);
        N = a.node_off[a.B];                       // (requested last: the wait for it is the wait for all)
        SCHED_FENCE();
        if (no_tiles(N)) return;
        fill_coefs_from<false>(Cb, rb, sc, (double)N, a.eps, (double *)part);
        fill_coefs_from<true>(Cc, rc, sc, (double)N, a.eps, (double *)part);
        __syncthreads();
    } else {
        request(tf * kTile);
        N = ((const volatile int32_t *)a.node_off)[a.B];      // (volatile: an ordinary load is hoisted above the branch, ahead of the other arm's requests)
        fill_coefs(Cb, a.bnb, nullptr, (double)N, a.eps, (double *)part);
        fill_coefs(Cc, a.bnc, a.bst_c, (double)N, a.eps, (double *)part);
    }
    const F4 pbs = ld4(&Cb[PS * H + 4 * t]), pbh = ld4(&Cb[PH * H + 4 * t]);
    const F4 rbs = ld4(&Cb[RS * H + 4 * t]), rbm = ld4(&Cb[RM * H + 4 * t]);
    const F4 k1 = ld4(&Cc[K1 * H + 4 * t]), k2 = ld4(&Cc[K2 * H + 4 * t]), k3 = ld4(&Cc[K3 * H + 4 * t]);
    F4 s1 = zero4(), s2 = zero4();
    bool any = false;
    for (TileWalk tw(N); tw.ti < tw.tend; tw.ti += tw.step) {
        const int tile0 = tw.ti * kTile;
        any = true;
        if (tw.ti != tf) request(tile0);
#pragma unroll
        for (int i = 0; i < kTile / 16; ++i) {
            const int v = tile0 + gi + 16 * i;
            if (v < N) {
                const F4 z = z4[i];
                const F4 pre = fma4(z, pbs, pbh);
                const F4 y2 = relu4(pre);
                const F4 dy2 = fma4(u4[i], k1, fma4(y2, k2, k3));   // BN_c backward
                const F4 vv = mask4(dy2, pre);
                st4(a.V + (int64_t)v * H + 4 * t, vv);
                s1 = add4(s1, vv);
                s2 = fma4(vv, fma4(z, rbs, rbm), s2);
            }
        }
    }
    if (!any) return;
    reduce_groups(part, s1, s2, a.bst_b, a.bst_b + H);
}

// =========================================================================
// Bk3 / Bk4 share one kernel: dz = BN'(gin) ; store dz ; S(dz) ; dx = dz W (MFMA);
//   kMask: out = dx * [pre(zout) > 0], S(out), S(out * xhat(zout))      (Bk3: through relu(bn_a))
//   else : out = dx                                                      (Bk4: dagg)
struct BwdLinArgs {
    const int32_t *node_off;
    const float *gin, *zin;   // upstream gradient and the Linear output feeding this BN (V,z2 | Wt,z1)
    BnDev bn_in;              // BN applied to zin (bn_b | bn_a)
    const double *bst_in;     // its backward sums [3][64] (slot 2 receives S(dz))
    double *bst_bias;         // == bst_in + 2*H
    const float *W;           // [64][kdim] the Linear being back-propagated through (W1 | W0)
    int32_t kdim;
    float *dz;                // [N][64] stored for the weight gradient
    float *out;               // Wt | D
    const float *zout;        // z1 (kMask) or NULL
    BnDev bn_out;             // bn_a (kMask)
    double *bst_out;          // [3][64] slots 0,1 (kMask)
    int32_t B;
    float eps;
    int32_t cap;              // node capacity of the per-node buffers
};

template <bool kMask>
__global__ __launch_bounds__(kThreads, 4) void gin_bwd_lin_kernel(BwdLinArgs a)
{
    TRAIN_STEP_WAVE_PRIORITY();
    __shared__ float Ci[kCoefRows * H], Co[kCoefRows * H];
    __shared__ __attribute__((aligned(16))) float red[4 * 3 * H];        // (also the fp64 scratch of fill_coefs)
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = tid >> 6, j = lane & 15, q = lane >> 4;
    __shared__ float Wt[H * kLdt];                 // W transposed, staged once per workgroup (every wave fetched all 16
                                                   // fragments as 64 strided 4-byte loads per lane before)
    const WStage wst = stage_weights_request(a.W, a.kdim);      // in flight with N and the statistics
    int N;
    // every load of the FIRST tile is issued with the node count, the statistics and the weights: one round trip.  Unconditional,
    // clamped to the capacity (a lane past the live rows reads a row that exists and drops it), because loads under per-block
    // `if (valid)` branches came out as one round trip each
    const int tf = first_tile();
    F4 gq[4], zq[4], zo[4];
    auto request = [&](int tile0) {
        const int64_t rbase = (int64_t)cap_row(tile0 + 16 * wv + j, a.cap) * H;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            gq[c] = ld4(a.gin + rbase + 16 * c + 4 * q);
            zq[c] = ld4(a.zin + rbase + 16 * c + 4 * q);
            zo[c] = kMask ? ld4(a.zout + rbase + 16 * c + 4 * q) : zero4();
        }
    };
    if (a.bn_in.totals && (!kMask || a.bn_out.totals)) {      // block-uniform; the usual case
        const CoefReq ri = coef_request(a.bn_in), ro = kMask ? coef_request(a.bn_out) : CoefReq();
        const RepReq si = rep_request(a.bst_in, 3 * H);
        request(tf * kTile);
        N = a.node_off[a.B];                       // (requested last: the wait for it is the wait for all)
        SCHED_FENCE();
        if (no_tiles(N)) return;
        fill_coefs_from<true>(Ci, ri, si, (double)N, a.eps, (double *)red);
        if (kMask) fill_coefs_from<false>(Co, ro, si, (double)N, a.eps, (double *)red);
    } else {
        request(tf * kTile);
        N = ((const volatile int32_t *)a.node_off)[a.B];      // (volatile: an ordinary load is hoisted above the branch, ahead of the other arm's requests)
        fill_coefs(Ci, a.bn_in, a.bst_in, (double)N, a.eps, (double *)red);
        if (kMask) fill_coefs(Co, a.bn_out, nullptr, (double)N, a.eps, (double *)red);
    }
    stage_weights_store_t(Wt, wst, a.kdim);
    __syncthreads();
    float *myred = &red[wv * 3 * H];
    for (TileWalk tw(N); tw.ti < tw.tend; tw.ti += tw.step) {
        const int tile0 = tw.ti * kTile;
        const int row = tile0 + 16 * wv + j;
        const bool valid = row < N;
        if (tw.ti != tf) request(tile0);
        F4 xb[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = 16 * c + 4 * q;
            F4 d = zero4();
            if (valid) {
                d = fma4(gq[c], ld4(&Ci[K1 * H + col]), fma4(zq[c], ld4(&Ci[K2 * H + col]), ld4(&Ci[K3 * H + col])));
                st4(a.dz + (int64_t)row * H + col, d);
            }
            xb[c] = d;
            const float bx = sum_rows16(d.x), by = sum_rows16(d.y), bz = sum_rows16(d.z), bw = sum_rows16(d.w);
            if (j == 15) { myred[2 * H + col] = bx; myred[2 * H + col + 1] = by; myred[2 * H + col + 2] = bz; myred[2 * H + col + 3] = bw; }
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {               // one 16-channel block at a time
            const int col = 16 * cb + 4 * q;
            F4 wf[4];                                  // output index = input column of W, reduction index = output row of W
#pragma unroll
            for (int c = 0; c < 4; ++c) wf[c] = ld4(&Wt[(16 * cb + j) * kLdt + 16 * c + 4 * q]);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc = mfma_16x16x4_f32(wf[c].x, xb[c].x, acc);
                acc = mfma_16x16x4_f32(wf[c].y, xb[c].y, acc);
                acc = mfma_16x16x4_f32(wf[c].z, xb[c].z, acc);
                acc = mfma_16x16x4_f32(wf[c].w, xb[c].w, acc);
            }
            F4 o = {acc[0], acc[1], acc[2], acc[3]};
            if (kMask) {
                const F4 z = zo[cb];
                o = valid ? mask4(o, fma4(z, ld4(&Co[PS * H + col]), ld4(&Co[PH * H + col]))) : zero4();
                const F4 xh = fma4(z, ld4(&Co[RS * H + col]), ld4(&Co[RM * H + col]));
                const float s0 = sum_rows16(o.x), s1 = sum_rows16(o.y), s2 = sum_rows16(o.z), s3 = sum_rows16(o.w);
                const float t0 = sum_rows16(o.x * xh.x), t1 = sum_rows16(o.y * xh.y), t2 = sum_rows16(o.z * xh.z),
                            t3 = sum_rows16(o.w * xh.w);
                if (j == 15) {
                    myred[col] = s0; myred[col + 1] = s1; myred[col + 2] = s2; myred[col + 3] = s3;
                    myred[H + col] = t0; myred[H + col + 1] = t1; myred[H + col + 2] = t2; myred[H + col + 3] = t3;
                }
            }
            if (valid) st4(a.out + (int64_t)row * H + col, o);
        }
        __syncthreads();
        if (tid < 3 * H) {
            const double v = (double)red[tid] + (double)red[3 * H + tid] + (double)red[6 * H + tid] + (double)red[9 * H + tid];
            const int rep = ((int)blockIdx.x % kRep) * 3 * H;
            if (tid >= 2 * H) atomicAdd(&a.bst_bias[rep + tid - 2 * H], v);
            else if (kMask) atomicAdd(&a.bst_out[rep + tid], v);
        }
        __syncthreads();
    }
}

// =========================================================================
// degree-embedding gradient: dx0 = (I + A) dagg_0 + dpooled_0[graph]; d emb[clamp(deg)] += dx0[:, pos:pos+emb].
// Many nodes share a (small) degree, so global atomics would serialise on a few cache lines:
// each block accumulates into an LDS table (column c is owned by thread c: no atomics, fixed order)
// and writes its table as one partial; the final kernel sums the kEmbBlocks partials.
constexpr int kEmbBlocks = 448;      // one 64-row tile per workgroup at bsz 256 (~390 tiles of the query view); 2 workgroups of 67 KiB fit a CU
constexpr int kEmbMaxElems = 10240;      // (max_degree + 1) * deg_emb_dim floats of LDS (40 KiB)

struct EmbArgs {
    const int32_t *node_off, *row_ptr, *col_idx, *graph_id;
    const float *D, *dpooled0;
    float *demb_parts;        // [kEmbBlocks][(max_degree + 1) * emb_dim]
    int32_t B, pos_dim, emb_dim, max_degree;
    int32_t cap;              // node capacity of the per-node buffers
};

__global__ __launch_bounds__(kThreads) void gin_bwd_emb_kernel(EmbArgs a)
{
    TRAIN_STEP_WAVE_PRIORITY();
    __shared__ float T[kTile * kLdt];
    __shared__ float part[32 * H];
    __shared__ float E[kEmbMaxElems];
    __shared__ int dclrow[kTile], gidl[kTile];
    __shared__ int prow[32];
    __shared__ int rpl[kTile + 1];
    const int tid = (int)threadIdx.x, t = tid & 15, gi = tid >> 4;
    // only the degree-embedding columns [pos_dim, pos_dim + emb_dim) of dx0 are used: the lanes whose four columns lie outside them
    // request nothing (a quarter of the gather's bytes at pos 32 / emb 16) and carry zeros
    const bool my_cols = 4 * t + 3 >= a.pos_dim && 4 * t < a.pos_dim + a.emb_dim;
    auto ident = [&](int u) -> F4 { return my_cols ? ld4(a.D + (int64_t)u * H + 4 * t) : zero4(); };
    // row pointers, graph ids and the lane group's 4 rows of the FIRST tile: requested together with the node count (clamped to
    // the capacity), stored afterwards (a load under `if (tid < ...)` next to its LDS store is a round trip of its own)
    const int tf = first_tile();
    int rp_own, gid_own;
    F4 own[kTile / 16];
    auto request = [&](int tile0) {
        rp_own = a.row_ptr[min(tile0 + min(tid, kTile), a.cap)];
        gid_own = a.graph_id[cap_row(tile0 + (tid & (kTile - 1)), a.cap)];
#pragma unroll
        for (int i = 0; i < kTile / 16; ++i) own[i] = ident(cap_row(tile0 + gi + 16 * i, a.cap));
    };
    request(tf * kTile);
    const int N = a.node_off[a.B];
    SCHED_FENCE();
    const int elems = (a.max_degree + 1) * a.emb_dim;
    for (int i = tid; i < elems; i += kThreads) E[i] = 0.f;
    for (TileWalk tw(N); tw.ti < tw.tend; tw.ti += tw.step) {
        const int tile0 = tw.ti * kTile;
        const int nrows = min(kTile, N - tile0);
        {
            if (tw.ti != tf) request(tile0);
            if (tid <= nrows) rpl[tid] = rp_own;
            if (tid >= 128 && tid - 128 < nrows) gidl[tid - 128] = gid_own;
#pragma unroll
            for (int i = 0; i < kTile / 16; ++i) st4(&T[(gi + 16 * i) * kLdt + 4 * t], gi + 16 * i < nrows ? own[i] : zero4());
        }
        __syncthreads();
        // (the pooled-path gradients of a batch requested together, graph ids from LDS: 8 dependent round trips before; the first
        //  kIt * kThreads of them -- all of them at emb_dim 16 -- go out behind the gather's first request of neighbour ids)
        constexpr int kIt = 4;                       // covers kTile rows x 16 columns with 256 threads; more columns loop below
        const int total = nrows * a.emb_dim;
        int rr[kIt], cc[kIt];
        float dv[kIt];
        auto pooled_request = [&](int base) {
#pragma unroll
            for (int i = 0; i < kIt; ++i) {
                const int idx = min(base + tid + i * kThreads, total - 1);
                rr[i] = idx / a.emb_dim; cc[i] = idx - rr[i] * a.emb_dim;
            }
#pragma unroll
            for (int i = 0; i < kIt; ++i) dv[i] = a.dpooled0[(int64_t)gidl[rr[i]] * H + a.pos_dim + cc[i]];
        };
        gather_tile<BWD_GATHER_J>(T, part, prow, nrows, a.col_idx, ident, [](F4 x) { return x; }, 1.0f, rpl, [&] { pooled_request(0); });
        // all threads: add the pooled-path gradient and look the clamped degree up ...
        if (tid < nrows) {
            const int deg = rpl[tid + 1] - rpl[tid];
            dclrow[tid] = deg < a.max_degree ? deg : a.max_degree;
        }
        for (int base = 0; base < total; base += kIt * kThreads) {
            if (base) pooled_request(base);
#pragma unroll
            for (int i = 0; i < kIt; ++i)
                if (base + tid + i * kThreads < total) T[rr[i] * kLdt + a.pos_dim + cc[i]] += dv[i];
        }
        __syncthreads();
        // ... then column c is owned by thread c: LDS only, fixed order, no atomics
        if (tid < a.emb_dim)
            for (int r = 0; r < nrows; ++r) E[dclrow[r] * a.emb_dim + tid] += T[r * kLdt + a.pos_dim + tid];
        __syncthreads();
    }
    for (int i = tid; i < elems; i += kThreads) a.demb_parts[(int64_t)blockIdx.x * elems + i] = E[i];
}

// =========================================================================
// weight gradients: slab[y][chunk] = sum over the chunk's rows of dZ^T X     (y = 2*l + which)
struct WgradJob {
    const float *dZ, *X;
    const double *Xd;         // prediction layers: X = pooled_i (fp64), rows = graphs
    BnDev bn;                 // which == 1: X = relu(bn_a(z1)); which == 0: X = agg (bn.weight == NULL)
    int32_t rows_fixed;       // > 0: number of rows (B); 0: node_off[B]
};
constexpr int kMaxWgJobs = 3 * GCC_GIN_MAX_LAYERS + 1;
struct WgradArgs {
    const int32_t *node_off;
    WgradJob job[kMaxWgJobs];
    float *slabs;             // [njobs][kWgChunks][64 * 64]
    float *bias_slabs;        // [njobs][kWgChunks][64]   column sums of dZ
    int32_t B;
    float eps;
    int32_t cap;              // node capacity of the per-node buffers
};

__global__ __launch_bounds__(kThreads, 3) void gin_wgrad_kernel(WgradArgs a)
{
    TRAIN_STEP_WAVE_PRIORITY();
    __shared__ float S[H * H];
    __shared__ float Cx[2 * H];
    const int jid = (int)blockIdx.y;
    const WgradJob &jb = a.job[jid];
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = tid >> 6, j = lane & 15, q = lane >> 4;
    __shared__ float Sb[4 * H];
    const bool act = jb.bn.weight != nullptr;
    const bool fast = act && jb.bn.totals != nullptr;    // block-uniform: the statistics' totals exist (the usual case)
    // the BatchNorm numbers of channel tid & 63, the node count and (below) the first tile's rows: one round trip
    // (the chain was node count -> totals -> weights -> first tile)
    // (unconditional: a job without a BatchNorm reads 4 numbers of the slab workspace and ignores them -- under `if (fast)`
    //  the requests wait inside the branch, ahead of the node count)
    const int cc = tid & (H - 1);
    const double *tp = fast ? jb.bn.totals : (const double *)a.slabs;
    const float *wp = fast ? jb.bn.weight : a.slabs, *bp = fast ? jb.bn.bias : a.slabs;
    const CoefReq cr = {tp[cc], tp[H + cc], wp[cc], bp[cc]};
    const int rcap = jb.rows_fixed > 0 ? jb.rows_fixed : a.cap;     // rows the job's operands hold
    f32x4 acc[4][4];
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[ob][kb] = z; }
    // a workgroup walks 6-7 row tiles; with the loads of a tile issued right before its products every tile cost a full
    // memory round trip (58 us for ~6 us of matrix work): the next tile's rows are requested before this tile's products,
    // and the FIRST tile's with the node count (clamped to the operands' capacity; rows >= N zeroed in mask_tile)
    float av[4][4], xv[4][4];
    // (unconditional loads from a clamped row, masked afterwards: under `row < N` branches every load waited for the one before)
    const bool xdouble = jb.Xd != nullptr;               // block-uniform: prediction layers read the fp64 pooled sums
    auto fetch = [&](int tile0, float (&A)[4][4], float (&X)[4][4]) {
        int64_t off[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int row = tile0 + 16 * wv + 4 * s + q;     // MFMA reduction index = row
            off[s] = (int64_t)cap_row(row, rcap) * H + j;
#pragma unroll
            for (int k = 0; k < 4; ++k) A[s][k] = jb.dZ[off[s] + 16 * k];
        }
        if (xdouble) {                                       // (8 requested before the first conversion waits: 16 would spill)
#pragma unroll
            for (int h = 0; h < 4; h += 2) {
                double xd[2][4];
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int k = 0; k < 4; ++k) xd[s][k] = jb.Xd[off[h + s] + 16 * k];
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int k = 0; k < 4; ++k) X[h + s][k] = (float)xd[s][k];
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int k = 0; k < 4; ++k) X[s][k] = jb.X[off[s] + 16 * k];
        }
    };
    auto mask_tile = [&](int tile0, int N, float (&A)[4][4], float (&X)[4][4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool in = tile0 + 16 * wv + 4 * s + q < N;
#pragma unroll
            for (int k = 0; k < 4; ++k) { A[s][k] = in ? A[s][k] : 0.f; X[s][k] = in ? X[s][k] : 0.f; }
        }
    };
    const int tf = first_tile();
    fetch(tf * kTile, av, xv);
    const int Nn = a.node_off[a.B];
    SCHED_FENCE();
    const int N = jb.rows_fixed > 0 ? jb.rows_fixed : Nn;
    TileWalk tw(N);
    if (tid < H) {
        float sc = 1.f, sh = 0.f;
        if (fast) {
            const double mean = cr.s1 / (double)Nn;
            double var = cr.s2 / (double)Nn - mean * mean;
            if (var < 0.0) var = 0.0;
            const double rstd = 1.0 / sqrt(var + (double)a.eps);
            sc = (float)((double)cr.gamma * rstd);
            sh = (float)((double)cr.beta - mean * (double)cr.gamma * rstd);
        } else if (act) {
            bn_scale_shift(jb.bn, tid, (double)Nn, a.eps, 1, sc, sh);
        }
        Cx[tid] = sc;
        Cx[H + tid] = sh;
    }
    __syncthreads();
    float xs[4], xh[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) { xs[kb] = Cx[16 * kb + j]; xh[kb] = Cx[H + 16 * kb + j]; }
    for (; tw.ti < tw.tend; tw.ti += tw.step) {
        const int tile0 = tw.ti * kTile;
        float an[4][4], xn[4][4];
        const bool more = tw.ti + tw.step < tw.tend;         // block-uniform
        if (more) fetch((tw.ti + tw.step) * kTile, an, xn);
        mask_tile(tile0, N, av, xv);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int row = tile0 + 16 * wv + 4 * s + q;
            float bv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                bsum[k] += av[s][k];
                bv[k] = act ? (row < N ? fmaxf(fmaf(xv[s][k], xs[k], xh[k]), 0.f) : 0.f) : xv[s][k];
            }
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) acc[ob][kb] = mfma_16x16x4_f32(av[s][ob], bv[kb], acc[ob][kb]);
        }
        if (more) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int k = 0; k < 4; ++k) { av[s][k] = an[s][k]; xv[s][k] = xn[s][k]; }
        }
    }
    // combine the 4 waves in a fixed order, then write this chunk's slab
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int o = 16 * ob + 4 * q + r, k = 16 * kb + j;
                        S[o * H + k] = (w == 0 ? 0.f : S[o * H + k]) + acc[ob][kb][r];
                    }
        }
        __syncthreads();
    }
    float *slab = a.slabs + ((int64_t)jid * kWgChunks + blockIdx.x) * H * H;
    for (int i = tid; i < H * H; i += kThreads) slab[i] = S[i];
    // bias gradient = column sums of dZ: over the 4 rows of a step (lanes q), then over the waves
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        bsum[k] += wave_shfl_xor(bsum[k], 16);
        bsum[k] += wave_shfl_xor(bsum[k], 32);
        if (q == 0) Sb[wv * H + 16 * k + j] = bsum[k];
    }
    __syncthreads();
    if (tid < H)
        a.bias_slabs[((int64_t)jid * kWgChunks + blockIdx.x) * H + tid] =
            (Sb[tid] + Sb[H + tid]) + (Sb[2 * H + tid] + Sb[3 * H + tid]);
}

// =========================================================================
// final: reduce slabs, prediction-layer / BatchNorm / bias / embedding gradients -> grads
struct FinalArgs {
    const float *slabs, *bias_slabs;
    const double *bst;        // [L][3 (a,b,c)][kRep][3][64]
    const float *demb_parts;  // [kEmbBlocks][emb_rows * emb_dim]
    gcc_gin_grads g;
    int32_t B, L, kdim0, emb_rows, emb_dim, accumulate, hid;
};

__device__ __forceinline__ void put(float *dst, float v, int acc)
{
    if (dst) *dst = acc ? *dst + v : v;
}

__global__ __launch_bounds__(kThreads) void gin_grad_final_kernel(FinalArgs a)
{
    TRAIN_STEP_WAVE_PRIORITY();
    const int64_t gid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int L = a.L;
    int64_t base = 0;
    // (1) weight gradients from the slabs: jobs [0, 2L) = linears.{0,1} of every layer, [2L, 3L+1) = linears_prediction
    const int njobs = 3 * L + 1;
    const int64_t n1 = (int64_t)njobs * H * H;
    if (gid < n1) {
        const int y = (int)(gid / (H * H)), idx = (int)(gid % (H * H));
        const int o = idx / H, k = idx % H;
        float s = 0.f;
        for (int c = 0; c < kWgChunks; ++c) s += a.slabs[((int64_t)y * kWgChunks + c) * H * H + idx];
        int kd = H;
        float *dst;
        if (y < 2 * L) {
            const int l = y >> 1, which = y & 1;
            kd = (which == 0 && l == 0) ? a.kdim0 : a.hid;
            dst = which ? a.g.lin1_w[l] : a.g.lin0_w[l];
        } else {
            const int i = y - 2 * L;
            kd = i == 0 ? a.kdim0 : a.hid;
            dst = a.g.pred_w[i];
        }
        if (k < kd && dst) put(dst + (int64_t)o * kd + k, s, a.accumulate);
        return;
    }
    base += n1;
    // (3) linears_prediction.i.bias = column sums of G_i
    const int64_t n3 = (int64_t)(L + 1) * H;
    if (gid < base + n3) {
        const int64_t r = gid - base;
        const int i = (int)(r / H), o = (int)(r % H);
        float s = 0.f;
        for (int c = 0; c < kWgChunks; ++c) s += a.bias_slabs[((int64_t)(2 * L + i) * kWgChunks + c) * H + o];
        if (a.g.pred_b[i]) put(a.g.pred_b[i] + o, s, a.accumulate);
        return;
    }
    base += n3;
    // (4) BatchNorm weight/bias and Linear bias gradients from the backward sums
    const int64_t n4 = (int64_t)L * 3 * 3 * H;
    if (gid < base + n4) {
        const int64_t r = gid - base;
        const int c = (int)(r % H), slot = (int)((r / H) % 3), which = (int)((r / (3 * H)) % 3), l = (int)(r / (9 * H));
        double acc = 0.0;
        const double *src = a.bst + ((int64_t)l * 3 + which) * kRep * 3 * H + slot * H + c;
#pragma unroll
        for (int rep = 0; rep < kRep; rep += 8) {          // batches of 8 independent loads (see replica_sums128)
            double v8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v8[u] = src[(int64_t)(rep + u) * 3 * H];
            acc += ((v8[0] + v8[1]) + (v8[2] + v8[3])) + ((v8[4] + v8[5]) + (v8[6] + v8[7]));
        }
        const float v = (float)acc;
        float *dst = nullptr;
        if (which == 0) dst = slot == 0 ? a.g.bn_a_b[l] : slot == 1 ? a.g.bn_a_w[l] : a.g.lin0_b[l];
        else if (which == 1) dst = slot == 0 ? a.g.bn_b_b[l] : slot == 1 ? a.g.bn_b_w[l] : a.g.lin1_b[l];
        else dst = slot == 0 ? a.g.bn_c_b[l] : slot == 1 ? a.g.bn_c_w[l] : nullptr;
        if (dst) put(dst + c, v, a.accumulate);
        return;
    }
    base += n4;
    // (5) degree embedding: 8 threads per element, each adding kEmbBlocks / 8 of the partial tables (8 loads in flight),
    // combined by a fixed shuffle tree.  The section starts on a wave boundary and is padded to whole waves (the shuffles
    // need every lane); one thread per element walked all 448 partials in 56 dependent rounds (17 of this kernel's 20 us).
    base = (base + 63) & ~(int64_t)63;
    const int64_t n5 = (int64_t)a.emb_rows * a.emb_dim;
    if (gid >= base) {                                   // wave-uniform
        static_assert(kEmbBlocks % 64 == 0, "8 threads x batches of 8 partials");
        const int64_t q = gid - base, r = q >> 3;
        const int part = (int)(q & 7);
        float s = 0.f;
        if (r < n5) {
            const float *src = a.demb_parts + (int64_t)part * (kEmbBlocks / 8) * n5 + r;
            for (int k = 0; k < kEmbBlocks / 8; k += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(k + u) * n5];
                s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
        }
        s += wave_shfl_xor(s, 1);
        s += wave_shfl_xor(s, 2);
        s += wave_shfl_xor(s, 4);
        if (part == 0 && r < n5 && a.g.degree_embedding) put(a.g.degree_embedding + r, s, a.accumulate);
    }
}

struct BwdWork {
    float *U, *V, *Wt, *D;
    float *dz1[GCC_GIN_MAX_LAYERS], *dz2[GCC_GIN_MAX_LAYERS];
    float *G, *dpooled, *slabs, *bias_slabs;
    double *bst;
    float *demb_parts;
    int64_t off_zero, zero_bytes, total;
};

inline BwdWork bwd_layout(char *base, int64_t node_cap, int32_t B, int32_t L, int32_t emb_elems)
{
    BwdWork w;
    auto al = [](int64_t x) { return (x + 255) & ~(int64_t)255; };
    int64_t o = 0;
    auto take = [&](int64_t bytes) { int64_t at = o; o = al(o + bytes); return base ? base + at : (char *)nullptr; };
    const int64_t act = node_cap * H * (int64_t)sizeof(float);
    w.U = (float *)take(act); w.V = (float *)take(act); w.Wt = (float *)take(act); w.D = (float *)take(act);
    for (int l = 0; l < GCC_GIN_MAX_LAYERS; ++l) {
        w.dz1[l] = l < L ? (float *)take(act) : nullptr;
        w.dz2[l] = l < L ? (float *)take(act) : nullptr;
    }
    w.G = (float *)take((int64_t)(L + 1) * B * H * sizeof(float));
    w.dpooled = (float *)take((int64_t)(L + 1) * B * H * sizeof(float));
    w.slabs = (float *)take((int64_t)(3 * L + 1) * kWgChunks * H * H * sizeof(float));
    w.bias_slabs = (float *)take((int64_t)(3 * L + 1) * kWgChunks * H * sizeof(float));
    w.demb_parts = (float *)take((int64_t)kEmbBlocks * emb_elems * sizeof(float));
    w.off_zero = o;                                   // everything from here on is zeroed per call
    w.bst = (double *)take((int64_t)L * 9 * kRep * H * sizeof(double));
    w.zero_bytes = o - w.off_zero;
    w.total = o;
    return w;
}

}  // namespace

extern "C" int64_t gcc_gin_backward_workspace_bytes(int64_t node_cap, int32_t batch_size, int32_t num_gin_layers)
{
    if (node_cap <= 0 || batch_size <= 0 || num_gin_layers < 1 || num_gin_layers > GCC_GIN_MAX_LAYERS) {
        snprintf(g_err, kErrLen, "gcc_gin_backward_workspace_bytes: bad argument");
        return -1;
    }
    return bwd_layout(nullptr, node_cap, batch_size, num_gin_layers, kEmbMaxElems).total;
}

extern "C" int32_t gcc_gin_backward(const gcc_gin_pass *pass, const float *dfeat, const gcc_gin_grads *grads,
                                    int32_t accumulate, void *workspace, int64_t workspace_bytes,
                                    int64_t node_cap, gcc_prof *prof, void *stream)
{
    if (!pass || !dfeat || !grads || !workspace) {
        snprintf(g_err, kErrLen, "gcc_gin_backward: null argument");
        return -1;
    }
    const gcc_gin_pass &p = *pass;
    if (p.edge_multiplicity > 1) {
        snprintf(g_err, kErrLen, "gcc_gin_backward: edge_multiplicity %d (inference-only feature)", p.edge_multiplicity);
        return -2;
    }
    const int L = p.w.num_gin_layers, B = p.batch_size;
    const int kdim0 = p.w.pos_dim + p.w.deg_emb_dim + 1;
    const int emb_elems = (p.w.max_degree + 1) * p.w.deg_emb_dim;
    if (!p.training || L < 1 || L > GCC_GIN_MAX_LAYERS || emb_elems > kEmbMaxElems) {
        snprintf(g_err, kErrLen, "gcc_gin_backward: needs a training-mode pass (layers=%d)", L);
        return -2;
    }
    for (int l = 0; l < L; ++l)
        if (!p.agg[l]) { snprintf(g_err, kErrLen, "gcc_gin_backward: agg[%d] was not saved", l); return -2; }
    const BwdWork w = bwd_layout((char *)workspace, node_cap, B, L, kEmbMaxElems);
    if (workspace_bytes < w.total) {
        snprintf(g_err, kErrLen, "gcc_gin_backward: workspace %lld < %lld bytes", (long long)workspace_bytes,
                 (long long)w.total);
        return -3;
    }
    if (node_cap < 1 || node_cap > 0x7fffffff || (p.node_cap > 0 && p.node_cap != node_cap)) {
        snprintf(g_err, kErrLen, "gcc_gin_backward: node_cap %lld (the pass says %lld)", (long long)node_cap, (long long)p.node_cap);
        return -2;
    }
    const int32_t cap = (int32_t)node_cap;           // the kernels clamp their speculative first-tile requests to it
    hipStream_t s = (hipStream_t)stream;
    prof_mark(prof, 0, s);
    const dim3 grid(tile_grid(p.rows_hint > 0 && p.rows_hint < node_cap ? p.rows_hint : node_cap)), block(kThreads);
    auto bst = [&](int l, int which) { return w.bst + ((int64_t)l * 3 + which) * kRep * 3 * H; };   // which: 0=a 1=b 2=c
    {
        ReadBwdArgs a;
        a.dfeat = dfeat; a.score = p.score; a.feat = p.feat; a.drop = drop_cfg(p);
        for (int i = 0; i <= L; ++i) a.pred_w[i] = p.w.pred_w[i];
        a.G = w.G; a.dpooled = w.dpooled; a.B = B; a.nlayers = L; a.kdim0 = kdim0; a.normalize = p.normalize; a.hid = hidden_of(p.w);
        a.norm_eps = p.w.norm_eps;
        // the per-call accumulators (BatchNorm-backward column sums) are cleared by extra workgroups of this launch
        a.zero = (float4 *)((char *)workspace + w.off_zero); a.zero16 = w.zero_bytes / 16; a.nread = ((B + 15) / 16 * (L + 1) + 3) / 4;
        hipLaunchKernelGGL(gin_bwd_readout_kernel, dim3(a.nread + 64), block, 0, s, a);
    }
    for (int l = L - 1; l >= 0; --l) {
        const BnDev bna = bn_of(p, p.w.bn_a[l], l, 0);
        const BnDev bnb = bn_of(p, p.w.bn_b[l], l, 1);
        const BnDev bnc = bn_of(p, p.w.bn_c[l], l, 2);
        {
            BwdCArgs a = {p.node_off, p.row_ptr, p.col_idx, p.graph_id, l == L - 1 ? nullptr : w.D,
                          w.dpooled + (int64_t)(l + 1) * B * H, p.z2[l], bnb, bnc, w.U, bst(l, 2), B, p.w.bn_eps, cap};
            hipLaunchKernelGGL(gin_bwd_c_kernel, grid, block, 0, s, a);
        }
        {
            BwdBArgs a = {p.node_off, w.U, p.z2[l], bnb, bnc, bst(l, 2), w.V, bst(l, 1), B, p.w.bn_eps, cap};
            hipLaunchKernelGGL(gin_bwd_b_kernel, grid, block, 0, s, a);
        }
        {
            BwdLinArgs a = {p.node_off, w.V, p.z2[l], bnb, bst(l, 1), bst(l, 1) + 2 * H, p.w.lin1_w[l], hidden_of(p.w),
                            w.dz2[l], w.Wt, p.z1[l], bna, bst(l, 0), B, p.w.bn_eps, cap};
            hipLaunchKernelGGL((gin_bwd_lin_kernel<true>), grid, block, 0, s, a);
        }
        {
            BwdLinArgs a = {p.node_off, w.Wt, p.z1[l], bna, bst(l, 0), bst(l, 0) + 2 * H, p.w.lin0_w[l],
                            l == 0 ? kdim0 : hidden_of(p.w), w.dz1[l], w.D, nullptr, BnDev(), nullptr, B, p.w.bn_eps, cap};
            hipLaunchKernelGGL((gin_bwd_lin_kernel<false>), grid, block, 0, s, a);
        }
    }
    {
        EmbArgs a = {p.node_off, p.row_ptr, p.col_idx, p.graph_id, w.D, w.dpooled, w.demb_parts, B, p.w.pos_dim,
                     p.w.deg_emb_dim, p.w.max_degree, cap};
        hipLaunchKernelGGL(gin_bwd_emb_kernel, dim3(kEmbBlocks), block, 0, s, a);
    }
    {
        // (one launch for all 3 L + 1 products: a launch per layer takes as long as this one -- every workgroup's chain of
        //  row tiles is the same -- so spreading them over a second stream bought nothing, profiles/r3_side_stream_probe.txt)
        WgradArgs a;
        a.node_off = p.node_off; a.slabs = w.slabs; a.bias_slabs = w.bias_slabs; a.B = B; a.eps = p.w.bn_eps; a.cap = cap;
        for (int l = 0; l < L; ++l) {
            a.job[2 * l + 0] = {w.dz1[l], p.agg[l], nullptr, BnDev(), 0};
            a.job[2 * l + 1] = {w.dz2[l], p.z1[l], nullptr, bn_of(p, p.w.bn_a[l], l, 0), 0};
        }
        for (int i = 0; i <= L; ++i)      // linears_prediction[i]: dW = G_i^T pooled_i over the B graphs
            a.job[2 * L + i] = {w.G + (int64_t)i * B * H, nullptr, p.pooled + (int64_t)i * B * H, BnDev(), B};
        hipLaunchKernelGGL(gin_wgrad_kernel, dim3(kWgChunks, 3 * L + 1), block, 0, s, a);
    }
    {
        FinalArgs a;
        a.slabs = w.slabs; a.bias_slabs = w.bias_slabs; a.bst = w.bst; a.demb_parts = w.demb_parts; a.g = *grads;
        a.B = B; a.L = L; a.kdim0 = kdim0; a.emb_rows = p.w.max_degree + 1; a.emb_dim = p.w.deg_emb_dim; a.hid = hidden_of(p.w);
        a.accumulate = accumulate;
        const int64_t upto4 = (int64_t)(3 * L + 1) * H * H + (int64_t)(L + 1) * H + (int64_t)L * 9 * H;
        const int64_t total = ((upto4 + 63) & ~(int64_t)63) + (((int64_t)a.emb_rows * a.emb_dim * 8 + 63) & ~(int64_t)63);
        hipLaunchKernelGGL(gin_grad_final_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)), block, 0, s, a);
    }
    prof_mark(prof, 1, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, kErrLen, "gcc_gin_backward: launch failed: %s", hipGetErrorString(e));
        return -10;
    }
    return 0;
}
