// gcc_amd/csrc/ginx.hip -- the GIN encoder at ANY hidden / output width, training mode included: forward with batch
// statistics and the full backward, fp32 on the matrix cores (v_mfma_f32_16x16x4_f32).  `--hidden-size` is part of the
// reference's flag surface (train.py:93; GraphEncoder node_hidden_dim / output_dim, graph_encoder.py:44-63); the fused
// 64-channel kernels of encoder.hip / encoder_bwd.hip serve widths up to 64 (zero-padded, exact), this file serves the
// rest.  Reference arithmetic, statement by statement:
//     features       graph_encoder.py:152-165   [pos_undirected | degree_embedding(clamp(in_deg)) | seed flag]
//     GINConv        gin.py:179-185,218         agg = h + sum over the in-neighbours (sum aggregation, eps = 0)
//     MLP            gin.py:107-116             linears.1(relu(batch_norms.0(linears.0(agg))))
//     ApplyNodeFunc  gin.py:54-58               relu(bn(mlp(.)))
//     outer BN/ReLU  gin.py:219-220
//     readout        gin.py:223-232             sum_i dropout(linears_prediction[i](SumPooling(hidden_rep[i])))
//     normalise      graph_encoder.py:195-196   F.normalize(p = 2, eps = 1e-5)
// and autograd's backward of exactly this graph (BatchNorm in training mode: biased batch variance for the
// normalisation, unbiased for running_var, momentum 0.1).
//
// Shape of the implementation: NOT fused.  One launch per operator -- a CSR gather, a strided MFMA GEMM used for every
// Linear (forward, data gradient, weight gradient with split-K atomics), column statistics in fp64, BatchNorm + ReLU
// forward / backward passes, per-graph pooling -- ~70 launches forward, ~150 backward at 4 layers.  Every activation is
// kept for the backward pass (6 [N, W] tensors per layer: HBM is 288 GB).  The batched subgraph of a symmetric parent is
// symmetric, so the gather's own transpose is the same gather (the sampler's contract; the 64-channel backward relies on
// it too).  Row counts live on the device (node_off[B]): every kernel is launched for node_cap rows and reads the live
// extent itself, no host synchronisation.
#include "host_common.h"

#include <stdlib.h>

namespace {

constexpr int kGT = 256;                 // threads of the GEMM workgroup: 4 waves, a 64 x 64 tile of C
constexpr int kBM = 64, kBN = 64, kBK = 16;
constexpr int kLdT = kBM + 4;            // LDS row stride of the k-major operand tiles
constexpr int kSplitRows = 1024;         // rows of the reduction dimension per workgroup when it is the node dimension (256 until round 6: 100 slabs of a
                                         // 25k-node batch queued on every fp64 accumulator, 2.3 ms per weight gradient at width 256): f32 on the matrix cores
                                         // inside a slab, fp64 atomics across slabs (a weight gradient sums over ~10^4 nodes whose terms largely cancel)

// C[m][n] (+)= sum_k A(m, k) B(k, n) [+ bias[n]];  A(m, k) = A[m * sam + k * sak], B(k, n) = B[k * sbk + n * sbn].
// rows_dim: 0 = all extents are the host's; 1 = M is *rows (device); 2 = K is *rows (device; split over gridDim.z with
// atomic adds into a zeroed fp64 accumulator); 3 = as 2 with the host's K (a long reduction over few output tiles: the head's
// d loss / d q = dlog [B, K] x queue [K, D] at K = 16384 was 16 workgroups x 1024 steps, 2.3 ms of an 11 ms step).
struct GemmArgs {
    const float *A, *B, *bias;
    float *C;
    int64_t sam, sak, sbk, sbn, ldc;
    int32_t M, N, K;
    const int32_t *rows;
    int32_t rows_dim, atomic;
    float alpha;
    double *Cd;                          // rows_dim >= 2: the fp64 accumulator the slabs add into (converted to C afterwards)
    int32_t vec;                         // bit 0 / 1: A / B may be read 16 bytes at a time along its contiguous index (strides and base aligned)
};

__global__ __launch_bounds__(kGT) void ginx_gemm_kernel(GemmArgs g)
{
    __shared__ float As[kBK][kLdT], Bs[kBK][kLdT];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int M = g.M, K = g.K;
    if (g.rows_dim == 1) M = *g.rows;
    if (g.rows_dim == 2) K = *g.rows;
    const int m0 = (int)blockIdx.x * kBM, n0 = (int)blockIdx.y * kBN;
    int k_lo = 0, k_hi = K;
    if (g.rows_dim >= 2) { k_lo = (int)blockIdx.z * kSplitRows; k_hi = min(K, k_lo + kSplitRows); }
    if (m0 >= M || k_lo >= k_hi) return;                     // (block-uniform)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[2][2] = {{zero4, zero4}, {zero4, zero4}};
    const int wr = (wv >> 1) * 32, wc = (wv & 1) * 32;       // this wave's 32 x 32 quarter of the tile
    const bool a_kc = g.sak == 1, b_nc = g.sbn == 1;         // which index of an operand is contiguous in memory
    // four elements of each operand tile per thread, along the contiguous index; the NEXT k-tile is requested before this one's
    // products (round 6: load -> barrier -> products -> barrier exposed a memory round trip per 16 columns of k), and where the
    // four elements are 16 contiguous, aligned bytes inside the operand they travel as one load (g.vec bit 0: A, bit 1: B)
    float ra[4], rb[4];
    auto fetch = [&](int k0) {
        if (a_kc) {
            const int mm = m0 + (tid >> 2), kk = k0 + (tid & 3) * 4;
            const float *p = g.A + (int64_t)mm * g.sam + kk;
            if ((g.vec & 1) && mm < M && kk + 3 < k_hi) {
                const float4 v = *(const float4 *)p;
                ra[0] = v.x; ra[1] = v.y; ra[2] = v.z; ra[3] = v.w;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) ra[u] = (mm < M && kk + u < k_hi) ? p[u] : 0.f;
            }
        } else {
            const int kk = k0 + (tid >> 4), mm = m0 + (tid & 15) * 4;
            const float *p = g.A + (int64_t)mm * g.sam + (int64_t)kk * g.sak;
            if ((g.vec & 1) && g.sam == 1 && mm + 3 < M && kk < k_hi) {
                const float4 v = *(const float4 *)p;
                ra[0] = v.x; ra[1] = v.y; ra[2] = v.z; ra[3] = v.w;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) ra[u] = (mm + u < M && kk < k_hi) ? p[(int64_t)u * g.sam] : 0.f;
            }
        }
        if (b_nc) {
            const int kk = k0 + (tid >> 4), nn = n0 + (tid & 15) * 4;
            const float *p = g.B + (int64_t)kk * g.sbk + nn;
            if ((g.vec & 2) && nn + 3 < g.N && kk < k_hi) {
                const float4 v = *(const float4 *)p;
                rb[0] = v.x; rb[1] = v.y; rb[2] = v.z; rb[3] = v.w;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) rb[u] = (nn + u < g.N && kk < k_hi) ? p[u] : 0.f;
            }
        } else {
            const int nn = n0 + (tid >> 2), kk = k0 + (tid & 3) * 4;
            const float *p = g.B + (int64_t)nn * g.sbn + (int64_t)kk * g.sbk;
            if ((g.vec & 2) && g.sbk == 1 && nn < g.N && kk + 3 < k_hi) {
                const float4 v = *(const float4 *)p;
                rb[0] = v.x; rb[1] = v.y; rb[2] = v.z; rb[3] = v.w;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) rb[u] = (nn < g.N && kk + u < k_hi) ? p[(int64_t)u * g.sbk] : 0.f;
            }
        }
    };
    auto store = [&]() {                                     // operand tiles -> LDS, k-major
        if (a_kc) {
            const int m = tid >> 2, kq = (tid & 3) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) As[kq + u][m] = ra[u];
        } else {
            const int kk_ = tid >> 4, m4 = (tid & 15) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) As[kk_][m4 + u] = ra[u];
        }
        if (b_nc) {
            const int kk_ = tid >> 4, n4 = (tid & 15) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) Bs[kk_][n4 + u] = rb[u];
        } else {
            const int n = tid >> 2, kq = (tid & 3) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) Bs[kq + u][n] = rb[u];
        }
    };
    fetch(k_lo);
    for (int k0 = k_lo; k0 < k_hi; k0 += kBK) {
        store();
        __syncthreads();
        if (k0 + kBK < k_hi) fetch(k0 + kBK);
#pragma unroll
        for (int ks = 0; ks < kBK; ks += 4) {
            const int kk = ks + (lane >> 4);
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[kk][wr + 16 * i + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[kk][wc + 16 * j + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma_16x16x4_f32(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr + 16 * i + 4 * (lane >> 4) + r, n = n0 + wc + 16 * j + (lane & 15);
                if (m < M && n < g.N) {
                    float v = acc[i][j][r] * g.alpha;
                    if (g.bias && (g.rows_dim < 2 || blockIdx.z == 0)) v += g.bias[n];
                    if (g.atomic) atomicAdd(g.Cd + (int64_t)m * g.ldc + n, (double)v); else g.C[(int64_t)m * g.ldc + n] = v;
                }
            }
}

// x0[v] = [pos_undirected[v] | degree_embedding[clamp(in_deg(v), 0, max_degree)] | seed flag]   (graph_encoder.py:152-165)
__global__ void ginx_feat_kernel(const int32_t *node_off, const int32_t *row_ptr, const int32_t *graph_id, const int32_t *seed_local,
                                 const float *pos, const float *emb, int B, int pos_dim, int de, int max_degree, int mult, float *x0)
{
    const int d_in = pos_dim + de + 1;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int N = node_off[B];
    const int v = (int)(i / d_in), c = (int)(i % d_in);
    if (v >= N) return;
    float val;
    if (c < pos_dim) val = pos[(int64_t)v * pos_dim + c];
    else if (c < pos_dim + de) {
        int d = (row_ptr[v + 1] - row_ptr[v]) * mult;
        d = d < 0 ? 0 : (d > max_degree ? max_degree : d);
        val = emb[(int64_t)d * de + (c - pos_dim)];
    } else {
        const int b = graph_id[v];
        val = (v == node_off[b] + (seed_local ? seed_local[b] : 0)) ? 1.f : 0.f;
    }
    x0[(int64_t)v * d_in + c] = val;
}

// d degree_embedding[clamp(deg(v))][c] += dx0[v][pos_dim + c]   (the gradient of nn.Embedding: a scatter-add).  Most nodes share a
// handful of small degrees: one global atomic per (node, column) queued 25 k nodes on a few cache lines (382 us at bsz 256).  A
// workgroup now adds its kFeatRows rows into an LDS copy of the table first and flushes the entries it touched.
constexpr int kFeatRows = 1024, kFeatMaxElems = 10240;     // rows per workgroup; (max_degree + 1) * de floats of LDS (40 KiB)
__global__ __launch_bounds__(256) void ginx_feat_bwd_kernel(const int32_t *node_off, const int32_t *row_ptr, int B, int pos_dim, int de, int max_degree,
                                                             const float *dx0, float *demb)
{
    __shared__ float E[kFeatMaxElems];
    const int d_in = pos_dim + de + 1;
    const int N = node_off[B];
    const int r0 = (int)blockIdx.x * kFeatRows, r1 = min(N, r0 + kFeatRows);
    if (r0 >= N) return;
    const int elems = (max_degree + 1) * de;
    const bool lds = elems <= kFeatMaxElems;                 // (block-uniform; larger tables: straight to global memory)
    if (lds) {
        for (int i = (int)threadIdx.x; i < elems; i += 256) E[i] = 0.f;
        __syncthreads();
    }
    for (int64_t i = (int64_t)r0 * de + threadIdx.x; i < (int64_t)r1 * de; i += 256) {
        const int v = (int)(i / de), c = (int)(i % de);
        int d = row_ptr[v + 1] - row_ptr[v];
        d = d < 0 ? 0 : (d > max_degree ? max_degree : d);
        const float g = dx0[(int64_t)v * d_in + pos_dim + c];
        if (lds) atomicAdd(&E[d * de + c], g); else atomicAdd(&demb[(int64_t)d * de + c], g);
    }
    if (lds) {
        __syncthreads();
        for (int i = (int)threadIdx.x; i < elems; i += 256)
            if (E[i] != 0.f) atomicAdd(&demb[i], E[i]);
    }
}

// out[v] = x[v] + mult * sum over row v of x[col]  (+ add[v] when add != NULL): one wave per row.  Rows whose width is a multiple of
// four are walked 16 bytes per lane with eight neighbour rows requested per round (round 6: one float per lane and one neighbour
// at a time -- a dependent col_idx -> row round trip per edge -- was 660 us per call at width 256, a third of the step); the
// neighbours are added in CSR order either way, so both paths give the same sums.
__global__ __launch_bounds__(256) void ginx_spmm_kernel(const int32_t *node_off, const int32_t *row_ptr, const int32_t *col_idx, int B,
                                                         const float *x, int D, const float *add, float *out, float mult)
{
    const int lane = (int)threadIdx.x & 63;
    const int v = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    if (v >= node_off[B]) return;
    const int e0 = row_ptr[v], e1 = row_ptr[v + 1];
    if ((D & 3) == 0) {
        constexpr int kJ = 8;
        for (int c = 4 * lane; c < D; c += 256) {
            float4 nb = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int e = e0; e < e1; e += kJ) {
                int u[kJ];
                float4 f[kJ];
#pragma unroll
                for (int j = 0; j < kJ; ++j) u[j] = col_idx[min(e + j, e1 - 1)];
#pragma unroll
                for (int j = 0; j < kJ; ++j) f[j] = *(const float4 *)(x + (int64_t)u[j] * D + c);
#pragma unroll
                for (int j = 0; j < kJ; ++j)
                    if (e + j < e1) { nb.x += f[j].x; nb.y += f[j].y; nb.z += f[j].z; nb.w += f[j].w; }
            }
            const float4 s = *(const float4 *)(x + (int64_t)v * D + c);
            float4 acc = make_float4(s.x + mult * nb.x, s.y + mult * nb.y, s.z + mult * nb.z, s.w + mult * nb.w);
            if (add) {
                const float4 a = *(const float4 *)(add + (int64_t)v * D + c);
                acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
            }
            *(float4 *)(out + (int64_t)v * D + c) = acc;
        }
        return;
    }
    for (int c = lane; c < D; c += 64) {                                 // (widths that are not a multiple of four: the 49 input features)
        constexpr int kJ = 8;
        float nb = 0.f;
        for (int e = e0; e < e1; e += kJ) {
            int u[kJ];
            float f[kJ];
#pragma unroll
            for (int j = 0; j < kJ; ++j) u[j] = col_idx[min(e + j, e1 - 1)];
#pragma unroll
            for (int j = 0; j < kJ; ++j) f[j] = x[(int64_t)u[j] * D + c];
#pragma unroll
            for (int j = 0; j < kJ; ++j)
                if (e + j < e1) nb += f[j];
        }
        float acc = x[(int64_t)v * D + c] + mult * nb;                   // (every edge counts `mult` times: gcc_gin_pass.edge_multiplicity)
        if (add) acc += add[(int64_t)v * D + c];
        out[(int64_t)v * D + c] = acc;
    }
}

// column sums over the live rows, fp64: sums[0][c] += sum_v f(v, c), sums[1][c] += sum_v g(v, c)
//   mode 0 (BatchNorm forward statistics):  f = x, g = x^2
//   mode 1 (BatchNorm + ReLU backward):     gr = dy * (y > 0);  f = gr,  g = gr * xhat,  xhat = (x - mean) * rstd
//   mode 2 (bias gradient):                 f = x              (sums[1] untouched)
__global__ __launch_bounds__(256) void ginx_colsum_kernel(const int32_t *node_off, int B, int fixed_rows, int mode, const float *x,
                                                           const float *y, const float *dy, const float *mr /* [2][D] mean, rstd */,
                                                           int D, double *sums)
{
    const int N = fixed_rows > 0 ? fixed_rows : node_off[B];
    const int r0 = (int)blockIdx.x * 128, r1 = min(N, r0 + 128);
    if (r0 >= N) return;
    for (int c = (int)threadIdx.x; c < D; c += 256) {
        double s0 = 0.0, s1 = 0.0;
        // (eight rows requested per round: a row at a time was a chain of 128 dependent-latency loads, 60 us per call)
        if (mode == 0) {
            for (int r = r0; r < r1; r += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = x[(int64_t)min(r + u, r1 - 1) * D + c];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (r + u < r1) { const double d = (double)v[u]; s0 += d; s1 += d * d; }
            }
        } else if (mode == 1) {
            const float mean = mr[c], rstd = mr[D + c];
            for (int r = r0; r < r1; r += 8) {
                float xv[8], yv[8], dv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t i = (int64_t)min(r + u, r1 - 1) * D + c;
                    xv[u] = x[i]; yv[u] = y[i]; dv[u] = dy[i];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (r + u < r1) {
                        const float gr = yv[u] > 0.f ? dv[u] : 0.f;
                        s0 += (double)gr;
                        s1 += (double)gr * (double)((xv[u] - mean) * rstd);
                    }
            }
        } else {
            for (int r = r0; r < r1; r += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = x[(int64_t)min(r + u, r1 - 1) * D + c];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (r + u < r1) s0 += (double)v[u];
            }
        }
        atomicAdd(&sums[c], s0);
        if (mode != 2) atomicAdd(&sums[D + c], s1);
    }
}

// per channel: batch mean / 1 / sqrt(biased variance + eps) from the sums (training) or from the running statistics
// (eval); training also moves the running statistics (momentum; unbiased variance) and counts the batch
__global__ void ginx_bn_prepare_kernel(const int32_t *node_off, int B, const double *sums, int D, float eps, float momentum, int training,
                                       int update, float *running_mean, float *running_var, int64_t *tracked, float *mr)
{
    const int c = (int)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= D) return;
    const double n = (double)node_off[B];
    if (training) {
        const double mean = n > 0 ? sums[c] / n : 0.0;
        double var = n > 0 ? sums[D + c] / n - mean * mean : 0.0;
        var = var < 0.0 ? 0.0 : var;
        mr[c] = (float)mean;
        mr[D + c] = (float)(1.0 / sqrt(var + (double)eps));
        if (update) {
            const double unb = n > 1 ? var * n / (n - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
            if (c == 0 && tracked) *tracked += 1;
        }
    } else {
        mr[c] = running_mean[c];
        mr[D + c] = 1.0f / sqrtf(running_var[c] + eps);
    }
}

// y = relu((x - mean) * rstd * gamma + beta).  Launched with one thread per FOUR elements when D is a multiple of four (16-byte
// accesses; `vec` = 1), else one per element
__global__ void ginx_bn_relu_kernel(const int32_t *node_off, int B, const float *x, const float *mr, const float *gamma, const float *beta,
                                    int D, float *y, int vec)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)node_off[B] * D;
    if (vec) {
        const int64_t i = 4 * t;
        if (i >= total) return;
        const int c = (int)(i % D);
        const float4 xv = *(const float4 *)(x + i), m = *(const float4 *)(mr + c), r = *(const float4 *)(mr + D + c);
        const float4 g = *(const float4 *)(gamma + c), b = *(const float4 *)(beta + c);
        float4 o;
        o.x = fmaxf((xv.x - m.x) * r.x * g.x + b.x, 0.f); o.y = fmaxf((xv.y - m.y) * r.y * g.y + b.y, 0.f);
        o.z = fmaxf((xv.z - m.z) * r.z * g.z + b.z, 0.f); o.w = fmaxf((xv.w - m.w) * r.w * g.w + b.w, 0.f);
        *(float4 *)(y + i) = o;
        return;
    }
    if (t >= total) return;
    const int c = (int)(t % D);
    const float v = (x[t] - mr[c]) * mr[D + c] * gamma[c] + beta[c];
    y[t] = v > 0.f ? v : 0.f;
}

// dx = gamma * rstd * (gr - mean(gr) - xhat * mean(gr * xhat)),  gr = dy * (y > 0)   (BatchNorm1d backward in training mode)
__device__ __forceinline__ float ginx_bn_bwd_one(float x, float y, float dy, float mean, float rstd, float gamma, float m0, float m1)
{
    const float gr = y > 0.f ? dy : 0.f;
    const float xhat = (x - mean) * rstd;
    return gamma * rstd * (gr - m0 - xhat * m1);
}
__global__ void ginx_bn_relu_bwd_kernel(const int32_t *node_off, int B, const float *x, const float *y, const float *dy, const float *mr,
                                        const float *gamma, const double *sums, int D, float *dx, int vec)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int N = node_off[B];
    const int64_t total = (int64_t)N * D;
    if (vec) {
        const int64_t i = 4 * t;
        if (i >= total) return;
        const int c = (int)(i % D);
        const float4 xv = *(const float4 *)(x + i), yv = *(const float4 *)(y + i), dv = *(const float4 *)(dy + i);
        const float4 m = *(const float4 *)(mr + c), r = *(const float4 *)(mr + D + c), g = *(const float4 *)(gamma + c);
        const double n = (double)N;
        float4 o;
        o.x = ginx_bn_bwd_one(xv.x, yv.x, dv.x, m.x, r.x, g.x, (float)(sums[c + 0] / n), (float)(sums[D + c + 0] / n));
        o.y = ginx_bn_bwd_one(xv.y, yv.y, dv.y, m.y, r.y, g.y, (float)(sums[c + 1] / n), (float)(sums[D + c + 1] / n));
        o.z = ginx_bn_bwd_one(xv.z, yv.z, dv.z, m.z, r.z, g.z, (float)(sums[c + 2] / n), (float)(sums[D + c + 2] / n));
        o.w = ginx_bn_bwd_one(xv.w, yv.w, dv.w, m.w, r.w, g.w, (float)(sums[c + 3] / n), (float)(sums[D + c + 3] / n));
        *(float4 *)(dx + i) = o;
        return;
    }
    if (t >= total) return;
    const int c = (int)(t % D);
    dx[t] = ginx_bn_bwd_one(x[t], y[t], dy[t], mr[c], mr[D + c], gamma[c], (float)(sums[c] / (double)N), (float)(sums[D + c] / (double)N));
}

// fp64 sums -> fp32 parameter gradients: dst[c] (+)= (float)src[c]   (also the weight gradients' fp64 accumulators, n = rows * cols)
__global__ void ginx_sums_to_grad_kernel(const double *src, int D, float *dst, int accumulate)
{
    const int c = (int)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < D) dst[c] = (accumulate ? dst[c] : 0.f) + (float)src[c];
}

// pooled[b][c] = sum of h over the nodes of graph b (SumPooling, gin.py:205,228): one workgroup per graph.  Widths that are a multiple
// of four: 64 lanes x 16 bytes cover 256 channels, the four waves take every fourth row (eight rows each in flight) and their fp64
// partial sums are added in wave order; otherwise a thread per channel walks the rows eight at a time.
__global__ __launch_bounds__(256) void ginx_pool_kernel(const int32_t *node_off, const float *h, int D, float *pooled)
{
    __shared__ double part[4][4 * 64];
    const int b = (int)blockIdx.x;
    const int r0 = node_off[b], r1 = node_off[b + 1];
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6;
    if ((D & 3) == 0) {
        for (int c0 = 0; c0 < D; c0 += 256) {                             // (block-uniform trip count)
            const int c = c0 + 4 * lane;
            double s[4] = {0.0, 0.0, 0.0, 0.0};
            if (c < D) {
                for (int r = r0 + wv; r < r1; r += 32) {                      // this wave's rows r0 + wv, + 4, ...: eight per round
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = *(const float4 *)(h + (int64_t)min(r + 4 * u, r1 - 1) * D + c);
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (r + 4 * u < r1) { s[0] += (double)v[u].x; s[1] += (double)v[u].y; s[2] += (double)v[u].z; s[3] += (double)v[u].w; }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) part[wv][4 * lane + e] = s[e];
            __syncthreads();
            if (wv == 0 && c < D) {
                float4 o;
                o.x = (float)((part[0][4 * lane + 0] + part[1][4 * lane + 0]) + (part[2][4 * lane + 0] + part[3][4 * lane + 0]));
                o.y = (float)((part[0][4 * lane + 1] + part[1][4 * lane + 1]) + (part[2][4 * lane + 1] + part[3][4 * lane + 1]));
                o.z = (float)((part[0][4 * lane + 2] + part[1][4 * lane + 2]) + (part[2][4 * lane + 2] + part[3][4 * lane + 2]));
                o.w = (float)((part[0][4 * lane + 3] + part[1][4 * lane + 3]) + (part[2][4 * lane + 3] + part[3][4 * lane + 3]));
                *(float4 *)(pooled + (int64_t)b * D + c) = o;
            }
            __syncthreads();
        }
        return;
    }
    for (int c = (int)threadIdx.x; c < D; c += 256) {
        double s = 0.0;
        for (int r = r0; r < r1; r += 8) {                                // (eight rows requested per round, added in row order)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = h[(int64_t)min(r + u, r1 - 1) * D + c];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (r + u < r1) s += (double)v[u];
        }
        pooled[(int64_t)b * D + c] = (float)s;
    }
}

// dh[v][c] (+)= dpooled[graph_id[v]][c]   (`vec`: one thread per four elements, D a multiple of four)
__global__ void ginx_pool_bwd_kernel(const int32_t *node_off, const int32_t *graph_id, int B, const float *dpooled, int D, float *dh, int accumulate, int vec)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)node_off[B] * D;
    if (vec) {
        const int64_t i = 4 * t;
        if (i >= total) return;
        const int v = (int)(i / D), c = (int)(i % D);
        float4 g = *(const float4 *)(dpooled + (int64_t)graph_id[v] * D + c);
        if (accumulate) { const float4 o = *(const float4 *)(dh + i); g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w; }
        *(float4 *)(dh + i) = g;
        return;
    }
    if (t >= total) return;
    const int v = (int)(t / D), c = (int)(t % D);
    const float g = dpooled[(int64_t)graph_id[v] * D + c];
    dh[t] = accumulate ? dh[t] + g : g;
}

// score (+)= y * keep * scale   (dropout with an explicit 0/1 keep mask, scale = 1 / (1 - p); keep == NULL: no dropout);
// backward: dy = dscore * keep * scale -- the same kernel with `first` = 1
__global__ void ginx_mask_acc_kernel(const float *y, const float *keep, float scale, int n, float *score, int first)
{
    const int i = (int)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = keep ? y[i] * keep[i] * scale : y[i];
    score[i] = first ? v : score[i] + v;
}

// feat = score / max(||score||_2, eps)   (one wave per row)
__global__ __launch_bounds__(64) void ginx_normalize_kernel(const float *score, int D, float eps, int normalize, float *feat)
{
    const int b = (int)blockIdx.x, lane = (int)threadIdx.x;
    float ss = 0.f;
    for (int c = lane; c < D; c += 64) { const float v = score[(int64_t)b * D + c]; ss += v * v; }
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss);
    const float inv = normalize ? 1.0f / fmaxf(nrm, eps) : 1.0f;
    for (int c = lane; c < D; c += 64) feat[(int64_t)b * D + c] = score[(int64_t)b * D + c] * inv;
}

// dscore = (dfeat - feat * <feat, dfeat>) / ||score||  when ||score|| >= eps, dfeat / eps below it; dfeat without normalisation
__global__ __launch_bounds__(64) void ginx_normalize_bwd_kernel(const float *score, const float *feat, const float *dfeat, int D, float eps,
                                                                 int normalize, float *dscore)
{
    const int b = (int)blockIdx.x, lane = (int)threadIdx.x;
    float ss = 0.f, dot = 0.f;
    for (int c = lane; c < D; c += 64) {
        const int64_t i = (int64_t)b * D + c;
        ss += score[i] * score[i];
        dot += feat[i] * dfeat[i];
    }
    ss = wave_sum(ss);
    dot = wave_sum(dot);
    const float nrm = sqrtf(ss);
    for (int c = lane; c < D; c += 64) {
        const int64_t i = (int64_t)b * D + c;
        float v = dfeat[i];
        if (normalize) v = nrm >= eps ? (dfeat[i] - feat[i] * dot) / nrm : dfeat[i] / eps;
        dscore[i] = v;
    }
}

// ---- MoCo / InfoNCE head at any feature size (memory_moco.py:26-63, criterions.py:5-33), dense: the logits are materialised
// out[b][0] = <q_b, k_b> / T   (memory_moco.py:33-34; the negatives come from the GEMM)
__global__ __launch_bounds__(64) void ginx_rowdot_kernel(const float *q, const float *k, int D, float inv_T, float *out, int64_t ldo)
{
    const int b = (int)blockIdx.x, lane = (int)threadIdx.x;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += q[(int64_t)b * D + c] * k[(int64_t)b * D + c];
    s = wave_sum(s);
    if (lane == 0) out[(int64_t)b * ldo] = s * inv_T;
}

// per row: lse, CE against column 0 (mode 0) or the row's own index (mode 1: criterions.py:27-33), d loss / d logits * B
// = softmax - onehot into dlog; acc[0] += CE, acc[1] += the label's logit (train.py:394,401 "prob")
__global__ __launch_bounds__(256) void ginx_ce_kernel(const float *out, int64_t ldo, int ncols, int mode, float *dlog, double *acc)
{
    __shared__ float red[4];
    const int b = (int)blockIdx.x, tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float *x = out + (int64_t)b * ldo;
    float m = -3.0e38f;
    for (int j = tid; j < ncols; j += 256) m = fmaxf(m, x[j]);
    m = wave_max(m);
    if (lane == 0) red[wv] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int j = tid; j < ncols; j += 256) s += expf(x[j] - m);
    s = wave_sum(s);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    const float lse = m + logf(s);
    const int label = mode == 0 ? 0 : b;
    for (int j = tid; j < ncols; j += 256) dlog[(int64_t)b * ldo + j] = expf(x[j] - lse) - (j == label ? 1.f : 0.f);
    if (tid == 0) {
        atomicAdd(&acc[0], (double)(lse - x[label]));
        atomicAdd(&acc[1], (double)x[label]);
    }
}

__global__ void ginx_ce_final_kernel(const double *acc, int B, float *loss, float *prob)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { *loss = (float)(acc[0] / B); *prob = (float)(acc[1] / B); }
}

// g[b][c] += coef * dlog[b * ldd] * k[b][c]   (the positive logit's share of d loss / d q)
__global__ void ginx_rank1_rows_kernel(const float *dlog, int64_t ldd, const float *k, int B, int D, float coef, float *g)
{
    const int i = (int)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const int b = i / D;
    g[i] += coef * dlog[(int64_t)b * ldd] * k[i];
}

// memory.index_copy_(0, (arange(n) + index) % K, keys)   (memory_moco.py:55-61)
__global__ void ginx_enqueue_kernel(float *mem, int K, int D, const float *keys, int n, int index)
{
    const int i = (int)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * D) return;
    const int r = i / D, c = i % D;
    mem[(int64_t)((index + r) % K) * D + c] = keys[i];
}

// ---------------------------------------------------------------- host side ----
struct XLayout {                         // float offsets inside the pass's workspace
    int64_t x0, agg[GCC_GIN_MAX_LAYERS], z1[GCC_GIN_MAX_LAYERS], a1[GCC_GIN_MAX_LAYERS], z2[GCC_GIN_MAX_LAYERS], a2[GCC_GIN_MAX_LAYERS],
        h[GCC_GIN_MAX_LAYERS];
    int64_t mr[GCC_GIN_MAX_LAYERS][3];   // [2][W] mean / rstd of the three BatchNorms of a layer
    int64_t pooled[GCC_GIN_MAX_LAYERS + 1], y, score;
    int64_t da, db, dc, dpool, dy, dscore;      // backward scratch: three [N, Wmax] buffers, [B, Wmax] x 3
    int64_t sums;                        // doubles: [2][Wmax] scratch of the statistics kernels (offset in FLOATS, 8-byte aligned)
    int64_t wg64;                        // doubles: [Wmax][Wmax] accumulator of a weight gradient
    int64_t total;
};

XLayout ginx_layout(int64_t N, int B, int L, int d_in, int W, int O)
{
    XLayout x;
    const int64_t Wm = W > d_in ? (W > O ? W : O) : (d_in > O ? d_in : O);
    int64_t o = 0;
    auto take = [&](int64_t n) { const int64_t at = o; o += (n + 63) / 64 * 64; return at; };
    x.x0 = take(N * d_in);
    for (int l = 0; l < L; ++l) {
        x.agg[l] = take(N * (l == 0 ? d_in : W));
        x.z1[l] = take(N * W); x.a1[l] = take(N * W); x.z2[l] = take(N * W); x.a2[l] = take(N * W); x.h[l] = take(N * W);
        for (int j = 0; j < 3; ++j) x.mr[l][j] = take(2 * W);
    }
    for (int l = 0; l <= L; ++l) x.pooled[l] = take((int64_t)B * (l == 0 ? d_in : W));
    x.y = take((int64_t)B * O);
    x.score = take((int64_t)B * O);
    x.da = take(N * Wm); x.db = take(N * Wm); x.dc = take(N * Wm);
    x.dpool = take((int64_t)B * Wm); x.dy = take((int64_t)B * O); x.dscore = take((int64_t)B * O);
    x.sums = take(4 * Wm + 16);
    x.wg64 = take(2 * Wm * Wm + 16);
    x.total = o;
    return x;
}

void gemm(hipStream_t s, const float *A, int64_t sam, int64_t sak, const float *B, int64_t sbk, int64_t sbn, float *C, int64_t ldc,
          int M, int N, int K, const float *bias, const int32_t *rows, int rows_dim, int64_t rows_cap, float alpha = 1.0f, double *acc64 = nullptr)
{
    // 16-byte loads along the contiguous index need the base and the OTHER index's stride to keep that alignment
    const auto al16 = [](const void *p_) { return (((uintptr_t)p_) & 15) == 0; };
    const int vec_a = (sak == 1 ? (sam % 4 == 0) : (sam == 1 && sak % 4 == 0)) && al16(A);
    const int vec_b = (sbn == 1 ? (sbk % 4 == 0) : (sbk == 1 && sbn % 4 == 0)) && al16(B);
    GemmArgs g = {A, B, bias, C, sam, sak, sbk, sbn, ldc, M, N, K, rows, rows_dim, rows_dim >= 2 ? 1 : 0, alpha, acc64, vec_a | (vec_b << 1)};
    const int mcap = rows_dim == 1 ? (int)rows_cap : M;
    const int64_t kcap = rows_dim == 2 ? rows_cap : K;
    dim3 grid((mcap + kBM - 1) / kBM, (N + kBN - 1) / kBN, rows_dim >= 2 ? (unsigned)((kcap + kSplitRows - 1) / kSplitRows) : 1u);
    if (rows_dim >= 2) (void)hipMemsetAsync(acc64, 0, sizeof(double) * (size_t)M * (size_t)ldc, s);     // (C is dense: ldc == N)
    hipLaunchKernelGGL(ginx_gemm_kernel, grid, dim3(kGT), 0, s, g);
    if (rows_dim >= 2) hipLaunchKernelGGL(ginx_sums_to_grad_kernel, dim3((unsigned)(((int64_t)M * ldc + 255) / 256)), dim3(256), 0, s, (const double *)acc64, (int)(M * ldc), C, 0);
}

inline unsigned blocks(int64_t n, int t = 256) { return (unsigned)((n + t - 1) / t); }

}  // namespace

extern "C" {

int64_t gcc_ginx_workspace_bytes(int64_t node_cap, int32_t batch_size, int32_t num_gin_layers, int32_t d_in, int32_t hidden, int32_t out_dim)
{
    if (node_cap <= 0 || batch_size <= 0 || num_gin_layers < 1 || num_gin_layers > GCC_GIN_MAX_LAYERS || d_in < 1 || hidden < 1 || out_dim < 1) {
        snprintf(g_err, kErrLen, "gcc_ginx_workspace_bytes: bad sizes");
        return -1;
    }
    return ginx_layout(node_cap, batch_size, num_gin_layers, d_in, hidden, out_dim).total * 4;
}

static int ginx_check(const gcc_ginx_pass *p, const char *who)
{
    const gcc_gin_weights &w = p->w;
    if (!p->node_off || !p->row_ptr || !p->col_idx || !p->graph_id || !p->pos || !p->workspace || !p->feat) {
        snprintf(g_err, kErrLen, "%s: NULL argument", who);
        return -1;
    }
    if (w.num_gin_layers < 1 || w.num_gin_layers > GCC_GIN_MAX_LAYERS || p->hidden < 1 || p->out_dim < 1 || p->batch_size < 1) {
        snprintf(g_err, kErrLen, "%s: bad sizes", who);
        return -2;
    }
    const int d_in = w.pos_dim + w.deg_emb_dim + 1;
    if (p->workspace_bytes < gcc_ginx_workspace_bytes(p->node_cap, p->batch_size, w.num_gin_layers, d_in, p->hidden, p->out_dim)) {
        snprintf(g_err, kErrLen, "%s: workspace too small", who);
        return -3;
    }
    return 0;
}

int32_t gcc_ginx_forward(const gcc_ginx_pass *p, void *stream)
{
    if (int rc = ginx_check(p, "gcc_ginx_forward")) return rc;
    hipStream_t s = (hipStream_t)stream;
    const gcc_gin_weights &w = p->w;
    const int L = w.num_gin_layers, B = p->batch_size, W = p->hidden, O = p->out_dim, d_in = w.pos_dim + w.deg_emb_dim + 1;
    const int64_t N = p->node_cap;
    const XLayout x = ginx_layout(N, B, L, d_in, W, O);
    float *ws = (float *)p->workspace;
    double *sums = (double *)(ws + x.sums);
    const int32_t *rows = p->node_off + B;                   // the live row count, on the device
    hipLaunchKernelGGL(ginx_feat_kernel, dim3(blocks(N * d_in)), dim3(256), 0, s, p->node_off, p->row_ptr, p->graph_id, p->seed_local, p->pos,
                       w.degree_embedding, B, w.pos_dim, w.deg_emb_dim, w.max_degree, p->edge_multiplicity > 0 ? p->edge_multiplicity : 1, ws + x.x0);
    const float *h = ws + x.x0;
    int D = d_in;
    auto bn = [&](const float *in, const gcc_bn &m, int64_t mr_off, float *out) {          // statistics -> (mean, rstd) -> y
        if (p->training) {
            (void)hipMemsetAsync(sums, 0, sizeof(double) * 2 * W, s);
            hipLaunchKernelGGL(ginx_colsum_kernel, dim3(blocks(N, 128)), dim3(256), 0, s, p->node_off, B, 0, 0, in, (const float *)nullptr,
                               (const float *)nullptr, (const float *)nullptr, W, sums);
        }
        hipLaunchKernelGGL(ginx_bn_prepare_kernel, dim3(blocks(W)), dim3(256), 0, s, p->node_off, B, sums, W, w.bn_eps, w.bn_momentum, p->training,
                           p->update_running_stats, m.running_mean, m.running_var, m.num_batches_tracked, ws + mr_off);
        // (workspace blocks are 256-byte aligned: rows of 4 k floats stay 16-byte aligned; the parameters are the caller's)
        const int vec = (W & 3) == 0 && ((((uintptr_t)m.weight) | ((uintptr_t)m.bias)) & 15) == 0;
        hipLaunchKernelGGL(ginx_bn_relu_kernel, dim3(blocks(vec ? N * W / 4 : N * W)), dim3(256), 0, s, p->node_off, B, in, ws + mr_off, m.weight, m.bias, W, out, vec);
    };
    hipLaunchKernelGGL(ginx_pool_kernel, dim3(B), dim3(256), 0, s, p->node_off, h, D, ws + x.pooled[0]);
    for (int l = 0; l < L; ++l) {
        hipLaunchKernelGGL(ginx_spmm_kernel, dim3(blocks(N, 4)), dim3(256), 0, s, p->node_off, p->row_ptr, p->col_idx, B, h, D, (const float *)nullptr,
                           ws + x.agg[l], (float)(p->edge_multiplicity > 1 ? p->edge_multiplicity : 1));
        gemm(s, ws + x.agg[l], D, 1, w.lin0_w[l], 1, D, ws + x.z1[l], W, 0, W, D, w.lin0_b[l], rows, 1, N);            // z1 = agg W0^T + b0
        bn(ws + x.z1[l], w.bn_a[l], x.mr[l][0], ws + x.a1[l]);
        gemm(s, ws + x.a1[l], W, 1, w.lin1_w[l], 1, W, ws + x.z2[l], W, 0, W, W, w.lin1_b[l], rows, 1, N);             // z2 = a1 W1^T + b1
        bn(ws + x.z2[l], w.bn_b[l], x.mr[l][1], ws + x.a2[l]);
        bn(ws + x.a2[l], w.bn_c[l], x.mr[l][2], ws + x.h[l]);
        h = ws + x.h[l];
        D = W;
        hipLaunchKernelGGL(ginx_pool_kernel, dim3(B), dim3(256), 0, s, p->node_off, h, D, ws + x.pooled[l + 1]);
    }
    for (int l = 0; l <= L; ++l) {                           // score = sum_l dropout(pooled_l Wp_l^T + bp_l)
        const int Dl = l == 0 ? d_in : W;
        gemm(s, ws + x.pooled[l], Dl, 1, w.pred_w[l], 1, Dl, ws + x.y, O, B, O, Dl, w.pred_b[l], nullptr, 0, 0);
        const float *keep = p->dropout_keep ? p->dropout_keep + (int64_t)l * B * O : nullptr;
        hipLaunchKernelGGL(ginx_mask_acc_kernel, dim3(blocks((int64_t)B * O)), dim3(256), 0, s, ws + x.y, keep, 1.0f / (1.0f - w.dropout_p), B * O,
                           ws + x.score, l == 0 ? 1 : 0);
    }
    hipLaunchKernelGGL(ginx_normalize_kernel, dim3(B), dim3(64), 0, s, ws + x.score, O, w.norm_eps, p->normalize, p->feat);
    if (p->pooled_out)
        for (int l = 1; l <= L; ++l)
            (void)hipMemcpyAsync(p->pooled_out + (int64_t)(l - 1) * B * W, ws + x.pooled[l], sizeof(float) * (size_t)B * W, hipMemcpyDeviceToDevice, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_err, kErrLen, "gcc_ginx_forward: launch failed: %s", hipGetErrorString(e)); return -10; }
    return 0;
}

int32_t gcc_ginx_backward(const gcc_ginx_pass *p, const float *dfeat, const gcc_gin_grads *gr, void *stream)
{
    if (int rc = ginx_check(p, "gcc_ginx_backward")) return rc;
    if (!dfeat || !gr || !p->training) { snprintf(g_err, kErrLen, "gcc_ginx_backward: needs dfeat, grads and a training-mode pass"); return -1; }
    if (p->edge_multiplicity > 1) { snprintf(g_err, kErrLen, "gcc_ginx_backward: edge_multiplicity must be 1 (as gcc_gin_backward)"); return -4; }
    hipStream_t s = (hipStream_t)stream;
    const gcc_gin_weights &w = p->w;
    const int L = w.num_gin_layers, B = p->batch_size, W = p->hidden, O = p->out_dim, d_in = w.pos_dim + w.deg_emb_dim + 1;
    const int64_t N = p->node_cap;
    const XLayout x = ginx_layout(N, B, L, d_in, W, O);
    float *ws = (float *)p->workspace;
    double *sums = (double *)(ws + x.sums);
    const int32_t *rows = p->node_off + B;
    float *dA = ws + x.da, *dB_ = ws + x.db, *dC = ws + x.dc;
    double *wg64 = (double *)(ws + x.wg64);
    // ---- readout: dscore -> per hidden_rep: dy = dscore * keep / (1 - p); dWp = dy^T pooled; dbp = colsum(dy); dpooled = dy Wp
    hipLaunchKernelGGL(ginx_normalize_bwd_kernel, dim3(B), dim3(64), 0, s, ws + x.score, p->feat, dfeat, O, w.norm_eps, p->normalize, ws + x.dscore);
    auto readout_bwd = [&](int l, float *dpool) {
        const int Dl = l == 0 ? d_in : W;
        const float *keep = p->dropout_keep ? p->dropout_keep + (int64_t)l * B * O : nullptr;
        hipLaunchKernelGGL(ginx_mask_acc_kernel, dim3(blocks((int64_t)B * O)), dim3(256), 0, s, ws + x.dscore, keep, 1.0f / (1.0f - w.dropout_p), B * O,
                           ws + x.dy, 1);
        gemm(s, ws + x.dy, 1, O, ws + x.pooled[l], Dl, 1, gr->pred_w[l], Dl, O, Dl, B, nullptr, nullptr, 0, 0);         // dWp [O, Dl] = dy^T pooled
        (void)hipMemsetAsync(sums, 0, sizeof(double) * O, s);
        hipLaunchKernelGGL(ginx_colsum_kernel, dim3(blocks(B, 128)), dim3(256), 0, s, p->node_off, B, B, 2, ws + x.dy, (const float *)nullptr,
                           (const float *)nullptr, (const float *)nullptr, O, sums);
        hipLaunchKernelGGL(ginx_sums_to_grad_kernel, dim3(blocks(O)), dim3(256), 0, s, sums, O, gr->pred_b[l], 0);
        gemm(s, ws + x.dy, O, 1, w.pred_w[l], Dl, 1, dpool, Dl, B, Dl, O, nullptr, nullptr, 0, 0);                        // dpooled [B, Dl] = dy Wp
    };
    // BatchNorm + ReLU backward: (x, y, dy) -> dx in place of dy's buffer `dx`; gamma / beta gradients from the two column sums
    auto bn_bwd = [&](const float *xin, const float *y, const float *dy, const gcc_bn &m, int64_t mr_off, float *dx, float *dgamma, float *dbeta) {
        (void)hipMemsetAsync(sums, 0, sizeof(double) * 2 * W, s);
        hipLaunchKernelGGL(ginx_colsum_kernel, dim3(blocks(N, 128)), dim3(256), 0, s, p->node_off, B, 0, 1, xin, y, dy, (const float *)(ws + mr_off), W, sums);
        hipLaunchKernelGGL(ginx_sums_to_grad_kernel, dim3(blocks(W)), dim3(256), 0, s, sums, W, dbeta, 0);
        hipLaunchKernelGGL(ginx_sums_to_grad_kernel, dim3(blocks(W)), dim3(256), 0, s, sums + W, W, dgamma, 0);
        const int vec = (W & 3) == 0 && (((uintptr_t)m.weight) & 15) == 0;
        hipLaunchKernelGGL(ginx_bn_relu_bwd_kernel, dim3(blocks(vec ? N * W / 4 : N * W)), dim3(256), 0, s, p->node_off, B, xin, y, dy, (const float *)(ws + mr_off), m.weight,
                           sums, W, dx, vec);
    };
    auto bias_grad = [&](const float *dz, float *db) {
        (void)hipMemsetAsync(sums, 0, sizeof(double) * W, s);
        hipLaunchKernelGGL(ginx_colsum_kernel, dim3(blocks(N, 128)), dim3(256), 0, s, p->node_off, B, 0, 2, dz, (const float *)nullptr,
                           (const float *)nullptr, (const float *)nullptr, W, sums);
        hipLaunchKernelGGL(ginx_sums_to_grad_kernel, dim3(blocks(W)), dim3(256), 0, s, sums, W, db, 0);
    };
    // dh of the last hidden representation: only its pooled readout feeds the loss
    readout_bwd(L, ws + x.dpool);
    hipLaunchKernelGGL(ginx_pool_bwd_kernel, dim3(blocks((W & 3) == 0 ? N * W / 4 : N * W)), dim3(256), 0, s, p->node_off, p->graph_id, B, ws + x.dpool, W, dA, 0, (W & 3) == 0);
    for (int l = L - 1; l >= 0; --l) {
        const int Din = l == 0 ? d_in : W;
        bn_bwd(ws + x.a2[l], ws + x.h[l], dA, w.bn_c[l], x.mr[l][2], dB_, gr->bn_c_w[l], gr->bn_c_b[l]);                  // -> d a2
        bn_bwd(ws + x.z2[l], ws + x.a2[l], dB_, w.bn_b[l], x.mr[l][1], dA, gr->bn_b_w[l], gr->bn_b_b[l]);                 // -> d z2
        gemm(s, dA, 1, W, ws + x.a1[l], W, 1, gr->lin1_w[l], W, W, W, 0, nullptr, rows, 2, N, 1.0f, wg64);                            // dW1 [W, W] = dz2^T a1
        bias_grad(dA, gr->lin1_b[l]);
        gemm(s, dA, W, 1, w.lin1_w[l], W, 1, dB_, W, 0, W, W, nullptr, rows, 1, N);                                        // d a1 = dz2 W1
        bn_bwd(ws + x.z1[l], ws + x.a1[l], dB_, w.bn_a[l], x.mr[l][0], dA, gr->bn_a_w[l], gr->bn_a_b[l]);                 // -> d z1
        gemm(s, dA, 1, W, ws + x.agg[l], Din, 1, gr->lin0_w[l], Din, W, Din, 0, nullptr, rows, 2, N, 1.0f, wg64);                     // dW0 [W, Din] = dz1^T agg
        bias_grad(dA, gr->lin0_b[l]);
        gemm(s, dA, W, 1, w.lin0_w[l], Din, 1, dB_, Din, 0, Din, W, nullptr, rows, 1, N);                                  // d agg = dz1 W0
        // d h_{l-1} = d agg + A d agg (the batched subgraph is symmetric) + the pooled readout of hidden_rep[l]'s input
        readout_bwd(l, ws + x.dpool);
        hipLaunchKernelGGL(ginx_pool_bwd_kernel, dim3(blocks((Din & 3) == 0 ? N * Din / 4 : N * Din)), dim3(256), 0, s, p->node_off, p->graph_id, B, ws + x.dpool, Din, dC, 0,
                           (Din & 3) == 0);
        hipLaunchKernelGGL(ginx_spmm_kernel, dim3(blocks(N, 4)), dim3(256), 0, s, p->node_off, p->row_ptr, p->col_idx, B, dB_, Din, (const float *)dC, dA, 1.0f);
    }
    // d x0 (in dA, width d_in) -> the degree embedding's rows
    (void)hipMemsetAsync(gr->degree_embedding, 0, sizeof(float) * (size_t)(w.max_degree + 1) * w.deg_emb_dim, s);
    hipLaunchKernelGGL(ginx_feat_bwd_kernel, dim3(blocks(N, kFeatRows)), dim3(256), 0, s, p->node_off, p->row_ptr, B, w.pos_dim, w.deg_emb_dim,
                       w.max_degree, dA, gr->degree_embedding);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_err, kErrLen, "gcc_ginx_backward: launch failed: %s", hipGetErrorString(e)); return -10; }
    return 0;
}

/* ---- the head at any feature size */
int32_t gcc_ncex_forward(const float *q, const float *k, const float *mem, int32_t B, int32_t K, int32_t D, float inv_T, int32_t mode,
                         float *out, float *dlog, float *grad_rows, float *grad_mem, float *loss, float *prob, double *acc, void *stream)
{
    if (!q || !mem || !out || !dlog || !grad_rows || !loss || !prob || !acc || B < 1 || K < 1 || D < 1 || (mode == 0 && !k) || (mode == 1 && (K != B || !grad_mem))) {
        snprintf(g_err, kErrLen, "gcc_ncex_forward: bad arguments");
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    const int ncols = mode == 0 ? K + 1 : K;
    const int64_t ld = ncols;
    float *neg = out + (mode == 0 ? 1 : 0);
    (void)hipMemsetAsync(acc, 0, 2 * sizeof(double), s);
    if (mode == 0) hipLaunchKernelGGL(ginx_rowdot_kernel, dim3(B), dim3(64), 0, s, q, k, D, inv_T, out, ld);
    gemm(s, q, D, 1, mem, 1, D, neg, ld, B, K, D, nullptr, nullptr, 0, 0, inv_T);                                  // q mem^T / T
    hipLaunchKernelGGL(ginx_ce_kernel, dim3(B), dim3(256), 0, s, (const float *)out, ld, ncols, mode, dlog, acc);
    hipLaunchKernelGGL(ginx_ce_final_kernel, dim3(1), dim3(64), 0, s, (const double *)acc, B, loss, prob);
    // the gradients for a unit upstream gradient, taken NOW -- before the caller enqueues the step's keys over queue rows
    // (memory_moco.py:55-61): d loss / d rows = (softmax - onehot) [k; mem] / (T B)
    const float coef = inv_T / (float)B;
    // (reduction over the K queue rows: split over workgroups when it is long, fp64 atomics into acc + 2)
    if (K >= 4 * kSplitRows) gemm(s, dlog + (mode == 0 ? 1 : 0), ld, 1, mem, D, 1, grad_rows, D, B, D, K, nullptr, nullptr, 3, 0, coef, acc + 2);
    else gemm(s, dlog + (mode == 0 ? 1 : 0), ld, 1, mem, D, 1, grad_rows, D, B, D, K, nullptr, nullptr, 0, 0, coef);
    if (mode == 0) hipLaunchKernelGGL(ginx_rank1_rows_kernel, dim3(blocks((int64_t)B * D)), dim3(256), 0, s, (const float *)dlog, ld, k, B, D, coef, grad_rows);
    else gemm(s, dlog, 1, ld, q, D, 1, grad_mem, D, K, D, B, nullptr, nullptr, 0, 0, coef);                         // d loss / d mem rows = dlog^T rows / (T B)
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_err, kErrLen, "gcc_ncex_forward: launch failed: %s", hipGetErrorString(e)); return -10; }
    return 0;
}

int32_t gcc_queue_enqueue_x(float *mem, int32_t K, int32_t D, const float *keys, int32_t nkeys, int32_t index, void *stream)
{
    if (!mem || !keys || nkeys < 0 || nkeys > K || index < 0 || index >= K) { snprintf(g_err, kErrLen, "gcc_queue_enqueue_x: bad arguments"); return -1; }
    hipLaunchKernelGGL(ginx_enqueue_kernel, dim3(blocks((int64_t)nkeys * D)), dim3(256), 0, (hipStream_t)stream, mem, K, D, keys, nkeys, index);
    return 0;
}

}  // extern "C"
