// gcc_amd/csrc/nce.hip -- MoCo / InfoNCE head, queue enqueue, EMA (gfx950).
//
// Replaces MemoryMoCo.forward (gcc/contrastive/memory_moco.py:26-63: bmm/mm/cat/div/
// clone/index_copy_), NCESoftmaxLoss / NCESoftmaxLossNS (gcc/contrastive/criterions.py:5-33:
// CrossEntropyLoss over [B, K+1]) and moment_update (train.py:169-172).
//
//   (dtype GCC_NCE_BF16: the same kernels with q / k / queue rounded to bf16 on load -- the forward logits then run on
//   v_mfma_f32_16x16x32_bf16, 2 instructions per 16 x 16 tile instead of 16; accumulation and softmax stay fp32)
//   nce_slice_kernel<false>  grid (S slices of the queue, B/64 query blocks).  The slice's rows
//       are staged through LDS in 64-row chunks shared by the 4 waves; each wave owns 16 queries:
//       logits tile = exact-f32 MFMA (queue rows x queries), online row-softmax (running max /
//       sum) in registers; per-(slice, query) partials go to HBM.  K = 16384, B = 256: 537 MFLOP,
//       4 MiB of queue read once per query block.
//   nce_combine_kernel       merges the partials: lse, positive logit, mean loss, mean prob.
//   nce_slice_kernel<true>   backward: recomputes the logits tile, p = exp(l - lse), and feeds it
//       straight into a second MFMA (p^T x queue rows) -- the 4 logits a lane holds after the
//       first MFMA are exactly the B-operand values it must supply to the second, so P never
//       leaves registers.  Per-slice dq slabs are reduced in a fixed order by nce_dq_kernel.
// [B, K+1] is only materialised when the caller asks for it (train.py indexes out[:, 0]).
#include "host_common.h"

#include <math.h>

namespace {

constexpr int D = GCC_NCE_DIM;    // 64
constexpr int kLd = 72;           // LDS row stride (floats)
constexpr int kChunk = 64;        // queue rows per staged chunk
constexpr int kThreads = 256;
constexpr int kQPerBlock = 64;    // 4 waves x 16 queries

struct F4 { float x, y, z, w; };
__device__ __forceinline__ F4 ld4(const float *p)
{
    const float4 v = *reinterpret_cast<const float4 *>(p);
    F4 r = {v.x, v.y, v.z, v.w};
    return r;
}
__device__ __forceinline__ void st4(float *p, F4 v) { *reinterpret_cast<float4 *>(p) = make_float4(v.x, v.y, v.z, v.w); }

struct NceDev {
    const float *q, *k, *mem, *patch;
    int32_t patch_index, patch_rows, B, K, pos_mode;
    int32_t bf16;            // operands rounded to bf16 (f32 accumulation): gcc_nce_args.dtype
    float inv_T;
    float *lse, *pos, *loss, *prob, *out_dense;
    int32_t S, R;             // slices and rows per slice (multiple of kChunk)
    float *pm, *ps;           // [S][B] partial max / sum
    float *slabs;             // [S][B][64] partial dq (backward)
    int32_t *ticket;          // arrival counter of nce_combine_kernel's workgroups (zeroed by the forward slice kernel)
    const float *dloss;
    int32_t by_mem_row;
    float *dq;
};

#ifndef GCC_NCE_WGS_DEFAULT
#define GCC_NCE_WGS_DEFAULT 256
#endif
constexpr int kNceWgs = GCC_NCE_WGS_DEFAULT;     // ~one workgroup per CU: enough to spread the queue, few enough slabs to reduce
struct Plan { int32_t S, R, QB; int64_t off_pm, off_ps, off_slabs, off_ticket, total; };

inline Plan make_plan(int32_t B, int32_t K)
{
    Plan p;
    p.QB = (B + kQPerBlock - 1) / kQPerBlock;
    static int wgs = 0;                      // workgroups of a slice launch (GCC_NCE_WGS: timing experiments)
    if (wgs == 0) { const char *e = getenv("GCC_NCE_WGS"); wgs = e ? atoi(e) : 0; if (wgs <= 0) wgs = kNceWgs; }
    int s = (wgs + p.QB - 1) / p.QB;
    const int maxs = (K + kChunk - 1) / kChunk;
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    p.R = (((K + s - 1) / s) + kChunk - 1) / kChunk * kChunk;
    p.S = (K + p.R - 1) / p.R;
    auto al = [](int64_t x) { return (x + 255) & ~(int64_t)255; };
    int64_t o = 0;
    p.off_pm = o; o = al(o + (int64_t)p.S * B * 4);
    p.off_ps = o; o = al(o + (int64_t)p.S * B * 4);
    p.off_slabs = o; o = al(o + (int64_t)p.S * B * D * 4);
    p.off_ticket = o; o = al(o + 4);
    p.total = o;
    return p;
}

__device__ __forceinline__ float rnd_bf16(float x) { return bf16_bits_to_f32(f32_to_bf16_bits(x)); }
__device__ __forceinline__ F4 rnd4(F4 v, bool on)
{
    if (on) { v.x = rnd_bf16(v.x); v.y = rnd_bf16(v.y); v.z = rnd_bf16(v.z); v.w = rnd_bf16(v.w); }
    return v;
}
// two fp32 values that are exactly representable in bf16 -> one packed pair (low half = first)
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) { return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xFFFF0000u); }

// One unconditional 16-byte load per call (the ADDRESS is selected, not the load: with the load under branches the compiler
// waited for each of a thread's four rows before requesting the next); rows >= K read row 0 and are zeroed by the caller.
__device__ __forceinline__ const float *mem_row_ptr(const NceDev &a, int r)
{
    const float *p = a.mem + (int64_t)(r < a.K ? r : 0) * D;
    if (a.patch) {                                   // (uniform) queue rows already overwritten by the enqueue
        int rel = r - a.patch_index;
        if (rel < 0) rel += a.K;
        if (r < a.K && rel < a.patch_rows) p = a.patch + (int64_t)rel * D;
    }
    return p;
}
// what follows a queue-row load once ALL loads of the batch have been requested: rows past the end are zeros, bf16 mode rounds
__device__ __forceinline__ F4 finish_mem4(const NceDev &a, F4 v, int r)
{
    if (r >= a.K) { F4 z = {0.f, 0.f, 0.f, 0.f}; v = z; }
    return rnd4(v, a.bf16 != 0);
}

template <bool kBwd, bool kBf16>
__global__ __launch_bounds__(kThreads) void nce_slice_kernel(NceDev a)
{
    TRAIN_STEP_WAVE_PRIORITY();
    __shared__ float Ms[kChunk * kLd];
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = tid >> 6, j = lane & 15, q = lane >> 4;
    const int s = (int)blockIdx.x;
    if (!kBwd && s == 0 && blockIdx.y == 0 && tid == 0) *a.ticket = 0;     // arrival counter of nce_combine_kernel
    const int qj = (int)blockIdx.y * kQPerBlock + 16 * wv + j;
    const bool qvalid = qj < a.B;
    F4 qf[4];
    const float *qrow = a.q + (int64_t)(qvalid ? qj : 0) * D;      // (unconditional loads, masked afterwards)
#pragma unroll
    for (int c = 0; c < 4; ++c) qf[c] = ld4(qrow + 16 * c + 4 * q);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        F4 z = {0.f, 0.f, 0.f, 0.f};
        qf[c] = qvalid ? rnd4(qf[c], kBf16) : z;
    }
    // bf16 mode, forward logits on v_mfma_f32_16x16x32_bf16: lane (j, q) supplies 8 consecutive k of query j for the k-group q of
    // each 32-wide slab (the queue rows come from LDS the same way); the products of bf16 values are exact in fp32, so the
    // result differs from the fp32-MFMA path on the rounded operands (what the backward recomputes) by summation order only
    u32x4 qb[2];
    if (kBf16 && !kBwd) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            F4 z = {0.f, 0.f, 0.f, 0.f};
            const F4 lo = qvalid ? rnd4(ld4(a.q + (int64_t)qj * D + 32 * sl + 8 * q), true) : z;
            const F4 hi = qvalid ? rnd4(ld4(a.q + (int64_t)qj * D + 32 * sl + 8 * q + 4), true) : z;
            qb[sl][0] = pack_bf16(lo.x, lo.y); qb[sl][1] = pack_bf16(lo.z, lo.w);
            qb[sl][2] = pack_bf16(hi.x, hi.y); qb[sl][3] = pack_bf16(hi.z, hi.w);
        }
    }
    const int row_beg = s * a.R, row_end = min(a.K, row_beg + a.R);
    const int ld_out = a.K + (a.pos_mode == 0 ? 1 : 0), off_out = a.pos_mode == 0 ? 1 : 0;
    float m = -INFINITY, ssum = 0.f;
    float my_lse = 0.f;
    if (kBwd && !a.by_mem_row && qvalid) my_lse = a.lse[qj];
    f32x4 acc2[4];
    if (kBwd) {
#pragma unroll
        for (int db = 0; db < 4; ++db) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc2[db] = z; }
    }
    // the next chunk of queue rows is requested while this one is being multiplied (a slice is 4 chunks: a round trip each)
    F4 nxt[4];
    auto request = [&](int c0) {                     // raw loads only: zeroing / rounding wait until the chunk is stored
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + kThreads * i, row = idx >> 4, c4 = (idx & 15) * 4;
            nxt[i] = ld4(mem_row_ptr(a, c0 + row) + c4);
        }
    };
    if (row_beg < row_end) request(row_beg);
    for (int c0 = row_beg; c0 < row_end; c0 += kChunk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + kThreads * i, row = idx >> 4, c4 = (idx & 15) * 4;
            st4(&Ms[row * kLd + c4], finish_mem4(a, nxt[i], c0 + row));
        }
        __syncthreads();
        if (c0 + kChunk < row_end) request(c0 + kChunk);
        for (int t = 0; t < 4; ++t) {
            if (c0 + 16 * t >= row_end) break;      // block-uniform
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (kBf16 && !kBwd) {
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const F4 lo = ld4(&Ms[(16 * t + j) * kLd + 32 * sl + 8 * q]), hi = ld4(&Ms[(16 * t + j) * kLd + 32 * sl + 8 * q + 4]);
                    u32x4 mb;
                    mb[0] = pack_bf16(lo.x, lo.y); mb[1] = pack_bf16(lo.z, lo.w);
                    mb[2] = pack_bf16(hi.x, hi.y); mb[3] = pack_bf16(hi.z, hi.w);
                    acc = mfma_16x16x32_bf16(mb, qb[sl], acc);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const F4 mf = ld4(&Ms[(16 * t + j) * kLd + 16 * c + 4 * q]);
                    acc = mfma_16x16x4_f32(mf.x, qf[c].x, acc);
                    acc = mfma_16x16x4_f32(mf.y, qf[c].y, acc);
                    acc = mfma_16x16x4_f32(mf.z, qf[c].z, acc);
                    acc = mfma_16x16x4_f32(mf.w, qf[c].w, acc);
                }
            }
            // acc[r] = mem[row0 + r] . q[qj], row0 = c0 + 16 t + 4 q
            const int row0 = c0 + 16 * t + 4 * q;
            float lv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float l = acc[r] * a.inv_T;                      // torch.div(out, self.T), memory_moco.py:43
                const bool valid = row0 + r < row_end;
                if (!kBwd && a.out_dense && qvalid && valid) a.out_dense[(int64_t)qj * ld_out + off_out + row0 + r] = l;
                lv[r] = valid ? l : -INFINITY;
            }
            if (!kBwd) {
                const float tmax = fmaxf(fmaxf(lv[0], lv[1]), fmaxf(lv[2], lv[3]));
                if (tmax > -INFINITY) {
                    const float mn = fmaxf(m, tmax);
                    ssum = ssum * expf(m - mn) + expf(lv[0] - mn) + expf(lv[1] - mn) + expf(lv[2] - mn) + expf(lv[3] - mn);
                    m = mn;
                }
            } else {
                float lse_r[4] = {my_lse, my_lse, my_lse, my_lse};
                if (a.by_mem_row) {                          // (uniform) E2E: the softmax runs over the other view's rows
#pragma unroll
                    for (int r = 0; r < 4; ++r) lse_r[r] = a.lse[min(row0 + r, a.K - 1)];      // four loads requested together
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = 0.f;
                    if (qvalid && lv[r] > -INFINITY) p = expf(lv[r] - lse_r[r]);
                    // second GEMM: dq[query][d] += p[row][query] * mem[row][d]; this lane's p is the
                    // B operand (k = q <-> row 4q + r, column j = query)
#pragma unroll
                    for (int db = 0; db < 4; ++db)
                        acc2[db] = mfma_16x16x4_f32(Ms[(16 * t + 4 * q + r) * kLd + 16 * db + j], p, acc2[db]);
                }
            }
        }
        __syncthreads();
    }
    if (!kBwd) {
#pragma unroll
        for (int d = 16; d <= 32; d <<= 1) {
            const float m2 = wave_shfl_xor(m, d), s2 = wave_shfl_xor(ssum, d);
            const float mn = fmaxf(m, m2);
            ssum = mn > -INFINITY ? ssum * expf(m - mn) + s2 * expf(m2 - mn) : 0.f;
            m = mn;
        }
        if (q == 0 && qvalid) {
            a.pm[(int64_t)s * a.B + qj] = m;
            a.ps[(int64_t)s * a.B + qj] = ssum;
        }
    } else if (qvalid) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            F4 o = {acc2[db][0], acc2[db][1], acc2[db][2], acc2[db][3]};
            st4(a.slabs + ((int64_t)s * a.B + qj) * D + 16 * db + 4 * q, o);
        }
    }
}

__global__ __launch_bounds__(kThreads) void nce_combine_kernel(NceDev a)
{
    TRAIN_STEP_WAVE_PRIORITY();
    // one wave per query (lanes over the 64 feature dims / the slices); the workgroup that arrives last adds the
    // per-query terms up in index order (deterministic) for the mean loss and the mean positive logit
    __shared__ double red[8];
    __shared__ int last;
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = tid >> 6;
    const int ld_out = a.K + (a.pos_mode == 0 ? 1 : 0);
    const int b = (int)blockIdx.x * (kThreads >> 6) + wv;
    if (b < a.B) {
        const float *other = a.pos_mode == 0 ? a.k : a.mem;        // l_pos = bmm(q, k) | diagonal of k q^T
        // (the query, the key and the first 64 slices' maxima and sums in one round trip; more slices loop below)
        float qv = a.q[(int64_t)b * D + lane], ov = other[(int64_t)b * D + lane];
        const int s0 = min(lane, a.S - 1);
        const float pm0r = a.pm[(int64_t)s0 * a.B + b], ps0r = a.ps[(int64_t)s0 * a.B + b];
        const float pm0 = lane < a.S ? pm0r : -INFINITY;
        if (a.bf16) { qv = rnd_bf16(qv); ov = rnd_bf16(ov); }
        const float pos = wave_sum(qv * ov) * a.inv_T;
        float m = pm0;
        for (int s = lane + 64; s < a.S; s += 64) m = fmaxf(m, a.pm[(int64_t)s * a.B + b]);
        m = wave_max(m);
        if (a.pos_mode == 0) m = fmaxf(m, pos);
        float sum = pm0 > -INFINITY ? ps0r * expf(pm0 - m) : 0.f;
        for (int s = lane + 64; s < a.S; s += 64) {
            const float pmv = a.pm[(int64_t)s * a.B + b], psv = a.ps[(int64_t)s * a.B + b];
            if (pmv > -INFINITY) sum += psv * expf(pmv - m);
        }
        sum = wave_sum(sum);
        if (a.pos_mode == 0) sum += expf(pos - m);
        const float lse = m + logf(sum);
        if (lane == 0) {
            a.lse[b] = lse;
            a.pos[b] = pos;
            if (a.pos_mode == 0 && a.out_dense) a.out_dense[(int64_t)b * ld_out] = pos;
        }
    }
    device_fence();                                              // lse / pos of this workgroup are visible device wide
    __syncthreads();
    if (tid == 0) last = atomicAdd(a.ticket, 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    device_fence();
    double lsum = 0.0, psum = 0.0;
    for (int i = tid; i < a.B; i += kThreads) {
        const float lse = load_fresh(a.lse + i), pos = load_fresh(a.pos + i);
        lsum += (double)(lse - pos);                             // CrossEntropyLoss row term
        psum += (double)pos;
    }
    lsum = wave_sum(lsum);
    psum = wave_sum(psum);
    if (lane == 0) { red[wv] = lsum; red[4 + wv] = psum; }
    __syncthreads();
    if (tid == 0) {
        a.loss[0] = (float)((red[0] + red[1] + red[2] + red[3]) / (double)a.B);    // reduction="mean"
        a.prob[0] = (float)((red[4] + red[5] + red[6] + red[7]) / (double)a.B);    // out[:, 0].mean(), train.py:394
        *a.ticket = 0;
    }
}

#ifndef NCE_DQ_BATCH
#define NCE_DQ_BATCH 16      // slabs requested per round
#endif
__global__ __launch_bounds__(kThreads) void nce_dq_kernel(NceDev a)
{
    TRAIN_STEP_WAVE_PRIORITY();
    const int gid = (int)blockIdx.x * kThreads + (int)threadIdx.x;
    const int b = gid >> 4, c4 = (gid & 15) * 4;
    if (b >= a.B) return;
    // everything the epilogue needs is requested first and the slabs 16 at a time: with 4 per round and the scalars
    // loaded at the end this was S / 4 + 3 = 19 dependent round trips for ~1 us of data
    const float dl = a.dloss[0];
    const float posb = a.pos_mode == 0 ? a.pos[b] : 0.f, lseb = a.pos_mode == 0 ? a.lse[b] : 0.f;     // (block-uniform branch)
    const F4 other = rnd4(ld4((a.pos_mode == 0 ? a.k : a.mem) + (int64_t)b * D + c4), a.bf16 != 0);
    F4 s = {0.f, 0.f, 0.f, 0.f};
    const float *sp = a.slabs + (int64_t)b * D + c4;
    const int64_t st = (int64_t)a.B * D;
    for (int sl = 0; sl < a.S; sl += NCE_DQ_BATCH) {  // fixed summation order: slab 0, 1, 2, ...
        F4 v[NCE_DQ_BATCH];
#pragma unroll
        for (int u = 0; u < NCE_DQ_BATCH; ++u) v[u] = ld4(sp + (int64_t)min(sl + u, a.S - 1) * st);
#pragma unroll
        for (int u = 0; u < NCE_DQ_BATCH; ++u)
            if (sl + u < a.S) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    const int nrows = a.by_mem_row ? a.K : a.B;              // rows of the softmax that is averaged
    const float coef = dl * a.inv_T / (float)nrows;
    F4 o;
    if (a.pos_mode == 0) {
        const float pp = expf(posb - lseb) - 1.f;
        o.x = coef * (s.x + pp * other.x); o.y = coef * (s.y + pp * other.y);
        o.z = coef * (s.z + pp * other.z); o.w = coef * (s.w + pp * other.w);
    } else {
        o.x = coef * (s.x - other.x); o.y = coef * (s.y - other.y); o.z = coef * (s.z - other.z); o.w = coef * (s.w - other.w);
    }
    st4(a.dq + (int64_t)b * D + c4, o);
}

__global__ __launch_bounds__(kThreads) void queue_enqueue_kernel(float *mem, int K, const float *keys, int nkeys,
                                                                 int index, float *saved, const gcc_step_scalars *sc)
{
    TRAIN_STEP_WAVE_PRIORITY();
    if (sc) index = sc->enqueue_index % K;                    // replayed step: the ring pointer lives on the device
    const int gid = (int)blockIdx.x * kThreads + (int)threadIdx.x;
    const int i = gid >> 4, c4 = (gid & 15) * 4;
    if (i >= nkeys) return;
    const int row = (index + i) % K;                          // torch.fmod(out_ids + index, queueSize)
    if (saved) st4(saved + (int64_t)i * D + c4, ld4(mem + (int64_t)row * D + c4));
    st4(mem + (int64_t)row * D + c4, ld4(keys + (int64_t)i * D + c4));
}

__global__ __launch_bounds__(kThreads) void ema_kernel(float *ema, const float *p, int64_t n, float m)
{
    TRAIN_STEP_WAVE_PRIORITY();
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride)
        ema[i] = ema[i] * m + (1.f - m) * p[i];               // p2.mul_(m).add_(1 - m, p1)
}

// ---- clip_grad_norm_ + Adam over one flat buffer (train.py:409,417)
constexpr int kNormBlocks = 32;
// sum of squares of the flat gradient: kNormBlocks partial sums; the workgroup that arrives last adds them up in index
// order (deterministic).  scratch: [0] = result, [1] = ticket (as int), [2 .. 2 + kNormBlocks) = partials.
__global__ __launch_bounds__(kThreads) void gradnorm_kernel(const float *g, int64_t n, double *scratch)
{
    TRAIN_STEP_WAVE_PRIORITY();
    static_assert(kNormBlocks <= 64, "the last workgroup adds the partials up with one wave");
    __shared__ double red[kThreads / 64];
    __shared__ int last;
    const int tid = (int)threadIdx.x;
    double s = 0.0;
    // 8 independent loads per round (clamped address, masked afterwards): one element per round was a dependent round
    // trip per element, ~12 in a row for the 100 k parameters of the GIN encoder
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t base = (int64_t)blockIdx.x * kThreads + tid; base < n; base += 8 * stride) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int64_t i = base + u * stride; x[u] = g[i < n ? i : n - 1]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (base + u * stride < n) s += (double)x[u] * (double)x[u];
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int i = 0; i < kThreads / 64; ++i) t += red[i];
        scratch[2 + blockIdx.x] = t;
        device_fence();
        last = atomicAdd((int *)(scratch + 1), 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (last && tid < 64) {                                  // block-uniform; one wave: the partials in one round trip, fixed tree
        device_fence();
        const double v = tid < (int)gridDim.x ? load_fresh_f64(scratch + 2 + tid) : 0.0;
        const double tot = wave_sum(v);
        if (tid == 0) {
            scratch[0] = tot;
            *(int *)(scratch + 1) = 0;
        }
    }
}

// optional extras of the Adam launch (gcc_adam_ema_step): the EMA copy of the parameters and the per-step meters
struct AdamExtras {
    float *ema; int64_t n_ema; float ema_m;
    const gcc_step_scalars *sc;      // replayed step: lr and the bias corrections come from here
    double *acc; int32_t *mx; const float *loss, *prob; const int32_t *node_off_q, *edge_off_q, *node_off_k; int32_t B;
};

__global__ __launch_bounds__(kThreads) void adam_kernel(float *p, float *g, float *m, float *v, int64_t n, float lr,
                                                        float b1, float b2, float eps, float wd, float bc1,
                                                        float bc2_sqrt, float max_norm, float grad_scale,
                                                        const double *sumsq, float *grad_norm, AdamExtras x)
{
    TRAIN_STEP_WAVE_PRIORITY();
    // grad_scale: the 1 / world of a summed (all-reduced) gradient, folded in here instead of a launch of its own
    const double ss = sumsq[0];
    if (x.sc) { lr = x.sc->lr; bc1 = x.sc->bias_corr1; bc2_sqrt = x.sc->bias_corr2_sqrt; }   // (uniform: one scalar load, in flight with the rest)
    // this thread's first element rides in the same round trip as the norm
    const int64_t stride = (int64_t)gridDim.x * kThreads, i0 = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t ic = i0 < n ? i0 : n - 1;
    float g0 = g[ic], p0 = p[ic], m0 = m[ic], v0 = v[ic];
    const float e0 = x.ema ? x.ema[i0 < x.n_ema ? i0 : x.n_ema - 1] : 0.f;      // (block-uniform branch)
    const float norm = grad_scale * (float)sqrt(ss);
    float coef = 1.f;
    if (max_norm > 0.f) {                                      // torch.nn.utils.clip_grad_norm_
        coef = max_norm / (norm + 1e-6f);
        coef = coef > 1.f ? 1.f : coef;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        grad_norm[0] = norm;
        if (x.acc) {                                           // train.py:418-428's meters (gcc_step_meters)
            const int32_t nq = x.node_off_q[x.B], nk = x.node_off_k[x.B], eq = x.edge_off_q[x.B];
            x.acc[0] += (double)x.loss[0];
            x.acc[1] += (double)x.prob[0];
            x.acc[2] += (double)norm;
            x.acc[3] += (double)nq + (double)nk;
            x.acc[4] += 1.0;
            x.mx[0] = nq > x.mx[0] ? nq : x.mx[0];
            x.mx[1] = eq > x.mx[1] ? eq : x.mx[1];
        }
    }
    coef *= grad_scale;
    const int64_t nmax = x.ema && x.n_ema > n ? x.n_ema : n;
    for (int64_t i = i0; i < nmax; i += stride) {
        float pi, ei = 0.f;
        if (i != i0) {
            const int64_t j = i < n ? i : n - 1;
            g0 = g[j]; p0 = p[j]; m0 = m[j]; v0 = v[j];
            if (x.ema) ei = x.ema[i < x.n_ema ? i : x.n_ema - 1];
        } else {
            ei = e0;
        }
        if (i < n) {
            const float gc = g0 * coef;
            g[i] = gc;                                         // the clipped gradient stays visible, as in torch
            const float gi = fmaf(wd, p0, gc);                 // weight_decay: grad = grad + wd * param
            const float mi = fmaf(b1, m0, (1.f - b1) * gi);
            const float vi = fmaf(b2, v0, (1.f - b2) * gi * gi);
            m[i] = mi;
            v[i] = vi;
            const float denom = sqrtf(vi) / bc2_sqrt + eps;
            pi = p0 - (lr / bc1) * (mi / denom);
            p[i] = pi;
        } else {
            pi = p[i];                                         // parameters past the live prefix: only averaged
        }
        if (x.ema && i < x.n_ema) x.ema[i] = ei * x.ema_m + (1.f - x.ema_m) * pi;   // p2.mul_(m).add_(1 - m, p1), train.py:169-172
    }
}

// ---- per-step meters of train.py:418-428 (loss / prob / gnorm / graph sizes) accumulated on the device: one thread
__global__ void step_meters_kernel(double *acc, int32_t *mx, const float *loss, const float *prob, const float *gnorm,
                                   const int32_t *node_off_q, const int32_t *edge_off_q, const int32_t *node_off_k,
                                   int32_t B)
{
    TRAIN_STEP_WAVE_PRIORITY();
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int32_t nq = node_off_q[B], nk = node_off_k[B], eq = edge_off_q[B];
    acc[0] += (double)loss[0];
    acc[1] += (double)prob[0];
    acc[2] += (double)gnorm[0];
    acc[3] += (double)nq + (double)nk;
    acc[4] += 1.0;
    mx[0] = nq > mx[0] ? nq : mx[0];
    mx[1] = eq > mx[1] ? eq : mx[1];
}

inline int fill_dev(const gcc_nce_args *a, void *workspace, int64_t workspace_bytes, NceDev &d, Plan &pl)
{
    if (!a || !a->q || !a->mem || a->B < 1 || a->K < 1 || (a->pos_mode == 0 && !a->k) ||
        (a->pos_mode == 1 && a->K < a->B) || !a->lse || !a->pos) {
        snprintf(g_err, kErrLen, "gcc_nce: bad argument");
        return -1;
    }
    pl = make_plan(a->B, a->K);
    if (!workspace || workspace_bytes < pl.total) {
        snprintf(g_err, kErrLen, "gcc_nce: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)pl.total);
        return -3;
    }
    d.q = a->q; d.k = a->k; d.mem = a->mem; d.patch = a->patch;
    d.patch_index = a->patch_index; d.patch_rows = a->patch ? a->patch_rows : 0;
    d.B = a->B; d.K = a->K; d.pos_mode = a->pos_mode; d.inv_T = a->inv_T;
    d.bf16 = a->dtype == GCC_NCE_BF16 ? 1 : 0;
    d.lse = a->lse; d.pos = a->pos; d.loss = a->loss; d.prob = a->prob; d.out_dense = a->out_dense;
    d.S = pl.S; d.R = pl.R;
    char *base = (char *)workspace;
    d.pm = (float *)(base + pl.off_pm); d.ps = (float *)(base + pl.off_ps); d.slabs = (float *)(base + pl.off_slabs);
    d.ticket = (int32_t *)(base + pl.off_ticket);
    d.dloss = nullptr; d.by_mem_row = 0; d.dq = nullptr;
    return 0;
}

}  // namespace

extern "C" {

int64_t gcc_nce_workspace_bytes(int32_t B, int32_t K)
{
    if (B < 1 || K < 1) { snprintf(g_err, kErrLen, "gcc_nce_workspace_bytes: bad argument"); return -1; }
    return make_plan(B, K).total;
}

int32_t gcc_nce_forward(const gcc_nce_args *a, void *workspace, int64_t workspace_bytes, gcc_prof *prof, void *stream)
{
    NceDev d;
    Plan pl;
    int rc = fill_dev(a, workspace, workspace_bytes, d, pl);
    if (rc) return rc;
    if (!a->loss || !a->prob) { snprintf(g_err, kErrLen, "gcc_nce_forward: loss/prob are required"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    prof_mark(prof, 0, s);
    if (d.bf16) hipLaunchKernelGGL((nce_slice_kernel<false, true>), dim3(pl.S, pl.QB), dim3(kThreads), 0, s, d);
    else hipLaunchKernelGGL((nce_slice_kernel<false, false>), dim3(pl.S, pl.QB), dim3(kThreads), 0, s, d);
    hipLaunchKernelGGL(nce_combine_kernel, dim3((d.B + (kThreads >> 6) - 1) / (kThreads >> 6)), dim3(kThreads), 0, s, d);
    prof_mark(prof, 1, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_err, kErrLen, "gcc_nce_forward: %s", hipGetErrorString(e)); return -10; }
    return 0;
}

int32_t gcc_nce_backward(const gcc_nce_args *a, const float *dloss, int32_t by_mem_row, float *dq, void *workspace,
                         int64_t workspace_bytes, gcc_prof *prof, void *stream)
{
    NceDev d;
    Plan pl;
    int rc = fill_dev(a, workspace, workspace_bytes, d, pl);
    if (rc) return rc;
    if (!dloss || !dq || (by_mem_row && a->pos_mode != 1)) {
        snprintf(g_err, kErrLen, "gcc_nce_backward: bad argument");
        return -1;
    }
    d.dloss = dloss; d.by_mem_row = by_mem_row; d.dq = dq;
    hipStream_t s = (hipStream_t)stream;
    prof_mark(prof, 0, s);
    if (d.bf16) hipLaunchKernelGGL((nce_slice_kernel<true, true>), dim3(pl.S, pl.QB), dim3(kThreads), 0, s, d);
    else hipLaunchKernelGGL((nce_slice_kernel<true, false>), dim3(pl.S, pl.QB), dim3(kThreads), 0, s, d);
    hipLaunchKernelGGL(nce_dq_kernel, dim3((a->B * 16 + kThreads - 1) / kThreads), dim3(kThreads), 0, s, d);
    prof_mark(prof, 1, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_err, kErrLen, "gcc_nce_backward: %s", hipGetErrorString(e)); return -10; }
    return 0;
}

int32_t gcc_queue_enqueue(float *mem, int32_t K, const float *keys, int32_t nkeys, int32_t index, float *saved,
                          void *stream)
{
    if (!mem || !keys || K < 1 || nkeys < 1 || nkeys > K || index < 0 || index >= K) {
        snprintf(g_err, kErrLen, "gcc_queue_enqueue: bad argument (K=%d n=%d index=%d)", K, nkeys, index);
        return -1;
    }
    hipLaunchKernelGGL(queue_enqueue_kernel, dim3((nkeys * 16 + kThreads - 1) / kThreads), dim3(kThreads), 0,
                       (hipStream_t)stream, mem, K, keys, nkeys, index, saved, (const gcc_step_scalars *)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

int32_t gcc_queue_enqueue_scalars(float *mem, int32_t K, const float *keys, int32_t nkeys, const gcc_step_scalars *scalars,
                                  void *stream)
{
    if (!mem || !keys || !scalars || K < 1 || nkeys < 1 || nkeys > K) {
        snprintf(g_err, kErrLen, "gcc_queue_enqueue_scalars: bad argument (K=%d n=%d)", K, nkeys);
        return -1;
    }
    hipLaunchKernelGGL(queue_enqueue_kernel, dim3((nkeys * 16 + kThreads - 1) / kThreads), dim3(kThreads), 0,
                       (hipStream_t)stream, mem, K, keys, nkeys, 0, (float *)nullptr, scalars);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

__global__ void step_scalars_kernel(gcc_step_scalars *dev, gcc_step_scalars v) { *dev = v; }

// one thread: ring entry (*counter mod ring_len) of the host-pinned ring -> the device struct; the entry was written by plain
// host stores before this launch was submitted, and is read word by word with system-scope loads (the ring's lines are
// reused every ring_len steps: nothing may be served from a stale cache line)
__global__ void step_scalars_fetch_kernel(gcc_step_scalars *dev, const gcc_step_scalars *ring, int ring_len,
                                          unsigned long long *counter)
{
    const unsigned long long n = *counter;
    const uint32_t *src = (const uint32_t *)(ring + (n % (unsigned long long)ring_len));
    uint32_t *dst = (uint32_t *)dev;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(gcc_step_scalars) / 4); ++i) dst[i] = load_system_u32(src + i);
    *counter = n + 1;
}

static void fill_scalars(gcc_step_scalars &v, float lr, float beta1, float beta2, int32_t adam_step, int32_t enqueue_index,
                         uint64_t dropout_seed)
{
    v.lr = lr;
    v.bias_corr1 = 1.0f - powf(beta1, (float)adam_step);          // exactly gcc_adam_step's host arithmetic
    v.bias_corr2_sqrt = sqrtf(1.0f - powf(beta2, (float)adam_step));
    v.enqueue_index = enqueue_index;
    v.dropout_seed = dropout_seed;
}

void gcc_step_scalars_fill(gcc_step_scalars *host_entry, float lr, float beta1, float beta2, int32_t adam_step,
                           int32_t enqueue_index, uint64_t dropout_seed)
{
    gcc_step_scalars v;
    fill_scalars(v, lr, beta1, beta2, adam_step, enqueue_index, dropout_seed);
    memcpy(host_entry, &v, sizeof(v));
    __atomic_thread_fence(__ATOMIC_RELEASE);                      // visible before the launch that follows is submitted
}

int32_t gcc_step_scalars_fetch(gcc_step_scalars *dev, const gcc_step_scalars *ring, int32_t ring_len,
                               unsigned long long *counter, void *stream)
{
    if (!dev || !ring || !counter || ring_len < 1) {
        snprintf(g_err, kErrLen, "gcc_step_scalars_fetch: bad argument");
        return -1;
    }
    hipLaunchKernelGGL(step_scalars_fetch_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dev, ring, (int)ring_len, counter);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

int32_t gcc_step_scalars_set(gcc_step_scalars *dev, float lr, float beta1, float beta2, int32_t adam_step,
                             int32_t enqueue_index, uint64_t dropout_seed, void *stream)
{
    if (!dev || adam_step < 1 || enqueue_index < 0) {
        snprintf(g_err, kErrLen, "gcc_step_scalars_set: bad argument");
        return -1;
    }
    gcc_step_scalars v;
    fill_scalars(v, lr, beta1, beta2, adam_step, enqueue_index, dropout_seed);
    hipLaunchKernelGGL(step_scalars_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dev, v);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

int32_t gcc_adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                      float beta1, float beta2, float eps, float weight_decay, int32_t step, float max_norm,
                      float grad_scale, float *grad_norm, double *scratch, void *stream)
{
    if (!param || !grad || !exp_avg || !exp_avg_sq || !grad_norm || !scratch || n < 1 || step < 1 || !(grad_scale > 0.f)) {
        snprintf(g_err, kErrLen, "gcc_adam_step: bad argument");
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(gradnorm_kernel, dim3(kNormBlocks), dim3(kThreads), 0, s, (const float *)grad, n, scratch);
    int blocks = (int)((n + kThreads - 1) / kThreads);
    if (blocks > 512) blocks = 512;
    const AdamExtras none = {};
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(kThreads), 0, s, param, grad, exp_avg, exp_avg_sq, n, lr, beta1,
                       beta2, eps, weight_decay, bc1, bc2_sqrt, max_norm, grad_scale, (const double *)scratch, grad_norm, none);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

static int32_t adam_ema_launch(const char *who, float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int32_t step, float max_norm,
                               float grad_scale, float *grad_norm, double *scratch, float *ema, int64_t n_ema, float ema_m,
                               const gcc_step_meters_args *meters, const gcc_step_scalars *scalars, void *stream)
{
    if (!param || !grad || !exp_avg || !exp_avg_sq || !grad_norm || !scratch || n < 1 || (!scalars && step < 1) || !(grad_scale > 0.f) ||
        (ema && n_ema < n) ||
        (meters && (!meters->acc || !meters->mx || !meters->loss || !meters->prob || !meters->node_off_q ||
                    !meters->edge_off_q || !meters->node_off_k || meters->batch_size < 1))) {
        snprintf(g_err, kErrLen, "%s: bad argument", who);
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    const float bc1 = scalars ? 1.f : 1.0f - powf(beta1, (float)step);
    const float bc2_sqrt = scalars ? 1.f : sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(gradnorm_kernel, dim3(kNormBlocks), dim3(kThreads), 0, s, (const float *)grad, n, scratch);
    AdamExtras x = {};
    x.sc = scalars;
    if (ema) { x.ema = ema; x.n_ema = n_ema; x.ema_m = ema_m; }
    if (meters) {
        x.acc = meters->acc; x.mx = meters->mx; x.loss = meters->loss; x.prob = meters->prob;
        x.node_off_q = meters->node_off_q; x.edge_off_q = meters->edge_off_q; x.node_off_k = meters->node_off_k;
        x.B = meters->batch_size;
    }
    const int64_t nmax = ema ? n_ema : n;
    int blocks = (int)((nmax + kThreads - 1) / kThreads);
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(kThreads), 0, s, param, grad, exp_avg, exp_avg_sq, n, lr, beta1,
                       beta2, eps, weight_decay, bc1, bc2_sqrt, max_norm, grad_scale, (const double *)scratch, grad_norm, x);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

int32_t gcc_adam_ema_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                          float beta1, float beta2, float eps, float weight_decay, int32_t step, float max_norm,
                          float grad_scale, float *grad_norm, double *scratch, float *ema, int64_t n_ema, float ema_m,
                          const gcc_step_meters_args *meters, void *stream)
{
    return adam_ema_launch("gcc_adam_ema_step", param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step,
                           max_norm, grad_scale, grad_norm, scratch, ema, n_ema, ema_m, meters, nullptr, stream);
}

int32_t gcc_adam_ema_step_scalars(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                                  float beta1, float beta2, float eps, float weight_decay, float max_norm,
                                  float grad_scale, float *grad_norm, double *scratch, float *ema, int64_t n_ema, float ema_m,
                                  const gcc_step_meters_args *meters, const gcc_step_scalars *scalars, void *stream)
{
    if (!scalars) { snprintf(g_err, kErrLen, "gcc_adam_ema_step_scalars: scalars is NULL"); return -1; }
    return adam_ema_launch("gcc_adam_ema_step_scalars", param, grad, exp_avg, exp_avg_sq, n, 0.f, beta1, beta2, eps, weight_decay,
                           0, max_norm, grad_scale, grad_norm, scratch, ema, n_ema, ema_m, meters, scalars, stream);
}

int32_t gcc_step_meters(double *acc, int32_t *mx, const float *loss, const float *prob, const float *grad_norm,
                        const int32_t *node_off_q, const int32_t *edge_off_q, const int32_t *node_off_k,
                        int32_t batch_size, void *stream)
{
    if (!acc || !mx || !loss || !prob || !grad_norm || !node_off_q || !edge_off_q || !node_off_k || batch_size < 1) {
        snprintf(g_err, kErrLen, "gcc_step_meters: bad argument");
        return -1;
    }
    hipLaunchKernelGGL(step_meters_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, acc, mx, loss, prob, grad_norm,
                       node_off_q, edge_off_q, node_off_k, batch_size);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

int32_t gcc_ema_update(float *ema, const float *p, int64_t n, float m, void *stream)
{
    if (!ema || !p || n < 1) { snprintf(g_err, kErrLen, "gcc_ema_update: bad argument"); return -1; }
    int blocks = (int)((n + kThreads - 1) / kThreads);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(ema_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, ema, p, n, m);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

}  // extern "C"
