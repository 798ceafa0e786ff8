// gcc_amd/csrc/sampler.hip -- RWR ego-net sampler + on-device batcher (gfx950).
//
// Replaces, for one DataLoader batch, the reference's CPU worker pipeline
//   LoadBalanceGraphDataset.__iter__/__getitem__  gcc/datasets/graph_dataset.py:85-179
//   _rwr_trace_to_dgl_graph (node set, subgraph)  gcc/datasets/data_util.py:218-239
//   batcher / dgl.batch                           gcc/datasets/data_util.py:26-32
// and the DGL C++ routines behind them (random_walk_with_restart, subgraph,
// batch).  RNG / ordering spec: oracle/sampler_oracle.c header; results are
// bit-exact against that oracle.
//
// Kernels (G = 2B subgraphs, g = view * B + b):
//   rwr_walk_kernel   1 wave per subgraph.  Walk lengths depend on the RNG only
//                     (no dead ends by contract), so 64 lanes run 64 walks at a
//                     time and a wave prefix sum over the lengths reproduces the
//                     sequential "stop after exactly L visited nodes" rule
//                     exactly.  Trace in LDS -> bitonic sort -> unique -> seed
//                     first; row extents of every member are fetched here.
//   induce_kernel     work unit = a 256-edge SEGMENT of a member's parent row (4
//                     coalesced dword loads per lane in flight); a subgraph gets
//                     ceil(segments / 16) virtual workgroups (device-side prefix +
//                     binary search), so hub-seed subgraphs with 10x the edges get
//                     10x the workgroups and every wave does <= 4 segments.  LDS hash
//                     map parent id -> local id; hits are ballot-compacted per
//                     segment in parent-row order (DGL VertexSubgraph) into that
//                     segment's scratch slot.  (Round-1 profile: the previous
//                     row-per-wave version was bound by one wave doing 78 iterations
//                     vs 5.5 on average: 2 % of the HBM roof.)
//   pack_kernel       1 workgroup per subgraph: prefix sums over subgraphs
//                     (dgl.batch offsets), row_ptr/col_idx with batched ids,
//                     parent_nid, graph_id.
// HBM-bound integer work; no MFMA anywhere in this file.
#include "device_compat.h"
#include "../../include/gcc_amd.h"

#include "host_common.h"

namespace {

constexpr uint32_t kEmpty = 0xFFFFFFFFu;
constexpr int kSeg = 256;           // parent edges per segment (64 lanes x 4 loads)
constexpr int kSegPerWg = 16;      // segments per virtual workgroup (4 per wave)
constexpr int kInduceThreads = 256;
constexpr int kPackParts = 4;      // pack workgroups per subgraph (hub-seed subgraphs have 10x the rows/edges)

__device__ __forceinline__ int pow2_ceil(int v)
{
    int p = 64;
    while (p < v) p <<= 1;
    return p;
}

// Per-call workspace carved from the caller's buffer.
struct Work {
    int32_t *seeds;       // [B]
    int32_t *sub_n;       // [G]
    int32_t *sub_cap;     // [G]   scratch slots: sum_i row_slots(deg_i, n)
    int32_t *sub_seg;     // [G]   number of row segments
    int32_t *sub_nnz;     // [G]
    int32_t *nodes;       // [G][ncap]   parent ids, seed first
    int32_t *rowbeg;      // [G][ncap]   row_ptr[node]
    int32_t *rowdeg;      // [G][ncap]   parent degree
    int32_t *rowoff;      // [G][ncap]   exclusive prefix of the rows' segment counts within the subgraph (walk kernel)
    int32_t *rowcap;      // [G][ncap]   exclusive prefix of the rows' scratch slots within the subgraph (walk kernel)
    int32_t *rowcnt;      // [G][ncap]   induced degree
    int32_t *vbp;         // [G + 1]     exclusive prefix of the subgraphs' virtual workgroups   (prefix kernel A)
    long long *sbp;       // [G + 1]     exclusive prefix of the subgraphs' scratch slots        (prefix kernel A)
    int32_t *nbp;         // [G + 1]     node offset of a subgraph inside its view's batch       (prefix kernel A)
    int32_t *ebp;         // [G + 1]     edge offset of a subgraph inside its view's batch       (prefix kernel B)
    int32_t *scratch;     // [scratch_entries] local col ids, row-sparse
    int32_t ncap;
};

struct WorkLayout {
    int64_t off_seeds, off_n, off_cap, off_seg, off_nnz, off_nodes, off_rowbeg, off_rowdeg, off_rowoff, off_rowcap,
        off_rowcnt, off_vbp, off_sbp, off_nbp, off_ebp, off_scratch, total;
    int32_t ncap;
};

inline WorkLayout work_layout(int32_t lmax, int32_t B, int64_t scratch_entries)
{
    WorkLayout w;
    auto al = [](int64_t x) { return (x + 255) & ~(int64_t)255; };
    const int64_t G = 2 * (int64_t)B;
    w.ncap = ((lmax + 1 + 63) / 64) * 64;
    int64_t o = 0;
    w.off_seeds = o;  o = al(o + 4 * (int64_t)B);
    w.off_n = o;      o = al(o + 4 * G);
    w.off_cap = o;    o = al(o + 4 * G);
    w.off_seg = o;    o = al(o + 4 * G);
    w.off_nnz = o;    o = al(o + 4 * G);
    w.off_nodes = o;  o = al(o + 4 * G * w.ncap);
    w.off_rowbeg = o; o = al(o + 4 * G * w.ncap);
    w.off_rowdeg = o; o = al(o + 4 * G * w.ncap);
    w.off_rowoff = o; o = al(o + 4 * G * w.ncap);
    w.off_rowcap = o; o = al(o + 4 * G * w.ncap);
    w.off_rowcnt = o; o = al(o + 4 * G * w.ncap);
    w.off_vbp = o;    o = al(o + 4 * (G + 1));
    w.off_sbp = o;    o = al(o + 8 * (G + 1));
    w.off_nbp = o;    o = al(o + 4 * (G + 1));
    w.off_ebp = o;    o = al(o + 4 * (G + 1));
    w.off_scratch = o; o = al(o + 4 * scratch_entries);
    w.total = o;
    return w;
}

struct BatchOutDev {
    int32_t *node_off, *edge_off, *parent_nid, *graph_id, *row_ptr, *col_idx;
    int64_t node_cap, edge_cap;
};

__device__ __forceinline__ int row_slots(int deg, int n);

// ------------------------------------------------------------------ K1 ----
__global__ __launch_bounds__(64) void rwr_walk_kernel(
    const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col_idx,
    const double *__restrict__ seed_cdf, const int32_t *__restrict__ ltab, int64_t num_nodes,
    int32_t ltab_len, int32_t p2max, uint64_t run_seed, int64_t first_sample_id, int32_t B,
    uint32_t restart_u32, const int32_t *__restrict__ seeds_in, Work w)
{
    DYN_SMEM(smem);
    uint32_t *buf = (uint32_t *)smem;        // [p2max] trace -> sorted trace
    int32_t *ldeg = (int32_t *)smem + p2max; // [p2max + 64] parent degree of kept nodes (n <= L + 1)
    const int lane = lane_id();
    const int g = (int)blockIdx.x;
    const int view = g / B, b = g - view * B;
    const uint64_t sid = (uint64_t)(first_sample_id + b);

    // ---- seed: graph_dataset.py:85-92 (p ~ deg^0.75; numpy choice == cdf upper bound)
    int32_t seed;
    if (seeds_in) {
        seed = seeds_in[b];
    } else {
        uint32_t x[4];
        philox4x32_10((uint32_t)sid, (uint32_t)(sid >> 32), 0u, 0u, (uint32_t)run_seed ^ 0x5EED5EEDu,
                      (uint32_t)(run_seed >> 32) ^ 0x00A11CE5u, x);
        const uint64_t u53 = ((uint64_t)x[0] << 21) | (uint64_t)(x[1] >> 11);
        const double u = (double)u53 * (1.0 / 9007199254740992.0);
        // wave-cooperative 64-ary upper bound: first index with cdf[i] > u
        int64_t lo = 0, hi = num_nodes;
        while (lo < hi) {
            const int64_t len = hi - lo;
            const int64_t step = (len + 63) >> 6;
            const int64_t idx = lo + (int64_t)lane * step;
            const bool le = (idx < hi) && (seed_cdf[idx] <= u);
            const int k = __popcll(wave_ballot(le));   // monotone: first k lanes true
            if (k == 0) { hi = lo; break; }
            const int64_t nhi = lo + (int64_t)k * step;
            lo = lo + (int64_t)(k - 1) * step + 1;
            if (nhi < hi) hi = nhi;
        }
        seed = (int32_t)(lo < num_nodes ? lo : num_nodes - 1);
    }
    if (view == 0 && lane == 0) w.seeds[b] = seed;

    const int32_t rp0 = row_ptr[seed];
    const int32_t deg0 = row_ptr[seed + 1] - rp0;
    const int32_t L = ltab[deg0 < ltab_len ? deg0 : ltab_len - 1];   // graph_dataset.py:113-124
    const int p2 = pow2_ceil(L);

    for (int i = lane; i < p2; i += 64) buf[i] = kEmpty;
    wave_sync();

    // ---- walks: graph_dataset.py:125-130 (DGL random_walk_with_restart)
    const uint64_t gid = sid * 2u + (uint64_t)view;
    const uint32_t k0 = (uint32_t)run_seed, k1 = (uint32_t)(run_seed >> 32);
    const uint32_t g0 = (uint32_t)gid, g1 = (uint32_t)(gid >> 32);
    int total = 0;   // wave-uniform: trace entries assigned so far
    for (int base = 0; total < L; base += 256) {
        uint32_t x0[4][4];
        int len[4], off[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t walk = (uint32_t)(base + j * 64 + lane);
            philox4x32_10(walk, 0u, g0, g1, k0, k1, x0[j]);
            // len = min{t >= 1 : word[2t-1] < restart_u32}, capped at L
            int l = 1;
            if (x0[j][1] >= restart_u32) {
                l = 2;
                if (x0[j][3] >= restart_u32) {
                    l = 3;
                    uint32_t y[4];
                    for (uint32_t blk = 1; l < L; ++blk) {
                        philox4x32_10(walk, blk, g0, g1, k0, k1, y);
                        if (y[1] < restart_u32) break;
                        ++l;
                        if (l >= L || y[3] < restart_u32) break;
                        ++l;
                    }
                }
            }
            len[j] = l;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int incl = wave_scan_incl(len[j]);
            off[j] = total + incl - len[j];
            total += wave_shfl(incl, 63);
        }
        // first step of every walk that is (at least partly) inside the budget
        int32_t cur[4];
        int allowed[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int a = L - off[j];
            a = a < 0 ? 0 : (a > len[j] ? len[j] : a);
            allowed[j] = a;
            cur[j] = seed;
            if (a > 0) cur[j] = col_idx[rp0 + (int32_t)__umulhi(x0[j][0], (uint32_t)deg0)];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (allowed[j] > 0) buf[off[j]] = (uint32_t)cur[j];
            if (allowed[j] > 1) {
                const uint32_t walk = (uint32_t)(base + j * 64 + lane);
                uint32_t y[4] = {x0[j][0], x0[j][1], x0[j][2], x0[j][3]};
                uint32_t yblk = 0;
                int32_t c = cur[j];
                for (int t = 1; t < allowed[j]; ++t) {
                    const uint32_t blk = (uint32_t)t >> 1;      // word 2t lives in block t/2
                    if (blk != yblk) { philox4x32_10(walk, blk, g0, g1, k0, k1, y); yblk = blk; }
                    const uint32_t r = (t & 1) ? y[2] : y[0];
                    const int32_t beg = row_ptr[c];
                    const int32_t d = row_ptr[c + 1] - beg;
                    c = col_idx[beg + (int32_t)__umulhi(r, (uint32_t)d)];
                    buf[off[j] + t] = (uint32_t)c;
                }
            }
        }
    }
    wave_sync();

    // ---- torch.unique (data_util.py:221): bitonic sort of the padded trace
    for (int k = 2; k <= p2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < (p2 >> 1); i += 64) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int hi = lo + j;
                const bool up = (lo & k) == 0;
                const uint32_t a = buf[lo], c = buf[hi];
                if ((a > c) == up) { buf[lo] = c; buf[hi] = a; }
            }
            wave_sync();
        }
    }

    // ---- subv = [seed] + (unique \ {seed})  (data_util.py:222-226); fetch row extents
    int32_t *nodes = w.nodes + (int64_t)g * w.ncap;
    int32_t *rowbeg = w.rowbeg + (int64_t)g * w.ncap;
    int32_t *rowdeg = w.rowdeg + (int64_t)g * w.ncap;
    if (lane == 0) { nodes[0] = seed; rowbeg[0] = rp0; rowdeg[0] = deg0; ldeg[0] = deg0; }
    int n = 1;   // wave-uniform
    for (int i0 = 0; i0 < L; i0 += 64) {
        const int i = i0 + lane;
        uint32_t v = kEmpty, prev = kEmpty;
        if (i < L) { v = buf[i]; if (i > 0) prev = buf[i - 1]; }
        const bool keep = (i < L) && (v != (uint32_t)seed) && (i == 0 || v != prev);
        const unsigned long long m = wave_ballot(keep);
        if (keep) {
            const int pos = n + __popcll(m & lanemask_lt());
            const int32_t rb = row_ptr[v];
            const int32_t d = row_ptr[v + 1] - rb;
            nodes[pos] = (int32_t)v;
            rowbeg[pos] = rb;
            rowdeg[pos] = d;
            ldeg[pos] = d;
        }
        n += __popcll(m);
    }
    wave_sync();
    // induction work units: row i contributes ceil(deg_i / kSeg) segments and row_slots() scratch slots
    // (the per-row exclusive prefixes are kept: induce_kernel and pack_kernel would otherwise redo these scans in
    // every workgroup)
    int run = 0, slots = 0;
    int32_t *rowoff = w.rowoff + (int64_t)g * w.ncap, *rowcap = w.rowcap + (int64_t)g * w.ncap;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const int c = i < n ? (ldeg[i] + kSeg - 1) / kSeg : 0;
        const int sl = i < n ? row_slots(ldeg[i], n) : 0;
        const int ci = wave_scan_incl(c), si = wave_scan_incl(sl);
        if (i < n) {
            rowoff[i] = run + ci - c;
            rowcap[i] = slots + si - sl;
        }
        run += wave_shfl(ci, 63);
        slots += wave_shfl(si, 63);
    }
    if (lane == 0) {
        w.sub_n[g] = n;
        w.sub_seg[g] = run;
        w.sub_cap[g] = slots;
        w.sub_nnz[g] = 0;
    }
}

// sum of arr[begin, end) by the whole workgroup (all threads get the result)
__device__ __forceinline__ long long block_range_sum(const int32_t *arr, int begin, int end, long long *red)
{
    long long s = 0;
    for (int i = begin + (int)threadIdx.x; i < end; i += (int)blockDim.x) s += arr[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int d = (int)blockDim.x >> 1; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    const long long r = red[0];
    __syncthreads();
    return r;
}

// scratch slots of one member row: every segment = 1 count header + at most min(segment length, n) hits
__device__ __forceinline__ int row_slots(int deg, int n)
{
    const int nseg = (deg + kSeg - 1) / kSeg;
    const int full = n < kSeg ? n : kSeg;
    const int rem = deg - (nseg - 1) * kSeg;
    return nseg + (nseg - 1) * full + (rem < n ? rem : n);
}

// block-wide exclusive scan of vals over [0, count) into LDS out[0..count] (out[count] = total);
// f(i) supplies the i-th value.  All threads call.
template <class T, class F>
__device__ __forceinline__ void block_exclusive_scan(T *out, int count, F f, T *wsum /* [5] */)
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) wsum[4] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < count; i0 += (int)blockDim.x) {
        const int i = i0 + tid;
        const T v = i < count ? f(i) : (T)0;
        T incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const T t = wave_shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        T base = wsum[4];
        for (int k = 0; k < wv; ++k) base += wsum[k];
        if (i < count) out[i] = base + incl - v;
        __syncthreads();
        if (tid == 0) wsum[4] += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (tid == 0) out[count] = wsum[4];
    __syncthreads();
}

// last index i in [0, count) with arr[i] <= key (arr ascending, arr[0] <= key)
__device__ __forceinline__ int upper_slot(const int32_t *arr, int count, int key)
{
    int lo = 0, hi = count;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (arr[mid] <= key) lo = mid; else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------------ K1b / K2b ----
// One workgroup: exclusive prefixes over the G = 2 B subgraphs that every workgroup of induce_kernel / pack_kernel
// needs (they used to recompute them: 19 % of induce_kernel's time).  kAfterInduce = false: virtual workgroups,
// scratch slots, node offsets (per view); true: edge offsets (per view; the induced edge counts exist only then).
template <bool kAfterInduce>
__global__ __launch_bounds__(256) void subgraph_prefix_kernel(int32_t B, Work w)
{
    DYN_SMEM(smem);
    __shared__ long long wsum64[5];
    __shared__ int32_t wsum32[5];
    const int G = 2 * B, tid = (int)threadIdx.x;
    int32_t *tmp = (int32_t *)smem;                  // [G + 1]
    if (!kAfterInduce) {
        long long *tmp64 = (long long *)(tmp + ((G + 2) & ~1));
        block_exclusive_scan<int32_t>(tmp, G, [&](int g) { return (w.sub_seg[g] + kSegPerWg - 1) / kSegPerWg; }, wsum32);
        for (int g = tid; g <= G; g += 256) w.vbp[g] = tmp[g];
        __syncthreads();
        block_exclusive_scan<long long>(tmp64, G, [&](int g) { return (long long)w.sub_cap[g]; }, wsum64);
        for (int g = tid; g <= G; g += 256) w.sbp[g] = tmp64[g];
        __syncthreads();
        block_exclusive_scan<int32_t>(tmp, G, [&](int g) { return w.sub_n[g]; }, wsum32);
        for (int g = tid; g <= G; g += 256) w.nbp[g] = tmp[g] - (g >= B && g < G ? tmp[B] : 0);   // restart at view k
    } else {
        block_exclusive_scan<int32_t>(tmp, G, [&](int g) { return w.sub_nnz[g]; }, wsum32);
        for (int g = tid; g <= G; g += 256) w.ebp[g] = tmp[g] - (g >= B && g < G ? tmp[B] : 0);
    }
}

// ------------------------------------------------------------------ K2 ----
static long long *g_induce_ticks = nullptr;      // diagnostics (gcc_sampler_debug_ticks): [0..3] phase ticks, [15] workgroups
#define IND_TICK(ph) do { if (ticks && tid == 0) { const long long now_ = device_ticks(); atomicAdd((unsigned long long *)&ticks[ph], (unsigned long long)(now_ - tick_)); tick_ = now_; } } while (0)
__global__ __launch_bounds__(kInduceThreads) void induce_kernel(
    const int32_t *__restrict__ col_idx, int32_t hcap_log2, int32_t G, int64_t scratch_entries, Work w,
    int32_t *__restrict__ status, long long *ticks)
{
    DYN_SMEM(smem);
    const int hcap = 1 << hcap_log2;
    uint32_t *hkey = (uint32_t *)smem;                       // [hcap]
    uint16_t *hval = (uint16_t *)(hkey + hcap);              // [hcap]
    int32_t *segoff = (int32_t *)(hval + hcap);              // [ncap + 1] exclusive prefix of segments per row
    int32_t *capoff = segoff + (w.ncap + 1);                 // [ncap + 1] exclusive prefix of scratch slots per row
    int32_t *vbp = capoff + (w.ncap + 1);                    // [G + 1]   exclusive prefix of virtual blocks
    long long *sbp = (long long *)(vbp + ((G + 2) & ~1));    // [G + 1]   exclusive prefix of scratch slots
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = tid >> 6;
    long long tick_ = ticks ? device_ticks() : 0;
    if (ticks && tid == 0) atomicAdd((unsigned long long *)&ticks[15], 1ull);
    for (int g = tid; g <= G; g += kInduceThreads) { vbp[g] = w.vbp[g]; sbp[g] = w.sbp[g]; }   // (subgraph_prefix_kernel)
    __syncthreads();
    IND_TICK(0);
    const int total_vb = vbp[G];
    const int shift = 32 - hcap_log2;
    int cur_g = -1;
    for (int vb = (int)blockIdx.x; vb < total_vb; vb += (int)gridDim.x) {
        const int g = upper_slot(vbp, G, vb);
        const int part = vb - vbp[g];
        const int n = w.sub_n[g], totseg = w.sub_seg[g];
        const long long sbase = sbp[g];
        if (sbase + (long long)w.sub_cap[g] > scratch_entries) {
            if (tid == 0) atomicOr(status, (int32_t)GCC_STATUS_SCRATCH_OVERFLOW);
            continue;
        }
        const int32_t *nodes = w.nodes + (int64_t)g * w.ncap;
        const int32_t *rowbeg = w.rowbeg + (int64_t)g * w.ncap;
        const int32_t *rowdeg = w.rowdeg + (int64_t)g * w.ncap;
        if (g != cur_g) {                                    // block-uniform
            __syncthreads();
            for (int i = tid; i < hcap; i += kInduceThreads) hkey[i] = kEmpty;
            __syncthreads();
            for (int i = tid; i < n; i += kInduceThreads) {
                const uint32_t key = (uint32_t)nodes[i];
                uint32_t h = (key * 0x9E3779B1u) >> shift;
                for (;;) {
                    const uint32_t old = atomicCAS(&hkey[h], kEmpty, key);
                    if (old == kEmpty) { hval[h] = (uint16_t)i; break; }
                    h = (h + 1) & (uint32_t)(hcap - 1);
                }
            }
            {                                                // per-row prefixes: written by the walk kernel
                const int32_t *ro = w.rowoff + (int64_t)g * w.ncap, *rc = w.rowcap + (int64_t)g * w.ncap;
                for (int i = tid; i <= n; i += kInduceThreads) {
                    segoff[i] = i < n ? ro[i] : totseg;
                    capoff[i] = i < n ? rc[i] : w.sub_cap[g];
                }
                __syncthreads();
            }
            cur_g = g;
            IND_TICK(1);
        }
        const int stride = 1 + (n < kSeg ? n : kSeg);
        int my_nnz = 0;
#pragma unroll 1
        for (int k = 0; k < kSegPerWg / 4; ++k) {
            const int s = part * kSegPerWg + wave * (kSegPerWg / 4) + k;
            if (s >= totseg) break;                          // wave-uniform
            const int i = upper_slot(segoff, n, s);
            const int e0 = (s - segoff[i]) * kSeg;
            const int32_t beg = rowbeg[i] + e0;
            const int len = min(kSeg, rowdeg[i] - e0);
            int32_t *out = w.scratch + sbase + capoff[i] + (s - segoff[i]) * stride;
            uint32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = u * 64 + lane;
                v[u] = e < len ? (uint32_t)col_idx[beg + e] : kEmpty;
            }
            int cnt = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int loc = -1;
                if (v[u] != kEmpty) {
                    uint32_t h = (v[u] * 0x9E3779B1u) >> shift;
                    for (;;) {
                        const uint32_t key = hkey[h];
                        if (key == v[u]) { loc = (int)hval[h]; break; }
                        if (key == kEmpty) break;
                        h = (h + 1) & (uint32_t)(hcap - 1);
                    }
                }
                const unsigned long long m = wave_ballot(loc >= 0);
                if (loc >= 0) out[1 + cnt + __popcll(m & lanemask_lt())] = loc;
                cnt += __popcll(m);
            }
            if (lane == 0) out[0] = cnt;
            my_nnz += cnt;
        }
        if (lane == 0 && my_nnz) atomicAdd(&w.sub_nnz[g], my_nnz);
        __syncthreads();
        IND_TICK(2);
    }
}

// ------------------------------------------------------------------ K3 ----
__global__ __launch_bounds__(256) void pack_kernel(int32_t B, Work w, BatchOutDev oq, BatchOutDev ok,
                                                    int64_t scratch_entries, int32_t *__restrict__ status)
{
    DYN_SMEM(smem);
    __shared__ int32_t wsum32[5];
    int32_t *segoff = (int32_t *)smem;            // [ncap + 1] exclusive prefix of segments per row
    int32_t *capoff = segoff + (w.ncap + 1);      // [ncap + 1] exclusive prefix of scratch slots per row
    int32_t *excl = capoff + (w.ncap + 1);        // [ncap + 1] exclusive prefix of induced degrees
    const int tid = (int)threadIdx.x;
    const int g = (int)blockIdx.x / kPackParts, part = (int)blockIdx.x % kPackParts;
    const int view = g / B, b = g - view * B;
    const BatchOutDev o = view ? ok : oq;
    const int n = w.sub_n[g];
    const int nnz = w.sub_nnz[g];
    const long long node_base = w.nbp[g];            // (subgraph_prefix_kernel<false / true>)
    const long long edge_base = w.ebp[g];
    const long long sbase = w.sbp[g];
    if (tid == 0 && part == 0) {
        o.node_off[b] = (int32_t)node_base;
        o.edge_off[b] = (int32_t)edge_base;
        if (b == B - 1) {
            o.node_off[B] = (int32_t)(node_base + n);
            o.edge_off[B] = (int32_t)(edge_base + nnz);
        }
    }
    const bool bad_scratch = sbase + (long long)w.sub_cap[g] > scratch_entries;
    const bool bad_nodes = node_base + n > o.node_cap;
    const bool bad_edges = edge_base + nnz > o.edge_cap;
    const int32_t *nodes = w.nodes + (int64_t)g * w.ncap;
    if (bad_scratch || bad_nodes || bad_edges) {
        if (tid == 0 && part == 0)
            atomicOr(status, (int32_t)((bad_scratch ? GCC_STATUS_SCRATCH_OVERFLOW : 0) |
                                       (bad_nodes ? GCC_STATUS_NODE_OVERFLOW : 0) |
                                       (bad_edges ? GCC_STATUS_EDGE_OVERFLOW : 0)));
        // leave a VALID structure behind (rows without edges, offsets clamped to the capacity) so that a
        // consumer that has not looked at `status` yet can never index out of bounds
        const long long ecl = edge_base < o.edge_cap ? edge_base : o.edge_cap;
        for (int i = part * 256 + tid; i < n; i += 256 * kPackParts) {
            if (node_base + i < o.node_cap) {
                o.parent_nid[node_base + i] = nodes[i];
                o.graph_id[node_base + i] = b;
                o.row_ptr[node_base + i] = (int32_t)ecl;
            }
        }
        if (b == B - 1 && tid == 0 && part == 0 && node_base + n <= o.node_cap) o.row_ptr[node_base + n] = (int32_t)ecl;
        return;
    }
    const int32_t *scratch = w.scratch + sbase;
    const int stride = 1 + (n < kSeg ? n : kSeg);

    {                                                        // per-row prefixes: written by the walk kernel
        const int32_t *ro = w.rowoff + (int64_t)g * w.ncap, *rc = w.rowcap + (int64_t)g * w.ncap;
        for (int i = tid; i <= n; i += 256) {
            segoff[i] = i < n ? ro[i] : w.sub_seg[g];
            capoff[i] = i < n ? rc[i] : w.sub_cap[g];
        }
        __syncthreads();
    }
    // induced degree of row i = sum of its segments' hit counts
    block_exclusive_scan<int32_t>(excl, n, [&](int i) {
        int c = 0;
        for (int k = 0; k < segoff[i + 1] - segoff[i]; ++k) c += scratch[capoff[i] + k * stride];
        return c;
    }, wsum32);

    for (int i = part * 256 + tid; i < n; i += 256 * kPackParts) {
        o.parent_nid[node_base + i] = nodes[i];
        o.graph_id[node_base + i] = b;
        o.row_ptr[node_base + i] = (int32_t)(edge_base + excl[i]);
    }
    if (b == B - 1 && tid == 0 && part == 0) o.row_ptr[node_base + n] = (int32_t)(edge_base + nnz);
    // one thread per output edge: its row by binary search, then its segment inside the row
    for (int e = part * 256 + tid; e < nnz; e += 256 * kPackParts) {
        const int i = upper_slot(excl, n, e);
        int off = e - excl[i];
        int at = capoff[i];
        for (;;) {
            const int c = scratch[at];
            if (off < c) break;
            off -= c;
            at += stride;
        }
        o.col_idx[edge_base + e] = (int32_t)node_base + scratch[at + 1 + off];
    }
}

}  // namespace

extern "C" {

void gcc_sampler_debug_ticks(long long *device_ticks64) { g_induce_ticks = device_ticks64; }   /* diagnostics only */


int64_t gcc_sampler_workspace_bytes(const gcc_graph *g, int32_t batch_size, int64_t scratch_entries)
{
    if (!g || batch_size <= 0 || scratch_entries <= 0 || g->lmax <= 0) {
        snprintf(g_err, kErrLen, "gcc_sampler_workspace_bytes: bad argument");
        return -1;
    }
    return work_layout(g->lmax, batch_size, scratch_entries).total;
}

int32_t gcc_sample_batch(const gcc_graph *g, const gcc_sample_params *p, const gcc_batch_out *out_q,
                         const gcc_batch_out *out_k, void *workspace, int64_t workspace_bytes,
                         int64_t scratch_entries, int32_t *status, void *stream)
{
    if (!g || !p || !out_q || !out_k || !workspace || !status) {
        snprintf(g_err, kErrLen, "gcc_sample_batch: null argument");
        return -1;
    }
    if (p->batch_size <= 0 || g->lmax <= 0 || g->lmax > 65534 || g->num_nodes <= 0 ||
        g->num_nodes > 0x7FFFFFFF || g->num_edges > 0x7FFFFFFF) {
        snprintf(g_err, kErrLen, "gcc_sample_batch: size out of range (B=%d lmax=%d V=%lld E=%lld)",
                 p->batch_size, g->lmax, (long long)g->num_nodes, (long long)g->num_edges);
        return -2;
    }
    const WorkLayout wl = work_layout(g->lmax, p->batch_size, scratch_entries);
    if (workspace_bytes < wl.total) {
        snprintf(g_err, kErrLen, "gcc_sample_batch: workspace %lld < %lld bytes",
                 (long long)workspace_bytes, (long long)wl.total);
        return -3;
    }
    char *base = (char *)workspace;
    Work w;
    w.seeds = (int32_t *)(base + wl.off_seeds);
    w.sub_n = (int32_t *)(base + wl.off_n);
    w.sub_cap = (int32_t *)(base + wl.off_cap);
    w.sub_seg = (int32_t *)(base + wl.off_seg);
    w.sub_nnz = (int32_t *)(base + wl.off_nnz);
    w.nodes = (int32_t *)(base + wl.off_nodes);
    w.rowbeg = (int32_t *)(base + wl.off_rowbeg);
    w.rowdeg = (int32_t *)(base + wl.off_rowdeg);
    w.rowoff = (int32_t *)(base + wl.off_rowoff);
    w.rowcap = (int32_t *)(base + wl.off_rowcap);
    w.rowcnt = (int32_t *)(base + wl.off_rowcnt);
    w.vbp = (int32_t *)(base + wl.off_vbp);
    w.sbp = (long long *)(base + wl.off_sbp);
    w.nbp = (int32_t *)(base + wl.off_nbp);
    w.ebp = (int32_t *)(base + wl.off_ebp);
    w.scratch = (int32_t *)(base + wl.off_scratch);
    w.ncap = wl.ncap;

    const int B = p->batch_size, G = 2 * B;
    hipStream_t s = (hipStream_t)stream;
    int p2max = 64;
    while (p2max < g->lmax) p2max <<= 1;
    int hlog = 7;
    while ((1 << hlog) < 2 * (g->lmax + 1)) ++hlog;
    const size_t lds1 = ((size_t)p2max * 2 + 64) * 4;
    const size_t lds2 = (size_t)(1 << hlog) * 6 + (size_t)(wl.ncap + 1) * 8 + (size_t)(G + 2) * 4 + (size_t)(G + 1) * 8 + 16;
    const size_t lds3 = (size_t)(wl.ncap + 1) * 12;
    if (lds1 > 160 * 1024 || lds2 > 160 * 1024 || lds3 > 160 * 1024) {
        snprintf(g_err, kErrLen, "gcc_sample_batch: lmax=%d / batch too large for 160 KiB of LDS", g->lmax);
        return -4;
    }
    BatchOutDev oq = {out_q->node_off, out_q->edge_off, out_q->parent_nid, out_q->graph_id,
                      out_q->row_ptr, out_q->col_idx, out_q->node_cap, out_q->edge_cap};
    BatchOutDev ok = {out_k->node_off, out_k->edge_off, out_k->parent_nid, out_k->graph_id,
                      out_k->row_ptr, out_k->col_idx, out_k->node_cap, out_k->edge_cap};

    prof_mark(p->prof, 0, s);
#ifndef GCC_AMD_HIPEMU
    // more than 64 KiB of dynamic LDS has to be opted into per kernel
    if (lds1 > 64 * 1024) (void)hipFuncSetAttribute((const void *)rwr_walk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    if (lds2 > 64 * 1024) (void)hipFuncSetAttribute((const void *)induce_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    if (lds3 > 64 * 1024) (void)hipFuncSetAttribute((const void *)pack_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
#endif
    hipLaunchKernelGGL(rwr_walk_kernel, dim3(G), dim3(64), lds1, s, g->row_ptr, g->col_idx, g->seed_cdf,
                       g->ltab, g->num_nodes, g->ltab_len, p2max, p->run_seed, p->first_sample_id, B,
                       p->restart_u32, p->seeds, w);
    prof_mark(p->prof, 1, s);
    const size_t lds_pref = (size_t)(G + 2) * 4 + (size_t)(G + 1) * 8 + 16;
    hipLaunchKernelGGL((subgraph_prefix_kernel<false>), dim3(1), dim3(256), lds_pref, s, B, w);
    hipLaunchKernelGGL(induce_kernel, dim3(G * 8), dim3(kInduceThreads), lds2, s, g->col_idx, hlog, G,
                       scratch_entries, w, status, g_induce_ticks);
    prof_mark(p->prof, 2, s);
    hipLaunchKernelGGL((subgraph_prefix_kernel<true>), dim3(1), dim3(256), lds_pref, s, B, w);
    hipLaunchKernelGGL(pack_kernel, dim3(G * kPackParts), dim3(256), lds3, s, B, w, oq, ok, scratch_entries, status);
    prof_mark(p->prof, 3, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, kErrLen, "gcc_sample_batch: launch failed: %s", hipGetErrorString(e));
        return -10;
    }
    return 0;
}

}  // extern "C"
