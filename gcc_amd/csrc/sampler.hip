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
// Kernels (G = 2 B S subgraphs of S consecutive steps, g = segment * B + b, segment = 2 * step + view):
//   rwr_walk_kernel   1 workgroup of 4 waves per subgraph (seeds whose trace budget exceeds
//                     1024 entries: 16 waves, a second launch).  Walk lengths depend on
//                     the RNG only (no dead ends by contract), so 256 threads run
//                     256 walks at a time and a block prefix sum over the lengths
//                     reproduces the sequential "stop after exactly L visited
//                     nodes" rule exactly.  Trace in LDS -> bitonic sort -> unique
//                     -> seed first; row extents of every member and the prefix of
//                     the rows' 16-byte QUADS of col_idx are written here.
//   induce_kernel     DGL VertexSubgraph = scan the members' parent rows for members
//                     (all but the subgraph's <= 32 longest: "Hub rows" below; one
//                     launch per size class; a sub-scan by binary search would touch the same
//                     cache lines: a subgraph has more members than a hub row has
//                     128-byte lines).  The rows of a subgraph form one flat space
//                     of aligned quads; a UNIT = 256 consecutive quads (1024 edges)
//                     = one wave x 4 coalesced dwordx4 loads per lane; a virtual
//                     workgroup = 2 units per wave (small class: 4 waves, big: 8).  The row of every quad of
//                     a unit comes from a 256-entry LDS strip (rows mark their first
//                     quad, a prefix maximum spreads the marks), not from a search
//                     per quad.  Members sit in an LDS Bloom bitmap (>= 64 bits per
//                     member): 98.5 % of the neighbours are non-members and cost one
//                     LDS word; survivors are queued per wave and looked up exactly
//                     (binary search in the sorted member list) in dense passes;
//                     hits go, in parent order, to the unit's private scratch slot.
//                     No atomics on shared counters inside the scan.  The kernel is
//                     VALU-bound: everything wave-uniform is
//                     kept in scalar registers on purpose.
//   prefix_a_kernel   one workgroup between the two: prefixes over the subgraphs;
//   records_kernel    the start record of every induce workgroup.
//   pack_kernel       prefix sums over subgraphs and units (dgl.batch offsets),
//                     row_ptr/col_idx with batched ids, parent_nid, graph_id.
//   hub_write_kernel  the rows of the members the induction did not scan (below).
// HBM-bound integer work; no MFMA anywhere in this file.
#include "device_compat.h"
#include "../../include/gcc_amd.h"

#include "host_common.h"

namespace {

constexpr uint32_t kEmpty = 0xFFFFFFFFu;
constexpr int kWalkThreads = 256;  // 4 waves per subgraph
constexpr int kUnitQuads = 256;    // 16-byte quads of col_idx per unit: 64 lanes x 4 dwordx4 loads
constexpr int kUnitElems = 4 * kUnitQuads;
// induce workgroups: the small class (subgraphs of at most kSmallMembers members) and the big class (induce_kernel).  A
// virtual workgroup is 2 units per wave; the static grids are kGridMult workgroups per subgraph / kBigGrid workgroups.
// Measured after hub rows stopped being scanned (profiles/r4_sampler_classes.md): 256 threads and 4 workgroups per
// subgraph for the small class (most of its subgraphs have one to four units left), 512 threads for the big one.
#ifndef GCC_INDUCE_THREADS
#define GCC_INDUCE_THREADS 256
#endif
#ifndef GCC_INDUCE_BIG_THREADS
#define GCC_INDUCE_BIG_THREADS 512
#endif
#ifndef GCC_INDUCE_GRID_MULT
#define GCC_INDUCE_GRID_MULT 4
#endif
#ifndef GCC_INDUCE_OCC
#define GCC_INDUCE_OCC
#endif
constexpr int kInduceThreads = GCC_INDUCE_THREADS;
constexpr int kInduceBigThreads = GCC_INDUCE_BIG_THREADS;
constexpr int kGridMult = GCC_INDUCE_GRID_MULT;   // induce workgroups per subgraph (a static grid; see induce_kernel)
__host__ __device__ constexpr int vwg_units(int threads) { return threads / 32; }    // units per virtual workgroup (2 per wave, one after the other)
#ifndef GCC_SMALL_MEMBERS
#define GCC_SMALL_MEMBERS 320
#endif
constexpr int kSmallMembers = GCC_SMALL_MEMBERS; // LDS tables of the small induce class (a multiple of 64)
#ifndef GCC_INDUCE_BIG_GRID
#define GCC_INDUCE_BIG_GRID 1024
#endif
constexpr int kBigGrid = GCC_INDUCE_BIG_GRID;     // induce workgroups of the big class (subgraphs with more members than the small class has LDS for)
constexpr int kRecInts = 12;       // per induce workgroup: {count, g, part, n, quads, unit base, scratch base (2), scanned rows, hubs, pad}
constexpr int kPrefixThreads = 1024;
constexpr int kCandCap = 256;      // per-wave queue of Bloom survivors (drained before every round of 256 that might not fit)
constexpr int kPackParts = 2;      // pack workgroups per subgraph.  4 while every row was scanned (hub-seed subgraphs had 100x the units); with the
                                                // hub rows out, measured per 16-step launch G1 / G2: 4 parts 0.606 / 1.486 ms, 2 parts 0.556 / 1.458, 1 part
                                                // 0.546 / 1.520; one part for small subgraphs and four for big ones with the idle parts leaving at once: 0.599 / 1.476
// Hub rows are NOT scanned (round 4).  The parent graph is symmetric (the input contract, x2dgl.py:43-47): member v's row
// holds hub H exactly when H's row holds v, so every edge (H -> v) of the induced subgraph is the mirror image of a hit
// (v -> H) found while scanning v's own -- short -- row (the induction marks it in H's neighbour bitmap), and edges
// between two hubs are found by one binary search per pair (tail of the walk kernel).  hub_write_kernel turns the bitmaps
// into rows after the pack.  The result is bit for bit the scanned one (same tests, same oracle); which rows are hubs
// only changes the cost.  On the bench graphs the rows of degree >= 1024 (G1: 10 per ego-net) / >= 4096 (G2: 14) hold
// 72 % of the entries a full scan reads.  Measured (profiles/r4_hub_rows.md, r4_sampler_classes.md): induction 437 ->
// 232 us (G1) and 1826 -> 1105 us (G2) per 16-step launch before the size classes, the pair searches and the row writer
// give 90 / 170 us back; the optimum over (threshold, slots) is flat around 512 .. 1024 x 32 (64 slots: no gain).
constexpr int kMaxHub = 32;   // (<= 64) most hub rows per subgraph (the workspace is laid out for this many)
constexpr int kHubDegreeDefault = 512;
constexpr int kMaxHubsDefault = 32; // with more rows over the threshold, the threshold of THAT subgraph rises to the power of two that
                                   // leaves at most this many: the pair searches grow with the square of the count, the bytes saved come
                                   // from the longest rows
constexpr uint32_t kHashMul = 0x9E3779u;    // 24-bit multiply (full rate; the 32-bit one is quarter rate): ids differing
                                            // only above bit 23 share a Bloom bit, which costs a look-up, not a result

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
    int32_t *sub_quads;   // [G]   aligned quads of col_idx covered by the member rows
    int32_t *sub_nnz;     // [G]
    int32_t *nodes;       // [G][ncap]   parent ids, seed first
    int32_t *rowbeg;      // [G][ncap]   row_ptr[node]
    int32_t *rowdeg;      // [G][ncap]   parent degree
    int32_t *rowq;        // [G][ncap]   exclusive prefix of the rows' quads within the subgraph (walk kernel)
    int32_t *vbp;         // [G + 1]     exclusive prefix of the subgraphs' virtual workgroups   (prefix kernel A)
    int32_t *ubp;         // [G + 1]     exclusive prefix of the subgraphs' units                (prefix kernel A)
    long long *sbp;       // [G + 1]     exclusive prefix of the subgraphs' scratch slots        (prefix kernel A)
    int32_t *nbp;         // [G + 1]     node offset of a subgraph inside its view's batch       (prefix kernel A)
    int32_t *ebp;         // [G + 1]     edge offset of a subgraph inside its view's batch       (prefix kernel B)
    int32_t *ucnt;        // [unit_cap]  hits of every unit
    int32_t *wrec;        // [G * kGridMult + kBigGrid][kRecInts] where each induce workgroup starts (prefix step A); the big class's after the small one's
    int32_t *vbpb;        // [G + 1]     the same prefix for the big class (subgraphs with more than lcap members; vbp counts the others)
    int32_t lcap;         // members a small-class induce workgroup has LDS tables for (>= ncap: one class)
    int32_t nsmall, nbig; // induce workgroups of the small / the big class (G * kGridMult, kBigGrid; fewer under gcc_sampler_debug_grids)
    int32_t *scratch;     // [scratch_entries] hits: (row << 16) | local column, one slot of 1024 per unit
    int32_t *srow;        // [G][ncap]   local id of the s-th SCANNED row (rowbeg / rowdeg / rowq are indexed by s, not by local id)
    int32_t *sub_ns;      // [G]         scanned rows
    int32_t *sub_nh;      // [G]         hub rows (not scanned)
    int32_t *big;         // [1 + G]     count, then the subgraphs the small walk launch left to the big one
    int32_t *sub_p0;      // [G]         members (other than the seed) whose parent id is below the seed's
    int32_t *hubloc;      // [G][kMaxHub] local id of hub k (ascending)
    int32_t *hubrb;       // [G][kMaxHub] its parent row's begin
    int32_t *hubdeg;      // [G][kMaxHub] and degree
    int32_t *hubcnt;      // [G][kMaxHub] entries of its induced row (counted by the walk kernel: hub pairs, and the induction: everything else)
    uint32_t *hubmark;    // [G][kMaxHub][mwords] bit i: member with local id i is a neighbour of the hub
    int32_t mwords;       // ncap / 32
    int32_t ncap;
    int32_t nseg;         // batch segments of this call: 2 (views q, k) per step; subgraph g = segment * B + b
    int64_t unit_cap;
    // adjacency among the parent's high-degree rows (gcc_graph.hub_index / hub_adj), or NULL: hub pairs are searched
    const int32_t *hub_index;
    const uint32_t *hub_adj;
    int32_t hub_words;
};

struct WorkLayout {
    int64_t off_seeds, off_n, off_quads, off_nnz, off_nodes, off_rowbeg, off_rowdeg, off_rowq,
        off_vbp, off_ubp, off_sbp, off_nbp, off_ebp, off_ucnt, off_wrec, off_scratch, total, unit_cap;
    int64_t off_srow, off_ns, off_nh, off_p0, off_big, off_vbpb, off_hubloc, off_hubrb, off_hubdeg, off_hubcnt, off_hubmark;
    int32_t ncap;
};

inline WorkLayout work_layout(int32_t lmax, int32_t B, int32_t nseg, int64_t scratch_entries)
{
    WorkLayout w;
    auto al = [](int64_t x) { return (x + 255) & ~(int64_t)255; };
    const int64_t G = (int64_t)nseg * B;
    w.ncap = ((lmax + 1 + 63) / 64) * 64;
    w.unit_cap = scratch_entries / kUnitElems + G + 1;    // a unit owns up to 1024 scratch slots; the last unit of a subgraph fewer
    int64_t o = 0;
    w.off_seeds = o;  o = al(o + 4 * (int64_t)B * (nseg / 2));
    w.off_n = o;      o = al(o + 4 * G);
    w.off_quads = o;  o = al(o + 4 * G);
    w.off_nnz = o;    o = al(o + 4 * G);
    w.off_nodes = o;  o = al(o + 4 * G * w.ncap);
    w.off_rowbeg = o; o = al(o + 4 * G * w.ncap);
    w.off_rowdeg = o; o = al(o + 4 * G * w.ncap);
    w.off_rowq = o;   o = al(o + 4 * G * w.ncap);
    w.off_vbp = o;    o = al(o + 4 * (G + 1));
    w.off_ubp = o;    o = al(o + 4 * (G + 1));
    w.off_sbp = o;    o = al(o + 8 * (G + 1));
    w.off_nbp = o;    o = al(o + 4 * (G + 1));
    w.off_ebp = o;    o = al(o + 4 * (G + 1));
    w.off_ucnt = o;   o = al(o + 4 * w.unit_cap);
    w.off_wrec = o;   o = al(o + 4 * (G * kGridMult + kBigGrid) * kRecInts);
    w.off_vbpb = o;   o = al(o + 4 * (G + 1));
    w.off_srow = o;   o = al(o + 4 * G * w.ncap);
    w.off_ns = o;     o = al(o + 4 * G);
    w.off_nh = o;     o = al(o + 4 * G);
    w.off_p0 = o;     o = al(o + 4 * G);
    w.off_big = o;    o = al(o + 4 * (G + 1));
    w.off_hubloc = o; o = al(o + 4 * G * kMaxHub);
    w.off_hubrb = o;  o = al(o + 4 * G * kMaxHub);
    w.off_hubdeg = o; o = al(o + 4 * G * kMaxHub);
    w.off_hubcnt = o; o = al(o + 4 * G * kMaxHub);
    w.off_hubmark = o; o = al(o + 4 * G * kMaxHub * (int64_t)(w.ncap / 32));
    w.off_scratch = o; o = al(o + 4 * scratch_entries);
    w.total = o;
    return w;
}

struct BatchOutDev {
    int32_t *node_off, *edge_off, *parent_nid, *graph_id, *row_ptr, *col_idx;
    int64_t node_cap, edge_cap;
};

// inclusive prefix sum over the workgroup's threads (thread order); *total = sum.  All threads call; wsum: LDS [5].
__device__ __forceinline__ int block_scan_incl(int v, int *total, int32_t *wsum)
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int incl = wave_scan_incl(v);
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int base = 0, all = 0;
    const int nw = (int)blockDim.x >> 6;
    for (int k = 0; k < nw; ++k) {
        const int t = wsum[k];
        base += k < wv ? t : 0;
        all += t;
    }
    __syncthreads();
    *total = all;
    return base + incl;
}

// two scans for one pair of barriers; wsum: LDS [2 * waves]
__device__ __forceinline__ void block_scan_incl2(int a, int b, int *ia, int *ib, int *ta, int *tb, int32_t *wsum)
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nw = (int)blockDim.x >> 6;
    const int sa = wave_scan_incl(a), sb = wave_scan_incl(b);
    if (lane == 63) { wsum[wv] = sa; wsum[nw + wv] = sb; }
    __syncthreads();
    int ba = 0, bb = 0, aa = 0, ab = 0;
    for (int k = 0; k < nw; ++k) {
        const int x = wsum[k], y = wsum[nw + k];
        ba += k < wv ? x : 0;
        bb += k < wv ? y : 0;
        aa += x;
        ab += y;
    }
    __syncthreads();
    *ia = ba + sa; *ib = bb + sb; *ta = aa; *tb = ab;
}

__device__ __forceinline__ int row_quads(int rb, int d) { return ((rb + d + 3) >> 2) - (rb >> 2); }


// ------------------------------------------------------------------ K1 ----
// Two launches: the small class (kBig = false: 256 threads, workgroup g = subgraph g, LDS for traces of at most p2cap
// entries -- 1024, 17 KiB, 8 workgroups per CU) hands the subgraphs whose seed allows a longer trace (graph_dataset.py:113-124:
// L grows with deg^0.75; 2.5 % of the seeds on the 10M / 200M graph) on through w.big; the big class (kBig = true: 1024
// threads, a resident grid over that list, LDS for the graph's longest trace) takes them.  With ONE launch sized for the
// longest trace every workgroup reserved 66 KiB there (two per CU) for the sake of those 2.5 %.
template <int kT, bool kBig>
__global__ __launch_bounds__(kT) void rwr_walk_kernel(
    const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col_idx,
    const double *__restrict__ seed_cdf, const int32_t *__restrict__ ltab, int64_t num_nodes,
    int32_t ltab_len, int32_t p2cap, uint64_t run_seed, int64_t first_sample_id, int64_t step_stride, int32_t B,
    uint32_t restart_u32, const int32_t *__restrict__ seeds_in, const int64_t *__restrict__ shard_off,
    int32_t num_shards, int32_t hub_degree, int32_t max_hubs, Work w)
{
    DYN_SMEM(smem);
    __shared__ int32_t wsum[2 * (kT / 64) + 1];
    __shared__ int32_t sh_want;                // members over the hub threshold
    __shared__ int32_t dhist[32];              // members by floor(log2(degree))
    uint32_t *buf = (uint32_t *)smem;          // [p2cap] trace -> sorted trace
    int32_t *ld = (int32_t *)smem + p2cap;     // [p2cap + 64] degrees of the kept rows (n <= L + 1)
    int32_t *lrb = ld + p2cap + 64;            // [p2cap + 64] their row begins          } small class only: the big class reads
    int32_t *lq = lrb + p2cap + 64;            // [p2cap + 64] and quads                 } row_ptr again (half the LDS)
    const int tid = (int)threadIdx.x;
    for (int item = (int)blockIdx.x;; item += (int)gridDim.x) {
    int g = item;
    if constexpr (kBig) {
        __syncthreads();                       // (the previous item's tables are no longer read)
        if (item >= w.big[0]) return;
        g = w.big[1 + item];
    }
    // subgraph g = segment * B + b, segment = 2 * step + view: a call covers the batches of several consecutive steps
    const int seg = g / B, b = g - seg * B;
    const int step = seg >> 1, view = seg & 1;
    const uint64_t sid = (uint64_t)(first_sample_id + (int64_t)step * step_stride + b);

    // ---- seed: graph_dataset.py:85-92 (p ~ deg^0.75; numpy choice == cdf upper bound)
    int32_t seed;
    if (seeds_in) {
        seed = seeds_in[step * B + b];
    } else {
        uint32_t x[4];
        philox4x32_10((uint32_t)sid, (uint32_t)(sid >> 32), 0u, 0u, (uint32_t)run_seed ^ 0x5EED5EEDu,
                      (uint32_t)(run_seed >> 32) ^ 0x00A11CE5u, x);
        const uint64_t u53 = ((uint64_t)x[0] << 21) | (uint64_t)(x[1] >> 11);
        const double u = (double)u53 * (1.0 / 9007199254740992.0);
        // the worker shard this DataLoader batch comes from (graph_dataset.py:23-30,63-76: batch i is produced by worker
        // i % num_workers out of ITS graphs; jobs repeat with period num_shards): the cdf is that shard's own
        int64_t lo = 0, hi = num_nodes;
        if (num_shards > 1) {
            const int64_t sh = (int64_t)((sid / (uint64_t)B) % (uint64_t)num_shards);
            lo = shard_off[sh];
            hi = shard_off[sh + 1];
        }
        const int64_t last = hi - 1;
        // workgroup-cooperative 256-ary upper bound: first index with cdf[i] > u (1M entries: 3 rounds)
        while (lo < hi) {
            const int64_t len = hi - lo;
            const int64_t step = (len + kT - 1) / kT;
            const int64_t idx = lo + (int64_t)tid * step;
            const bool le = (idx < hi) && (seed_cdf[idx] <= u);
            int k;
            (void)block_scan_incl(le ? 1 : 0, &k, wsum);       // monotone: the first k threads are true
            if (k == 0) { hi = lo; break; }
            const int64_t nhi = lo + (int64_t)k * step;
            lo = lo + (int64_t)(k - 1) * step + 1;
            if (nhi < hi) hi = nhi;
        }
        seed = (int32_t)(lo <= last ? lo : last);
    }
    if (view == 0 && tid == 0) w.seeds[step * B + b] = seed;

    const int32_t rp0 = row_ptr[seed];
    const int32_t deg0 = row_ptr[seed + 1] - rp0;
    const int32_t L = ltab[deg0 < ltab_len ? deg0 : ltab_len - 1];   // graph_dataset.py:113-124
    const int p2 = pow2_ceil(L);
    if constexpr (!kBig) {
        if (p2 > p2cap) {                      // (block-uniform) a long trace: the big class's
            if (tid == 0) w.big[1 + atomicAdd(&w.big[0], 1)] = g;
            return;
        }
    }

    for (int i = tid; i < p2; i += kT) buf[i] = kEmpty;
    if (tid < 32) dhist[tid] = 0;
    __syncthreads();

    // ---- walks: graph_dataset.py:125-130 (DGL random_walk_with_restart); walk id = base + thread
    const uint64_t gid = sid * 2u + (uint64_t)view;
    const uint32_t k0 = (uint32_t)run_seed, k1 = (uint32_t)(run_seed >> 32);
    const uint32_t g0 = (uint32_t)gid, g1 = (uint32_t)(gid >> 32);
    int total = 0;   // block-uniform: trace entries assigned so far
    for (int base = 0; total < L; base += kT) {
        const uint32_t walk = (uint32_t)(base + tid);
        uint32_t x0[4];
        philox4x32_10(walk, 0u, g0, g1, k0, k1, x0);
        // len = min{t >= 1 : word[2t-1] < restart_u32}, capped at L
        int len = 1;
        if (x0[1] >= restart_u32) {
            len = 2;
            if (x0[3] >= restart_u32) {
                len = 3;
                uint32_t y[4];
                for (uint32_t blk = 1; len < L; ++blk) {
                    philox4x32_10(walk, blk, g0, g1, k0, k1, y);
                    if (y[1] < restart_u32) break;
                    ++len;
                    if (len >= L || y[3] < restart_u32) break;
                    ++len;
                }
            }
        }
        int sum;
        const int incl = block_scan_incl(len, &sum, wsum);
        const int off = total + incl - len;
        total += sum;
        // the part of this walk that is inside the budget
        int allowed = L - off;
        allowed = allowed < 0 ? 0 : (allowed > len ? len : allowed);
        if (allowed > 0) {
            int32_t c = col_idx[rp0 + (int32_t)__umulhi(x0[0], (uint32_t)deg0)];
            buf[off] = (uint32_t)c;
            uint32_t y[4] = {x0[0], x0[1], x0[2], x0[3]};
            uint32_t yblk = 0;
            for (int t = 1; t < allowed; ++t) {
                const uint32_t blk = (uint32_t)t >> 1;      // word 2t lives in block t/2
                if (blk != yblk) { philox4x32_10(walk, blk, g0, g1, k0, k1, y); yblk = blk; }
                const uint32_t r = (t & 1) ? y[2] : y[0];
                const int32_t beg = row_ptr[c];
                const int32_t d = row_ptr[c + 1] - beg;
                c = col_idx[beg + (int32_t)__umulhi(r, (uint32_t)d)];
                buf[off + t] = (uint32_t)c;
            }
        }
    }
    __syncthreads();

    // ---- torch.unique (data_util.py:221): bitonic sort of the padded trace, all four waves
    for (int k = 2; k <= p2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < (p2 >> 1); i += kT) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int hi = lo + j;
                const bool up = (lo & k) == 0;
                const uint32_t a = buf[lo], c = buf[hi];
                if ((a > c) == up) { buf[lo] = c; buf[hi] = a; }
            }
            __syncthreads();
        }
    }

    // ---- subv = [seed] + (unique \ {seed})  (data_util.py:222-226); fetch row extents
    int32_t *nodes = w.nodes + (int64_t)g * w.ncap;
    int32_t *rowbeg = w.rowbeg + (int64_t)g * w.ncap;
    int32_t *rowdeg = w.rowdeg + (int64_t)g * w.ncap;
    if (tid == 0) { nodes[0] = seed; ld[0] = deg0; if constexpr (!kBig) { lrb[0] = rp0; lq[0] = row_quads(rp0, deg0); } sh_want = deg0 >= hub_degree ? 1 : 0; if (deg0 >= hub_degree) atomicAdd(&dhist[31 - __builtin_clz((uint32_t)deg0 | 1u)], 1); }
    __syncthreads();
    int n = 1, p0 = 0;   // block-uniform: members; members with a parent id below the seed's (they are sorted: locals 1 .. p0)
    for (int i0 = 0; i0 < L; i0 += kT) {
        const int i = i0 + tid;
        uint32_t v = kEmpty, prev = kEmpty;
        if (i < L) { v = buf[i]; if (i > 0) prev = buf[i - 1]; }
        const bool keep = (i < L) && (v != (uint32_t)seed) && (i == 0 || v != prev);
        int kept;
        const int incl2 = block_scan_incl(keep ? 1 + ((int32_t)v < seed ? 1 << 16 : 0) : 0, &kept, wsum);
        const int incl = incl2 & 0xFFFF;
        p0 += kept >> 16;
        kept &= 0xFFFF;
        if (keep) {
            const int pos = n + incl - 1;
            const int32_t rb = row_ptr[v];
            const int32_t d = row_ptr[v + 1] - rb;
            nodes[pos] = (int32_t)v;
            ld[pos] = d;
            if constexpr (!kBig) { lrb[pos] = rb; lq[pos] = row_quads(rb, d); }
            if (d >= hub_degree) { atomicAdd(&sh_want, 1); atomicAdd(&dhist[31 - __builtin_clz((uint32_t)d | 1u)], 1); }
        }
        n += kept;
    }
    __syncthreads();
    // induction work.  Rows of degree >= the subgraph's threshold (below: at most max_hubs of them) are hubs: not scanned, their
    // induced rows come from the mirror images of the other rows' hits (hub_write_kernel).  The scanned rows
    // are compacted: scanned row s has local id srow[s], covers row_quads() aligned quads of col_idx, rowq = their
    // exclusive prefix.
    int run = 0, ns = 0, nhw = 0;                        // block-uniform: quads, scanned rows, rows over the threshold so far
    int32_t *rowq = w.rowq + (int64_t)g * w.ncap;
    int32_t *srow = w.srow + (int64_t)g * w.ncap;
    int32_t *hubloc = w.hubloc + (int64_t)g * kMaxHub, *hubrb = w.hubrb + (int64_t)g * kMaxHub, *hubdeg = w.hubdeg + (int64_t)g * kMaxHub;
    // this subgraph's threshold: hub_degree, or -- with more than max_hubs rows over it -- the smallest power of two that
    // leaves at most max_hubs (block-uniform; which rows are hubs changes the cost only, never the result)
    int thr = hub_degree;
    if (sh_want > max_hubs) {
        int cnt = 0, bb = 31;
        while (bb > 0 && cnt + dhist[bb - 1] <= max_hubs) { cnt += dhist[bb - 1]; --bb; }   // (members with d >= 2^bb: dhist[bb ..]; none has bit 31)
        thr = bb >= 31 ? 0x7FFFFFFF : (1 << bb);
        if (thr < hub_degree) thr = hub_degree;          // (the histogram holds the rows over hub_degree only: 2^bb may undercut it)
    }
    for (int i0 = 0; i0 < n; i0 += kT) {
        const int i = i0 + tid;
        const bool in = i < n;
        const int d = in ? ld[i] : 0;
        const bool is_hub = in && d >= thr;
        int rbi = 0;
        if (in) { if constexpr (kBig) rbi = row_ptr[nodes[i]]; else rbi = lrb[i]; }   // (nodes: this workgroup's own stores, before a barrier)
        int c = 0;
        if (in && !is_hub) { if constexpr (kBig) c = row_quads(rbi, d); else c = lq[i]; }
        int qincl, cnts, qsum, tots;                     // two scans, one pair of barriers
        block_scan_incl2(c, in ? (is_hub ? 1 << 16 : 1) : 0, &qincl, &cnts, &qsum, &tots, wsum);
        if (in && !is_hub) {
            const int p = ns + (cnts & 0xFFFF) - 1;
            srow[p] = i;
            rowbeg[p] = rbi;
            rowdeg[p] = d;
            rowq[p] = run + qincl - c;
        }
        if (is_hub) { const int hidx = nhw + (cnts >> 16) - 1; hubloc[hidx] = i; hubrb[hidx] = rbi; hubdeg[hidx] = d; }
        run += qsum;
        ns += tots & 0xFFFF;
        nhw += tots >> 16;
    }
    const int nh = nhw;                                  // <= max_hubs <= kMaxHub
    uint32_t *hm = w.hubmark + (int64_t)g * kMaxHub * w.mwords;
    int32_t *hubcnt = w.hubcnt + (int64_t)g * kMaxHub;
    {   // the hubs' neighbour bitmaps and entry counts start empty (only the words this subgraph can touch)
        const int words = (n + 31) >> 5;
        for (int i = tid; i < nh * words; i += kT) hm[(i / words) * w.mwords + (i % words)] = 0u;
        if (tid < nh) hubcnt[tid] = 0;
    }
    if (tid == 0) {
        w.sub_n[g] = n;
        w.sub_ns[g] = ns;
        w.sub_nh[g] = nh;
        w.sub_p0[g] = p0;
        w.sub_quads[g] = run;
        w.sub_nnz[g] = 0;
    }
    __syncthreads();
    // edges between two hubs (neither row will be scanned): one search per pair, in the shorter of the two rows (rows are
    // sorted: the input contract); both mirror images are marked and counted here
    const int npairs = nh * (nh - 1) / 2;
    auto found = [&](int a, int c) {
        const int la = hubloc[a], lc = hubloc[c];
        atomicOr(&hm[a * w.mwords + (lc >> 5)], 1u << (lc & 31));
        atomicOr(&hm[c * w.mwords + (la >> 5)], 1u << (la & 31));
        atomicAdd(&hubcnt[a], 1);
        atomicAdd(&hubcnt[c], 1);
        atomicAdd(&w.sub_nnz[g], 2);
    };
    if (npairs > 0 && w.hub_index) {                     // (block-uniform) one bit probe per pair in the parent's hub-hub table
        for (int pr = tid; pr < nh * nh; pr += kT) {
            const int a = pr / nh, c = pr - a * nh;
            if (a >= c) continue;
            const int ia = w.hub_index[nodes[hubloc[a]]], ic = w.hub_index[nodes[hubloc[c]]];   // (both >= 0: degree >= the table's threshold)
            if ((w.hub_adj[(int64_t)ia * w.hub_words + (ic >> 5)] >> (ic & 31)) & 1u) found(a, c);
        }
    } else if (npairs > 0 && npairs <= 48) {             // (block-uniform)
        // few pairs (the usual case): a team of 16 lanes per pair narrows the range 16-fold per round of loads -- 2 to 5
        // dependent loads for rows of 256 .. 1M entries where a one-lane binary search has 8 to 20
        const int lane = tid & 63, team = tid >> 4, sl = tid & 15, tsh = (lane >> 4) * 16;
        for (int pr0 = 0; pr0 < npairs; pr0 += kT / 16) {
            const int pr = pr0 + team;
            const bool valid = pr < npairs;
            int a = 0, c = 1, lo = 0, hi = 0;
            int32_t key = 0;
            if (valid) {
                int rem = pr;
                while (rem >= nh - 1 - a) { rem -= nh - 1 - a; ++a; }
                c = a + 1 + rem;
                const int da = hubdeg[a], dc = hubdeg[c];
                const int sx = da <= dc ? a : c, tx = da <= dc ? c : a;
                key = nodes[hubloc[tx]];
                lo = hubrb[sx];
                hi = lo + hubdeg[sx];
            }
            // invariant: the key, if present, lies in [lo, hi)
            while (wave_ballot(hi - lo > 16) != 0ull) {
                const bool act = hi - lo > 16;
                const int stp = (hi - lo + 15) >> 4;
                const int idx = lo + sl * stp;
                const bool le = act && idx < hi && col_idx[idx] <= key;      // monotone over the team's lanes
                const int k = __popcll((wave_ballot(le) >> tsh) & 0xFFFFull);
                if (act) {
                    if (k == 0) { hi = lo; }                 // below the first entry
                    else {
                        const int nlo = lo + (k - 1) * stp, nhi = lo + k * stp;
                        lo = nlo;
                        hi = nhi < hi ? nhi : hi;
                    }
                }
            }
            const bool hit = valid && lo + sl < hi && col_idx[lo + sl] == key;
            const bool any = ((wave_ballot(hit) >> tsh) & 0xFFFFull) != 0ull;
            if (any && sl == 0) found(a, c);
        }
    } else if (npairs > 0) {
    for (int pr = tid; pr < nh * nh; pr += kT) {
        const int a = pr / nh, c = pr - a * nh;
        if (a >= c) continue;
        const int da = hubdeg[a], dc = hubdeg[c];
        const int sx = da <= dc ? a : c, tx = da <= dc ? c : a;     // look hub tx's id up in hub sx's (shorter) row
        const int32_t key = nodes[hubloc[tx]];
        const int rb0 = hubrb[sx], re0 = rb0 + hubdeg[sx];
        int lo = rb0, hi = re0;                          // lower bound of the key (three independent probes per round were
                                                         // measured slower than this: the searches are bound by requests, not latency)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (col_idx[mid] < key) lo = mid + 1; else hi = mid;
        }
        if (lo < re0 && col_idx[lo] == key) found(a, c);
    }
    }
    if constexpr (!kBig) return;
    }
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

__device__ __forceinline__ int units_of(int quads) { return (quads + kUnitQuads - 1) / kUnitQuads; }

// ------------------------------------------------------------------ prefix steps ----
// Exclusive prefixes over the G = 2 B subgraphs that every workgroup of induce_kernel / pack_kernel needs, one small
// workgroup each: step A after the walks (virtual workgroups, units, scratch slots, node offsets per view, subgraph of
// every virtual workgroup), step B after the induction (edge offsets per view).  (Folding them into the last workgroup
// of the kernel before was measured: the agent-scope release every workgroup needs for the hand-off writes back L2 on
// this multi-die part, 4096 times per launch -- the induction went from 45 to 305 us.  A kernel boundary is cheaper.)
// Prefixes over the G subgraphs of a launch on ONE workgroup: wave k owns the k-th contiguous chunk of subgraphs and walks it
// 64 at a time (coalesced loads, a wave scan per step, no barrier), one block step joins the waves' totals, a second walk
// writes.  (One block scan per 1024 subgraphs and per view -- 32 views in a 16-step launch -- were 54 + 21 us per launch, a
// tenth of a G1 launch; `per` consecutive subgraphs per thread with one block scan: 38 + 10, the stride-`per` loads.)
// dst[view * B + b] = exclusive prefix of src within each view (dgl.batch offsets restart per view) = the prefix over all
// G subgraphs minus its value at the view's first subgraph.  All threads call.
__device__ void view_prefix(int32_t B, int32_t nseg, const int32_t *src, int32_t *dst, int32_t *wsum /* LDS [waves] */,
                            int32_t *pre /* LDS [G + 1] */)
{
    const int G = nseg * B, tid = (int)threadIdx.x, T = (int)blockDim.x, lane = tid & 63, wv = tid >> 6, nw = T >> 6;
    const int chunk = ((G + nw - 1) / nw + 63) & ~63, c0 = wv * chunk, c1 = min(G, c0 + chunk);
    int sum = 0;
    for (int g = c0 + lane; g < c1; g += 64) sum += src[g];
    sum = wave_last(wave_scan_incl(sum));
    if (lane == 0) wsum[wv] = sum;
    __syncthreads();
    int run = 0;
    for (int k = 0; k < wv; ++k) run += wsum[k];
    for (int g0 = c0; g0 < c1; g0 += 64) {
        const int g = g0 + lane;
        const int v = g < c1 ? src[g] : 0;
        const int incl = wave_scan_incl(v);
        if (g < c1) pre[g] = run + incl - v;
        run += wave_last(incl);
    }
    __syncthreads();
    for (int g = tid; g < G; g += T) dst[g] = pre[g] - pre[(g / B) * B];
}

__device__ __forceinline__ long long wave_scan_incl64(long long v)
{
    const int lane = (int)threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long t = wave_shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

// step A, after the walks: virtual workgroups of either induce class, units, scratch slots (all over the whole launch), node
// offsets per view.  Same shape as view_prefix.
__device__ void prefix_step_a(int32_t B, const Work &w, int32_t *wsum /* LDS [64] */, long long *wsum64 /* LDS [16] */,
                              int32_t *pre /* LDS [G + 1] */)
{
    const int G = w.nseg * B, tid = (int)threadIdx.x, T = (int)blockDim.x, lane = tid & 63, wv = tid >> 6, nw = T >> 6;
    const int chunk = ((G + nw - 1) / nw + 63) & ~63, c0 = wv * chunk, c1 = min(G, c0 + chunk);
    auto parts = [&](int g, int &v, int &vb, int &u, int &q) {
        q = w.sub_quads[g];
        const bool isbig = w.sub_n[g] > w.lcap;
        u = units_of(q);
        v = isbig ? 0 : (u + vwg_units(kInduceThreads) - 1) / vwg_units(kInduceThreads);
        vb = isbig ? (u + vwg_units(kInduceBigThreads) - 1) / vwg_units(kInduceBigThreads) : 0;
    };
    int sv = 0, sb = 0, su = 0;
    long long ss = 0;
    for (int g = c0 + lane; g < c1; g += 64) {
        int v, vb, u, q;
        parts(g, v, vb, u, q);
        sv += v; sb += vb; su += u; ss += 4ll * q;
    }
    sv = wave_last(wave_scan_incl(sv));
    sb = wave_last(wave_scan_incl(sb));
    su = wave_last(wave_scan_incl(su));
    ss = wave_shfl(wave_scan_incl64(ss), 63);
    if (lane == 0) { wsum[wv * 4 + 0] = sv; wsum[wv * 4 + 1] = sb; wsum[wv * 4 + 2] = su; wsum64[wv] = ss; }
    __syncthreads();
    int rv = 0, rb = 0, ru = 0, tv = 0, tb = 0, tu = 0;
    long long rs = 0, ts = 0;
    for (int k = 0; k < nw; ++k) {
        if (k < wv) { rv += wsum[k * 4]; rb += wsum[k * 4 + 1]; ru += wsum[k * 4 + 2]; rs += wsum64[k]; }
        tv += wsum[k * 4]; tb += wsum[k * 4 + 1]; tu += wsum[k * 4 + 2]; ts += wsum64[k];
    }
    for (int g0 = c0; g0 < c1; g0 += 64) {
        const int g = g0 + lane;
        int v = 0, vb = 0, u = 0, q = 0;
        if (g < c1) parts(g, v, vb, u, q);
        const int iv = wave_scan_incl(v), ib = wave_scan_incl(vb), iu = wave_scan_incl(u);
        const long long is = wave_scan_incl64(4ll * q);
        if (g < c1) { w.vbp[g] = rv + iv - v; w.vbpb[g] = rb + ib - vb; w.ubp[g] = ru + iu - u; w.sbp[g] = rs + is - 4ll * q; }
        rv += wave_last(iv); rb += wave_last(ib); ru += wave_last(iu); rs += wave_shfl(is, 63);
    }
    if (tid == 0) { w.vbp[G] = tv; w.vbpb[G] = tb; w.ubp[G] = tu; w.sbp[G] = ts; }
    __syncthreads();                                   // (wsum is reused)
    view_prefix(B, w.nseg, w.sub_n, w.nbp, wsum, pre);
}

__global__ __launch_bounds__(kPrefixThreads) void prefix_a_kernel(int32_t B, Work w)
{
    DYN_SMEM(smem);
    __shared__ int32_t wsum[64];
    __shared__ long long wsum64[16];
    prefix_step_a(B, w, wsum, wsum64, (int32_t *)smem);
}
// The start record of every induce workgroup (see prefix_step_a): one thread per record, on as many workgroups as it
// takes -- as the tail of the single-workgroup prefix kernel this was 9 us per step, and 104 us of a 16-step launch
// (65 k records by 1024 threads).
__global__ __launch_bounds__(256) void records_kernel(int32_t B, Work w)
{
    const int G = w.nseg * B;
    const int nsmall = w.nsmall;
    const int bb = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (bb >= nsmall + w.nbig) return;
    const bool big = bb >= nsmall;                    // the big class's records follow the small one's
    const int b = big ? bb - nsmall : bb, nwg = big ? w.nbig : nsmall;
    const int32_t *vbp = big ? w.vbpb : w.vbp;
    const int cv = vbp[G];
    const int chunk = (cv + nwg - 1) / nwg;
    const int vb0 = b * chunk;
    int32_t *rec = w.wrec + (int64_t)bb * kRecInts;
    const int count = min(chunk, cv - vb0);
    if (count <= 0) { rec[0] = 0; return; }
    int lo = 0, hi = G;                               // last g with vbp[g] <= vb0 (it has virtual workgroups OF THIS CLASS: vbp[g + 1] > vb0)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (vbp[mid] <= vb0) lo = mid; else hi = mid;
    }
    const long long sb = w.sbp[lo];
    rec[0] = count;
    rec[1] = lo;
    rec[2] = vb0 - vbp[lo];
    rec[3] = w.sub_n[lo];
    rec[4] = w.sub_quads[lo];
    rec[5] = w.ubp[lo];
    rec[6] = (int32_t)(sb & 0xFFFFFFFFll);
    rec[7] = (int32_t)(sb >> 32);
    rec[8] = w.sub_ns[lo];
    rec[9] = w.sub_nh[lo];
}
__global__ __launch_bounds__(kPrefixThreads) void prefix_b_kernel(int32_t B, Work w)
{
    DYN_SMEM(smem);
    __shared__ int32_t wsum[16];
    view_prefix(B, w.nseg, w.sub_nnz, w.ebp, wsum, (int32_t *)smem);
}

__device__ __forceinline__ bool scratch_overflows(const Work &w, int g, int64_t scratch_entries)
{
    return w.sbp[g] + 4ll * (long long)w.sub_quads[g] > scratch_entries ||
           (int64_t)w.ubp[g] + units_of(w.sub_quads[g]) > w.unit_cap;
}

// ------------------------------------------------------------------ K2 ----
static int g_dbg_grids[3] = {0, 0, 0};           // diagnostics (gcc_sampler_debug_grids): small / big induce grid, big walk grid
static const bool g_use_hub_table = [] { const char *e = getenv("GCC_SAMPLER_HUB_TABLE"); return !e || atoi(e) != 0; }();   // A/B knob: 0 = search every hub pair
static long long *g_induce_ticks = nullptr;      // diagnostics (gcc_sampler_debug_ticks): [0..2] phase ticks, [15] workgroups
#define IND_TICK(ph) do { if (ticks && tid == 0) { const long long now_ = device_ticks(); atomicAdd((unsigned long long *)&ticks[ph], (unsigned long long)(now_ - tick_)); tick_ = now_; } } while (0)
template <int kT>
__global__ __launch_bounds__(kT) GCC_INDUCE_OCC void induce_kernel(
    const int32_t *__restrict__ col_idx, int64_t num_edges, int32_t bm_log2_cap, int32_t B, int64_t scratch_entries,
    Work w, int32_t *__restrict__ status, long long *ticks, int32_t big, int32_t lcap)
{
    // Two launches (like the walks): the small class has LDS tables for lcap = w.lcap members (320: every subgraph whose
    // seed has the rw_hops trace budget, 91 % on the bench graphs; 27 KiB, six workgroups per CU), the big class for the
    // graph's longest trace (ncap; 70 KiB at lmax 2348: two per CU, which every workgroup paid before the split).
    DYN_SMEM(smem);
    constexpr int kW = kT / 64, kUV = vwg_units(kT);
    const int ncap = w.ncap, G = w.nseg * B;                 // (ncap: the stride of the per-subgraph arrays in the workspace)
    uint32_t *snodes = (uint32_t *)smem;                     // [lcap]     members (seed first, the rest ascending)
    int32_t *sq = (int32_t *)(snodes + lcap);                // [lcap + 1] exclusive prefix of quads per row
    int32_t *srb = sq + (lcap + 2);                          // [lcap]     row begin  (sq padded: what follows stays 8-byte aligned)
    int32_t *srd = srb + lcap;                               // [lcap]     row degree   (sq / srb / srd: SCANNED rows, by scan position)
    uint16_t *srow16 = (uint16_t *)(srd + lcap);             // [lcap]     local id of a scanned row
    uint8_t *hubslot = (uint8_t *)(srow16 + lcap);           // [lcap]     hub index of a member, 255 = not a hub
    uint32_t *bm = (uint32_t *)(hubslot + lcap);             // [1 << (bm_log2_cap - 5)] Bloom bitmap of the members   (lcap % 64 == 0)
    uint32_t *candv_all = bm + (1u << (bm_log2_cap - 5));    // [4][kCandCap] Bloom survivors (parent ids) ...
    uint16_t *candr_all = (uint16_t *)(candv_all + kW * kCandCap);   // [waves][kCandCap] ... and their row
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = wave_uniform(tid >> 6);
    uint32_t *candv = candv_all + wave * kCandCap;
    uint16_t *candr = candr_all + wave * kCandCap;
    uint16_t *rowl = candr_all + kW * kCandCap + wave * kUnitQuads;   // [waves][256] row of every quad of a unit
    long long tick_ = ticks ? device_ticks() : 0;
    if (ticks && tid == 0) atomicAdd((unsigned long long *)&ticks[15], 1ull);
    const int32_t *myrec = w.wrec + ((int64_t)blockIdx.x + (big ? (int64_t)w.nsmall : 0)) * kRecInts;   // (prefix step A)
    const uint4 *recp = (const uint4 *)myrec;
    const uint4 ra = recp[0], rb4 = recp[1];
    const uint2 rc2 = *(const uint2 *)(myrec + 8);
    const int count = wave_uniform((int)ra.x);               // (uniform by construction: into scalar registers)
    int g = wave_uniform((int)ra.y), part = wave_uniform((int)ra.z), n = wave_uniform((int)ra.w);
    int ns = wave_uniform((int)rc2.x), nh = wave_uniform((int)rc2.y);   // scanned rows, hubs
    int totq = wave_uniform((int)rb4.x), ubase = wave_uniform((int)rb4.y);
    long long sbase = (long long)(((unsigned long long)(uint32_t)wave_uniform((int)rb4.w) << 32) |
                                  (unsigned long long)(uint32_t)wave_uniform((int)rb4.z));
    int cur_g = -1, bshift = 0;
    // A workgroup takes CONSECUTIVE virtual workgroups when there are more of them than workgroups (the 10M/200M graph:
    // 2.5 x), so that the member tables are rebuilt only at a subgraph boundary.  (A resident grid with virtual
    // workgroups handed out through a counter was measured slower: 52 against 45 us.)
    for (int it = 0; it < count; ++it) {
        if (it) {
            if (++part * kUV >= units_of(totq)) {   // on to the next subgraph that has edges (block-uniform)
                part = 0;
                do {                                         // ... of this class
                    ++g;
                    totq = g < G ? wave_uniform(w.sub_quads[g]) : 1;
                    n = g < G ? wave_uniform(w.sub_n[g]) : (big ? 0x7FFFFFFF : 0);
                } while (totq == 0 || (n > w.lcap) != (big != 0));
                if (g >= G) break;                           // (the records and the prefixes come from the same pass)
                ns = wave_uniform(w.sub_ns[g]);
                nh = wave_uniform(w.sub_nh[g]);
                ubase = wave_uniform(w.ubp[g]);
                const long long sb = w.sbp[g];
                sbase = (long long)(((unsigned long long)(uint32_t)wave_uniform((int)(sb >> 32)) << 32) |
                                    (unsigned long long)(uint32_t)wave_uniform((int)(sb & 0xFFFFFFFFll)));
            }
        }
        const int nunits = units_of(totq);
        if (sbase + 4ll * (long long)totq > scratch_entries || (int64_t)ubase + nunits > w.unit_cap) {   // block-uniform
            if (tid == 0) atomicOr(status, (int32_t)GCC_STATUS_SCRATCH_OVERFLOW);
            continue;
        }
        IND_TICK(0);
        // this wave's units are wave, wave + 4, ... of the virtual workgroup's kUV.  issue(): the rows of a
        // unit's quads (binary search in the LDS prefix), then its 4 dwordx4 loads per lane.  (Two units in flight per
        // wave were measured: 112 VGPRs, 4 workgroups per CU instead of 6, 42 against 37 us.)
        uint4 v[4];
        int r_[4];                                          // (row bounds, quad address: re-derived from LDS when the data is in)
        auto issue = [&](int unit) {
            // Rows of the unit's 256 consecutive quads without a search per quad (4 searches x 10 dependent steps per
            // lane were a third of the kernel's instructions, and the kernel is VALU-bound): the wave finds the row of
            // the unit's first quad together (64-ary, two rounds up to 4096 members), the rows that START inside the
            // unit mark their first quad in a 256-entry LDS strip, and a prefix maximum spreads the marks.
            const int qb = unit * kUnitQuads;
            {
                int r0 = 0, span = ns;                               // last slot with sq[slot] <= qb is in [r0, r0 + span)
                while (span > 1) {
                    const int stride = (span + 63) >> 6;
                    const int idx = r0 + lane * stride;
                    const bool le = lane * stride < span && sq[idx] <= qb;   // true for lane 0, monotone in the lane
                    const int c = __popcll(wave_ballot(le));
                    const int adv = (c - 1) * stride;
                    r0 += adv;
                    span = min(stride, span - adv);
                }
                ((uint2 *)rowl)[lane] = make_uint2(0u, 0u);          // entries 4 lane .. 4 lane + 3
                wave_sync();
                for (int i0 = r0 + 1;; i0 += 64) {                   // rows r0 + 1 ... start after qb (sq ascends strictly)
                    const int i = i0 + lane;
                    const int q = i < ns ? sq[i] : 0x7FFFFFFF;
                    const bool in = q < qb + kUnitQuads;
                    if (in) rowl[q - qb] = (uint16_t)(i - r0);       // <= 256: every row has at least one quad
                    if (wave_ballot(in) != ~0ull) break;             // wave-uniform
                }
                wave_sync();
                const uint2 mk = ((const uint2 *)rowl)[lane];
                int e0 = (int)(mk.x & 0xFFFFu), e1 = (int)(mk.x >> 16), e2 = (int)(mk.y & 0xFFFFu), e3 = (int)(mk.y >> 16);
                e1 = max(e1, e0);
                e2 = max(e2, e1);
                e3 = max(e3, e2);
                const int incl = wave_scan_max_incl(e3);
                int before = wave_shfl_up(incl, 1);
                if (lane == 0) before = 0;
                e0 = max(e0, before);
                e1 = max(e1, before);
                e2 = max(e2, before);
                e3 = max(e3, before);
                ((uint2 *)rowl)[lane] = make_uint2((uint32_t)e0 | ((uint32_t)e1 << 16), (uint32_t)e2 | ((uint32_t)e3 << 16));
                wave_sync();
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int fq = qb + u * 64 + lane;
                    const bool ok = fq < totq;
                    const int r = r0 + (int)rowl[u * 64 + lane];
                    const int rbv = srb[r];
                    const int a0 = (((rbv >> 2) + (fq - sq[r])) << 2);   // element index of the aligned quad
                    r_[u] = ok ? r : -1;
                    if (ok && (int64_t)a0 + 4 <= num_edges) {
                        v[u] = *(const uint4 *)(col_idx + a0);
                    } else {                                         // the array's last quad may be partial
                        uint32_t t[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] = (ok && (int64_t)a0 + e < num_edges) ? (uint32_t)col_idx[a0 + e] : kEmpty;
                        v[u] = make_uint4(t[0], t[1], t[2], t[3]);
                    }
                }
            }
        };
        const int unit0 = part * kUV + wave;
        if (g != cur_g) {                                    // block-uniform
            __syncthreads();                                 // the previous subgraph's tables are no longer read
            int bl = 11;                                     // >= 64 bits per member, >= 2048 bits
            while ((1 << bl) < 64 * n && bl < bm_log2_cap) ++bl;
            bshift = 32 - bl;
            for (int i = tid; i < (1 << (bl - 5)); i += kT) bm[i] = 0u;
            const int32_t *nodes = w.nodes + (int64_t)g * ncap;
            const int32_t *rq = w.rowq + (int64_t)g * ncap;
            const int32_t *rb = w.rowbeg + (int64_t)g * ncap;
            const int32_t *rd = w.rowdeg + (int64_t)g * ncap;
            const int32_t *sr = w.srow + (int64_t)g * ncap;
            for (int i = tid; i < n; i += kT) {
                snodes[i] = (uint32_t)nodes[i];
                hubslot[i] = 255;
            }
            for (int i = tid; i < ns; i += kT) {
                sq[i] = rq[i];
                srb[i] = rb[i];
                srd[i] = rd[i];
                srow16[i] = (uint16_t)sr[i];
            }
            if (tid == 0) sq[ns] = totq;
            __syncthreads();
            if (tid < nh) hubslot[w.hubloc[(int64_t)g * kMaxHub + tid]] = (uint8_t)tid;      // (nh <= kMaxHub <= threads)
            __syncthreads();
        }
        int my_nnz = 0, nhub_hits = 0;                       // wave-uniform: this wave's hits, and how many of them have a hub as target
        const int unit_end = min(nunits, (part + 1) * kUV);
#pragma unroll 1
        for (int unit = unit0;; unit += kW) {              // wave-uniform; every wave enters once
            if (unit < unit_end) issue(unit);                        // (one call site: the body is large)
            if (g != cur_g) {                                        // block-uniform, first pass only: the first unit's
                for (int i = tid; i < n; i += kT) {      // loads fly while the bitmap is built
                    const uint32_t h = umul24(snodes[i], kHashMul) >> bshift;
                    atomicOr(&bm[h >> 5], 1u << (h & 31));
                }
                __syncthreads();
                cur_g = g;
                IND_TICK(1);
            }
            if (unit >= unit_end) break;
            int32_t *out = w.scratch + sbase + (long long)unit * kUnitElems;
            uint32_t *hmark = w.hubmark + (int64_t)g * kMaxHub * w.mwords;
            int ncand = 0, nout = 0;                                 // wave-uniform
            auto drain = [&]() {
                wave_sync();
                for (int c0 = 0; c0 < ncand; c0 += 64) {
                    const int c = c0 + lane;
                    int loc = -1;
                    if (c < ncand) {
                        const uint32_t val = candv[c];
                        if (val == snodes[0]) {
                            loc = 0;
                        } else {
                            int lo = 1, hi = n;                      // first index in [1, n) with snodes[idx] >= val
                            while (lo < hi) {
                                const int mid = (lo + hi) >> 1;
                                if (snodes[mid] < val) lo = mid + 1; else hi = mid;
                            }
                            if (lo < n && snodes[lo] == val) loc = lo;
                        }
                    }
                    const unsigned long long m = wave_ballot(loc >= 0);
                    uint32_t hs = 255u;
                    if (loc >= 0) {
                        const uint32_t lrow = srow16[candr[c]];      // the scanned row's local id
                        out[nout + __popcll(m & lanemask_lt())] = (int32_t)((lrow << 16) | (uint32_t)loc);
                        hs = hubslot[loc];
                        // the hit's mirror image: row `loc` is a hub and is not scanned -- (hub -> this row) is an edge too
                        // (every (row, hub) pair is hit once -- rows hold no duplicates --, so every mark is a new entry)
                        if (hs != 255u) {
                            atomicOr(&hmark[hs * (uint32_t)w.mwords + (lrow >> 5)], 1u << (lrow & 31));
                            atomicAdd(&w.hubcnt[(int64_t)g * kMaxHub + hs], 1);
                        }
                    }
                    nhub_hits += __popcll(wave_ballot(hs != 255u));
                    nout += __popcll(m);
                }
                wave_sync();
                ncand = 0;
            };
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t vals[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                const int rr = r_[u];
                const int rc = max(rr, 0);
                const int rbv = srb[rc];
                const int a0 = (((rbv >> 2) + (unit * kUnitQuads + u * 64 + lane - sq[rc])) << 2);
                const int d0 = a0 - rbv;                             // elements outside the row belong to other rows:
                const uint32_t deg = rr >= 0 ? (uint32_t)srd[rc] : 0u;   // 0 <= d0 + e < deg, one unsigned compare
                uint32_t pass = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t h = umul24(vals[e], kHashMul) >> bshift;
                    const uint32_t in_row = (uint32_t)(d0 + e) < deg ? 1u : 0u;
                    pass |= ((bm[h >> 5] >> (h & 31)) & in_row) << e;     // (always read: no branch per element)
                }
                const int cnt = __popc(pass);
                const int incl = wave_scan_incl(cnt);
                const int tot = wave_last(incl);
                if (ncand + tot > kCandCap) drain();                 // wave-uniform; a round adds at most 256 = kCandCap
                int at = ncand + incl - cnt;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (pass & (1u << e)) {
                        candv[at] = vals[e];
                        candr[at] = (uint16_t)r_[u];
                        ++at;
                    }
                }
                ncand += tot;
            }
            drain();
            if (lane == 0) w.ucnt[ubase + unit] = nout;
            my_nnz += nout;
        }
        if (lane == 0 && my_nnz + nhub_hits) atomicAdd(&w.sub_nnz[g], my_nnz + nhub_hits);   // (+ the mirror images: the hubs' own rows)
        IND_TICK(2);
    }
}

// ------------------------------------------------------------------ K3 ----
struct PackOuts { BatchOutDev o[2 * GCC_SAMPLE_MAX_STEPS]; };   // one batch per segment: q, k of step 0, q, k of step 1, ...
__global__ __launch_bounds__(256) void pack_kernel(int32_t B, Work w, PackOuts outs,
                                                    int64_t scratch_entries, int32_t *__restrict__ status)
{
    __shared__ int32_t wsum[5];
    __shared__ int32_t uoff[257];                 // exclusive prefix of the hit counts of a chunk of 256 units
    __shared__ int32_t sh_base, sh_carry;
    __shared__ int32_t hl[kMaxHub + 1], hc[kMaxHub + 1];   // hub rows (ascending local ids) and the exclusive prefix of their entry counts
    const int tid = (int)threadIdx.x;
    const int g = (int)blockIdx.x / kPackParts, part = (int)blockIdx.x % kPackParts;
    const int nunits = units_of(w.sub_quads[g]);
    constexpr int nparts = kPackParts;
    const int seg = g / B, b = g - seg * B;
    const BatchOutDev o = outs.o[seg];
    const int n = w.sub_n[g];
    const int nnz = w.sub_nnz[g];                    // scanned hits + the hub rows' entries
    const int nh = w.sub_nh[g];
    if (tid < 64) {                                  // (kMaxHub <= 64: one wave, one round of loads)
        const int cnt = tid < nh ? w.hubcnt[(int64_t)g * kMaxHub + tid] : 0;
        const int incl = wave_scan_incl(cnt);
        if (tid <= kMaxHub) hl[tid] = tid < nh ? w.hubloc[(int64_t)g * kMaxHub + tid] : 0x7FFFFFFF;
        if (kMaxHub == 64 && tid == 0) hl[64] = 0x7FFFFFFF;
        if (tid < nh) hc[tid] = incl - cnt;
        if (tid == 63) hc[nh] = incl;
    }
    __syncthreads();
    const int hs0 = nh > 32 ? 32 : (nh > 16 ? 16 : (nh > 8 ? 8 : (nh > 4 ? 4 : (nh > 2 ? 2 : 1))));   // (2 * hs0 >= nh)
    // Hub rows are not in the flat sequence of scanned hits: everything at or after row i sits hshift(i) entries further on
    // (the entries of the hub rows before row i), and a hub row H itself starts at (scanned hits of rows < H) + hshift(H)
    auto hshift = [&](int i) -> int {
        if (nh == 0) return 0;
        int k = 0;                                   // hubs with a local id below i (hl is padded with INT_MAX)
        for (int st = hs0; st >= 1; st >>= 1) k += hl[k + st - 1] < i ? st : 0;
        k += hl[k] < i ? 1 : 0;
        return hc[k];
    };
    const long long node_base = w.nbp[g];            // (prefix steps A / B)
    const long long edge_base = w.ebp[g];
    const long long sbase = w.sbp[g];
    const int ubase = w.ubp[g];
    if (tid == 0 && part == 0) {
        o.node_off[b] = (int32_t)node_base;
        o.edge_off[b] = (int32_t)edge_base;
        if (b == B - 1) {
            o.node_off[B] = (int32_t)(node_base + n);
            o.edge_off[B] = (int32_t)(edge_base + nnz);
        }
    }
    const bool bad_scratch = scratch_overflows(w, g, scratch_entries);
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
        for (int i = part * 256 + tid; i < n; i += 256 * nparts) {
            if (node_base + i < o.node_cap) {
                o.parent_nid[node_base + i] = nodes[i];
                o.graph_id[node_base + i] = b;
                o.row_ptr[node_base + i] = (int32_t)ecl;
            }
        }
        if (b == B - 1 && tid == 0 && part == 0 && node_base + n <= o.node_cap) o.row_ptr[node_base + n] = (int32_t)ecl;
        return;
    }
    for (int i = part * 256 + tid; i < n; i += 256 * nparts) {
        o.parent_nid[node_base + i] = nodes[i];
        o.graph_id[node_base + i] = b;
    }
    if (b == B - 1 && tid == 0 && part == 0) o.row_ptr[node_base + n] = (int32_t)(edge_base + nnz);

    // this part's units; hits before them and the row of the last of those hits
    const int u0 = (int)((long long)nunits * part / nparts), u1 = (int)((long long)nunits * (part + 1) / nparts);
    const int32_t *ucnt = w.ucnt + ubase;
    const int32_t *scratch = w.scratch + sbase;
    {
        int s = 0, last = -1;
        for (int j = tid; j < u0; j += 256) {
            const int c = ucnt[j];
            s += c;
            if (c > 0) last = j;
        }
        int tot;
        (void)block_scan_incl(s, &tot, wsum);
        // block max of `last`
        for (int d = 32; d >= 1; d >>= 1) { const int t = wave_shfl_xor(last, d); last = t > last ? t : last; }
        if ((tid & 63) == 0) wsum[tid >> 6] = last;
        __syncthreads();
        if (tid == 0) {
            int m = wsum[0];
            for (int k = 1; k < 4; ++k) m = wsum[k] > m ? wsum[k] : m;
            sh_base = tot;
            sh_carry = m >= 0 ? (int)((uint32_t)scratch[(long long)m * kUnitElems + ucnt[m] - 1] >> 16) : -1;
        }
        __syncthreads();
    }
    int e_base = sh_base;                              // block-uniform: hits before the current chunk
    for (int c0 = u0; c0 < u1; c0 += 256) {
        const int j = c0 + tid;
        const int cnt = j < u1 ? ucnt[j] : 0;
        int chunk_total;
        const int incl = block_scan_incl(cnt, &chunk_total, wsum);
        uoff[tid] = incl - cnt;
        if (tid == 0) uoff[256] = chunk_total;
        __syncthreads();
        const int carry = sh_carry;
        for (int x = tid; x < chunk_total; x += 256) {
            const int jj = upper_slot(uoff, 256, x);   // among equal offsets the last one owns x (the others are empty)
            const int kk = x - uoff[jj];
            const int32_t *slot = scratch + (long long)(c0 + jj) * kUnitElems;
            const uint32_t hv = (uint32_t)slot[kk];
            const int row = (int)(hv >> 16);
            const long long e = (long long)e_base + x;
            o.col_idx[edge_base + e + hshift(row)] = (int32_t)node_base + (int32_t)(hv & 0xFFFFu);
            int prow;
            if (kk > 0) {
                prow = (int)((uint32_t)slot[kk - 1] >> 16);
            } else if (x > 0) {
                const int pj = upper_slot(uoff, 256, x - 1);
                prow = (int)((uint32_t)scratch[(long long)(c0 + pj) * kUnitElems + (x - 1 - uoff[pj])] >> 16);
            } else {
                prow = carry;
            }
            for (int i = prow + 1; i <= row; ++i) o.row_ptr[node_base + i] = (int32_t)(edge_base + e + hshift(i));
        }
        __syncthreads();
        if (tid == 0 && chunk_total > 0) {
            const int pj = upper_slot(uoff, 256, chunk_total - 1);
            sh_carry = (int)((uint32_t)scratch[(long long)(c0 + pj) * kUnitElems + (chunk_total - 1 - uoff[pj])] >> 16);
        }
        e_base += chunk_total;
        __syncthreads();
    }
    if (part == nparts - 1) {                          // rows after the last hit have no induced edges
        const int carry = sh_carry;
        const int scanned = nnz - hc[nh];              // (all scanned hits lie before these rows)
        for (int i = carry + 1 + tid; i < n; i += 256) o.row_ptr[node_base + i] = (int32_t)(edge_base + scanned + hshift(i));
    }
}

// ------------------------------------------------------------------ hub rows ----
// hub_write_kernel (after the pack): the entries of every hub row -- the set bits of its neighbour bitmap (marked by the
// walk kernel for hub pairs and by the induction for everything else) -- in the order of the parent row: ascending parent
// id, i.e. ascending local id except that the seed (local id 0) stands where ITS parent id belongs.  One WAVE per hub (no
// workgroup barrier): a lane takes the bitmap words lane, lane + 64, ...; a wave scan of the popcounts places them.
__global__ __launch_bounds__(256) void hub_write_kernel(int32_t B, Work w, PackOuts outs, int64_t scratch_entries)
{
    const int g = (int)blockIdx.x, tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nh = w.sub_nh[g];
    if (nh == 0) return;                                 // (block-uniform)
    const int seg = g / B;
    const BatchOutDev o = outs.o[seg];
    const int n = w.sub_n[g], nnz = w.sub_nnz[g];
    const long long node_base = w.nbp[g], edge_base = w.ebp[g];
    if (scratch_overflows(w, g, scratch_entries) || node_base + n > o.node_cap || edge_base + nnz > o.edge_cap) return;   // (pack left clamped rows)
    const int p0 = w.sub_p0[g];                          // locals 1 .. p0 have smaller parent ids than the seed (walk kernel)
    const int words = (n + 31) >> 5;
    const uint32_t *hm = w.hubmark + (int64_t)g * kMaxHub * w.mwords;
    // every load this wave needs, up front: where its hubs' rows start (lanes 0 .. 7) and the first 64 words of their
    // bitmaps (all there is up to 2048 members) -- one hub after the other was three dependent loads per hub, 24 in a row
    // for a wave with 8 hubs
    constexpr int kPerWave = kMaxHub / 4;
    const int mine = nh > wv ? (nh - wv + 3) >> 2 : 0;   // hubs wv, wv + 4, ...
    int basel = 0;
    if (lane < mine) basel = o.row_ptr[node_base + w.hubloc[(int64_t)g * kMaxHub + wv + 4 * lane]];      // (written by the pack)
    uint32_t first[kPerWave];
#pragma unroll
    for (int j = 0; j < kPerWave; ++j) first[j] = (j < mine && lane < words) ? hm[(wv + 4 * j) * w.mwords + lane] : 0u;
#pragma unroll
    for (int j = 0; j < kPerWave; ++j) {
        if (j >= mine) break;                            // (wave-uniform)
        const long long base = (long long)wave_shfl(basel, j);
        const uint32_t *bits = hm + (wv + 4 * j) * w.mwords;
        const int seed_bit = (int)(wave_shfl(first[j], 0) & 1u);
        int before = 0, seed_at = 0;                     // wave-uniform: entries of locals >= 1 in the words done; of locals 1 .. p0
        for (int w0 = 0; w0 < words; w0 += 64) {
            const int wi = w0 + lane;
            uint32_t v = w0 == 0 ? first[j] : (wi < words ? bits[wi] : 0u);
            if (wi == 0) v &= ~1u;                       // the seed is placed separately
            // bits of this word that belong to locals <= p0
            const int lo_bit = wi * 32;
            const uint32_t m_le = p0 >= lo_bit + 31 ? 0xFFFFFFFFu : (p0 < lo_bit ? 0u : (0xFFFFFFFFu >> (31 - (p0 - lo_bit))));
            const int pc = __popc(v), pl = __popc(v & m_le);
            const int incl = wave_scan_incl(pc | (pl << 16));            // (n <= 65535: both sums fit 16 bits)
            const int tot = wave_last(incl);
            int at = before + (incl & 0xFFFF) - pc;      // entries of locals >= 1 before this word
            while (v) {
                const int bit = __ffsll((unsigned long long)v) - 1;
                v &= v - 1;
                const int loc = lo_bit + bit;
                o.col_idx[base + at + (loc > p0 ? seed_bit : 0)] = (int32_t)node_base + loc;
                ++at;
            }
            before += tot & 0xFFFF;
            seed_at += tot >> 16;
        }
        if (seed_bit && lane == 0) o.col_idx[base + seed_at] = (int32_t)node_base;
    }
}

}  // namespace

extern "C" {

void gcc_sampler_debug_ticks(long long *device_ticks64) { g_induce_ticks = device_ticks64; }   /* diagnostics only */
void gcc_sampler_debug_grids(int32_t small_grid, int32_t big_grid, int32_t walk_big_grid)          /* tests only */
{
    g_dbg_grids[0] = small_grid; g_dbg_grids[1] = big_grid; g_dbg_grids[2] = walk_big_grid;
}


static int32_t check_steps(const char *who, int32_t batch_size, int32_t num_steps)
{
    // prefix step A keeps one LDS word per subgraph of the call: 2 * batch_size * num_steps + 1 of them PLUS the kernel's
    // static LDS (wsum[64] + wsum64[16] = 384 B, budgeted as 512) must fit the 64 KiB a launch gets without an opt-in
    if (num_steps < 1 || num_steps > GCC_SAMPLE_MAX_STEPS || (2ll * batch_size * num_steps + 1) * 4 + 512 > 64 * 1024) {
        snprintf(g_err, kErrLen, "%s: num_steps = %d (1 .. %d, and 2 * batch_size * num_steps <= 16255)", who, num_steps,
                 GCC_SAMPLE_MAX_STEPS);
        return -1;
    }
    return 0;
}

int64_t gcc_sampler_workspace_bytes_multi(const gcc_graph *g, int32_t batch_size, int32_t num_steps, int64_t scratch_entries)
{
    if (!g || batch_size <= 0 || scratch_entries <= 0 || g->lmax <= 0) {
        snprintf(g_err, kErrLen, "gcc_sampler_workspace_bytes: bad argument");
        return -1;
    }
    if (check_steps("gcc_sampler_workspace_bytes_multi", batch_size, num_steps)) return -1;
    return work_layout(g->lmax, batch_size, 2 * num_steps, scratch_entries).total;
}

int64_t gcc_sampler_workspace_bytes(const gcc_graph *g, int32_t batch_size, int64_t scratch_entries)
{
    return gcc_sampler_workspace_bytes_multi(g, batch_size, 1, scratch_entries);
}

int32_t gcc_sample_multi(const gcc_graph *g, const gcc_sample_params *p, int32_t num_steps, int64_t sample_id_stride,
                         const gcc_batch_out *outs, void *workspace, int64_t workspace_bytes,
                         int64_t scratch_entries, int32_t *status, void *stream)
{
    if (!g || !p || !outs || !workspace || !status) {
        snprintf(g_err, kErrLen, "gcc_sample_multi: null argument");
        return -1;
    }
    if (g->num_shards > 1 && !g->shard_off) {
        snprintf(g_err, kErrLen, "gcc_sample_multi: num_shards = %d without shard_off", g->num_shards);
        return -1;
    }
    if (p->batch_size <= 0 || g->lmax <= 0 || g->lmax > 65534 || g->num_nodes <= 0 ||
        g->num_nodes > 0x7FFFFFFF || g->num_edges > 0x7FFFFFFF) {
        snprintf(g_err, kErrLen, "gcc_sample_multi: size out of range (B=%d lmax=%d V=%lld E=%lld)",
                 p->batch_size, g->lmax, (long long)g->num_nodes, (long long)g->num_edges);
        return -2;
    }
    if (check_steps("gcc_sample_multi", p->batch_size, num_steps)) return -1;
    const int nseg = 2 * num_steps;
    const WorkLayout wl = work_layout(g->lmax, p->batch_size, nseg, scratch_entries);
    if (workspace_bytes < wl.total) {
        snprintf(g_err, kErrLen, "gcc_sample_multi: workspace %lld < %lld bytes",
                 (long long)workspace_bytes, (long long)wl.total);
        return -3;
    }
    if (((uintptr_t)g->col_idx & 15u) != 0) {
        snprintf(g_err, kErrLen, "gcc_sample_multi: col_idx must be 16-byte aligned (the induction streams it in dwordx4 quads)");
        return -5;
    }
    char *base = (char *)workspace;
    Work w;
    w.seeds = (int32_t *)(base + wl.off_seeds);
    w.sub_n = (int32_t *)(base + wl.off_n);
    w.sub_quads = (int32_t *)(base + wl.off_quads);
    w.sub_nnz = (int32_t *)(base + wl.off_nnz);
    w.nodes = (int32_t *)(base + wl.off_nodes);
    w.rowbeg = (int32_t *)(base + wl.off_rowbeg);
    w.rowdeg = (int32_t *)(base + wl.off_rowdeg);
    w.rowq = (int32_t *)(base + wl.off_rowq);
    w.srow = (int32_t *)(base + wl.off_srow);
    w.sub_ns = (int32_t *)(base + wl.off_ns);
    w.sub_nh = (int32_t *)(base + wl.off_nh);
    w.sub_p0 = (int32_t *)(base + wl.off_p0);
    w.big = (int32_t *)(base + wl.off_big);
    w.hubloc = (int32_t *)(base + wl.off_hubloc);
    w.hubrb = (int32_t *)(base + wl.off_hubrb);
    w.hubdeg = (int32_t *)(base + wl.off_hubdeg);
    w.hubcnt = (int32_t *)(base + wl.off_hubcnt);
    w.hubmark = (uint32_t *)(base + wl.off_hubmark);
    w.mwords = wl.ncap / 32;
    w.vbp = (int32_t *)(base + wl.off_vbp);
    w.ubp = (int32_t *)(base + wl.off_ubp);
    w.sbp = (long long *)(base + wl.off_sbp);
    w.nbp = (int32_t *)(base + wl.off_nbp);
    w.ebp = (int32_t *)(base + wl.off_ebp);
    w.ucnt = (int32_t *)(base + wl.off_ucnt);
    w.wrec = (int32_t *)(base + wl.off_wrec);
    w.vbpb = (int32_t *)(base + wl.off_vbpb);
    w.scratch = (int32_t *)(base + wl.off_scratch);
    w.ncap = wl.ncap;
    w.lcap = wl.ncap;
    w.nseg = nseg;
    w.unit_cap = wl.unit_cap;

    const int B = p->batch_size, G = nseg * B;
    hipStream_t s = (hipStream_t)stream;
    int p2max = 64;
    while (p2max < g->lmax) p2max <<= 1;
    int bmlog = 11;                                  // Bloom bitmap: 64 bits per member up to 1024 members, 8 KiB at most (LDS per
    while ((1 << bmlog) < 64 * (g->lmax + 1) && bmlog < 16) ++bmlog;   // workgroup decides how many are resident: 82 KiB at lmax 2360 left one per CU)
    const int p2small = p2max < 1024 ? p2max : 1024;      // the small walk class: traces up to this many entries
    const size_t lds1 = ((size_t)p2small * 4 + 192) * 4;   //   trace, degrees, row begins, quads
    const size_t lds1b = ((size_t)p2max * 2 + 64) * 4;     // the big class: trace, degrees
    const size_t lds2 = (size_t)wl.ncap * 19 + 8 + ((size_t)1 << (bmlog - 3)) + (size_t)(kInduceBigThreads / 64) * (kCandCap * 6 + kUnitQuads * 2) + 16;
    // the small induce class: subgraphs of at most kSmallMembers members (a seed with the rw_hops budget: <= rw_hops + 1)
    const int lcap_small = wl.ncap > kSmallMembers ? kSmallMembers : wl.ncap;
    int bmlog_s = 11;
    while ((1 << bmlog_s) < 64 * lcap_small && bmlog_s < bmlog) ++bmlog_s;
    const size_t lds2s = (size_t)lcap_small * 19 + 8 + ((size_t)1 << (bmlog_s - 3)) + (size_t)(kInduceThreads / 64) * (kCandCap * 6 + kUnitQuads * 2) + 16;
    w.lcap = lcap_small;
    w.nsmall = G * kGridMult;
    w.nbig = kBigGrid;
    int walk_big_grid = G < 512 ? G : 512;
    if (g_dbg_grids[0] > 0 && g_dbg_grids[0] < w.nsmall) w.nsmall = g_dbg_grids[0];     // (tests: tiny grids make every workgroup walk
    if (g_dbg_grids[1] > 0 && g_dbg_grids[1] < w.nbig) w.nbig = g_dbg_grids[1];          //  through several virtual workgroups / subgraphs /
    if (g_dbg_grids[2] > 0 && g_dbg_grids[2] < walk_big_grid) walk_big_grid = g_dbg_grids[2];   //  list entries)
    // rows of at least this degree are not scanned (kMaxHub per subgraph): hub_degree 0 = default, < 0 = scan everything
    // (hub rows are rebuilt from mirror images: exact on a symmetric, sorted, duplicate- and loop-free parent only, so the
    //  short cut needs the caller's word that the contract was checked -- gcc_graph.flags; without it every row is scanned)
    const bool contract = (g->flags & GCC_GRAPH_CONTRACT_CHECKED) != 0;
    const int32_t hub_degree = (p->hub_degree < 0 || !contract) ? 0x7FFFFFFF : (p->hub_degree == 0 ? kHubDegreeDefault : p->hub_degree);
    const int32_t max_hubs = p->max_hubs <= 0 ? kMaxHubsDefault : (p->max_hubs > kMaxHub ? kMaxHub : p->max_hubs);
    if (lds1b > 160 * 1024 || lds2 > 160 * 1024 - 256) {
        snprintf(g_err, kErrLen, "gcc_sample_multi: lmax=%d / batch too large for 160 KiB of LDS", g->lmax);
        return -4;
    }
    PackOuts po;
    memset(&po, 0, sizeof(po));
    for (int i = 0; i < nseg; ++i) {
        const gcc_batch_out &o = outs[i];
        po.o[i] = {o.node_off, o.edge_off, o.parent_nid, o.graph_id, o.row_ptr, o.col_idx, o.node_cap, o.edge_cap};
    }

    prof_mark(p->prof, 0, s);
#ifndef GCC_AMD_HIPEMU
    // more than 64 KiB of dynamic LDS has to be opted into per kernel
    if (lds1b > 64 * 1024) (void)hipFuncSetAttribute((const void *)rwr_walk_kernel<1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1b);
    if (lds2 > 64 * 1024) (void)hipFuncSetAttribute((const void *)induce_kernel<kInduceBigThreads>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
#endif
    const int64_t *shards = g->num_shards > 1 ? g->shard_off : nullptr;
    // the hub-hub table serves a call whose hubs all have at least the table's degree (GCC_SAMPLER_HUB_TABLE=0: searches, for A/B)
    const bool table = g->hub_index && g->hub_adj && g->num_hubs > 0 && hub_degree != 0x7FFFFFFF && hub_degree >= g->hub_table_degree && g_use_hub_table;
    w.hub_index = table ? g->hub_index : nullptr;
    w.hub_adj = table ? g->hub_adj : nullptr;
    w.hub_words = g->hub_words;
    if (p2max > p2small) (void)hipMemsetAsync(w.big, 0, 4, s);
    hipLaunchKernelGGL((rwr_walk_kernel<kWalkThreads, false>), dim3(G), dim3(kWalkThreads), lds1, s, g->row_ptr, g->col_idx, g->seed_cdf,
                       g->ltab, g->num_nodes, g->ltab_len, p2small, p->run_seed, p->first_sample_id, sample_id_stride, B,
                       p->restart_u32, p->seeds, shards, g->num_shards, hub_degree, max_hubs, w);
    if (p2max > p2small)                             // seeds with longer traces: 1024 threads each, a resident grid over the list
        hipLaunchKernelGGL((rwr_walk_kernel<1024, true>), dim3(walk_big_grid), dim3(1024), lds1b, s, g->row_ptr, g->col_idx, g->seed_cdf,
                           g->ltab, g->num_nodes, g->ltab_len, p2max, p->run_seed, p->first_sample_id, sample_id_stride, B,
                           p->restart_u32, p->seeds, shards, g->num_shards, hub_degree, max_hubs, w);
    hipLaunchKernelGGL(prefix_a_kernel, dim3(1), dim3(kPrefixThreads), (size_t)(G + 1) * 4, s, B, w);
    hipLaunchKernelGGL(records_kernel, dim3((w.nsmall + w.nbig + 255) / 256), dim3(256), 0, s, B, w);
    prof_mark(p->prof, 1, s);                        // marks 1 -> 2 bracket the induction alone (bench.py's roofline interval)
    hipLaunchKernelGGL(induce_kernel<kInduceThreads>, dim3(w.nsmall), dim3(kInduceThreads), lds2s, s, g->col_idx, g->num_edges, bmlog_s, B,
                       scratch_entries, w, status, g_induce_ticks, 0, lcap_small);
    if (wl.ncap > lcap_small)                        // subgraphs with more members: tables for the graph's longest trace
        hipLaunchKernelGGL(induce_kernel<kInduceBigThreads>, dim3(w.nbig), dim3(kInduceBigThreads), lds2, s, g->col_idx, g->num_edges, bmlog, B,
                           scratch_entries, w, status, g_induce_ticks, 1, wl.ncap);
    prof_mark(p->prof, 2, s);
    hipLaunchKernelGGL(prefix_b_kernel, dim3(1), dim3(kPrefixThreads), (size_t)(G + 1) * 4, s, B, w);
    hipLaunchKernelGGL(pack_kernel, dim3(G * kPackParts), dim3(256), 0, s, B, w, po, scratch_entries, status);
    hipLaunchKernelGGL(hub_write_kernel, dim3(G), dim3(256), 0, s, B, w, po, scratch_entries);
    prof_mark(p->prof, 3, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, kErrLen, "gcc_sample_multi: launch failed: %s", hipGetErrorString(e));
        return -10;
    }
    return 0;
}

int32_t gcc_sample_batch(const gcc_graph *g, const gcc_sample_params *p, const gcc_batch_out *out_q,
                         const gcc_batch_out *out_k, void *workspace, int64_t workspace_bytes,
                         int64_t scratch_entries, int32_t *status, void *stream)
{
    if (!out_q || !out_k) {
        snprintf(g_err, kErrLen, "gcc_sample_batch: null argument");
        return -1;
    }
    const gcc_batch_out outs[2] = {*out_q, *out_k};
    return gcc_sample_multi(g, p, 1, 0, outs, workspace, workspace_bytes, scratch_entries, status, stream);
}

}  // extern "C"
