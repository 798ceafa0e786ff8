// gcc_amd/csrc/posemb.hip -- positional embedding of sampled subgraphs (gfx950).
//
// Replaces _add_undirected_graph_positional_embedding / eigen_decomposision
// (gcc/datasets/data_util.py:242-281: scipy.sparse.linalg.eigsh = ARPACK, ~35 ms per
// subgraph per CPU core -- the dominant cost of the reference's data pipeline).
//
// Sampled ego-nets are star-like: ~75 % of them have FEWER than 32 positive eigenvalues, the
// rest of the "top 32" is a large degenerate null space, and exact multiplicities occur among
// the positive eigenvalues too.  Pipeline of one call (any number of views, see gcc_posemb_multi):
//   posemb_classify_kernel  deflated size n' of every subgraph (twin leaves collapse) -> work lists
//   posemb_direct_kernel    n' <= 384: dense symmetric eigensolver (tridiagonalise, bisect, inverse
//                           iteration): exact multiplicities, deterministic run time; three size classes
//   posemb_krylov_kernel    larger ones: thick-restart Krylov-Schur, Ritz problem by the same solver core
#include <mutex>
#include "host_common.h"
#include <type_traits>

namespace {

constexpr int kJMax = GCC_POSEMB_LDS_MAX;
#ifndef GCC_POSEMB_SMALL_T
#define GCC_POSEMB_SMALL_T 256
#endif
constexpr int kSmallT = GCC_POSEMB_SMALL_T;   // threads of a small-class workgroup
constexpr int kJSmall = 64;        // LDS-resident size classes of the direct solver: n' <= 64 (256 threads, ~60 KiB of LDS)
                                   // and 65..kJMax (1024 threads, ~150 KiB)

constexpr int kMaxViews = GCC_POSEMB_MAX_VIEWS;
struct PosView {
    const int32_t *node_off, *row_ptr, *col_idx;
    float *pos, *evals, *raw;
};
struct PosMulti {                    // one launch covers the subgraphs of all views: item id = view * B + subgraph
    PosView v[kMaxViews];
    int32_t nviews, B, hidden;
    uint64_t seed;
    int32_t *status;
    long long *ticks;                // diagnostics: [class][phase] wall-clock ticks (NULL = off), gcc_posemb_debug_ticks
};
#define PHASE_TICK(ph) do { if (m.ticks && tid == 0) { const long long now_ = device_ticks(); atomicAdd((unsigned long long *)&m.ticks[kCls * 16 + (ph)], (unsigned long long)(now_ - tick_)); tick_ = now_; } } while (0)

// executed f32 FLOPs of one dense solve (diagnostics, ticks[class][14]): tridiagonalisation 2 n^3 (p = A v and the rank-2
// update on both triangles), bisection 5 flops per Sturm row, an inverse-iteration solve ~16 n per vector, the cluster
// sweep is not counted (data dependent, small), back-transformation 2 n^2 per vector, expansion 3 per output element
__device__ __forceinline__ unsigned long long dense_solve_flops(int nr, int kq, int na, int its, int probes_times_rounds, int n, int k)
{
    const unsigned long long N = (unsigned long long)nr;
    return 2ull * N * N * N + 5ull * (unsigned long long)probes_times_rounds * (unsigned long long)kq * N
           + 16ull * (unsigned long long)its * (unsigned long long)na * N + 2ull * N * N * (unsigned long long)na
           + 3ull * (unsigned long long)n * (unsigned long long)k;
}

struct PosArgs {
    const int32_t *node_off, *row_ptr, *col_idx;
    float *pos, *evals, *raw;
    int32_t B, hidden;
    uint64_t seed;
    int32_t *status;
    const int32_t *list;     // subgraph indices handled by this launch, or NULL = all
    const int32_t *count;    // device count for `list`
};

__device__ __forceinline__ void item_args(const PosMulti &m, int item, struct PosArgs &a, int &b);

// ---- exact leaf deflation.  Degree-1 nodes that share a parent p are twins: with t of them, the
// t - 1 contrast vectors on the leaves are exact null vectors of M = D^-1/2 A D^-1/2, and the rest of
// the spectrum is that of the quotient matrix M' in which the t leaves are one "super-leaf" coupled
// to p with sqrt(t / d_p).  Sampled ego-nets are star-like (a typical n = 92 has ~35 such null
// vectors), so M' is ~40 % smaller and the O(n^3) dense solve ~5x cheaper; an eigenvector y of M' expands
// to the leaves as y[super-leaf] / sqrt(t).  Any orthonormal basis of a degenerate eigenspace is
// as good as ARPACK's, so when the top-k reaches into the null space the contrasts are used.
#ifndef GCC_POSEMB_MID_THREADS
#define GCC_POSEMB_MID_THREADS 1024
#endif
constexpr int kMidT = GCC_POSEMB_MID_THREADS;    // workgroup size of the 65..kJMax class
constexpr int kNodeMax = 1024;       // largest subgraph the deflating direct kernels look at (per-node LDS tables)
constexpr uint16_t kNone = 0xFFFFu;
constexpr float kZeroEig = 1e-5f;    // |lambda| below this is "the null space" when ranking

//
// ---- exact stalk deflation.  A "stalk" of a hub h is a pendant two-path h - a - b (deg a = 2, deg b = 1); hub seeds
// of the 1M-node graph carry hundreds of them.  With s >= 2 stalks on h, the vectors c_i (x_a, x_b) = c_i (1, +-1) / sqrt(2)
// with sum c_i = 0 are exact eigenvectors for +-1/sqrt(2) (they do not couple to h), s - 1 of each sign, and the rest of
// the spectrum is that of the quotient in which the s stalks are ONE stalk whose middle node couples to h with
// sqrt(s / (2 d_h)) (the a - b coupling 1/sqrt(2) is unchanged); a quotient eigenvector expands to every stalk divided by
// sqrt(s).  The +1/sqrt(2) copies usually ARE among the top 32, so their contrasts (the same Helmert basis as for twin
// leaves, over the stalks of a hub in index order) are merged into the ranking by value.  Twin leaves and stalks never
// overlap: a stalk's middle node has exactly one leaf, a hub of >= 2 stalks has >= 2 non-leaf neighbours.
struct Defl {
    int32_t *tcnt, *cbase, *pcnt;       // [cap] leaves per parent | contrast bases: twin groups (low 16 bits), stalk groups (high 16) | stalks per hub
    uint16_t *par, *rep, *ridx, *ord;   // [cap] parent of a leaf | first leaf of a parent | reduced index | order inside the group
    uint16_t *phub, *prep;              // [cap] hub of a stalk's middle node | middle node of a hub's first stalk
};
constexpr int kDeflNodeBytes = 24;
constexpr float kStalkEig = 0.70710678f;
constexpr int kStalkSrc = 4097;         // column codes: j >= 0 eigenvector j | -(c + 1) twin contrast c | -(kStalkSrc + 2 c + sign) stalk contrast c

__device__ __forceinline__ void defl_bind(Defl &d, void *base, int cap)
{
    d.tcnt = (int32_t *)base;
    d.cbase = d.tcnt + cap;
    d.pcnt = d.cbase + cap;
    d.par = (uint16_t *)(d.pcnt + cap);
    d.rep = d.par + cap;
    d.ridx = d.rep + cap;
    d.ord = d.ridx + cap;
    d.phub = d.ord + cap;
    d.prep = d.phub + cap;
}

// the three per-node passes of the table build (a barrier between them; rp = the subgraph's row pointers, col = the batch's column ids)
__device__ __forceinline__ void defl_init_node(const Defl &d, int i, const int32_t *rp, const int32_t *col, int n0)
{
    d.par[i] = rp[i + 1] - rp[i] == 1 ? (uint16_t)(col[rp[i]] - n0) : kNone;
    d.tcnt[i] = 0;
    d.pcnt[i] = 0;
    d.rep[i] = kNone;
    d.prep[i] = kNone;
    d.phub[i] = kNone;
    d.ord[i] = 0;
}
__device__ __forceinline__ void defl_count_node(const Defl &d, int i, const int32_t *rp, const int32_t *col, int n0, bool stalks)
{
    if (d.par[i] != kNone) {
        atomicAdd(&d.tcnt[d.par[i]], 1);
    } else if (stalks && rp[i + 1] - rp[i] == 2) {
        const int x = col[rp[i]] - n0, y = col[rp[i] + 1] - n0;
        const bool lx = rp[x + 1] - rp[x] == 1, ly = rp[y + 1] - rp[y] == 1;
        if (lx != ly) {
            const int h = lx ? y : x;
            d.phub[i] = (uint16_t)h;
            atomicAdd(&d.pcnt[h], 1);
        }
    }
}
__device__ __forceinline__ void defl_order_node(const Defl &d, int p, const int32_t *rp, const int32_t *col, int n0)
{
    const bool lt = d.tcnt[p] >= 2, st = d.pcnt[p] >= 2;
    if (!lt && !st) return;
    int o = 0, q = 0;
    for (int e = rp[p]; e < rp[p + 1]; ++e) {      // rows are sorted: the members of a group in index order
        const int j = col[e] - n0;
        if (lt && d.par[j] == (uint16_t)p) {
            if (o == 0) d.rep[p] = (uint16_t)j;
            d.ord[j] = (uint16_t)o++;
        } else if (st && d.phub[j] == (uint16_t)p) {
            if (q == 0) d.prep[p] = (uint16_t)j;
            d.ord[j] = (uint16_t)q++;
        }
    }
}
// node i is represented by another node of its group in the quotient
__device__ __forceinline__ bool defl_collapsed(const Defl &d, int i)
{
    const int pi = d.par[i];
    if (pi != (int)kNone) {
        if (d.tcnt[pi] >= 2) return d.rep[pi] != (uint16_t)i;                    // twin leaf
        const int h = d.phub[pi];                                                 // the leaf of a stalk?
        return h != (int)kNone && d.pcnt[h] >= 2 && d.prep[h] != (uint16_t)pi;
    }
    const int h = d.phub[i];
    return h != (int)kNone && d.pcnt[h] >= 2 && d.prep[h] != (uint16_t)i;
}
// Reduced indices and contrast bases of all n nodes (after the three table passes): block-wide prefix sums over kept
// flags / twin contrasts / stalk contrasts, kT nodes at a time, in node order.  (Thread 0 walking the n nodes alone was
// 17 us of the 65..128 class's items and 150 us of the sparse block class's, n up to 1024.)  ALL threads call it;
// tot[0..2] (LDS) = n', z, zp after the trailing barrier; wsum = LDS scratch [3 * kT / 64].
template <int kT>
__device__ __forceinline__ void defl_prefix_block(const Defl &d, int n, int *tot /* [3] */, int *wsum)
{
    constexpr int kNW = kT / 64;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 3) tot[tid] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += kT) {
        const int i = i0 + tid;
        const bool valid = i < n;
        const int tc = valid ? d.tcnt[i] : 0, pc = valid ? d.pcnt[i] : 0;
        const int extra = tc >= 2 ? tc - 1 : 0, pextra = pc >= 2 ? pc - 1 : 0;
        const int kept = valid && !defl_collapsed(d, i) ? 1 : 0;
        const int sa = wave_scan_incl(kept), sb = wave_scan_incl(extra), sc = wave_scan_incl(pextra);
        if (lane == 63) { wsum[wv] = sa; wsum[kNW + wv] = sb; wsum[2 * kNW + wv] = sc; }
        __syncthreads();
        int pa = tot[0], pb = tot[1], pcs = tot[2];
        for (int q = 0; q < wv; ++q) { pa += wsum[q]; pb += wsum[kNW + q]; pcs += wsum[2 * kNW + q]; }
        if (valid) {
            d.cbase[i] = (pb + sb - extra) | ((pcs + sc - pextra) << 16);
            d.ridx[i] = kept ? (uint16_t)(pa + sa - 1) : kNone;
        }
        __syncthreads();
        if (tid < 3) {
            int t = tot[tid];
            for (int q = 0; q < kNW; ++q) t += wsum[tid * kNW + q];
            tot[tid] = t;
        }
        __syncthreads();
    }
}

// factor on 1 / sqrt(d_i d_j) for the coupling of two KEPT neighbours
__device__ __forceinline__ float defl_coupling(const Defl &d, int i, int j)
{
    float f = 1.0f;
    if (d.par[j] == (uint16_t)i && d.tcnt[i] >= 2) f = sqrtf((float)d.tcnt[i]);
    else if (d.par[i] == (uint16_t)j && d.tcnt[j] >= 2) f = sqrtf((float)d.tcnt[j]);
    else if (d.phub[j] == (uint16_t)i && d.pcnt[i] >= 2) f = sqrtf((float)d.pcnt[i]);
    else if (d.phub[i] == (uint16_t)j && d.pcnt[j] >= 2) f = sqrtf((float)d.pcnt[j]);
    return f;
}
// what the expansion needs of the tables, 8 bytes per node: quotient row | order inside the group | group size (0 = not
// grouped; bit 15 = stalk member, bit 14 = the stalk's leaf) | contrast base of the group
constexpr int kRecStalk = 0x8000, kRecStalkLeaf = 0x4000, kRecSize = 0x3FFF;
__device__ __forceinline__ void defl_record(const Defl &d, int v, const int32_t *rp, const int32_t *col, int n0, uint16_t *rec)
{
    const int pv = d.par[v];
    int rsrc = d.ridx[v], o = 0, g = 0, cb = 0;
    if (pv != (int)kNone && d.tcnt[pv] >= 2) {
        rsrc = d.ridx[d.rep[pv]]; o = d.ord[v]; g = d.tcnt[pv]; cb = d.cbase[pv] & 0xFFFF;
    } else {
        const int mid = pv != (int)kNone ? pv : v;                               // the middle node if v belongs to a stalk
        const int h = d.phub[mid];
        if (h != (int)kNone && d.pcnt[h] >= 2) {
            int r = d.prep[h];
            if (pv != (int)kNone) {                                              // the first stalk's leaf
                const int x = col[rp[r]] - n0, y = col[rp[r] + 1] - n0;
                r = rp[x + 1] - rp[x] == 1 ? x : y;
            }
            rsrc = d.ridx[r]; o = d.ord[mid]; cb = (int)((uint32_t)d.cbase[h] >> 16);
            g = d.pcnt[h] | kRecStalk | (pv != (int)kNone ? kRecStalkLeaf : 0);
        }
    }
    rec[0] = (uint16_t)rsrc; rec[1] = (uint16_t)o; rec[2] = (uint16_t)g; rec[3] = (uint16_t)cb;
}
// entry (node, output column) of the raw eigenvector matrix; Y = the quotient's eigenvectors [row][ldy]
__device__ __forceinline__ float defl_expand(int rsrc, int o, int g, int cb, int src, const float *Y, int ldy)
{
    const int tp = g & kRecSize;
    if (src >= 0) return Y[rsrc * ldy + src] * (tp ? 1.0f / sqrtf((float)tp) : 1.0f);
    if (!tp) return 0.f;
    int c;
    float f = 1.0f;
    if (src > -kStalkSrc) {
        if (g & kRecStalk) return 0.f;
        c = -src - 1;
    } else {
        if (!(g & kRecStalk)) return 0.f;
        const int q = -src - kStalkSrc;
        c = q >> 1;
        f = ((q & 1) && (g & kRecStalkLeaf)) ? -kStalkEig : kStalkEig;
    }
    const int jm1 = c - cb;                                                      // contrast j = jm1 + 1 of the group
    if (jm1 < 0 || jm1 >= tp - 1) return 0.f;
    const int j = jm1 + 1;
    const float nrm = f / sqrtf((float)(j * (j + 1)));
    return o < j ? nrm : (o == j ? -(float)j * nrm : 0.f);
}
// Ranks of the merged spectrum: lam[0 .. kq) (descending: the top of the quotient's spectrum), zp stalk contrasts at
// +1/sqrt(2), the null space = zeros of M' then the z twin contrasts, zp stalk contrasts at -1/sqrt(2).  eigsh(which="LA")
// returns the k largest in ascending order (data_util.py:251).  One thread.  -> number of quotient eigenvectors needed.
__device__ int rank_columns(const float *lam, int kq, int k, int z, int zp, int *colsrc, float *ev)
{
    int nge_s = 0, npz = 0, nge_ms = 0, na = 0;
    for (int j = 0; j < kq; ++j) {
        const float l = lam[j];
        if (l >= kStalkEig - kZeroEig) nge_s = j + 1;
        if (l >= -kZeroEig) npz = j + 1;
        if (l >= -kStalkEig - kZeroEig) nge_ms = j + 1;
    }
    for (int j = 0; j < kq; ++j) {
        const float l = lam[j];
        const int r = j + (j >= nge_s ? zp : 0) + (j >= npz ? z : 0) + (j >= nge_ms ? zp : 0);
        if (r < k) {
            na = j + 1;
            colsrc[k - 1 - r] = j;
            if (ev) ev[k - 1 - r] = fabsf(l) <= kZeroEig ? 0.f : l;
        }
    }
    for (int c = 0; c < zp && nge_s + c < k; ++c) {
        colsrc[k - 1 - (nge_s + c)] = -(kStalkSrc + 2 * c);
        if (ev) ev[k - 1 - (nge_s + c)] = kStalkEig;
    }
    for (int c = 0; c < z && npz + zp + c < k; ++c) {
        colsrc[k - 1 - (npz + zp + c)] = -(c + 1);
        if (ev) ev[k - 1 - (npz + zp + c)] = 0.f;
    }
    for (int c = 0; c < zp && nge_ms + zp + z + c < k; ++c) {
        colsrc[k - 1 - (nge_ms + zp + z + c)] = -(kStalkSrc + 2 * c + 1);
        if (ev) ev[k - 1 - (nge_ms + zp + z + c)] = -kStalkEig;
    }
    return na;
}

// =========================================================================
// Direct solver: Householder tridiagonalisation of the dense deflated matrix (one wave per row, the
// reflector replicated in registers: 2 barriers per column), multi-section bisection on Sturm counts
// for the k wanted eigenvalues (kT / 32 probe points per eigenvalue and round), inverse iteration on
// the tridiagonal matrix with partial pivoting (one thread per eigenvalue; three solve +
// re-orthogonalise rounds, classical Gram-Schmidt twice inside clusters of eigenvalues closer than
// 1e-3: the LAPACK stein recipe) and the back-transformation by the stored reflectors (one wave per
// vector pair, vectors in registers, no barriers).  ~25x less LDS traffic than two-sided Jacobi sweeps,
// deterministic run time, and exact multiplicities are resolved.
// Three instantiations by deflated size n':  <= 64 and 65..128 keep the matrix in LDS;  129..384 keep
// it in a workspace slot (rows are streamed coalesced, 4 n'^3 bytes in total; one CU sustains ~40 GB/s
// from beyond its L2, which is what bounds this class and why it stops at 384) with the tridiagonal
// data and the eigenvectors in LDS.
constexpr int kYld = 33;             // row stride of Y ([i][j], j < 32): odd, so conflict free both for "lane = vector"
                                     // (solves) and for "lane = row" (dots, back-transformation); the Krylov kernel's
                                     // Ritz problem wants up to 40 vectors and uses stride 65 (TriLds::ldy)
constexpr int kMaxVec = 32;          // hidden <= 32 on this path (GCC: positional_embedding_size = 32)
constexpr int kVecCap = 64;          // most eigenvectors the solver core handles
constexpr float kOrtol = 4e-3f;      // eigenvalues closer than this are re-orthogonalised against each other (LAPACK stein: 1e-3; in fp32 an
                                     // inverse-iteration residual of ~3e-7 over a gap of 3e-3 already leaves 1e-4..2.4e-4 between two vectors)
constexpr float kSep = 2e-6f;        // minimum distance between two inverse-iteration shifts
constexpr float kPivTiny = 1.2e-7f;  // pivots of T - shift are clamped to eps * ||T||  (||T|| <= 1)
constexpr int kGMax = GCC_POSEMB_DIRECT_MAX;   // largest deflated size of the workspace-resident class
constexpr int kGLds = 128 * 1024;    // dynamic LDS of that class (of 160 KiB per CU)
constexpr int kBMax = GCC_POSEMB_BIG_MAX;      // second workspace-resident class: kGMax < n' <= kBMax (hub seeds; a handful per
                                               // batch view at rw_hops 256 on a 1M-node power-law graph) -- exact multiplicities
                                               // (1/sqrt(2) occurs 30+ times in such ego-nets) are beyond a single-vector Krylov
                                               // iteration, ARPACK included
constexpr int kBLds = 160 * 1024 - 2048;       // its dynamic LDS: the whole CU minus the kernel's static __shared__ objects

__device__ __forceinline__ void item_args(const PosMulti &m, int item, PosArgs &a, int &b)
{
    const int view = item / m.B;
    b = item - view * m.B;
    const PosView &pv = m.v[view];
    a.node_off = pv.node_off; a.row_ptr = pv.row_ptr; a.col_idx = pv.col_idx;
    a.pos = pv.pos; a.evals = pv.evals; a.raw = pv.raw;
    a.B = m.B; a.hidden = m.hidden; a.seed = m.seed; a.status = m.status;
    a.list = nullptr; a.count = nullptr;
}

// Work lists.  posemb_classify_kernel sorts the subgraphs of a batch into four classes by deflated size; every
// solver kernel is launched with a SMALL fixed grid whose workgroups pull items from their class list.  (A grid of
// one fat workgroup per subgraph that exits early when the class does not match keeps the workgroup dispatcher
// busy placing 160-KiB-LDS / 1024-thread workgroups that do nothing, which delays every other queue.)
enum { kClsSmall = 0, kClsMid = 1, kClsSlot = 2, kClsKrylov = 3, kClsBig = 4, kClsCheb = 5, kClsW48 = 6, kClsW64 = 7, kNumCls = 8 };
// one-wave teams (posemb_wave_kernel): deflated size <= 48 / <= 64 with at most kWaveNodes original nodes; the
// 256-thread small class stays behind them for the (rare) leafier subgraphs
constexpr int kWaveNodes = 256;
// Both measured on the device in round 4 (scripts/gpu/r4_call1.sh, profiles/r4_posemb_phases_protos.txt): 'matrix' 17.2 -> 12.6 us
// (n' <= 48) / 32.3 -> 20.9 us (<= 64) of wave time per item, 'expand' 12.7 -> 10.7 / 19.5 -> 15.9; strict device tests green.
#ifndef GCC_POSEMB_EXPAND4
#define GCC_POSEMB_EXPAND4 1         // expansion of the one-wave teams four nodes per iteration (0: two)
#endif
#ifndef GCC_POSEMB_EDGE_FILL
#define GCC_POSEMB_EDGE_FILL 1       // matrix fill of the one-wave teams by entry instead of by row (0: lane = row)
#endif
constexpr int kWaveTeams = 4;        // teams (waves) per workgroup
static_assert(kNumCls == GCC_POSEMB_TICK_CLASSES, "include/gcc_amd.h: tick buffer classes");
struct PosHead {                     // head of the caller's workspace (zeroed per call)
    int32_t *count;                  // [4] items per class
    int32_t *next;                   // [4] work counters
    int32_t *list;                   // [4][T] item ids, T = views * B
    float *slots;                    // [workgroups of the slot class][slot_floats]: matrix (kGMax x kGMax) + deflation tables
    float *bslots;                   // [workgroups of the big class][bslot_floats]: matrix (kBMax x kBMax) + deflation tables
    int64_t bslot_floats;
    float *tabs;                     // [workgroups of the mid class, then of the small class][kNodeMax * 4]: deflation tables
    int32_t tabs_small_off;          // first small-class table (= workgroups of the mid class)
    int32_t T;
    int32_t ldv;                     // longest subgraph the Krylov class has room for (node_cap / batch_size, rounded up)
    int32_t use_cheb;                // deflated sizes above GCC_POSEMB_LDS_MAX try the sparse Chebyshev class first
    int32_t use_wave;                // deflated sizes <= 64 go to the one-wave teams
    int32_t use_stalks;              // pendant two-paths of a hub are deflated too (GCC_POSEMB_STALKS, default 1)
    int64_t slot_floats;
    float *pslots;                   // [workgroups of the register-resident 65..128 class (two / four waves)][kPairSlotFloats]: matrix / reflectors + expansion records
    int32_t use_pair;
};

// The deflation tables (24 KiB) are built in LDS where the eigenvector arrays go later; the 8 bytes per node that the
// expansion at the end needs of them (defl_record) go to the workspace: it keeps the mid class at 132 KiB, so that a
// workgroup of the training step (26 KiB) still fits on the same CU, and the small class at 50 KiB (3 per CU).
template <int kNMax, int kT, bool kGlobalA>
__host__ __device__ constexpr int direct_lds_bytes()
{
    return kGlobalA ? (kNMax > kGMax ? kBLds : kGLds)
                    : (int)(sizeof(float) * (6 * kNMax + 32 * kYld + kT + kNMax * (kNMax + 1) + kNMax * kYld)
                            + kNMax * (33 * 8 + 32));
}

struct TriLds {
    float *dg, *of, *of2, *tau;      // [kNMax] diagonal, off-diagonal, its square, reflector scales
    float *pbuf, *vbuf;              // [kNMax]
    float *coef;                     // [nv][ldy] Gram-Schmidt coefficients; column ldy - 1 = squared norm  (nv = ldy - 1 vectors)
    int *cnt;                        // [kT] Sturm counts of one bisection round
    float *Y;                        // [n'][ldy] eigenvectors
    int ldy;
    float *Ud, *Us;                  // [n'][bw + 1] LU factors of the current batch of bw inverse iterations
    uint8_t *Uf;                     // [n'][bw]
    int bw, ldu;
};

struct EigShared {                   // per-workgroup scratch of the solver core (a __shared__ object of the kernel)
    float lamv[kVecCap], shiftv[kVecCap], lo[kVecCap], hi[kVecCap];
    int cs[kVecCap], posi[kVecCap];
    int na, maxpos, bad;
    int diag_lost, diag_its;         // diagnostics of the last eig_top_vectors call
    int gs_lost;                     // result of the cluster sweep done by wave 0 (block classes)
    float gs_left;
};

// A (n x n, symmetric, both triangles kept, row stride lda) -> T = Q^T A Q; Q = H_0 H_1 ... H_{n-3},
// H_k = I - tau_k v_k v_k^T with v_k = (0, ..., 0, 1, A[k][k+2], ..., A[k][n-1]).  All threads call it; ends with a barrier.
// (Round 5: a fused variant -- every wave forms reflector k + 1 for itself from the row as it will be after the update, the rows are
//  multiplied with it while they are updated: one pass and one barrier per column instead of two -- was built, passed the strict tests
//  and measured 151 against 161 us per 65..128-class item and 283 against 268 us in the block class's Ritz problems
//  (scripts/gpu/r5_call7.sh, third run): with 16 waves on 4 SIMDs the redundant per-wave work costs what the barrier saved.  Not kept.)
template <int kCPL, int kT, int kR>
__device__ void tridiagonalize(float *A, int lda, int n, const TriLds &w)
{
    constexpr int kNW = kT / 64;                     // kR rows per wave are in flight (memory-level parallelism)
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int k = 0; k + 2 < n; ++k) {
        // every wave forms reflector k from row k (columns k+1 .. n-1): identical arithmetic, no barrier
        float v[kCPL];
        float sig = 0.f;
#pragma unroll
        for (int u = 0; u < kCPL; ++u) {
            const int c = lane + 64 * u;
            v[u] = (c > k && c < n) ? A[(int64_t)k * lda + c] : 0.f;
            sig += c > k + 1 ? v[u] * v[u] : 0.f;
        }
        sig = wave_sum(sig);
        const float x0 = A[(int64_t)k * lda + k + 1];
        if (sig <= 1e-30f) {                         // block-uniform: the column is already tridiagonal, H_k = I
            if (tid == 0) { w.dg[k] = A[(int64_t)k * lda + k]; w.of[k] = x0; w.tau[k] = 0.f; }
            continue;
        }
        const float mu = sqrtf(x0 * x0 + sig);
        const float beta = x0 > 0.f ? -mu : mu;
        const float t = (beta - x0) / beta;
        const float scale = 1.0f / (x0 - beta);
#pragma unroll
        for (int u = 0; u < kCPL; ++u) {
            const int c = lane + 64 * u;
            v[u] = c == k + 1 ? 1.0f : v[u] * scale;
        }
        if (wv == 0) {
#pragma unroll
            for (int u = 0; u < kCPL; ++u)
                if (lane + 64 * u < n) w.vbuf[lane + 64 * u] = v[u];
        }
        // p = tau A v on the trailing block
        for (int i0 = k + 1 + wv * kR; i0 < n; i0 += kNW * kR) {
            float s[kR];
#pragma unroll
            for (int r = 0; r < kR; ++r) {
                s[r] = 0.f;
                const int i = i0 + r < n ? i0 + r : n - 1;
#pragma unroll
                for (int u = 0; u < kCPL; ++u) {
                    const int c = lane + 64 * u;
                    s[r] += (c > k && c < n) ? A[(int64_t)i * lda + c] * v[u] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < kR; ++r) {
                const float sr = wave_sum(s[r]);
                if (lane == 0 && i0 + r < n) w.pbuf[i0 + r] = t * sr;
            }
        }
        __syncthreads();
        if (wv == 0) {                               // row k is dead now: it stores the reflector
#pragma unroll
            for (int u = 0; u < kCPL; ++u) {
                const int c = lane + 64 * u;
                if (c > k + 1 && c < n) A[(int64_t)k * lda + c] = v[u];
            }
            if (lane == 0) { w.dg[k] = A[(int64_t)k * lda + k]; w.of[k] = beta; w.tau[k] = t; }
        }
        float pc[kCPL], pv = 0.f;
#pragma unroll
        for (int u = 0; u < kCPL; ++u) {
            const int c = lane + 64 * u;
            pc[u] = (c > k && c < n) ? w.pbuf[c] : 0.f;
            pv += pc[u] * v[u];
        }
        const float K = 0.5f * t * wave_sum(pv);
#pragma unroll
        for (int u = 0; u < kCPL; ++u) pc[u] -= K * v[u];          // w = p - K v
        for (int i0 = k + 1 + wv * kR; i0 < n; i0 += kNW * kR) {    // A -= v w^T + w v^T
            float arow[kR][kCPL];
#pragma unroll
            for (int r = 0; r < kR; ++r) {
                const int i = i0 + r < n ? i0 + r : n - 1;
#pragma unroll
                for (int u = 0; u < kCPL; ++u) {
                    const int c = lane + 64 * u;
                    arow[r][u] = (c > k && c < n) ? A[(int64_t)i * lda + c] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < kR; ++r) {
                const int i = i0 + r;
                if (i >= n) continue;
                const float vi = w.vbuf[i], wi = w.pbuf[i] - K * vi;
#pragma unroll
                for (int u = 0; u < kCPL; ++u) {
                    const int c = lane + 64 * u;
                    if (c > k && c < n) A[(int64_t)i * lda + c] = arow[r][u] - (vi * pc[u] + wi * v[u]);
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (n >= 2) { w.dg[n - 2] = A[(int64_t)(n - 2) * lda + n - 2]; w.of[n - 2] = A[(int64_t)(n - 2) * lda + n - 1]; }
        w.dg[n - 1] = A[(int64_t)(n - 1) * lda + n - 1];
        w.of[n - 1] = 0.f;
    }
    __syncthreads();
}

// The same reduction streaming ONLY THE LOWER TRIANGLE (row i: columns <= i): half the memory traffic, which is what
// bounds the workspace-resident class.  p = A v is assembled from the row part (sum over c <= i of A[i][c] v_c) and the
// column part (row i adds A[i][c] v_i to p_c for c < i; per-wave partial sums in `slab`, combined in wave order).
// Column k of the trailing matrix (the next reflector's input) is captured into xcol while the rows are updated, so
// no strided column read is needed.  Row k's (unused) upper part stores reflector k as in tridiagonalize().
// slab: LDS [kT / 64][ldslab]; xcol: LDS [n].  All threads call it; ends with a barrier.
template <int kCPL, int kT, int kR>
__device__ void tridiagonalize_lower(float *A, int lda, int n, const TriLds &w, float *slab, int ldslab, float *xcol)
{
    constexpr int kNW = kT / 64;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < n; i += kT) xcol[i] = i >= 1 ? A[(int64_t)i * lda] : 0.f;        // column 0 below the diagonal
    __syncthreads();
    for (int k = 0; k + 2 < n; ++k) {
        float v[kCPL];
        float sig = 0.f;
#pragma unroll
        for (int u = 0; u < kCPL; ++u) {
            const int c = lane + 64 * u;
            v[u] = (c > k && c < n) ? xcol[c] : 0.f;
            sig += c > k + 1 ? v[u] * v[u] : 0.f;
        }
        sig = wave_sum(sig);
        const float x0 = xcol[k + 1];
        if (sig <= 1e-30f) {                         // block-uniform: the column is already tridiagonal, H_k = I
            if (tid == 0) { w.dg[k] = A[(int64_t)k * lda + k]; w.of[k] = x0; w.tau[k] = 0.f; }
            __syncthreads();                         // xcol is rewritten below
            for (int i = k + 2 + tid; i < n; i += kT) xcol[i] = A[(int64_t)i * lda + k + 1];
            __syncthreads();
            continue;
        }
        const float mu = sqrtf(x0 * x0 + sig);
        const float beta = x0 > 0.f ? -mu : mu;
        const float t = (beta - x0) / beta;
        const float scale = 1.0f / (x0 - beta);
#pragma unroll
        for (int u = 0; u < kCPL; ++u) {
            const int c = lane + 64 * u;
            v[u] = c == k + 1 ? 1.0f : v[u] * scale;
        }
        if (wv == 0) {
#pragma unroll
            for (int u = 0; u < kCPL; ++u)
                if (lane + 64 * u < n) w.vbuf[lane + 64 * u] = v[u];
        }
        __syncthreads();                             // vbuf visible (the column part needs v_i of other rows)
        float colacc[kCPL];
#pragma unroll
        for (int u = 0; u < kCPL; ++u) colacc[u] = 0.f;
        for (int i0 = k + 1 + wv * kR; i0 < n; i0 += kNW * kR) {
            float s[kR];
#pragma unroll
            for (int r = 0; r < kR; ++r) {
                s[r] = 0.f;
                const int i = i0 + r < n ? i0 + r : n - 1;
                const float vi = i0 + r < n ? w.vbuf[i] : 0.f;
#pragma unroll
                for (int u = 0; u < kCPL; ++u) {
                    const int c = lane + 64 * u;
                    const float aic = (c > k && c <= i) ? A[(int64_t)i * lda + c] : 0.f;
                    s[r] = fmaf(aic, v[u], s[r]);
                    colacc[u] = fmaf(c < i ? aic : 0.f, vi, colacc[u]);
                }
            }
#pragma unroll
            for (int r = 0; r < kR; ++r) {
                const float sr = wave_sum(s[r]);
                if (lane == 0 && i0 + r < n) w.pbuf[i0 + r] = sr;        // row part
            }
        }
#pragma unroll
        for (int u = 0; u < kCPL; ++u)
            if (lane + 64 * u < n) slab[wv * ldslab + lane + 64 * u] = colacc[u];
        __syncthreads();
        for (int c = k + 1 + tid; c < n; c += kT) {   // p = tau (row part + column parts in wave order)
            float pc = w.pbuf[c];
            for (int q = 0; q < kNW; ++q) pc += slab[q * ldslab + c];
            w.pbuf[c] = t * pc;
        }
        if (wv == 0) {                               // row k is dead now: its upper part stores the reflector
#pragma unroll
            for (int u = 0; u < kCPL; ++u) {
                const int c = lane + 64 * u;
                if (c > k + 1 && c < n) A[(int64_t)k * lda + c] = v[u];
            }
            if (lane == 0) { w.dg[k] = A[(int64_t)k * lda + k]; w.of[k] = beta; w.tau[k] = t; }
        }
        __syncthreads();
        float pc[kCPL], pv = 0.f;
#pragma unroll
        for (int u = 0; u < kCPL; ++u) {
            const int c = lane + 64 * u;
            pc[u] = (c > k && c < n) ? w.pbuf[c] : 0.f;
            pv += pc[u] * v[u];
        }
        const float K = 0.5f * t * wave_sum(pv);
#pragma unroll
        for (int u = 0; u < kCPL; ++u) pc[u] -= K * v[u];          // w = p - K v
        for (int i0 = k + 1 + wv * kR; i0 < n; i0 += kNW * kR) {    // A -= v w^T + w v^T on the lower triangle
            float arow[kR][kCPL];
#pragma unroll
            for (int r = 0; r < kR; ++r) {
                const int i = i0 + r < n ? i0 + r : n - 1;
#pragma unroll
                for (int u = 0; u < kCPL; ++u) {
                    const int c = lane + 64 * u;
                    arow[r][u] = (c > k && c <= i) ? A[(int64_t)i * lda + c] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < kR; ++r) {
                const int i = i0 + r;
                if (i >= n) continue;
                const float vi = w.vbuf[i], wi = w.pbuf[i] - K * vi;
#pragma unroll
                for (int u = 0; u < kCPL; ++u) {
                    const int c = lane + 64 * u;
                    if (c > k && c <= i) {
                        const float nv = arow[r][u] - (vi * pc[u] + wi * v[u]);
                        A[(int64_t)i * lda + c] = nv;
                        if (c == k + 1) xcol[i] = nv;               // column k + 1: the next reflector's input
                    }
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (n >= 2) { w.dg[n - 2] = A[(int64_t)(n - 2) * lda + n - 2]; w.of[n - 2] = A[(int64_t)(n - 1) * lda + n - 2]; }
        w.dg[n - 1] = A[(int64_t)(n - 1) * lda + n - 1];
        w.of[n - 1] = 0.f;
    }
    __syncthreads();
}

// number of eigenvalues of T below x (Sturm sequence of the LDL^T pivots, as LAPACK's dlaebz)
__device__ __forceinline__ int sturm_count(const float *dg, const float *of2, int n, float x)
{
    float q = dg[0] - x;
    if (fabsf(q) < 1e-30f) q = -1e-30f;
    int c = q < 0.f ? 1 : 0;
    for (int i = 1; i < n; ++i) {                    // counts tolerate the 1-ulp reciprocal
        q = dg[i] - x - of2[i - 1] * fast_rcp(q);
        if (fabsf(q) < 1e-30f) q = -1e-30f;
        c += q < 0.f ? 1 : 0;
    }
    return c;
}

__device__ __forceinline__ float hash_unit(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return (float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f;
}

// one inverse-iteration step for eigenvector j (called by ONE thread, slot jl of the LU batch): solve
// (T - shift) x = y in place in column j of Y, by Gaussian elimination with partial pivoting fused with the
// right-hand side; x is normalised.  Returns false if the solution is not finite.
template <bool kDeep = false>
__device__ bool inverse_iteration_step(const TriLds &w, int n, int j, int jl, float shift, bool random_rhs, uint32_t hseed)
{
    float *Y = w.Y + j, *Ud = w.Ud + jl, *Us = w.Us + jl;
    uint8_t *Uf = w.Uf + jl;
    const int ldu = w.ldu, ldf = w.bw, ldy = w.ldy;
    float cd = w.dg[0] - shift, cs = n > 1 ? w.of[0] : 0.f;
    float cy = random_rhs ? hash_unit(hseed, (uint32_t)j, 0u) : Y[0];
    // forward elimination, branch free: row i is either the running row (cd, cs, 0 | cy) or, when the sub-diagonal
    // entry is larger, the next row of T - shift (sub, nd, ns | by) and the running row is eliminated instead.
    // The operands of row i + 2 are requested BEFORE row i's dependent arithmetic (round 5; the compiler cannot move the loads
    // over the loop's own stores: Y is read and written).  Measured: NO gain -- 'invit' of the 65..128 class 86 against 83 us
    // per item (scripts/gpu/r5_call7.sh) --, i.e. a row's ~240 cycles are the lone half-wave's own instruction issue (7 LDS
    // operations + ~15 dependent VALU operations per row), not the latency of its loads.  Kept: same arithmetic, same results.
    float sub = cs, nd = n > 1 ? w.dg[1] - shift : 0.f, ns = n > 2 ? w.of[1] : 0.f;
    float by = n > 1 ? (random_rhs ? hash_unit(hseed, (uint32_t)j, 1u) : Y[ldy]) : 0.f;
#pragma unroll 4
    for (int i = 0; i + 1 < n; ++i) {
        float nd2 = 0.f, ns2 = 0.f, by2 = 0.f;               // row i + 2 (the next iteration's "next row")
        if (i + 2 < n) {
            nd2 = w.dg[i + 2] - shift;
            ns2 = i + 3 < n ? w.of[i + 2] : 0.f;
            by2 = random_rhs ? hash_unit(hseed, (uint32_t)j, (uint32_t)(i + 2)) : Y[(i + 2) * ldy];
        }
        const bool swap = fabsf(cd) < fabsf(sub);
        const float piv = swap ? sub : cd, oth = swap ? cd : sub;
        const float mult = piv != 0.f ? oth * fast_rcp(piv) : 0.f;
        Ud[i * ldu] = piv;
        Us[i * ldu] = swap ? nd : cs;
        Uf[i * ldf] = swap ? 1 : 0;
        Y[i * ldy] = swap ? by : cy;
        const float ncd = swap ? cs - mult * nd : nd - mult * cs;
        const float ncs = swap ? -mult * ns : ns;
        const float ncy = swap ? cy - mult * by : by - mult * cy;
        cd = ncd; cs = ncs; cy = ncy;
        sub = ns;                                            // of[i + 1]: the sub-diagonal entry under the new running row (i + 2 < n here whenever it is used)
        nd = nd2; ns = ns2; by = by2;
    }
    Ud[(n - 1) * ldu] = cd; Us[(n - 1) * ldu] = 0.f; Uf[(n - 1) * ldf] = 0; Y[(n - 1) * ldy] = cy;
    float x1 = 0.f, x2 = 0.f, ss = 0.f;
    // back substitution, the next row's operands requested ahead in the same way
    float d = cd, us = 0.f, s2 = 0.f, yi = cy;               // row n - 1
    if constexpr (kDeep) {
        // the factors live in the workspace (two- / four-wave teams): a row's operands come from L2, so they are requested FOUR rows (one
        // chunk) ahead; same arithmetic in the same order
        float dq[4], uq[4];
        uint8_t fq[4];
        auto fetch = [&](int top) {                          // rows top, top - 1, .. top - 3 (clamped; rows below 0 are not used)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ii = top - u > 0 ? top - u : 0;
                dq[u] = Ud[ii * ldu]; uq[u] = Us[ii * ldu]; fq[u] = Uf[ii * ldf];
            }
        };
        fetch(n - 1);
        for (int top = n - 1; top >= 0; top -= 4) {
            float dc[4], uc[4];
            uint8_t fc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { dc[u] = dq[u]; uc[u] = uq[u]; fc[u] = fq[u]; }
            if (top >= 4) fetch(top - 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = top - u;
                if (i >= 0) {
                    float dd = dc[u];
                    const float s2c = (fc[u] && i + 2 < n) ? w.of[i + 1] : 0.f;
                    if (fabsf(dd) < kPivTiny) dd = dd < 0.f ? -kPivTiny : kPivTiny;
                    const float x = (Y[i * ldy] - uc[u] * x1 - s2c * x2) * fast_rcp(dd);
                    Y[i * ldy] = x;
                    x2 = x1; x1 = x;
                    ss = fmaf(x, x, ss);
                }
            }
        }
    } else
#pragma unroll 4
    for (int i = n - 1; i >= 0; --i) {
        float dn = 0.f, usn = 0.f, s2n = 0.f, yn = 0.f;      // row i - 1
        if (i > 0) {
            dn = Ud[(i - 1) * ldu];
            usn = Us[(i - 1) * ldu];
            s2n = (Uf[(i - 1) * ldf] && i + 1 < n) ? w.of[i] : 0.f;
            yn = Y[(i - 1) * ldy];
        }
        if (fabsf(d) < kPivTiny) d = d < 0.f ? -kPivTiny : kPivTiny;
        const float x = (yi - us * x1 - s2 * x2) * fast_rcp(d);
        Y[i * ldy] = x;
        x2 = x1; x1 = x;
        ss = fmaf(x, x, ss);
        d = dn; us = usn; s2 = s2n; yi = yn;
    }
    const bool ok = ss > 0.f && ss < 3.0e38f;
    const float inv = ok ? 1.0f / sqrtf(ss) : 0.f;
#pragma unroll 4
    for (int i = 0; i < n; ++i) Y[i * ldy] *= inv;
    return ok;
}

// Gram-Schmidt inside every cluster of close eigenvalues; the t-th members of all clusters are processed together
// (2 barriers per t).  Vectors stay un-normalised while the sweep runs (projections divide by the squared norms kept
// in coef[.][32]); a second pass follows only where the first one removed more than half of a vector ("twice is
// enough", Kahan / Parlett).  Returns (block-uniform) the number of vectors that vanished, i.e. were in the span of
// their predecessors.  All threads call it; ends with a barrier.
constexpr int kMaxInvIt = 12;         // solve + sweep rounds of eig_top_vectors (3 as a rule)
constexpr float kVanish = 1e-4f;     // squared norm left of a unit vector below which it counts as "in the span of its predecessors"
constexpr float kHeavy = 1e-2f;      // ... below which what is left is too noisy to be final

// Returns (block-uniform) the number of vectors that vanished, i.e. were in the span of their predecessors; those are
// refilled with fresh pseudo-random numbers (LAPACK stein restarts them the same way) and *min_left is the smallest
// squared norm any member kept (1 = nothing removed).  All threads call it; ends with a barrier.
// (one-wave solver core, further below)
struct WaveTri {
    float *dg, *of, *of2, *tau;      // [kNMax] diagonal, off-diagonal, its square, reflector scales
    float *nrm;                      // [64] squared norms inside the Gram-Schmidt sweep
    float *Y;                        // [n'][ldy] eigenvectors
    int ldy;
};

// cluster_orthonormalize() for one wave: lane = rows lane, lane + 64, ... of Y (n <= 64 kCPL).  Members are taken in index
// order (the members of a cluster are consecutive and its predecessors finished), projections on all predecessors are
// computed from the same vector (classical Gram-Schmidt, a second pass where the first removed more than half: "twice is
// enough").  No barrier: a sweep over a cluster of m copies is m^2 / 2 wave reductions instead of 3 m workgroup barriers.
template <int kCPL>
__device__ __forceinline__ int wave_cluster_orthonormalize(const WaveTri &w, int n, int na, const int *cs, const int *posi, int maxpos,
                                                           float *min_left, uint32_t hseed)
{
    *min_left = 1.0f;
    if (maxpos == 0) return 0;
    const int lane = lane_id();
    float *Y = w.Y;
    const int ldy = w.ldy;
    if (lane < na) w.nrm[lane] = 1.0f;                    // unit vectors come out of the solves
    wave_sync();
    int lost = 0;
    float left = 1.0f;
    for (int j = 0; j < na; ++j) {
        if (posi[j] == 0) continue;                       // wave-uniform (LDS broadcast)
        const int c0 = cs[j];
        float y[kCPL];
#pragma unroll
        for (int u = 0; u < kCPL; ++u) y[u] = lane + 64 * u < n ? Y[(lane + 64 * u) * ldy + j] : 0.f;
        float now = 1.0f;
        for (int pass = 0; pass < 2; ++pass) {
            float acc[kCPL];
#pragma unroll
            for (int u = 0; u < kCPL; ++u) acc[u] = 0.f;
            for (int l = c0; l < j; ++l) {
                const float nl = w.nrm[l];
                float yl[kCPL], d = 0.f;
#pragma unroll
                for (int u = 0; u < kCPL; ++u) {
                    yl[u] = lane + 64 * u < n ? Y[(lane + 64 * u) * ldy + l] : 0.f;
                    d = fmaf(y[u], yl[u], d);
                }
                const float s = wave_sum(d);
                const float cf = nl >= kVanish ? s / nl : 0.f;   // a vanished predecessor spans nothing
#pragma unroll
                for (int u = 0; u < kCPL; ++u) acc[u] = fmaf(cf, yl[u], acc[u]);
            }
            float d = 0.f;
#pragma unroll
            for (int u = 0; u < kCPL; ++u) { y[u] -= acc[u]; d = fmaf(y[u], y[u], d); }
            now = wave_sum(d);
            if (!(pass == 0 && now < 0.5f)) break;
        }
#pragma unroll
        for (int u = 0; u < kCPL; ++u)
            if (lane + 64 * u < n) Y[(lane + 64 * u) * ldy + j] = y[u];
        if (lane == 0) w.nrm[j] = now;
        left = now < left ? now : left;
        if (now < kVanish) ++lost;
        wave_sync();
    }
    for (int j = 0; j < na; ++j) {                        // normalise; vanished members restart from pseudo-random numbers
        if (posi[j] == 0) continue;
        const float c = w.nrm[j];
#pragma unroll
        for (int u = 0; u < kCPL; ++u) {
            const int r = lane + 64 * u;
            if (r < n) {
                if (c < kVanish) Y[r * ldy + j] = hash_unit(hseed, (uint32_t)j, (uint32_t)r);
                else Y[r * ldy + j] *= 1.0f / sqrtf(c);
            }
        }
    }
    wave_sync();
    *min_left = left;
    return lost;
}

template <int kT>
__device__ int cluster_orthonormalize(const TriLds &w, int n, int na, const int *cs, const int *posi, int maxpos,
                                      float *min_left, uint32_t hseed)
{
    *min_left = 1.0f;
    constexpr int kNW = kT / 64;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float *Y = w.Y, *coef = w.coef;
    const int ldy = w.ldy, nc = w.ldy - 1;                  // coefficient rows have stride ldy, the norms sit in column nc
    if (maxpos == 0) return 0;
    if (tid < na) coef[tid * ldy + nc] = 1.0f;      // squared norms: unit vectors come out of the solves
    __syncthreads();
    int lost = 0;
    float left = 1.0f;
    for (int t = 1; t <= maxpos; ++t) {
        for (int pass = 0; pass < 2; ++pass) {
            // projections on the predecessors in the cluster (a vanished predecessor spans nothing)
            for (int j = 0; j < na; ++j) {
                if (posi[j] != t) continue;
                for (int l = cs[j] + wv; l < j; l += kNW) {
                    float s = 0.f;
                    for (int r = lane; r < n; r += 64) s = fmaf(Y[r * ldy + j], Y[r * ldy + l], s);
                    s = wave_sum(s);
                    const float nl = coef[l * ldy + nc];
                    if (lane == 0) coef[j * ldy + l] = nl >= kVanish ? s / nl : 0.f;
                }
            }
            __syncthreads();
            float *part_sq = (float *)w.cnt;                           // [vector][wave] partial squared norms
            for (int j = 0; j < na; ++j) {
                if (posi[j] != t) continue;
                float part = 0.f;
                for (int i = tid; i < n; i += kT) {
                    float acc = 0.f;
                    for (int l = cs[j]; l < j; ++l) acc = fmaf(coef[j * ldy + l], Y[i * ldy + l], acc);
                    const float y = Y[i * ldy + j] - acc;
                    Y[i * ldy + j] = y;
                    part = fmaf(y, y, part);
                }
                part = wave_sum(part);
                if (lane == 0) part_sq[j * kNW + wv] = part;
            }
            __syncthreads();
            // every wave adds the partials up in the same order (lane = vector): identical, deterministic results
            float now = 0.f;
            const bool mine = lane < na && posi[lane] == t;
            if (mine)
                for (int q = 0; q < kNW; ++q) now += part_sq[lane * kNW + q];
            // (the vectors enter with unit norm: "before" is 1 in the first pass; nobody reads coef[.][32] of step t here)
            const bool again = pass == 0 && wave_ballot(mine && now < 0.5f) != 0ull;
            if (wv == 0 && mine) coef[lane * ldy + nc] = now;
#ifdef GCC_POSEMB_DEVDEBUG
            if (wv == 0 && mine) printf("gs t=%d pass=%d j=%d now=%.6f\n", t, pass, lane, now);
#endif
            if (!again) break;
        }
        __syncthreads();
        for (int j = 0; j < na; ++j) {
            if (posi[j] != t) continue;
            const float c = coef[j * ldy + nc];
            left = c < left ? c : left;
            if (c < kVanish) ++lost;
        }
    }
    // normalise the members of the clusters; vanished ones restart from pseudo-random numbers
    for (int j = 0; j < na; ++j) {
        if (posi[j] == 0) continue;
        const float c = coef[j * ldy + nc];
        if (c < kVanish) {
            for (int i = tid; i < n; i += kT) Y[i * ldy + j] = hash_unit(hseed, (uint32_t)j, (uint32_t)i);
        } else {
            const float inv = 1.0f / sqrtf(c);
            for (int i = tid; i < n; i += kT) Y[i * ldy + j] *= inv;
        }
    }
    __syncthreads();
    *min_left = left;
    return lost;
}

// ---- the kq largest eigenvalues of T: eigenvalue j (descending) has ascending index nr - 1 - j and lies in [lo, hi]
// with count(lo) <= nr - 1 - j < count(hi); every round probes kT / kVec interior points per eigenvalue.  On exit
// es.lamv[0..kq) holds them in descending order.  The spectrum must lie inside (-1.001, 1.001) (normalised adjacency
// matrices and their Rayleigh-Ritz projections).  All threads call it; ends with a barrier.
template <int kT, int kVec>
__device__ void eig_top_values(const TriLds &w, int nr, int kq, EigShared &es)
{
    constexpr int kP = kT / kVec;
    constexpr int kRounds = kP >= 32 ? 6 : (kP >= 16 ? 7 : 9);         // 2.002 / (kP + 1)^rounds < 1e-8
    const int tid = (int)threadIdx.x;
    for (int i = tid; i < nr; i += kT) w.of2[i] = w.of[i] * w.of[i];
    if (tid < kVec) { es.lo[tid] = -1.001f; es.hi[tid] = 1.001f; }
    __syncthreads();
    const int j = tid / kP, ip = tid - j * kP;
    for (int round = 0; round < kRounds; ++round) {
        const float xlo = es.lo[j], xhi = es.hi[j];
        const float step = (xhi - xlo) * (1.0f / (float)(kP + 1));
        const float x = xlo + step * (float)(ip + 1);
        const int c = j < kq ? sturm_count(w.dg, w.of2, nr, x) : 0;
        w.cnt[tid] = c;
        __syncthreads();
        if (j < kq) {
            const int tgt = nr - 1 - j;
            const bool above = c > tgt;
            const bool prev_above = ip > 0 && w.cnt[tid - 1] > tgt;
            if (above && !prev_above) {
                es.hi[j] = x;
                if (ip > 0) es.lo[j] = xlo + step * (float)ip;
            }
            if (ip == kP - 1 && !above) es.lo[j] = x;
        }
        __syncthreads();
    }
    if (tid == 0) {
        for (int q = 0; q < kq; ++q) {
            float l = 0.5f * (es.lo[q] + es.hi[q]);
            if (q > 0 && l > es.lamv[q - 1]) l = es.lamv[q - 1];
            es.lamv[q] = l;
        }
        es.bad = 0;
    }
    __syncthreads();
}

__device__ __forceinline__ void phase_tick(long long *row, int ph, long long &tick)
{
    if (row && threadIdx.x == 0) {
        const long long now = device_ticks();
        atomicAdd((unsigned long long *)&row[ph], (unsigned long long)(now - tick));
        tick = now;
    }
}

// ---- eigenvectors 0 .. na-1 (of the eigenvalues es.lamv, descending) of the matrix that tridiagonalize() reduced:
// inverse iteration on T (shifts at least kSep apart; three solves, Gram-Schmidt inside clusters of eigenvalues closer
// than kOrtol after the second and third), then x = H_0 ... H_{nr-3} y with one wave per pair of vectors, the vectors in
// registers.  Result in w.Y[i * ldy + j].  Returns (block-uniform) true if a vector could not be produced.
template <int kCPL, int kT, bool kPair = false>
__device__ bool eig_top_vectors(const float *A, int lda, int nr, int na, const TriLds &w, EigShared &es, uint32_t hseed,
                                long long *tick_row, long long &tick)
{
    static_assert(!kPair || ((kT == 128 || kT == 256) && kCPL == 2), "two- / four-wave teams");
    constexpr int kNW = kT / 64;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ldy = w.ldy;
    if (tid == 0) {
        // Orthogonalisation clusters: chains of eigenvalues closer than kOrtol.  Shifts: an eigenvalue keeps its own
        // value unless it belongs to a CLUMP (a numerically multiple eigenvalue: copies closer than kSep); the first
        // copy keeps the value, the others are displaced by multiples of kSep AWAY from the nearest other eigenvalue --
        // displaced towards it they amplify that neighbour more than their own eigenspace (device trace: 12 copies of
        // 1/sqrt(2) with an eigenvalue 3.5e-5 below; the last four members all converged to the neighbour's
        // eigenvector and cancelled each other in every sweep).  The step shrinks when there is little room.
        int maxpos = 0;
        for (int j = 0; j < na; ++j) {
            es.shiftv[j] = es.lamv[j];
            es.cs[j] = (j > 0 && es.lamv[j - 1] - es.lamv[j] <= kOrtol) ? es.cs[j - 1] : j;
            es.posi[j] = j - es.cs[j];
            maxpos = es.posi[j] > maxpos ? es.posi[j] : maxpos;
        }
        for (int j = 0; j < na;) {
            int b = j;
            while (b + 1 < na && es.lamv[b] - es.lamv[b + 1] < kSep) ++b;
            const int c = b - j + 1;
            if (c > 1) {
                float room_up = 2.0f;
                if (j > 0) room_up = fminf(es.lamv[j - 1], es.shiftv[j - 1]) - es.lamv[j];
                const float room_dn = b + 1 < na ? es.lamv[b] - es.lamv[b + 1] : (na >= nr ? 2.0f : 0.0f);   // below the computed ones: unknown
                const bool up = room_up >= room_dn;
                const float room = up ? room_up : room_dn;
                float step = kSep;
                if ((float)c * step > 0.5f * room) step = fmaxf(0.5f * room / (float)c, 2.5e-7f);
                for (int m = 1; m < c; ++m) es.shiftv[j + m] = up ? es.lamv[j] + (float)m * step : es.lamv[b] - (float)m * step;
            }
            j = b + 1;
        }
        es.maxpos = maxpos;
    }
    __syncthreads();
#ifdef GCC_POSEMB_DEVDEBUG
    if (tid == 0) {
        printf("eig nr=%d na=%d maxpos=%d\n", nr, na, es.maxpos);
        for (int j = 0; j < na; ++j) printf("  j=%d lam=%.7f shift=%.7f cs=%d posi=%d\n", j, es.lamv[j], es.shiftv[j], es.cs[j], es.posi[j]);
    }
#endif
    const int maxpos = es.maxpos;
    int lost = 0;
    // Three solves as a rule.  A Gram-Schmidt sweep that cancels most of a cluster member leaves the other eigenvectors'
    // admixture (one factor displacement / gap ~ 1e-2 per solve) magnified by the cancellation: the last of 16 copies of
    // 1/sqrt(2) in a hub ego-net came out with 1.5e-3 of an eigenvector 3e-3 away.  So a member that kept less than 1 %
    // of its squared norm buys one more solve + sweep round, and a vanished one (span of its predecessors: the 16 start
    // vectors of a 16-dimensional eigenspace are never well conditioned) restarts from random numbers and buys three --
    // LAPACK's stein iterates on the same criterion.
    int need_until = 2;
    for (int it = 0; it < kMaxInvIt; ++it) {
        // (one half wave runs the batch's 32 solves.  Dealing them out over all 16 waves -- two lanes each -- was measured and is
        //  SLOWER: 'invit' 86 -> 111 us per 65..128-class item (scripts/gpu/r5_call7.sh, second run): a row's cost is its ~7 LDS
        //  instructions, and one instruction serving 32 solves beats sixteen serving two each)
        const int slot = tid;
        for (int j0 = 0; j0 < na; j0 += w.bw) {
            if (slot < w.bw && j0 + slot < na) {
                // The head of a cluster keeps its own eigenvalue as shift.  When the cluster is a numerically multiple
                // eigenvalue (copies within a few ulp), T - shift is singular to working precision in several directions at
                // once and every further solve returns a DIFFERENT vector of that eigenspace (pivot clamping / rounding
                // decide): the members, orthonormal against the old head, then lose most of themselves against the new one,
                // round after round (device trace of a 54-node ego-net with six copies of 1/sqrt(2): the sixth member kept
                // 3e-4 of its squared norm in every sweep).  The same holds for the first copy of a second multiple eigenvalue
                // inside one chain of close eigenvalues.  Two solves make such a vector an eigenvector to 1e-14; it is frozen
                // from the third round on and the others converge against a fixed reference.
                const int j = j0 + slot;
                const bool frozen = it >= 2 && es.shiftv[j] == es.lamv[j] && j + 1 < na && es.lamv[j] - es.lamv[j + 1] < kSep;
                if (!frozen) {
                    const bool ok = inverse_iteration_step<kPair>(w, nr, j, slot, es.shiftv[j], it == 0, hseed);
                    if (!ok) es.bad = 1;
                }
            }
            __syncthreads();
        }
        phase_tick(tick_row, 3, tick);             // inverse iteration
#ifdef GCC_POSEMB_DEVDEBUG
        if (tid == 0 && (it == 1 || it == 2))
            for (int j = 10; j < 16 && j < na; ++j)
                for (int i = 0; i < nr; ++i) printf("P it=%d j=%d i=%d %.9g\n", it, j, i, w.Y[i * ldy + j]);
        __syncthreads();
#endif
        if (it > 0) {                              // the first solve only enters the cluster subspaces
            float left;
#ifdef GCC_POSEMB_BLOCK_GS
            lost = cluster_orthonormalize<kT>(w, nr, na, es.cs, es.posi, maxpos, &left, hseed ^ (0x51ED27u * (uint32_t)(it + 1)));
#else
            // The sweep is done by ONE wave without barriers (m^2 / 2 wave reductions for a cluster of m copies) while the
            // others wait: the block version above needs 3 workgroup barriers per cluster position and pass, which cost
            // 119 of the 484 us of a mid-class item (hub ego-nets carry clusters of 30-60 copies).
            if (wv == 0) {
                WaveTri ww;
                ww.Y = w.Y; ww.ldy = w.ldy; ww.nrm = w.coef; ww.dg = ww.of = ww.of2 = ww.tau = nullptr;
                float lf;
                const int ls = wave_cluster_orthonormalize<kCPL>(ww, nr, na, es.cs, es.posi, maxpos, &lf,
                                                                 hseed ^ (0x51ED27u * (uint32_t)(it + 1)));
                if (lane == 0) { es.gs_lost = ls; es.gs_left = lf; }
            }
            __syncthreads();
            lost = es.gs_lost;
            left = es.gs_left;
            __syncthreads();
#endif
            if (lost > 0) need_until = it + 3 > need_until ? it + 3 : need_until;
            else if (left < kHeavy) need_until = it + 1 > need_until ? it + 1 : need_until;
        }
        phase_tick(tick_row, 4, tick);
        if (tid == 0) { es.diag_lost = lost; es.diag_its = it + 1; }
#ifdef GCC_POSEMB_DEVDEBUG
        if (tid == 0) printf("it=%d lost=%d need_until=%d bad=%d\n", it, lost, need_until, es.bad);
        if (tid == 0 && it == 1)
            for (int j = 10; j < 16 && j < na; ++j)
                for (int l = 10; l < j; ++l) printf("C j=%d l=%d %.9g norm=%.9g\n", j, l, w.coef[j * ldy + l], w.coef[l * ldy + ldy - 1]);
        if (tid == 0 && it <= 1) {
            if (it == 0) for (int i = 0; i < nr; ++i) printf("T %d %.9g %.9g\n", i, w.dg[i], w.of[i]);
            for (int j = 10; j < 16 && j < na; ++j)
                for (int i = 0; i < nr; ++i) printf("Y it=%d j=%d i=%d %.9g\n", it, j, i, w.Y[i * ldy + j]);
        }
        __syncthreads();
#endif
        if (it >= need_until) break;
    }
    if constexpr (kPair) {
        // x = H_0 ... H_{nr-3} y for a two- / four-wave team: kLPV = kT / 32 lanes per vector (4 / 8) -- lane group q of a wave holds the rows
        // kLPV r + q of its vector in registers --, so a reflector costs (nr - kk) / kLPV multiply-adds per lane twice and 2 / 3 cross-lane
        // additions; the reflectors come from the workspace one step ahead, through a double-buffered LDS copy (one team barrier per
        // reflector).  (The version below gives a wave two vectors at a time: 8 passes over all reflectors with two full wave
        // reductions each, 152 us per item on two waves.)
        constexpr int kLPV = kT / 32, kVPW = 64 / kLPV, kR = 128 / kLPV;
        const int j = kVPW * wv + (lane & (kVPW - 1)), q = lane / kVPW;
        const bool act = j < na;
        float *Yj = w.Y + (act ? j : 0);
        float y[kR], vc[kR];
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const int c = kLPV * r + q;
            y[r] = (act && c < nr) ? Yj[c * ldy] : 0.f;
            vc[r] = 0.f;
        }
        auto fetch = [&](int kk) -> float { return (tid > kk + 1 && tid < nr) ? A[(int64_t)kk * lda + tid] : (tid == kk + 1 ? 1.0f : 0.f); };
        float nxt = (nr >= 3 && tid < 128) ? fetch(nr - 3) : 0.f;
        int par = 0;
        for (int kk = nr - 3; kk >= 0; --kk) {
            float *vb = w.pbuf + par * 128;              // (pbuf and vbuf are adjacent: 2 x 128 floats)
            if (tid < 128) vb[tid] = nxt;
            __syncthreads();
            if (kk > 0 && tid < 128) nxt = fetch(kk - 1);
            par ^= 1;
            const float t = w.tau[kk];
            if (t == 0.f) continue;                      // uniform
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < kR / 4; ++g) {
                if (4 * kLPV * g + 4 * kLPV - 1 > kk) {  // uniform: the rows of these four registers reach beyond kk
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        vc[4 * g + u] = vb[kLPV * (4 * g + u) + q];
                        s = fmaf(vc[4 * g + u], y[4 * g + u], s);
                    }
                }
            }
#pragma unroll
            for (int m = kVPW; m < 64; m <<= 1) s += wave_shfl_xor(s, m);
            s *= t;
#pragma unroll
            for (int g = 0; g < kR / 4; ++g) {
                if (4 * kLPV * g + 4 * kLPV - 1 > kk) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) y[4 * g + u] = fmaf(-s, vc[4 * g + u], y[4 * g + u]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const int c = kLPV * r + q;
            if (act && c < nr) Yj[c * ldy] = y[r];
        }
    } else
    for (int j = 2 * wv; j < na; j += 2 * kNW) {
        const bool two = j + 1 < na;
        float y0[kCPL], y1[kCPL], v[kCPL], vn[kCPL];
#pragma unroll
        for (int u = 0; u < kCPL; ++u) {
            const int c = lane + 64 * u;
            y0[u] = c < nr ? w.Y[c * ldy + j] : 0.f;
            y1[u] = (c < nr && two) ? w.Y[c * ldy + j + 1] : 0.f;
            vn[u] = (nr >= 3 && c > nr - 2 && c < nr) ? A[(int64_t)(nr - 3) * lda + c] : 0.f;
        }
        for (int kk = nr - 3; kk >= 0; --kk) {
#pragma unroll
            for (int u = 0; u < kCPL; ++u) {
                const int c = lane + 64 * u;
                v[u] = c == kk + 1 ? 1.0f : (c > kk + 1 ? vn[u] : 0.f);
                vn[u] = (kk > 0 && c > kk && c < nr) ? A[(int64_t)(kk - 1) * lda + c] : 0.f;   // prefetch reflector kk - 1
            }
            const float t = w.tau[kk];
            if (t == 0.f) continue;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int u = 0; u < kCPL; ++u) { s0 += v[u] * y0[u]; s1 += v[u] * y1[u]; }
            s0 = t * wave_sum(s0);
            s1 = t * wave_sum(s1);
#pragma unroll
            for (int u = 0; u < kCPL; ++u) { y0[u] -= s0 * v[u]; y1[u] -= s1 * v[u]; }
        }
#pragma unroll
        for (int u = 0; u < kCPL; ++u) {
            const int c = lane + 64 * u;
            if (c < nr) {
                w.Y[c * ldy + j] = y0[u];
                if (two) w.Y[c * ldy + j + 1] = y1[u];
            }
        }
    }
    __syncthreads();
    phase_tick(tick_row, 5, tick);                 // back-transformation
    return lost > 0 || es.bad != 0;
}

// ---- two-wave teams (65 <= n' <= 128): tridiagonalize() for a workgroup of TWO waves with the matrix in REGISTERS -- thread r owns
// row r (128 floats).  The 16-wave version above is bound by its two barriers and ~6 LDS round trips per column while it owns a whole
// CU (1,024 threads at 128 registers, 132 KiB of LDS); here a column costs two 2-wave barriers, the FLOPs run out of registers with
// the other operand broadcast from LDS (16 bytes per instruction), and a workgroup takes 128 threads and kPairLds of LDS, so that
// four of them share a CU.  A lives in the workgroup's workspace slot (L2): read once (by columns -- A is symmetric -- so that the
// read is coalesced), its rows then receive the reflectors as in the other versions.
//   column k:  the owners of rows k and k + 1 put them into LDS                                  | barrier
//              all: sigma, x0, a_kk from row k (a 2 x 64 wave reduction, the same in both waves); v_r; S_r = sum_{c>k+1} a_rc row_k[c];
//              p_r = tau (a_{r,k+1} + scale S_r)  [a_{r,k+1} = row_{k+1}[r]];  p_r, v_r and the wave's part of p.v into LDS  | barrier
//              K = tau/2 p.v;  a_rc -= v_r p_c + (p_r - 2 K v_r) v_c   [= v_r w_c + w_r v_c with w = p - K v]
#ifndef GCC_POSEMB_PAIR_THREADS
#define GCC_POSEMB_PAIR_THREADS 256  // 128: two waves, thread = row (tridiagonalize_pair); 256: four waves, thread = half a row (tridiagonalize_quad)
#endif
#ifndef GCC_POSEMB_QUAD_OCC
#define GCC_POSEMB_QUAD_OCC 3        // waves per SIMD the four-wave kernel is compiled for: 3 = 168 registers, three workgroups per CU (423 us per item,
                                     // 0.827 ms per step); 4 = 128 registers with ~50 spilled values (467 us, 0.830); 2 = 208 registers (424 us, 0.837): scripts/gpu/r5_call24.sh, r5_call25.sh
#endif
constexpr int kPairT = GCC_POSEMB_PAIR_THREADS;
static_assert(kPairT == 128 || kPairT == 256, "two- or four-wave teams");
constexpr int kPairLds = (kPairT == 128 ? 32 : 33) * 1024;   // dynamic LDS of such a workgroup (+ ~2 KiB static): the deflation tables (24 KiB), later the eigenvectors (17 KiB)
constexpr int kPairSlotFloats = 128 * 128 + 2048 + 2 * 128 * 33 + 1024 + 64;   // matrix / reflectors | expansion records (8 bytes per node) | LU factors
template <int kNMax>
__device__ void tridiagonalize_pair(float *A, int lda, int n, const TriLds &w, float *xb /* LDS, 16-byte aligned, 6 kNMax + 8 floats */)
{
    static_assert(kNMax == 128, "two waves, one row per thread");
    constexpr int kB = kNMax / 8, kG = kNMax / 32;
    const int r = (int)threadIdx.x, lane = r & 63, h = r >> 6;
    float a[kNMax];
    {   // unconditional loads (the slot holds kNMax rows) and a bit mask: a select is turned back into a branch per element, and
        // addresses that do not depend on the item are hoisted out of the item loop -- 128 of them -- and spilled
        uint64_t ap = (uint64_t)(A + r);
        opaque_u64(ap);
        const float *Ar = (const float *)ap;
        const uint32_t mr = r < n ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int c = 0; c < kNMax; ++c) a[c] = __uint_as_float(__float_as_uint(Ar[c * lda]) & (c < n ? mr : 0u));
    }
    __syncthreads();                                     // all rows are in registers: A's rows may now receive the reflectors
    float *pv = xb + 4 * kNMax, *vv = xb + 5 * kNMax, *kpart = xb + 6 * kNMax;
    const int ncols = (n + 31) & ~31;                    // the readers take groups of 32 columns: zeros up to the group's end
    auto put_row = [&](float *dst, int from) {           // one lane: its row, blocks of eight columns from `from` on
#pragma unroll
        for (int cb = 0; cb < kB; ++cb) {
            if (8 * cb + 7 >= from && 8 * cb < ncols) {
                *(float4 *)(dst + 8 * cb) = make_float4(a[8 * cb], a[8 * cb + 1], a[8 * cb + 2], a[8 * cb + 3]);
                *(float4 *)(dst + 8 * cb + 4) = make_float4(a[8 * cb + 4], a[8 * cb + 5], a[8 * cb + 6], a[8 * cb + 7]);
            }
        }
    };
#pragma unroll 1
    for (int k = 0; k + 2 < n; ++k) {
        float *rk = xb + (k & 1) * 2 * kNMax, *rk1 = rk + kNMax;     // (two buffers: a column without a reflector has one barrier only)
        if (h == (k >> 6) && lane == (k & 63)) put_row(rk, k);
        if (h == ((k + 1) >> 6) && lane == ((k + 1) & 63)) put_row(rk1, k);
        __syncthreads();
        const float c0 = lane < n ? rk[lane] : 0.f, c1 = lane + 64 < n ? rk[64 + lane] : 0.f;   // (blocks of columns beyond n are not written)
        const float sig = wave_sum((lane > k + 1 ? c0 * c0 : 0.f) + (lane + 64 > k + 1 ? c1 * c1 : 0.f));
        const float x0 = rk[k + 1], akk = rk[k];
        if (sig <= 1e-30f) {                             // uniform over the workgroup: the column is already tridiagonal, H_k = I
            if (r == 0) { w.dg[k] = akk; w.of[k] = x0; w.tau[k] = 0.f; }
            continue;
        }
        const float mu = sqrtf(x0 * x0 + sig);
        const float beta = x0 > 0.f ? -mu : mu;
        const float t = (beta - x0) / beta;
        const float scale = 1.0f / (x0 - beta);
        const float vr = r == k + 1 ? 1.0f : (r > k + 1 ? (h ? c1 : c0) * scale : 0.f);
        // (groups of 32 columns under one uniform branch: the operands of a group are requested together -- a branch per block of
        //  eight made every block wait for its own LDS reads -- at the price of up to 31 columns of multiply-adds with zeros)
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int g = 0; g < kG; ++g) {
            if (32 * g + 31 > k + 1 && 32 * g < n) {     // uniform
                float q[32];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float4 t4 = *(const float4 *)(rk + 32 * g + 4 * u);
                    q[4 * u] = t4.x; q[4 * u + 1] = t4.y; q[4 * u + 2] = t4.z; q[4 * u + 3] = t4.w;
                }
                if (32 * g > k + 1) {
#pragma unroll
                    for (int u = 0; u < 32; u += 2) { s0 = fmaf(a[32 * g + u], q[u], s0); s1 = fmaf(a[32 * g + u + 1], q[u + 1], s1); }
                } else {                                 // the group the reflector starts in
#pragma unroll
                    for (int u = 0; u < 32; u += 2) {
                        s0 = fmaf(a[32 * g + u], 32 * g + u > k + 1 ? q[u] : 0.f, s0);
                        s1 = fmaf(a[32 * g + u + 1], 32 * g + u + 1 > k + 1 ? q[u + 1] : 0.f, s1);
                    }
                }
            }
        }
        const float p = (r > k && r < n) ? t * fmaf(scale, s0 + s1, rk1[r]) : 0.f;
        const float kp = wave_sum(p * vr);
        pv[r] = p;
        vv[r] = vr;
        if (lane == 0) kpart[h] = kp;
        if (r > k + 1 && r < n) A[(int64_t)k * lda + r] = vr;        // row k of A stores the reflector
        if (r == 0) { w.dg[k] = akk; w.of[k] = beta; w.tau[k] = t; }
        __syncthreads();
        const float K = 0.5f * t * (kpart[0] + kpart[1]);
        const float wr = fmaf(-2.0f * K, vr, p);
#pragma unroll
        for (int g = 0; g < kG; ++g) {
            if (32 * g + 31 > k && 32 * g < n) {         // (p and v are zero left of column k + 1 and beyond n)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {         // (two halves: 32 + 32 operand registers beside the 128 of the row spill a few)
                    float pc[16], vc[16];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float4 p4 = *(const float4 *)(pv + 32 * g + 16 * hf + 4 * u), v4 = *(const float4 *)(vv + 32 * g + 16 * hf + 4 * u);
                        pc[4 * u] = p4.x; pc[4 * u + 1] = p4.y; pc[4 * u + 2] = p4.z; pc[4 * u + 3] = p4.w;
                        vc[4 * u] = v4.x; vc[4 * u + 1] = v4.y; vc[4 * u + 2] = v4.z; vc[4 * u + 3] = v4.w;
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) a[32 * g + 16 * hf + u] = fmaf(-vr, pc[u], fmaf(-wr, vc[u], a[32 * g + 16 * hf + u]));
                }
            }
        }
    }
    {   // the last 2 x 2 block
        __syncthreads();
        float *rk = xb, *rk1 = xb + kNMax;
        if (n >= 2 && r == n - 2) put_row(rk, n - 2);
        if (r == n - 1) put_row(rk1, n - 2);
        __syncthreads();
        if (r == 0) {
            if (n >= 2) { w.dg[n - 2] = rk[n - 2]; w.of[n - 2] = rk[n - 1]; }
            w.dg[n - 1] = rk1[n - 1];
            w.of[n - 1] = 0.f;
        }
    }
    __syncthreads();
}

// The same for FOUR waves: thread t owns HALF of row r = t >> 1 -- the columns of parity t & 1, 64 registers -- so that a column's
// multiply-adds and LDS broadcasts per thread are half the two-wave version's, the two halves of a row's product meet with one lane
// exchange, and a workgroup's four waves at 168 registers take a third of a CU where the two at 227 took a quarter -- for 423 instead of 556 us.  Rows and the p / v
// vectors lie in LDS de-interleaved (even columns | 16 bytes | odd columns: the two parities of a 16-lane group read different banks).
constexpr int kQuadHalf = 68;            // floats from the even half of a row in LDS to its odd half
template <int kNMax>
__device__ void tridiagonalize_quad(float *A, int lda, int n, const TriLds &w, float *xb /* LDS, 16-byte aligned, 6 * 136 + 8 floats */)
{
    static_assert(kNMax == 128, "four waves, half a row per thread");
    constexpr int kRow = 2 * kQuadHalf, kG = kNMax / 32;
    const int t = (int)threadIdx.x, r = t >> 1, hh = t & 1, lane = t & 63, h = t >> 6;
    float a[64];                                         // a[j] = A[r][2 j + hh]
    {
        uint64_t ap = (uint64_t)(A + r);
        opaque_u64(ap);
        const float *Ar = (const float *)ap;
        const uint32_t mr = r < n ? 0xFFFFFFFFu : 0u;
        const int jmax = (n - hh + 1) >> 1;              // 2 j + hh < n  <=>  j < jmax  (a compare against a constant per element: no index vectors)
        const float *Ah = Ar + hh * lda;
#pragma unroll
        for (int j = 0; j < 64; ++j) a[j] = __uint_as_float(__float_as_uint(Ah[2 * j * lda]) & (j < jmax ? mr : 0u));
    }
    __syncthreads();                                     // all rows are in registers: A's rows may now receive the reflectors
    float *pv = xb + 4 * kRow, *vv = xb + 5 * kRow, *kpart = xb + 6 * kRow;
    auto flat = [&](const float *row, int c) -> float { return row[(c & 1) * kQuadHalf + (c >> 1)]; };
    const int ncols = (n + 31) & ~31;
    auto put_half = [&](float *dst, int from) {          // one thread: its half of the row, blocks of 8 registers = 16 columns
        float *d = dst + hh * kQuadHalf;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if (16 * b + 15 >= from && 16 * b < ncols) {
                *(float4 *)(d + 8 * b) = make_float4(a[8 * b], a[8 * b + 1], a[8 * b + 2], a[8 * b + 3]);
                *(float4 *)(d + 8 * b + 4) = make_float4(a[8 * b + 4], a[8 * b + 5], a[8 * b + 6], a[8 * b + 7]);
            }
        }
    };
#pragma unroll 1
    for (int k = 0; k + 2 < n; ++k) {
        float *rk = xb + (k & 1) * 2 * kRow, *rk1 = rk + kRow;
        if (r == k) put_half(rk, k);
        if (r == k + 1) put_half(rk1, k);
        __syncthreads();
        const float c0 = lane < n ? flat(rk, lane) : 0.f, c1 = lane + 64 < n ? flat(rk, lane + 64) : 0.f;
        const float sig = wave_sum((lane > k + 1 ? c0 * c0 : 0.f) + (lane + 64 > k + 1 ? c1 * c1 : 0.f));
        const float x0 = flat(rk, k + 1), akk = flat(rk, k);
        if (sig <= 1e-30f) {                             // uniform over the workgroup
            if (t == 0) { w.dg[k] = akk; w.of[k] = x0; w.tau[k] = 0.f; }
            continue;
        }
        const float mu = sqrtf(x0 * x0 + sig);
        const float beta = x0 > 0.f ? -mu : mu;
        const float tt = (beta - x0) / beta;
        const float scale = 1.0f / (x0 - beta);
        const float vr = r == k + 1 ? 1.0f : (r > k + 1 && r < n ? flat(rk, r) * scale : 0.f);
        float s0 = 0.f, s1 = 0.f;
        const float *rkh = rk + hh * kQuadHalf;
        const int jlo = (k + 1 - hh) >> 1;               // 2 j + hh > k + 1  <=>  j > jlo
#pragma unroll
        for (int g = 0; g < kG; ++g) {
            if (32 * g + 31 > k + 1 && 32 * g < n) {     // uniform
                float q[16];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 t4 = *(const float4 *)(rkh + 16 * g + 4 * u);
                    q[4 * u] = t4.x; q[4 * u + 1] = t4.y; q[4 * u + 2] = t4.z; q[4 * u + 3] = t4.w;
                }
                if (32 * g > k + 1) {
#pragma unroll
                    for (int u = 0; u < 16; u += 2) { s0 = fmaf(a[16 * g + u], q[u], s0); s1 = fmaf(a[16 * g + u + 1], q[u + 1], s1); }
                } else {                                 // the group the reflector starts in
#pragma unroll
                    for (int u = 0; u < 16; ++u) s0 = fmaf(a[16 * g + u], 16 * g + u > jlo ? q[u] : 0.f, s0);
                }
            }
        }
        float S = s0 + s1;
        S += wave_shfl_xor(S, 1);                        // the two halves of the row
        const float p = (r > k && r < n) ? tt * fmaf(scale, S, flat(rk1, r)) : 0.f;
        const float kp = wave_sum(hh == 0 ? p * vr : 0.f);
        if (hh == 0) {
            pv[(r & 1) * kQuadHalf + (r >> 1)] = p;
            vv[(r & 1) * kQuadHalf + (r >> 1)] = vr;
            if (r > k + 1 && r < n) A[(int64_t)k * lda + r] = vr;    // row k of A stores the reflector
        }
        if (lane == 0) kpart[h] = kp;
        if (t == 0) { w.dg[k] = akk; w.of[k] = beta; w.tau[k] = tt; }
        __syncthreads();
        const float K = 0.5f * tt * ((kpart[0] + kpart[1]) + (kpart[2] + kpart[3]));
        const float wr = fmaf(-2.0f * K, vr, p);
        const float *pvh = pv + hh * kQuadHalf, *vvh = vv + hh * kQuadHalf;
#pragma unroll
        for (int g = 0; g < kG; ++g) {
            if (32 * g + 31 > k && 32 * g < n) {         // (p and v are zero left of column k + 1 and beyond n)
                float pc[16], vc[16];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 p4 = *(const float4 *)(pvh + 16 * g + 4 * u), v4 = *(const float4 *)(vvh + 16 * g + 4 * u);
                    pc[4 * u] = p4.x; pc[4 * u + 1] = p4.y; pc[4 * u + 2] = p4.z; pc[4 * u + 3] = p4.w;
                    vc[4 * u] = v4.x; vc[4 * u + 1] = v4.y; vc[4 * u + 2] = v4.z; vc[4 * u + 3] = v4.w;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) a[16 * g + u] = fmaf(-vr, pc[u], fmaf(-wr, vc[u], a[16 * g + u]));
            }
        }
    }
    {   // the last 2 x 2 block
        __syncthreads();
        float *rk = xb, *rk1 = xb + kRow;
        if (n >= 2 && r == n - 2) put_half(rk, n - 2);
        if (r == n - 1) put_half(rk1, n - 2);
        __syncthreads();
        if (t == 0) {
            if (n >= 2) { w.dg[n - 2] = flat(rk, n - 2); w.of[n - 2] = flat(rk, n - 1); }
            w.dg[n - 1] = flat(rk1, n - 1);
            w.of[n - 1] = 0.f;
        }
    }
    __syncthreads();
}

// =========================================================================
// ONE-WAVE solver core (n' <= kNMax <= 64): the same algorithms as above -- Householder tridiagonalisation, Sturm
// bisection, inverse iteration with partial pivoting, Gram-Schmidt inside clusters, back-transformation -- for a
// team of ONE wave, with no workgroup barrier anywhere.  The 256-thread version above spends an item's 0.2 ms in
// barriers and LDS round trips (62 Householder steps x 3-4 __syncthreads for ~1 MFLOP of work) while it owns a
// quarter of a CU's LDS; here a matrix row lives in one lane (p = A v and the rank-2 update walk the row in LDS,
// the reflector and w are broadcast with v_readlane), an eigenvalue pair of probes or an eigenvector lives in one
// lane, the LU factors of an inverse iteration live in that lane's REGISTERS (fully unrolled elimination; no LDS
// slots), and several independent teams share a workgroup.  Dot products over rows are DPP wave reductions.

// A (n x n, symmetric, both triangles, row stride lda, LDS) -> T = Q^T A Q as tridiagonalize().  Lane i owns row i and
// keeps it in REGISTERS: the loops over the columns are unrolled in blocks of eight (wave-uniform branches skip the
// blocks left of k and right of n), the reflector v and w = p - K v are broadcast with v_readlane, and the column the
// NEXT reflector is made of (A[.][k+1], by symmetry row k+1) is captured while the rows are updated -- so a step touches
// the LDS only to store its reflector into the (dead) row k of A.
template <int kNMax>
__device__ __forceinline__ void wave_tridiagonalize(float *A, int lda, int n, const WaveTri &w)
{
    static_assert(kNMax <= 64 && kNMax % 8 == 0, "one row per lane, blocks of eight columns");
    constexpr int kB = kNMax / 8;
    const int lane = lane_id();
    float a[kNMax];
    {
        const float *Ai = A + (lane < n ? lane : 0) * lda;
#pragma unroll
        for (int c = 0; c < kNMax; ++c) a[c] = (lane < n && c < n) ? Ai[c] : 0.f;
    }
    wave_sync();                                         // the rows are in registers: A's rows may be overwritten by reflectors
    float cap = a[0];                                    // column k of the current matrix, entry i in lane i
#pragma unroll 1
    for (int k = 0; k + 2 < n; ++k) {
        const float arow = (lane > k && lane < n) ? cap : 0.f;
        const float akk = wave_readlane(cap, k);
        const float sig = wave_sum(lane > k + 1 ? arow * arow : 0.f);
        const float x0 = wave_readlane(arow, k + 1);
        if (sig <= 1e-30f) {                             // wave-uniform: the column is already tridiagonal, H_k = I
            if (lane == 0) { w.dg[k] = akk; w.of[k] = x0; w.tau[k] = 0.f; }
#pragma unroll
            for (int c = 0; c < kNMax; ++c)
                if (c == k + 1) cap = a[c];
            continue;
        }
        const float mu = sqrtf(x0 * x0 + sig);
        const float beta = x0 > 0.f ? -mu : mu;
        const float t = (beta - x0) / beta;
        const float scale = 1.0f / (x0 - beta);
        const float v = lane == k + 1 ? 1.0f : arow * scale;                      // v_c in lane c (0 outside k+1 .. n-1)
        // p_i = t sum_c A[i][c] v_c  (v is zero outside the trailing block, so whole blocks need no masks)
        float p0 = 0.f, p1 = 0.f;
#pragma unroll
        for (int cb = 0; cb < kB; ++cb) {
            if (8 * cb + 7 > k && 8 * cb < n) {
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    p0 = fmaf(a[8 * cb + u], wave_readlane(v, 8 * cb + u), p0);
                    p1 = fmaf(a[8 * cb + u + 1], wave_readlane(v, 8 * cb + u + 1), p1);
                }
            }
        }
        const bool mine = lane > k && lane < n;
        const float p = mine ? t * (p0 + p1) : 0.f;
        const float K = 0.5f * t * wave_sum(p * v);
        const float wv = p - K * v;                                               // w = p - K v  (0 outside the block)
        if (lane > k + 1 && lane < n) A[k * lda + lane] = v;                      // row k of A stores the reflector
        if (lane == 0) { w.dg[k] = akk; w.of[k] = beta; w.tau[k] = t; }
        // A[i][c] -= v_i w_c + w_i v_c  (dead rows have v_i = w_i = 0); column k + 1 is captured for the next step
#pragma unroll
        for (int cb = 0; cb < kB; ++cb) {
            if (8 * cb + 7 > k && 8 * cb < n) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = 8 * cb + u;
                    a[c] -= v * wave_readlane(wv, c) + wv * wave_readlane(v, c);
                    if (c == k + 1) cap = a[c];
                }
            }
        }
    }
    {   // the last 2 x 2 block: column n - 2 is in cap (column 0 when no step ran), A[n-1][n-1] in lane n - 1
        float last = 0.f;
#pragma unroll
        for (int c = 0; c < kNMax; ++c)
            if (c == n - 1) last = a[c];
        const float d2 = wave_readlane(cap, n >= 2 ? n - 2 : 0), o2 = wave_readlane(cap, n - 1), d1 = wave_readlane(last, n - 1);
        if (lane == 0) {
            if (n >= 2) { w.dg[n - 2] = d2; w.of[n - 2] = o2; }
            w.dg[n - 1] = d1;
            w.of[n - 1] = 0.f;
        }
    }
    wave_sync();
}

// Sturm counts of TWO probe points at once: the recurrence of one point is a chain of dependent operations (reciprocal,
// fused multiply-add, compare: ~40 cycles per row) that a single wave on its SIMD cannot hide; a second, independent chain
// rides in its shadow, and the rows of T are requested four at a time.
__device__ __forceinline__ void sturm_count2(const float *dg, const float *of2, int n, float xa, float xb, int &ca, int &cb)
{
    float qa = dg[0] - xa, qb = dg[0] - xb;
    if (fabsf(qa) < 1e-30f) qa = -1e-30f;
    if (fabsf(qb) < 1e-30f) qb = -1e-30f;
    int na = qa < 0.f ? 1 : 0, nb = qb < 0.f ? 1 : 0;
    int i = 1;
    for (; i + 4 <= n; i += 4) {
        float d[4], o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { d[u] = dg[i + u]; o[u] = of2[i + u - 1]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            qa = d[u] - xa - o[u] * fast_rcp(qa);
            qb = d[u] - xb - o[u] * fast_rcp(qb);
            if (fabsf(qa) < 1e-30f) qa = -1e-30f;
            if (fabsf(qb) < 1e-30f) qb = -1e-30f;
            na += qa < 0.f ? 1 : 0;
            nb += qb < 0.f ? 1 : 0;
        }
    }
    for (; i < n; ++i) {
        const float d = dg[i], o = of2[i - 1];
        qa = d - xa - o * fast_rcp(qa);
        qb = d - xb - o * fast_rcp(qb);
        if (fabsf(qa) < 1e-30f) qa = -1e-30f;
        if (fabsf(qb) < 1e-30f) qb = -1e-30f;
        na += qa < 0.f ? 1 : 0;
        nb += qb < 0.f ? 1 : 0;
    }
    ca = na; cb = nb;
}

// the kq largest eigenvalues of T (as eig_top_values): 64 / kVec lanes per eigenvalue, two probe points per lane and round
// (kVec = 32: four probes = 5-section, 12 rounds; kVec = 64: two = trisection, 18 rounds); brackets live in registers, the
// partner lane's counts come by a shuffle.
template <int kVec>
__device__ __forceinline__ void wave_eig_top_values(const WaveTri &w, int nr, int kq, EigShared &es)
{
    constexpr int kL = 64 / kVec;                         // lanes per eigenvalue
    constexpr int kP = 2 * kL;                            // probes per eigenvalue and round
    constexpr int kRounds = kP == 4 ? 12 : 18;            // 2.002 / (kP + 1)^rounds < 1e-8
    const int lane = lane_id();
    if (lane < nr) w.of2[lane] = w.of[lane] * w.of[lane];
    wave_sync();
    const int j = lane / kL, ip = lane - j * kL;
    const int tgt = nr - 1 - j;
    float lo = -1.001f, hi = 1.001f;
#pragma unroll 1
    for (int round = 0; round < kRounds; ++round) {
        const float step = (hi - lo) * (1.0f / (float)(kP + 1));
        // this lane's probes are number 2 ip + 1 and 2 ip + 2 of the kP interior points
        const float xa = lo + step * (float)(2 * ip + 1), xb = lo + step * (float)(2 * ip + 2);
        int ca = 0, cb = 0;
        if (j < kq) sturm_count2(w.dg, w.of2, nr, xa, xb, ca, cb);
        int cnt[kP];
        if (kL == 2) {
            const int oa = wave_shfl_xor(ca, 1), ob = wave_shfl_xor(cb, 1);
            cnt[0] = ip == 0 ? ca : oa; cnt[1] = ip == 0 ? cb : ob;
            cnt[2] = ip == 0 ? oa : ca; cnt[3] = ip == 0 ? ob : cb;
        } else {
            cnt[0] = ca; cnt[1] = cb;
        }
        // the eigenvalue lies left of the first probe whose count exceeds the target
        float nlo = lo + step * (float)kP, nhi = hi;
#pragma unroll
        for (int mm = kP - 1; mm >= 0; --mm)
            if (cnt[mm] > tgt) { nhi = lo + step * (float)(mm + 1); nlo = lo + step * (float)mm; }
        lo = nlo; hi = nhi;
    }
    if (ip == 0 && j < kq) es.lamv[j] = 0.5f * (lo + hi);
    wave_sync();
    if (lane == 0) {
        for (int q = 1; q < kq; ++q)
            if (es.lamv[q] > es.lamv[q - 1]) es.lamv[q] = es.lamv[q - 1];
        es.bad = 0;
    }
    wave_sync();
}

// inverse_iteration_step() with the LU factors in REGISTERS: the eliminations are fully unrolled over kNMax rows --
// T is padded with decoupled rows (diagonal kPadDiag far outside the spectrum, zero off-diagonals, zero right-hand
// side; see wave_pad_tridiagonal) so that the unrolled code needs no per-row guards -- and the pivots, super-diagonals
// and swap flags of one vector never leave its lane.
constexpr float kPadDiag = 4.0f;
template <int kNMax>
__device__ __forceinline__ void wave_pad_tridiagonal(const WaveTri &w, int n)
{
    const int lane = lane_id();
    if (lane >= n && lane < kNMax) { w.dg[lane] = kPadDiag; w.of[lane] = 0.f; }
    for (int i = n * w.ldy + lane; i < kNMax * w.ldy; i += 64) w.Y[i] = 0.f;      // rows n .. kNMax-1 of every vector
    wave_sync();
}

template <int kNMax>
__device__ __forceinline__ bool wave_inverse_iteration_step(const WaveTri &w, int n, int j, float shift, bool random_rhs, uint32_t hseed)
{
    float *Y = w.Y + j;
    const int ldy = w.ldy;
    float ud[kNMax], us[kNMax];
    uint32_t swp[2] = {0u, 0u};                      // row i swapped <=> bit i & 31 of swp[i >> 5]
    COMPILER_MEMORY_FENCE();                         // T is re-read per solve: hoisted out of the caller's loop it would sit in 2 kNMax registers
    if (random_rhs)                                  // (outside the unrolled elimination: the hashes would be kept in registers)
        for (int i = 0; i < n; ++i) Y[i * ldy] = hash_unit(hseed, (uint32_t)j, (uint32_t)i);
    float cd = w.dg[0] - shift, cs = w.of[0];
    float cy = Y[0];
#pragma unroll
    for (int i = 0; i < kNMax - 1; ++i) {
        const float sub = w.of[i], nd = w.dg[i + 1] - shift, ns = i + 2 < kNMax ? w.of[i + 1] : 0.f;
        const float by = Y[(i + 1) * ldy];
        const bool swap = fabsf(cd) < fabsf(sub);
        const float piv = swap ? sub : cd, oth = swap ? cd : sub;
        const float mult = piv != 0.f ? oth * fast_rcp(piv) : 0.f;
        ud[i] = piv;
        us[i] = swap ? nd : cs;
        swp[i >> 5] |= swap ? (1u << (i & 31)) : 0u;
        opaque_u32(swp[i >> 5]);                     // accumulated here and now: not kNMax separate registers OR-ed at the end
        Y[i * ldy] = swap ? by : cy;
        const float ncd = swap ? cs - mult * nd : nd - mult * cs;
        const float ncs = swap ? -mult * ns : ns;
        const float ncy = swap ? cy - mult * by : by - mult * cy;
        cd = ncd; cs = ncs; cy = ncy;
        if ((i & 3) == 3) SCHED_FENCE();             // keeps the scheduler from issuing all rows' LDS reads up front (registers)
    }
    // (opaque: were the bits recognisable as the comparisons above, the compiler would keep all kNMax lane masks alive in
    //  scalar register pairs for the back substitution instead of these two words, and spill them)
    COMPILER_MEMORY_FENCE();                         // the off-diagonals are read again below: reloaded, not kept in kNMax registers
    float x1, x2 = 0.f, ss;
    {   // last row
        float d = cd;
        if (fabsf(d) < kPivTiny) d = d < 0.f ? -kPivTiny : kPivTiny;
        const float x = cy * fast_rcp(d);
        Y[(kNMax - 1) * ldy] = x;
        x1 = x;
        ss = x * x;
    }
#pragma unroll
    for (int i = kNMax - 2; i >= 0; --i) {
        float d = ud[i];
        if (fabsf(d) < kPivTiny) d = d < 0.f ? -kPivTiny : kPivTiny;
        const float s2 = (((swp[i >> 5] >> (i & 31)) & 1u) && i + 2 < kNMax) ? w.of[i + 1] : 0.f;
        const float x = (Y[i * ldy] - us[i] * x1 - s2 * x2) * fast_rcp(d);
        Y[i * ldy] = x;
        x2 = x1; x1 = x;
        ss = fmaf(x, x, ss);
        if ((i & 3) == 0) SCHED_FENCE();
    }
    const bool ok = ss > 0.f && ss < 3.0e38f;
    const float inv = ok ? 1.0f / sqrtf(ss) : 0.f;
#pragma unroll 4
    for (int i = 0; i < n; ++i) Y[i * ldy] *= inv;
    return ok;
}

// eig_top_vectors() for one wave: eigenvectors 0 .. na-1 of the matrix wave_tridiagonalize() reduced, in w.Y[i * ldy + j].
// kVec = 32: na <= 32 and the back-transformation splits the rows of a reflector between the two half-waves.
template <int kNMax, int kVec>
__device__ __forceinline__ bool wave_eig_top_vectors(const float *A, int lda, int nr, int na, const WaveTri &w, EigShared &es, uint32_t hseed,
                                                     long long *tick_row, long long &tick)
{
    const int lane = lane_id();
    const int ldy = w.ldy;
    if (lane == 0) {
        // clusters and shifts exactly as eig_top_vectors(): chains of eigenvalues closer than kOrtol are orthogonalised
        // against each other; the copies of a numerically multiple eigenvalue are displaced away from the nearest other one
        int maxpos = 0;
        for (int j = 0; j < na; ++j) {
            es.shiftv[j] = es.lamv[j];
            es.cs[j] = (j > 0 && es.lamv[j - 1] - es.lamv[j] <= kOrtol) ? es.cs[j - 1] : j;
            es.posi[j] = j - es.cs[j];
            maxpos = es.posi[j] > maxpos ? es.posi[j] : maxpos;
        }
        for (int j = 0; j < na;) {
            int b = j;
            while (b + 1 < na && es.lamv[b] - es.lamv[b + 1] < kSep) ++b;
            const int c = b - j + 1;
            if (c > 1) {
                float room_up = 2.0f;
                if (j > 0) room_up = fminf(es.lamv[j - 1], es.shiftv[j - 1]) - es.lamv[j];
                const float room_dn = b + 1 < na ? es.lamv[b] - es.lamv[b + 1] : (na >= nr ? 2.0f : 0.0f);
                const bool up = room_up >= room_dn;
                const float room = up ? room_up : room_dn;
                float step = kSep;
                if ((float)c * step > 0.5f * room) step = fmaxf(0.5f * room / (float)c, 2.5e-7f);
                for (int mm = 1; mm < c; ++mm) es.shiftv[j + mm] = up ? es.lamv[j] + (float)mm * step : es.lamv[b] - (float)mm * step;
            }
            j = b + 1;
        }
        es.maxpos = maxpos;
    }
    wave_sync();
    wave_pad_tridiagonal<kNMax>(w, nr);
    const int maxpos = es.maxpos;
    int lost = 0;
    int need_until = 2;
#pragma unroll 1
    for (int it = 0; it < kMaxInvIt; ++it) {
        if (lane < na) {
            const int j = lane;
            const bool frozen = it >= 2 && es.shiftv[j] == es.lamv[j] && j + 1 < na && es.lamv[j] - es.lamv[j + 1] < kSep;
            if (!frozen) {
                const bool ok = wave_inverse_iteration_step<kNMax>(w, nr, j, es.shiftv[j], it == 0, hseed);
                if (!ok) es.bad = 1;
            }
        }
        wave_sync();
        if (tick_row && lane == 0) { const long long now_ = device_ticks(); atomicAdd((unsigned long long *)&tick_row[3], (unsigned long long)(now_ - tick)); tick = now_; }
        if (it > 0) {
            float left;
            lost = wave_cluster_orthonormalize<1>(w, nr, na, es.cs, es.posi, maxpos, &left, hseed ^ (0x51ED27u * (uint32_t)(it + 1)));
            if (lost > 0) need_until = it + 3 > need_until ? it + 3 : need_until;
            else if (left < kHeavy) need_until = it + 1 > need_until ? it + 1 : need_until;
        }
        if (tick_row && lane == 0) { const long long now_ = device_ticks(); atomicAdd((unsigned long long *)&tick_row[4], (unsigned long long)(now_ - tick)); tick = now_; }
        if (lane == 0) { es.diag_lost = lost; es.diag_its = it + 1; }
        if (it >= need_until) break;
    }
    // x = H_0 ... H_{nr-3} y: lane = vector, the vector in REGISTERS (the LU registers are dead); with na <= 32 the two
    // half-waves share a vector, half h holding the rows c = 2 r + h.  The reflector entries come as LDS broadcasts, blocks
    // of rows the reflector does not reach are skipped by wave-uniform branches.
    {
        constexpr int kH = kVec == 32 ? 2 : 1;
        constexpr int kR = (kNMax + kH - 1) / kH;
        const int j = kH == 2 ? (lane & 31) : lane, h = kH == 2 ? (lane >> 5) : 0;
        const bool act = j < na;
        float *Yj = w.Y + (act ? j : 0);
        float y[kR], vc[kR];
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const int c = kH * r + h;
            y[r] = (act && c < nr) ? Yj[c * ldy] : 0.f;
            vc[r] = 0.f;
        }
        for (int kk = nr - 3; kk >= 0; --kk) {
            const float t = w.tau[kk];
            if (t == 0.f) continue;                      // wave-uniform
            const float *vk = A + kk * lda;              // reflector kk: v[kk+1] = 1, v[c] = vk[c] for c > kk + 1 (0 beyond nr)
            float s = 0.f;
#pragma unroll
            for (int r0 = 0; r0 < kR; r0 += 4) {
                if (kH * (r0 + 3) + (kH - 1) > kk) {     // some row of this block is beyond kk
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (r0 + u < kR) {
                            const int c = kH * (r0 + u) + h;
                            const float ld = c < kNMax ? vk[c] : 0.f;
                            vc[r0 + u] = c > kk + 1 ? ld : (c == kk + 1 ? 1.0f : 0.f);
                            s = fmaf(vc[r0 + u], y[r0 + u], s);
                        }
                    }
                }
            }
            if (kH == 2) s += wave_shfl_xor(s, 32);
            s *= t;
#pragma unroll
            for (int r0 = 0; r0 < kR; r0 += 4) {
                if (kH * (r0 + 3) + (kH - 1) > kk) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (r0 + u < kR) y[r0 + u] = fmaf(-s, vc[r0 + u], y[r0 + u]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const int c = kH * r + h;
            if (act && c < nr) Yj[c * ldy] = y[r];
        }
    }
    wave_sync();
    if (tick_row && lane == 0) { const long long now_ = device_ticks(); atomicAdd((unsigned long long *)&tick_row[5], (unsigned long long)(now_ - tick)); tick = now_; }
    return lost > 0 || es.bad != 0;
}

// one wave per subgraph: deflated size -> class list; k <= 0 subgraphs are finished here (zeros, data_util.py:243-244)
constexpr int kClsThreads = 256;
__global__ __launch_bounds__(kClsThreads) void posemb_classify_kernel(PosMulti m, PosHead hd)
{
    __shared__ int32_t tc[kClsThreads / 64][kNodeMax];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int item = (int)blockIdx.x * (kClsThreads / 64) + wv;
    if (item >= hd.T) return;
    PosArgs a;
    int b;
    item_args(m, item, a, b);
    const int n0 = a.node_off[b], n = a.node_off[b + 1] - n0;
    const int k = min(min(n - 2, a.hidden), kMaxVec);   // data_util.py:278
    if (k <= 0) {
        for (int i = lane; i < n * a.hidden; i += 64) a.pos[(int64_t)n0 * a.hidden + i] = 0.f;
        if (a.raw) for (int i = lane; i < n * a.hidden; i += 64) a.raw[(int64_t)n0 * a.hidden + i] = 0.f;
        if (a.evals) for (int i = lane; i < a.hidden; i += 64) a.evals[(int64_t)b * a.hidden + i] = 0.f;
        return;
    }
    int cls = kClsKrylov;
    if (n <= kNodeMax) {
        const int32_t *rp = a.row_ptr + n0;
        int32_t *t = tc[wv];
        for (int i = lane; i < n; i += 64) t[i] = 0;
        wave_sync();
        for (int i = lane; i < n; i += 64) {           // twin leaves per parent (low half), stalks per hub (high half)
            const int dg = rp[i + 1] - rp[i];
            if (dg == 1) {
                atomicAdd(&t[a.col_idx[rp[i]] - n0], 1);
            } else if (dg == 2 && hd.use_stalks) {
                const int x = a.col_idx[rp[i]] - n0, y = a.col_idx[rp[i] + 1] - n0;
                const bool lx = rp[x + 1] - rp[x] == 1, ly = rp[y + 1] - rp[y] == 1;
                if (lx != ly) atomicAdd(&t[lx ? y : x], 1 << 16);
            }
        }
        wave_sync();
        int zz = 0;
        for (int i = lane; i < n; i += 64) {
            const int tl = t[i] & 0xFFFF, ts = t[i] >> 16;
            zz += (tl >= 2 ? tl - 1 : 0) + (ts >= 2 ? 2 * (ts - 1) : 0);
        }
        for (int dd = 32; dd >= 1; dd >>= 1) zz += wave_shfl_xor(zz, dd);
        const int nr = n - zz;                         // t >= 2 leaves of one parent count once, s >= 2 stalks of one hub as one stalk
        cls = (hd.use_wave && nr <= 64 && n <= kWaveNodes) ? (nr <= 48 ? kClsW48 : kClsW64)
            : nr <= kJSmall ? kClsSmall : nr <= kJMax ? kClsMid : hd.use_cheb ? kClsCheb : nr <= kGMax ? kClsSlot : nr <= kBMax ? kClsBig : kClsKrylov;
    }
    if (cls == kClsKrylov && n >= hd.ldv) {          // no room: the caller's node_cap / batch_size must bound every subgraph
        for (int i = lane; i < n * a.hidden; i += 64) a.pos[(int64_t)n0 * a.hidden + i] = 0.f;
        if (a.evals) for (int i = lane; i < a.hidden; i += 64) a.evals[(int64_t)b * a.hidden + i] = 0.f;
        if (lane == 0) atomicOr(a.status, (int32_t)GCC_STATUS_POSEMB_TOO_LARGE);
        return;
    }
    if (lane == 0) hd.list[(int64_t)cls * hd.T + atomicAdd(hd.count + cls, 1)] = item;
}

// ---- work lists, largest first.  The class kernels' workgroups pull items off their list with an atomic counter; the classify
// kernel appends in arrival order.  An item of the block class runs 0.6 .. 2.5 ms, 194 of them per call of 16 views go to 64
// workgroups: in arrival order the last long item starts when most workgroups have already run dry, and the kernels behind it in
// the stream wait for it.  One workgroup per class sorts its list by node count, descending (counting sort, ties in any order --
// the order of the items never mattered to their results): longest-processing-time-first over the same atomic counter.
// Items a later kernel appends (the block class hands on what it cannot solve) stay behind the sorted ones.
constexpr int kSortMax = 4096, kSortBins = 1024, kSortThreads = 256;
__global__ __launch_bounds__(kSortThreads) void posemb_sort_kernel(PosMulti m, PosHead hd)
{
    __shared__ int items[kSortMax];
    __shared__ short keys[kSortMax];
    __shared__ int hist[kSortBins], start[kSortBins], wsum[kSortThreads / 64];
    const int cls = (int)blockIdx.x, tid = (int)threadIdx.x, cnt = hd.count[cls];
    if (cnt < 2 || cnt > kSortMax) return;                                   // (block-uniform)
    int32_t *list = hd.list + (int64_t)cls * hd.T;
    for (int i = tid; i < kSortBins; i += kSortThreads) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < cnt; i += kSortThreads) {
        const int gb = list[i];
        PosArgs a;
        int b;
        item_args(m, gb, a, b);
        const int n = a.node_off[b + 1] - a.node_off[b];
        const int key = n < kSortBins ? n : kSortBins - 1;
        items[i] = gb;
        keys[i] = (short)key;
        atomicAdd(&hist[key], 1);
    }
    __syncthreads();
    {                                                                        // start[key] = items with a larger key: 4 bins per thread, descending
        int mine = 0;
        for (int k = 0; k < 4; ++k) mine += hist[kSortBins - 1 - (4 * tid + k)];
        const int incl = wave_scan_incl(mine);
        if ((tid & 63) == 63) wsum[tid >> 6] = incl;
        __syncthreads();
        int before = incl - mine;
        for (int w = 0; w < (tid >> 6); ++w) before += wsum[w];
        for (int k = 0; k < 4; ++k) {
            const int key = kSortBins - 1 - (4 * tid + k);
            start[key] = before;
            before += hist[key];
        }
    }
    __syncthreads();
    for (int i = tid; i < kSortBins; i += kSortThreads) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < cnt; i += kSortThreads) {
        const int key = keys[i];
        list[start[key] + atomicAdd(&hist[key], 1)] = items[i];
    }
}

template <int kCls, int kNMin, int kNMax, int kT, bool kGlobalA, bool kPair = false>
__global__ __launch_bounds__(kT, kPair ? (kT == 256 ? GCC_POSEMB_QUAD_OCC : 2) : 1) void posemb_direct_kernel(PosMulti m, PosHead hd)
{
    static_assert(kNMax % 64 == 0 && kT % 64 == 0 && kT >= 64, "size class");
    static_assert(!kPair || (kGlobalA && kT == kPairT && kNMax == 128), "two- / four-wave teams: the matrix and the reflectors live in the workspace");
    DYN_SMEM(smem);
    __shared__ EigShared es;
    __shared__ int colsrc[64];                  // per output column: eigenvector j >= 0, or -(c + 1) for contrast c
    __shared__ int sh_tot[3], sh_na, sh_item;
    constexpr int kNW = kT / 64, kCPL = kNMax / 64;
    constexpr int lda = kGlobalA ? kNMax : kNMax + 1;   // LDS: odd stride; workspace: rows start on 256-byte boundaries
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (;;) {                                     // items of this class
    __syncthreads();
    if (tid == 0) sh_item = atomicAdd(hd.next + kCls, 1);
    __syncthreads();
    if (sh_item >= hd.count[kCls]) return;
    const int gb = hd.list[(int64_t)kCls * hd.T + sh_item];
    PosArgs a;
    int b;
    item_args(m, gb, a, b);
    const int n0 = a.node_off[b], n = a.node_off[b + 1] - n0;
#ifdef GCC_POSEMB_ABLATE_MID                          // timing experiments only (tools/build_variant.sh): the class costs nothing, its rows stay zero
    if (kCls == kClsMid) continue;
#endif
    const int k = min(min(n - 2, a.hidden), kMaxVec);   // data_util.py:278; k >= 1 (classify kernel)
    long long tick_ = m.ticks ? device_ticks() : 0;
    if (m.ticks && tid == 0) atomicAdd((unsigned long long *)&m.ticks[kCls * 16 + 15], 1ull);   // items
    // ---- LDS carve-up
    TriLds w;
    w.dg = (float *)smem;
    w.of = w.dg + kNMax;
    w.of2 = w.of + kNMax;
    w.tau = w.of2 + kNMax;
    w.pbuf = w.tau + kNMax;
    w.vbuf = w.pbuf + kNMax;
    w.coef = w.vbuf + kNMax;
    w.cnt = (int *)(w.coef + 32 * kYld);
    float *lds_rest = (float *)(w.cnt + kT);
    Defl d;
    float *A = nullptr;
    if (!kGlobalA) {
        A = lds_rest;
        lds_rest += kNMax * lda;
    }
    // the deflation tables (24 KiB) are built in LDS (the eigenvector arrays overlay them); what the expansion at the end
    // needs of them goes to the workgroup's table in the workspace once the matrix is filled
    constexpr int lds_total = kPair ? kPairLds : direct_lds_bytes<kNMax, kT, kGlobalA>();
    static_assert(lds_total - (int)sizeof(float) * (6 * kNMax + 32 * kYld + kT + (kGlobalA ? 0 : kNMax * (kNMax + 1)))
                  >= kNodeMax * kDeflNodeBytes, "the deflation tables overlay the eigenvector / LU region");
    defl_bind(d, lds_rest, kNodeMax);
    const int32_t *rp = a.row_ptr + n0;

    // ---- twin-leaf and stalk groups
    for (int i = tid; i < n; i += kT) defl_init_node(d, i, rp, a.col_idx, n0);
    __syncthreads();
    for (int i = tid; i < n; i += kT) defl_count_node(d, i, rp, a.col_idx, n0, hd.use_stalks != 0);
    __syncthreads();
    for (int p = tid; p < n; p += kT) defl_order_node(d, p, rp, a.col_idx, n0);
    __syncthreads();
    defl_prefix_block<kT>(d, n, sh_tot, w.cnt);    // (w.cnt: kT ints of LDS that are free until the bisection)
    const int nr = sh_tot[0], z = sh_tot[1], zp = sh_tot[2];    // reduced size n', number of twin / stalk contrasts
    if (nr > kNMax || nr < kNMin) continue;        // cannot happen: the classify kernel computed the same size
    if (kGlobalA) A = kPair ? hd.pslots + (int64_t)blockIdx.x * kPairSlotFloats
                            : kCls == kClsBig ? hd.bslots + (int64_t)blockIdx.x * hd.bslot_floats
                                              : hd.slots + (int64_t)blockIdx.x * hd.slot_floats;   // the workgroup's own slot
    // ---- rest of the carve-up: eigenvectors and as many LU slots as fit
    w.Y = lds_rest;
    w.ldy = kYld;
    {
        static_assert(!kGlobalA || (int)sizeof(float) * (6 * kNMax + 32 * kYld + kT + kNMax * kYld) + kNMax * (5 * 8 + 4) <= lds_total,
                      "eigenvectors + the narrowest LU batch must fit");
        if constexpr (kPair) {                           // the LU factors of all 32 inverse iterations: in the slot (L2), behind the records
            w.bw = 32;
            w.ldu = 33;
            w.Ud = A + kNMax * lda + kNodeMax * 2;
            w.Us = w.Ud + kNMax * 33;
            w.Uf = (uint8_t *)(w.Us + kNMax * 33);
        } else {
        const int used = (int)((unsigned char *)(w.Y + nr * kYld) - smem);
        const int left = lds_total - used;
        int bw = 32;
        while (bw > 4 && nr * ((bw + 1) * 8 + bw) > left) bw >>= 1;
        w.bw = bw;
        w.ldu = bw + 1;
        w.Ud = w.Y + nr * kYld;
        w.Us = w.Ud + nr * w.ldu;
        w.Uf = (uint8_t *)(w.Us + nr * w.ldu);
        }
    }
    for (int i = tid; i < nr * lda; i += kT) A[i] = 0.f;
    __syncthreads();
    // M' = norm * adj * norm on the kept nodes (data_util.py:273-277), group couplings scaled by sqrt(group size)
    for (int i = wv; i < n; i += kNW) {
        if (d.ridx[i] == kNone) continue;          // wave-uniform
        const int ri = d.ridx[i];
        const int di = rp[i + 1] - rp[i];
        for (int e = rp[i] + lane; e < rp[i + 1]; e += 64) {
            const int j = a.col_idx[e] - n0;
            if (d.ridx[j] == kNone) continue;
            const int dj = rp[j + 1] - rp[j];
            A[ri * lda + d.ridx[j]] = defl_coupling(d, i, j) / sqrtf((float)di * (float)dj);   // in_degrees().clip(1) ** -0.5 on both sides
        }
    }
    // what the expansion at the end needs of the tables: 8 bytes per node, in the workgroup's table in the workspace
    uint16_t *xrec = kGlobalA ? (uint16_t *)(A + (int64_t)kNMax * lda)
                              : (uint16_t *)(hd.tabs + ((int64_t)(kCls == kClsMid ? 0 : hd.tabs_small_off) + blockIdx.x) * kNodeMax * 4);
    for (int v = tid; v < n; v += kT) defl_record(d, v, rp, a.col_idx, n0, xrec + 4 * v);
    __syncthreads();

    PHASE_TICK(0);                                 // deflation + matrix
    if constexpr (kPair && kT == 256) {
        tridiagonalize_quad<kNMax>(A, lda, nr, w, lds_rest);      // (the eigenvector / LU region of the LDS is free until the bisection)
    } else if constexpr (kPair) {
        tridiagonalize_pair<kNMax>(A, lda, nr, w, lds_rest);      // (the eigenvector / LU region of the LDS is free until the bisection)
    } else if (kGlobalA) {
        // (the eigenvector / LU region of the LDS is free until the bisection: per-wave column sums and the pivot column)
        float *xcol = lds_rest, *slab = lds_rest + kNMax;
        tridiagonalize_lower<kCPL, kT, 4>(A, lda, nr, w, slab, kNMax, xcol);
    } else {
        tridiagonalize<kCPL, kT, 2>(A, lda, nr, w);
    }
    PHASE_TICK(1);
    const int kq = min(k, nr);
    eig_top_values<kT, kMaxVec>(w, nr, kq, es);
    PHASE_TICK(2);                                 // bisection
    // ---- ranks of the merged spectrum (rank_columns)
    if (tid < 64) colsrc[tid] = 0;
    __syncthreads();
    if (tid == 0) sh_na = rank_columns(es.lamv, kq, k, z, zp, colsrc, a.evals ? a.evals + (int64_t)b * a.hidden : nullptr);
    if (a.evals) for (int i = k + tid; i < a.hidden; i += kT) a.evals[(int64_t)b * a.hidden + i] = 0.f;
    __syncthreads();
    const bool failed = eig_top_vectors<kCPL, kT, kPair>(A, lda, nr, sh_na, w, es, (uint32_t)a.seed ^ ((uint32_t)gb * 0x9E3779B1u),
                                                  m.ticks ? m.ticks + kCls * 16 : nullptr, tick_);
    if (tid == 0 && failed) {
        atomicOr(a.status, (int32_t)GCC_STATUS_POSEMB_NOT_CONVERGED);
        if (atomicAdd(a.status + 4, 1) == 0) {               // diagnostics: the first item that failed
            a.status[5] = gb; a.status[6] = kCls; a.status[7] = nr; a.status[8] = es.bad; a.status[9] = n;
            a.status[10] = es.diag_lost; a.status[11] = es.diag_its; a.status[12] = es.maxpos; a.status[13] = sh_na;
        }
    }
    // ---- expand to the n original nodes; x = normalize(u, "l2") row-wise, zero padded (data_util.py:260-262)
    for (int v = wv; v < n; v += kNW) {
        const int rsrc = xrec[4 * v], o = xrec[4 * v + 1], g = xrec[4 * v + 2], cb = xrec[4 * v + 3];
        const float val = lane < k ? defl_expand(rsrc, o, g, cb, colsrc[lane], w.Y, kYld) : 0.f;
        const float s2 = wave_sum(val * val);
        const float inv = s2 > 0.f ? 1.0f / sqrtf(s2) : 1.0f;
        if (lane < a.hidden) {
            a.pos[(int64_t)(n0 + v) * a.hidden + lane] = val * inv;
            if (a.raw) a.raw[(int64_t)(n0 + v) * a.hidden + lane] = val;
        }
    }
    PHASE_TICK(6);                                 // expansion
    if (m.ticks && tid == 0)
        atomicAdd((unsigned long long *)&m.ticks[kCls * 16 + 14],
                  dense_solve_flops(nr, kq, sh_na, es.diag_its, (kT / kMaxVec) * ((kT / kMaxVec) >= 32 ? 6 : ((kT / kMaxVec) >= 16 ? 7 : 9)), n, k));
    }   // next item
}


// ---- one subgraph per WAVE (deflated size <= kNMax <= 64, at most kWaveNodes original nodes): kWaveTeams independent
// teams per workgroup, each with its own LDS carve-up, pulling items from the class list; no workgroup barrier.
template <int kNMax>
__host__ __device__ constexpr int wave_team_bytes()
{
    // A | Y | dg, of, of2, tau | nrm | per-node expansion records (8 bytes)   [the deflation tables overlay Y]
    return (int)sizeof(float) * (kNMax * (kNMax + 1) + kNMax * kYld + 4 * kNMax + 64) + kWaveNodes * 8;
}

// (register budget: the LU factors of an inverse iteration take 2 kNMax registers per lane; without an occupancy
//  target the scheduler spreads the unrolled eliminations over all 512)
#ifndef GCC_POSEMB_W48_OCC
#define GCC_POSEMB_W48_OCC 1      // 2 spills ~40 registers; measured 40.6 vs 37.9 us per item (scripts/gpu/r3_call5.sh)
#endif
template <int kCls, int kNMax>
__global__ __launch_bounds__(kWaveTeams * 64, kNMax <= 48 ? GCC_POSEMB_W48_OCC : 1) void posemb_wave_kernel(PosMulti m, PosHead hd)
{
    static_assert(kNMax <= 64 && kNMax * kYld * 4 >= kWaveNodes * kDeflNodeBytes, "the deflation tables overlay Y");
    DYN_SMEM(smem);
    __shared__ EigShared es_all[kWaveTeams];
    __shared__ int colsrc_all[kWaveTeams][64];
    constexpr int lda = kNMax + 1;                   // odd: lane = row and lane = column are both conflict free
    const int lane = lane_id(), team = (int)threadIdx.x >> 6;
    float *A = (float *)(smem + (size_t)team * wave_team_bytes<kNMax>());
    WaveTri w;
    w.Y = A + kNMax * lda;
    w.ldy = kYld;
    w.dg = w.Y + kNMax * kYld;
    w.of = w.dg + kNMax;
    w.of2 = w.of + kNMax;
    w.tau = w.of2 + kNMax;
    w.nrm = w.tau + kNMax;
    uint16_t *xinfo = (uint16_t *)(w.nrm + 64);      // [kWaveNodes][4]: defl_record
    EigShared &es = es_all[team];
    int *colsrc = colsrc_all[team];
    for (;;) {                                       // items of this class, one per wave
    int item = 0;
    if (lane == 0) item = atomicAdd(hd.next + kCls, 1);
    item = wave_bcast_first(item);
    if (item >= hd.count[kCls]) return;
    const int gb = hd.list[(int64_t)kCls * hd.T + item];
    PosArgs a;
    int b;
    item_args(m, gb, a, b);
    const int n0 = a.node_off[b], n = a.node_off[b + 1] - n0;
#ifdef GCC_POSEMB_ABLATE_WAVE                         // timing experiments only (tools/build_variant.sh)
    continue;
#endif
    const int k = min(min(n - 2, a.hidden), kMaxVec);   // data_util.py:278; k >= 1 (classify kernel)
    long long tick_ = m.ticks ? device_ticks() : 0;
    if (m.ticks && lane == 0) atomicAdd((unsigned long long *)&m.ticks[kCls * 16 + 15], 1ull);   // items
#define WAVE_TICK(ph) do { if (m.ticks && lane == 0) { const long long now_ = device_ticks(); atomicAdd((unsigned long long *)&m.ticks[kCls * 16 + (ph)], (unsigned long long)(now_ - tick_)); tick_ = now_; } } while (0)
    const int32_t *rp = a.row_ptr + n0;
    // ---- twin-leaf and stalk groups (tables in the Y region: dead again before the first eigenvector is written)
    Defl d;
    defl_bind(d, w.Y, kWaveNodes);
    for (int i = lane; i < n; i += 64) defl_init_node(d, i, rp, a.col_idx, n0);
    wave_sync();
    for (int i = lane; i < n; i += 64) defl_count_node(d, i, rp, a.col_idx, n0, hd.use_stalks != 0);
    wave_sync();
    for (int p = lane; p < n; p += 64) defl_order_node(d, p, rp, a.col_idx, n0);
    wave_sync();
    int nr = 0, z = 0, zp = 0;                       // reduced size n', number of twin / stalk contrasts
    for (int i0 = 0; i0 < n; i0 += 64) {             // prefixes over the nodes, 64 at a time
        const int i = i0 + lane;
        const bool valid = i < n;
        const int tc = valid ? d.tcnt[i] : 0, pc = valid ? d.pcnt[i] : 0;
        const int extra = tc >= 2 ? tc - 1 : 0, pextra = pc >= 2 ? pc - 1 : 0;
        const bool kept = valid && !defl_collapsed(d, i);
        const int incl = wave_scan_incl(extra), pincl = wave_scan_incl(pextra);
        const unsigned long long km = wave_ballot(kept);
        if (valid) {
            d.cbase[i] = (z + incl - extra) | ((zp + pincl - pextra) << 16);
            d.ridx[i] = kept ? (uint16_t)(nr + __popcll(km & lanemask_lt())) : kNone;
        }
        nr += __popcll(km);
        z += wave_last(incl);
        zp += wave_last(pincl);
    }
    wave_sync();
    if (nr > kNMax || nr < 1) continue;              // cannot happen: the classify kernel computed the same size
    for (int i = lane; i < nr * lda; i += 64) A[i] = 0.f;
    wave_sync();
    // M' = norm * adj * norm on the kept nodes (data_util.py:273-277), group couplings scaled by sqrt(group size)
#if GCC_POSEMB_EDGE_FILL
    {   // lane = ENTRY of the ego-net's CSR: the entries are split evenly over the lanes (row by bisection over the row
        // pointers, rebased and parked in the record area, which is written only afterwards); with lane = row the lane that
        // holds the hub walks its 100+ entries one by one while the others idle (15 / 32 us of an item's 38 / 69)
        int32_t *rpl = (int32_t *)xinfo;             // [n + 1] <= kWaveNodes + 1 ints of the 8 * kWaveNodes bytes
        const int e0 = rp[0];
        for (int i = lane; i <= n; i += 64) rpl[i] = rp[i] - e0;
        wave_sync();
        const int E = rpl[n];
        for (int eb = 0; eb < E; eb += 4 * 64) {
            int jj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) jj[u] = a.col_idx[e0 + min(eb + u * 64 + lane, E - 1)] - n0;   // 4 coalesced requests in flight
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = eb + u * 64 + lane;
                if (e >= E) continue;
                int lo = 0, hi = n;                  // largest i with rpl[i] <= e
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (rpl[mid] <= e) lo = mid; else hi = mid;
                }
                const int i = lo, j = jj[u];
                if (d.ridx[i] == kNone || d.ridx[j] == kNone) continue;
                const int di = rpl[i + 1] - rpl[i], dj = rpl[j + 1] - rpl[j];
                A[d.ridx[i] * lda + d.ridx[j]] = defl_coupling(d, i, j) / sqrtf((float)di * (float)dj);
            }
        }
        wave_sync();                                 // rpl is done with before the records overwrite it
    }
#else
    for (int i = lane; i < n; i += 64) {             // lane = row
        if (d.ridx[i] == kNone) continue;
        const int ri = d.ridx[i];
        const int di = rp[i + 1] - rp[i];
        for (int e = rp[i]; e < rp[i + 1]; ++e) {
            const int j = a.col_idx[e] - n0;
            if (d.ridx[j] == kNone) continue;
            const int dj = rp[j + 1] - rp[j];
            A[ri * lda + d.ridx[j]] = defl_coupling(d, i, j) / sqrtf((float)di * (float)dj);   // in_degrees().clip(1) ** -0.5 on both sides
        }
    }
#endif
    // what the expansion at the end needs of the tables, 8 bytes per node
    for (int v = lane; v < n; v += 64) defl_record(d, v, rp, a.col_idx, n0, xinfo + 4 * v);
    wave_sync();
    WAVE_TICK(0);                                    // deflation + matrix
    wave_tridiagonalize<kNMax>(A, lda, nr, w);
    WAVE_TICK(1);
    const int kq = min(k, nr);
    wave_eig_top_values<kMaxVec>(w, nr, kq, es);
    WAVE_TICK(2);                                    // bisection
    // ---- ranks of the merged spectrum (rank_columns)
    colsrc[lane] = 0;
    wave_sync();
    int na = 0;
    if (lane == 0) na = rank_columns(es.lamv, kq, k, z, zp, colsrc, a.evals ? a.evals + (int64_t)b * a.hidden : nullptr);
    na = wave_bcast_first(na);
    if (a.evals) for (int i = k + lane; i < a.hidden; i += 64) a.evals[(int64_t)b * a.hidden + i] = 0.f;
    wave_sync();
    const bool failed = wave_eig_top_vectors<kNMax, kMaxVec>(A, lda, nr, na, w, es, (uint32_t)a.seed ^ ((uint32_t)gb * 0x9E3779B1u),
                                                            m.ticks ? m.ticks + kCls * 16 : nullptr, tick_);
    if (lane == 0 && failed) {
        atomicOr(a.status, (int32_t)GCC_STATUS_POSEMB_NOT_CONVERGED);
        if (atomicAdd(a.status + 4, 1) == 0) {               // diagnostics: the first item that failed
            a.status[5] = gb; a.status[6] = kCls; a.status[7] = nr; a.status[8] = es.bad; a.status[9] = n;
            a.status[10] = es.diag_lost; a.status[11] = es.diag_its; a.status[12] = es.maxpos; a.status[13] = na;
        }
    }
    // ---- expand to the n original nodes; x = normalize(u, "l2") row-wise, zero padded (data_util.py:260-262); lane = column
#if GCC_POSEMB_EXPAND4
    {   // four nodes per iteration: 16 lanes per node, columns c and c + 16 per lane (half the iterations of the loop below)
        const int c16 = lane & 15, nv = lane >> 4;
        const int src0 = c16 < k ? colsrc[c16] : 0, src1 = c16 + 16 < k ? colsrc[c16 + 16] : 0;
        for (int v0 = 0; v0 < n; v0 += 4) {
            const int v = v0 + nv;
            const bool valid = v < n;
            const int vv = valid ? v : 0;
            const int rsrc = xinfo[4 * vv + 0], o = xinfo[4 * vv + 1], g = xinfo[4 * vv + 2], cb = xinfo[4 * vv + 3];
            const float val0 = valid && c16 < k ? defl_expand(rsrc, o, g, cb, src0, w.Y, kYld) : 0.f;
            const float val1 = valid && c16 + 16 < k ? defl_expand(rsrc, o, g, cb, src1, w.Y, kYld) : 0.f;
            const float s2 = wave_shfl(row16_sum_last(fmaf(val0, val0, val1 * val1)), lane | 15);
            const float inv = s2 > 0.f ? 1.0f / sqrtf(s2) : 1.0f;
            if (valid) {
                if (c16 < a.hidden) {
                    a.pos[(int64_t)(n0 + v) * a.hidden + c16] = val0 * inv;
                    if (a.raw) a.raw[(int64_t)(n0 + v) * a.hidden + c16] = val0;
                }
                if (c16 + 16 < a.hidden) {
                    a.pos[(int64_t)(n0 + v) * a.hidden + c16 + 16] = val1 * inv;
                    if (a.raw) a.raw[(int64_t)(n0 + v) * a.hidden + c16 + 16] = val1;
                }
            }
        }
    }
#else
    // (two nodes per iteration: lanes 0-31 write node v, lanes 32-63 node v + 1)
    const int col = lane & 31, hv = lane >> 5;
    const int src = col < k ? colsrc[col] : 0;
    for (int v0 = 0; v0 < n; v0 += 2) {
        const int v = v0 + hv;
        const bool valid = v < n;
        const int vv = valid ? v : 0;
        const int rsrc = xinfo[4 * vv + 0], o = xinfo[4 * vv + 1], g = xinfo[4 * vv + 2], cb = xinfo[4 * vv + 3];
        const float val = valid && col < k ? defl_expand(rsrc, o, g, cb, src, w.Y, kYld) : 0.f;
        const float s2 = half32_sum(val * val);
        const float inv = s2 > 0.f ? 1.0f / sqrtf(s2) : 1.0f;
        if (valid && col < a.hidden) {
            a.pos[(int64_t)(n0 + v) * a.hidden + col] = val * inv;
            if (a.raw) a.raw[(int64_t)(n0 + v) * a.hidden + col] = val;
        }
    }
#endif
    WAVE_TICK(6);                                    // expansion
    if (m.ticks && lane == 0)
        atomicAdd((unsigned long long *)&m.ticks[kCls * 16 + 14], dense_solve_flops(nr, kq, na, es.diag_its, 4 * 12, n, k));
    wave_sync();                                     // the next item reuses the team's LDS
    }   // next item
#undef WAVE_TICK
}


// =========================================================================
// Large subgraphs (n > 128; hub seeds, graph_dataset.py:113-124 lets L grow with the seed degree):
// thick-restart Krylov-Schur.  A symmetric Arnoldi process with classical Gram-Schmidt applied
// twice builds V (n x (m+1), column-major in HBM/L2) and the dense projected matrix H = V^T M V
// (LDS); every m = 64 columns the Ritz pairs of H come from the direct solver core, convergence is judged by
// |beta_m * y_{m,i}| for the k wanted pairs, and the basis is compressed to the k + 8 best Ritz
// vectors plus the residual direction (the coupling entries of H are regenerated by the next
// projection, so no arrowhead bookkeeping is needed).  One start vector ~ U[0,1)^n like the
// reference (np.random.rand, data_util.py:248); as with ARPACK, exact multiplicities beyond the
// first copy are only found through rounding.
constexpr int kM = 64;
constexpr int kKeepExtra = 8;
constexpr int kMaxCycles = 16;        // restarts; pairs next to a tiny spectral gap converge (and are defined) poorly
constexpr float kRitzTol = 1e-4f;     // |beta_m y_mi| of the wanted pairs (ARPACK: machine eps; tests: 1e-3)
constexpr int kLongDeg = 32;
constexpr int kMaxLong = 512;
constexpr int kCsrCap = 10240;        // edges of a subgraph kept in LDS (uint16 local ids) for the matrix-vector products
constexpr int kKThreads = 1024;      // 16 waves: the long-vector work is bound by L2 latency, not by FLOPs
constexpr int kKWaves = kKThreads / 64;

struct KryArgs {
    PosMulti m;
    PosHead hd;
    float *vws;              // [workgroups][2][(kM + 1) * ldv]   ping-pong basis (the restart rotation is out of place)
    int32_t ldv;             // column stride (>= max n, multiple of 64)
};

__device__ __forceinline__ float block_sum(float v, float *red /* [kKWaves] */)
{
    const int tid = (int)threadIdx.x;
    v = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < ((int)blockDim.x >> 6); ++i) r += red[i];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(kKThreads) void posemb_krylov_kernel(KryArgs ka)
{
    DYN_SMEM(smem);
    __shared__ float H[(kM + 1) * kM];              // H[i * kM + j], i <= j + 1
    __shared__ float Aj[kM * (kM + 1)], Yj[kM * (kM + 1)];
    __shared__ float theta[kM], hbuf[kM + 1], red[kKWaves];
    __shared__ int sel[kM], colsrc[kM], longrows[kMaxLong];
    __shared__ EigShared es;
    __shared__ int flag, nlong, done, sh_item;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv_id = tid >> 6;
    for (;;) {                                       // items of the Krylov class
    __syncthreads();
    if (tid == 0) sh_item = atomicAdd(ka.hd.next + kClsKrylov, 1);
    __syncthreads();
    if (sh_item >= ka.hd.count[kClsKrylov]) return;
    const int gb = ka.hd.list[(int64_t)kClsKrylov * ka.hd.T + sh_item];
    PosArgs a;
    int b;
    item_args(ka.m, gb, a, b);
    const PosMulti &m = ka.m;
    constexpr int kCls = kClsKrylov;
    long long tick_ = m.ticks ? device_ticks() : 0;
    if (m.ticks && tid == 0) atomicAdd((unsigned long long *)&m.ticks[kCls * 16 + 15], 1ull);   // items
    const int n0 = a.node_off[b], n = a.node_off[b + 1] - n0;
    const int ldv = ka.ldv;
    float *V = ka.vws + (int64_t)blockIdx.x * 2 * (kM + 1) * ldv;     // the workgroup's own basis storage
    float *Valt = V + (int64_t)(kM + 1) * ldv;
    float *x = (float *)smem, *w = x + ldv, *dinv = w + ldv;
    TriLds tw;                                       // Rayleigh-Ritz problem (kM x kM) for the direct solver core
    tw.dg = dinv + ldv;
    tw.of = tw.dg + kM;
    tw.of2 = tw.of + kM;
    tw.tau = tw.of2 + kM;
    tw.pbuf = tw.tau + kM;
    tw.vbuf = tw.pbuf + kM;
    tw.coef = tw.vbuf + kM;
    tw.cnt = (int *)(tw.coef + kVecCap * (kVecCap + 1));
    tw.Y = Yj;
    tw.ldy = kVecCap + 1;
    tw.bw = kVecCap;
    tw.ldu = kVecCap + 1;
    tw.Ud = (float *)(tw.cnt + kKThreads);
    tw.Us = tw.Ud + kM * tw.ldu;
    tw.Uf = (uint8_t *)(tw.Us + kM * tw.ldu);
    uint16_t *ccol = (uint16_t *)(tw.Uf + kM * kVecCap);    // [kCsrCap] local column ids of the subgraph's CSR
    uint16_t *crow = ccol + kCsrCap;                        // [ldv] row offsets
    const int k = min(n - 2, a.hidden);
    const int keep = min(k + kKeepExtra, kM - 8);
    const int lda = kM + 1;

    if (tid == 0) nlong = 0;
    for (int i = tid; i < (kM + 1) * kM; i += kKThreads) H[i] = 0.f;
    __syncthreads();
    for (int r = tid; r < n; r += kKThreads) {
        const int d = a.row_ptr[n0 + r + 1] - a.row_ptr[n0 + r];
        dinv[r] = 1.0f / sqrtf((float)(d < 1 ? 1 : d));
        if (d > kLongDeg) {
            const int slot = atomicAdd(&nlong, 1);
            if (slot < kMaxLong) longrows[slot] = r;
        }
    }
    // v0 = U[0,1)^n (np.random.rand(n)), normalised
    float ss = 0.f;
    for (int r = tid; r < n; r += kKThreads) {
        uint32_t rnd[4];
        philox4x32_10((uint32_t)r, 0u, (uint32_t)gb, 0x9E0B5EEDu, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), rnd);
        const float u = (float)(rnd[0] >> 8) * (1.0f / 16777216.0f);
        w[r] = u;
        ss += u * u;
    }
    __syncthreads();
    float nrm = sqrtf(block_sum(ss, red));
    for (int r = tid; r < n; r += kKThreads) {
        const float vn = w[r] / nrm;
        V[r] = vn;
        x[r] = vn * dinv[r];                         // x = D^-1/2 v_j is kept up to date where v_j is written
    }
    // the subgraph's CSR in LDS (two dependent global round trips per matrix-vector product otherwise)
    const int rp0 = a.row_ptr[n0], nnz = a.row_ptr[n0 + n] - rp0;
    const bool csr_lds = nnz <= kCsrCap && nnz < 65536 && n < ldv;
    if (csr_lds) {
        for (int e = tid; e < nnz; e += kKThreads) ccol[e] = (uint16_t)(a.col_idx[rp0 + e] - n0);
        for (int r = tid; r <= n; r += kKThreads) crow[r] = (uint16_t)(a.row_ptr[n0 + r] - rp0);
    }
    __syncthreads();
    const int nl = nlong < kMaxLong ? nlong : kMaxLong;
    const bool long_overflow = nlong > kMaxLong;

    // orthogonalise w against V[:, 0..ncols) twice (CGS2); optionally accumulate the coefficients into H[:, hcol]
    auto orth = [&](int ncols, int hcol) {
        for (int pass = 0; pass < 2; ++pass) {
            // two basis columns per wave at a time, four 64-row slices each: 8 loads in flight per lane (the basis
            // lives in L2: the passes are bound by the latency of these reads)
            for (int i = 2 * wv_id; i < ncols; i += 2 * kKWaves) {
                const float *va = V + (int64_t)i * ldv;
                const bool two = i + 1 < ncols;
                const float *vb = two ? va + ldv : va;
                float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
                for (int r = lane; r < n; r += 256) {
                    const int r1 = r + 64, r2 = r + 128, r3 = r + 192;
                    const float x0 = va[r], y0 = vb[r];
                    const float x1 = r1 < n ? va[r1] : 0.f, y1 = r1 < n ? vb[r1] : 0.f;
                    const float x2 = r2 < n ? va[r2] : 0.f, y2 = r2 < n ? vb[r2] : 0.f;
                    const float x3 = r3 < n ? va[r3] : 0.f, y3 = r3 < n ? vb[r3] : 0.f;
                    const float w0 = w[r], w1 = r1 < n ? w[r1] : 0.f, w2 = r2 < n ? w[r2] : 0.f, w3 = r3 < n ? w[r3] : 0.f;
                    a0 = fmaf(x0, w0, a0); a1 = fmaf(x1, w1, a1); a0 = fmaf(x2, w2, a0); a1 = fmaf(x3, w3, a1);
                    b0 = fmaf(y0, w0, b0); b1 = fmaf(y1, w1, b1); b0 = fmaf(y2, w2, b0); b1 = fmaf(y3, w3, b1);
                }
                const float sa = wave_sum(a0 + a1), sb = wave_sum(b0 + b1);
                if (lane == 0) {
                    hbuf[i] = sa;
                    if (two) hbuf[i + 1] = sb;
                }
            }
            __syncthreads();
            if (hcol >= 0 && tid < ncols) H[tid * kM + hcol] += hbuf[tid];
            for (int r = tid; r < n; r += kKThreads) {
                float a0 = w[r], a1 = 0.f, a2 = 0.f, a3 = 0.f;
                int i = 0;
                for (; i + 8 <= ncols; i += 8) {             // 8 basis reads in flight
                    const float v0 = V[(int64_t)i * ldv + r], v1 = V[(int64_t)(i + 1) * ldv + r];
                    const float v2 = V[(int64_t)(i + 2) * ldv + r], v3 = V[(int64_t)(i + 3) * ldv + r];
                    const float v4 = V[(int64_t)(i + 4) * ldv + r], v5 = V[(int64_t)(i + 5) * ldv + r];
                    const float v6 = V[(int64_t)(i + 6) * ldv + r], v7 = V[(int64_t)(i + 7) * ldv + r];
                    a0 = fmaf(-v0, hbuf[i], a0); a1 = fmaf(-v1, hbuf[i + 1], a1);
                    a2 = fmaf(-v2, hbuf[i + 2], a2); a3 = fmaf(-v3, hbuf[i + 3], a3);
                    a0 = fmaf(-v4, hbuf[i + 4], a0); a1 = fmaf(-v5, hbuf[i + 5], a1);
                    a2 = fmaf(-v6, hbuf[i + 6], a2); a3 = fmaf(-v7, hbuf[i + 7], a3);
                }
                for (; i + 4 <= ncols; i += 4) {
                    a0 = fmaf(-V[(int64_t)i * ldv + r], hbuf[i], a0);
                    a1 = fmaf(-V[(int64_t)(i + 1) * ldv + r], hbuf[i + 1], a1);
                    a2 = fmaf(-V[(int64_t)(i + 2) * ldv + r], hbuf[i + 2], a2);
                    a3 = fmaf(-V[(int64_t)(i + 3) * ldv + r], hbuf[i + 3], a3);
                }
                for (; i < ncols; ++i) a0 = fmaf(-V[(int64_t)i * ldv + r], hbuf[i], a0);
                w[r] = (a0 + a1) + (a2 + a3);
            }
            __syncthreads();
        }
    };

    int j = 0, steps = 0, cycle = 0;
    for (; cycle < kMaxCycles; ++cycle) {
        for (; j < kM; ++j, ++steps) {
            // w = D^-1/2 A D^-1/2 v_j   (short rows: one thread per row; long rows: one wave per row)
            for (int r = tid; r < n; r += kKThreads) {
                const int beg = csr_lds ? (int)crow[r] : a.row_ptr[n0 + r] - rp0;
                const int end = csr_lds ? (int)crow[r + 1] : a.row_ptr[n0 + r + 1] - rp0;
                if (end - beg > kLongDeg && !long_overflow) continue;
                float s = 0.f;
                if (csr_lds) for (int e = beg; e < end; ++e) s += x[ccol[e]];
                else for (int e = beg; e < end; ++e) s += x[a.col_idx[rp0 + e] - n0];
                w[r] = s * dinv[r];
            }
            if (!long_overflow) {
                for (int i = wv_id; i < nl; i += kKWaves) {
                    const int r = longrows[i];
                    const int beg = csr_lds ? (int)crow[r] : a.row_ptr[n0 + r] - rp0;
                    const int end = csr_lds ? (int)crow[r + 1] : a.row_ptr[n0 + r + 1] - rp0;
                    float s = 0.f;
                    if (csr_lds) for (int e = beg + lane; e < end; e += 64) s += x[ccol[e]];
                    else for (int e = beg + lane; e < end; e += 64) s += x[a.col_idx[rp0 + e] - n0];
                    s = wave_sum(s);
                    if (lane == 0) w[r] = s * dinv[r];
                }
            }
            __syncthreads();
            orth(j + 1, j);
            float s2 = 0.f;
            for (int r = tid; r < n; r += kKThreads) s2 = fmaf(w[r], w[r], s2);
            float beta = sqrtf(block_sum(s2, red));
            if (beta < 1e-6f) {                      // invariant subspace: continue with a fresh direction
                for (int r = tid; r < n; r += kKThreads) {
                    uint32_t rnd[4];
                    philox4x32_10((uint32_t)r, (uint32_t)(steps + 1), (uint32_t)gb, 0x9E0B5EEDu,
                                  (uint32_t)a.seed, (uint32_t)(a.seed >> 32), rnd);
                    w[r] = (float)(rnd[0] >> 8) * (1.0f / 16777216.0f) - 0.5f;
                }
                __syncthreads();
                orth(j + 1, -1);
                float s3 = 0.f;
                for (int r = tid; r < n; r += kKThreads) s3 = fmaf(w[r], w[r], s3);
                nrm = sqrtf(block_sum(s3, red));
                beta = 0.f;
            } else {
                nrm = beta;
            }
            if (tid == 0) H[(j + 1) * kM + j] = beta;
            for (int r = tid; r < n; r += kKThreads) {
                const float vn = w[r] / nrm;
                V[(int64_t)(j + 1) * ldv + r] = vn;
                x[r] = vn * dinv[r];                 // for the next product (after a restart the next v_j is this vector too)
            }
            __syncthreads();
        }
        PHASE_TICK(0);                               // Arnoldi steps
        // Rayleigh-Ritz on the symmetric part of H
        for (int i = tid; i < kM * lda; i += kKThreads) {
            const int r = i / lda, c = i - r * lda;
            float v = 0.f;
            if (c < kM) v = r <= c ? H[r * kM + c] : H[c * kM + r];
            Aj[i] = v;
        }
        __syncthreads();
        // the `keep` largest Ritz pairs by the direct solver core (tridiagonalise, bisect, inverse iteration): ~4x
        // cheaper than Jacobi sweeps on the 64 x 64 matrix, and only the pairs that are used are computed
        tridiagonalize<1, kKThreads, 2>(Aj, lda, kM, tw);
        eig_top_values<kKThreads, kVecCap>(tw, kM, keep, es);
        const bool ritz_failed = eig_top_vectors<1, kKThreads>(Aj, lda, kM, keep, tw, es,
                                                               (uint32_t)a.seed ^ ((uint32_t)gb * 0x9E3779B1u) ^ (uint32_t)cycle,
                                                               nullptr, tick_);
        PHASE_TICK(1);                               // Ritz pairs of H
        if (tid < kM) {
            theta[tid] = tid < keep ? es.lamv[tid] : -2.f;
            sel[tid] = tid < keep ? tid : kM;           // Ritz pair tid has rank tid (0 = largest); the others are not computed
        }
        if (tid == 0) { done = 1; flag = ritz_failed ? 2 : 0; }
        __syncthreads();
        const float beta_m = H[kM * kM + (kM - 1)];
        if (tid < k) {
            const float res = fabsf(beta_m * Yj[(kM - 1) * lda + tid]);
            if (res > kRitzTol) done = 0;
            if (res > 10.f * kRitzTol) flag = 2;         // far from converged (reported if the cycle cap is hit)
        }
        __syncthreads();
        const bool finished = done != 0 || cycle == kMaxCycles - 1;
#ifdef GCC_AMD_HIPEMU
        if (getenv("GCC_POSEMB_DEBUG") && tid == 0) {
            float worst = 0.f; int nbad = 0;
            for (int i = 0; i < kM; ++i) if (sel[i] < k) { const float rs = fabsf(beta_m * Yj[(kM - 1) * lda + i]); worst = rs > worst ? rs : worst; nbad += rs > kRitzTol; }
            fprintf(stderr, "kry b=%d n=%d cycle=%d worst=%.2e nbad=%d theta_k=%.4f\n", b, n, cycle + 1, worst, nbad, 0.f);
        }
#endif
        if (finished && done == 0 && flag == 2 && tid == 0) {
            atomicOr(a.status, (int32_t)GCC_STATUS_POSEMB_NOT_CONVERGED);
            if (atomicAdd(a.status + 4, 1) == 0) { a.status[5] = gb; a.status[6] = kClsKrylov; a.status[7] = n; a.status[8] = -1; a.status[9] = n; }
        }
        if (finished && done == 0 && tid == 0) atomicAdd(a.status + 3, 1);      // diagnostics: items that stopped at the cycle cap
        const int nout = finished ? k : keep;
        // output column t <- Ritz vector: restart keeps rank t, the final result is ascending (rank k-1-t)
        if (tid < kM) {
            const int t = finished ? k - 1 - sel[tid] : sel[tid];
            if (t >= 0 && t < nout) colsrc[t] = tid;
        }
        __syncthreads();
        if (!finished) {
            // Valt[:, t] = V[:, 0..m) Y[:, colsrc[t]]: one thread per (row, 8 output columns)
            const int nchunk = (nout + 7) >> 3;
            for (int task = tid; task < n * nchunk; task += kKThreads) {
                const int ch = task / n, r = task - ch * n;       // consecutive threads -> consecutive rows
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                int src[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) src[u] = colsrc[min(8 * ch + u, nout - 1)];
                for (int c = 0; c < kM; ++c) {
                    const float v = V[(int64_t)c * ldv + r];
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc[u] = fmaf(v, Yj[c * lda + src[u]], acc[u]);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (8 * ch + u < nout) Valt[(int64_t)(8 * ch + u) * ldv + r] = acc[u];
            }
            // the residual direction becomes column `keep`; H = diag(theta_kept)
            for (int r = tid; r < n; r += kKThreads) Valt[(int64_t)keep * ldv + r] = V[(int64_t)kM * ldv + r];
            for (int i = tid; i < (kM + 1) * kM; i += kKThreads) H[i] = 0.f;
            __syncthreads();
            if (tid < kM && sel[tid] < keep) H[sel[tid] * kM + sel[tid]] = theta[tid];
            __syncthreads();
            float *tmp = V; V = Valt; Valt = tmp;
            j = keep;
            PHASE_TICK(2);                           // restart compression
            continue;
        }
        // final: u_t = V Y[:, colsrc[t]] -> raw eigenvectors, then normalize(u, "l2") rows, zero padding
        float *rawout = a.raw ? a.raw : a.pos;
        const int nchunk = (k + 7) >> 3;
        for (int task = tid; task < n * nchunk; task += kKThreads) {
            const int ch = task / n, r = task - ch * n;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int src[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) src[u] = colsrc[min(8 * ch + u, k - 1)];
            for (int c = 0; c < kM; ++c) {
                const float v = V[(int64_t)c * ldv + r];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] = fmaf(v, Yj[c * lda + src[u]], acc[u]);
            }
            float *ro = rawout + (int64_t)(n0 + r) * a.hidden + 8 * ch;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (8 * ch + u < k) ro[u] = acc[u];
        }
        __syncthreads();
        for (int r = tid; r < n; r += kKThreads) {
            const float *ro = rawout + (int64_t)(n0 + r) * a.hidden;
            float *po = a.pos + (int64_t)(n0 + r) * a.hidden;
            float s2 = 0.f;
            for (int c = 0; c < k; ++c) s2 = fmaf(ro[c], ro[c], s2);
            const float inv = s2 > 0.f ? 1.0f / sqrtf(s2) : 1.0f;
            for (int c = 0; c < a.hidden; ++c) {
                const float v = c < k ? ro[c] : 0.f;
                po[c] = v * inv;
                if (a.raw) a.raw[(int64_t)(n0 + r) * a.hidden + c] = v;
            }
        }
        if (a.evals && tid < kM && sel[tid] < k) a.evals[(int64_t)b * a.hidden + (k - 1 - sel[tid])] = theta[tid];
        if (a.evals) for (int c = k + tid; c < a.hidden; c += kKThreads) a.evals[(int64_t)b * a.hidden + c] = 0.f;
        break;
    }
    PHASE_TICK(3);                                   // final Ritz vectors
    if (tid == 0) {                                  // diagnostics: status[1] = max restart cycles, status[2] = Arnoldi steps
        atomicMax(a.status + 1, cycle + 1);
        atomicAdd(a.status + 2, steps);
    }
    }   // next item
}


// =========================================================================
// Large deflated subgraphs (n' > 128), first choice: Chebyshev-filtered subspace iteration on the SPARSE deflated
// matrix with a guard block.  A dense reduction streams the matrix once or twice per column (the workspace classes
// above: 2.4 ms at n' = 250, 19 ms at n' = 580, bound by one CU's path to L2), but M' has ~6 entries per row and the
// ego-net spectra are spread: with 64 block vectors (k <= 32 wanted + guards) a degree-d Chebyshev polynomial on
// [-1, cut] separates the wanted end by exp(-0.4 d .. -0.6 d) on the hub ego-nets of the 1M-node graph, so ~26 sparse
// products converge to 1e-5.  A block also takes exact multiplicities in its stride: whatever it holds of a repeated
// eigenvalue's eigenspace are eigenvectors (the 30+ copies of 1/sqrt(2) that defeat single-vector Krylov/ARPACK).
//   round: X <- p_d(M') X (scaled three-term recurrence, two n' x 64 buffers in L2, in place) ; W = M' X ;
//          G = X^T X, K = X^T W (fp64 accumulation) ; G = R^T R (Jacobi-scaled Cholesky) ; H = R^-T K R^-1 ;
//          H = Y Theta Y^T (the dense solver core, all 64 pairs) ; X <- X R^-1 Y, W <- W R^-1 Y ; residuals
//          ||W_i - theta_i X_i|| of the wanted pairs ; cut <- theta_64 - 0.02.
// Items that do not converge in kChRounds, whose top k reaches the null space (the contrast bookkeeping of the
// direct solver is needed then), or whose edges do not fit the LDS are appended to the work lists of the dense
// classes, which run afterwards in the same stream: the exact solver remains the reference for every corner.
constexpr int kChP = 64;             // widest block (sizes the workspace and the LDS)
constexpr int kChPWide = 64, kChPNarrow = 32;   // the two block widths of the solve (posemb_cheb_kernel: `solve`)
#ifndef GCC_POSEMB_CH_NARROW_WANT
#define GCC_POSEMB_CH_NARROW_WANT 20
#endif
constexpr int kChNarrowWant = GCC_POSEMB_CH_NARROW_WANT;   // items whose quotient must deliver at most this many pairs (k - zp) start with the narrow block (0: never)
constexpr int kChNarrowGuards = 8;   // ... and move to the wide one when the first Ritz values show more than 32 - 8 wanted pairs
#ifndef GCC_POSEMB_CH_THREADS
#define GCC_POSEMB_CH_THREADS 1024
#endif
constexpr int kChThreads = GCC_POSEMB_CH_THREADS;
constexpr int kChCsrCap = 12288;     // directed edges of the deflated subgraph (uint16 column ids in LDS)
#ifndef GCC_POSEMB_CH_LONGDEG
#define GCC_POSEMB_CH_LONGDEG 32
#endif
// Rows with more kept entries than this are "long": cut into chunks whose partial sums go through the slab, so that a
// product's critical path is a chunk (or a short row), not the subgraph's densest row.  (Rounds 2-4 used 96 for both the
// threshold and the chunk length: a hub row of 450 entries was five chunks of 24 dependent gather rounds each on five
// 8-thread groups, with the other 1000 threads waiting at the barrier behind them, and every row of 33..96 entries was a
// serial chain of its own in the row pass -- ~28 us per product at n' ~ 300 whether the block sat in LDS or in L2.)
// The threshold doubles per item until at most half of the chunk slots are long rows; the chunk length is what spreads
// the long rows' entries over the remaining slots (a multiple of 4, at least 8).
constexpr int kChLongDeg = GCC_POSEMB_CH_LONGDEG;
constexpr int kChSlabFloats = 4096;  // 16 KB of partial sums: 64 chunk slots for a 64-column block, 128 for a 32-column one
constexpr int kChMaxChunks = kChSlabFloats / 32;
constexpr int kChMaxLong = kChMaxChunks / 2;
constexpr int kChRounds = 16;      // filter rounds of an item
constexpr int kChMaxRitz = 5;      // Ritz steps of an item (the first one: values only)
constexpr float kChTol = 2e-5f;      // residual norm of the wanted Ritz pairs
constexpr double kChShift = 1e-8;
constexpr float kChAmpLog = 12.9f;  // ln of the largest filter amplification between two re-orthonormalisations (4e5; at 1e7 fp32 loses the guard end and one hub ego-net in 29 misses the strict invariants)
constexpr int kChLdy = kChP + 1;
constexpr int kChBw = 32;            // inverse iterations of the Ritz problem per batch

struct LdsYes { static constexpr bool value = true; };
struct LdsNo { static constexpr bool value = false; };

struct ChebArgs {
    PosMulti m;
    PosHead hd;
    float *xws;              // [workgroups][3][kNodeMax * kChP]  the block buffers
    uint32_t *tabs;          // [workgroups][kNodeMax * 4]        deflation tables once the matrix is built
};

__host__ __device__ constexpr int cheb_region_bytes()
{
    // the largest of: deflation tables (16 KiB) | two fp64 64 x 64 matrices (64 KiB) | the Ritz problem: L, H/C, Y and the
    // Gram-Schmidt coefficients (64 x 65 each), the solver's vectors, Sturm counts, LU slots
    constexpr int ritz = (int)sizeof(float) * (4 * kChP * kChLdy + 7 * kChP + kChThreads + 2 * kChP * (kChBw + 1)) + kChP * kChBw;
    return ritz > 3 * 32768 ? ritz : 3 * 32768;              // ... | three fp64 64 x 64 matrices (G, K, L^-1)
}
__host__ __device__ constexpr int cheb_lds_bytes()
{
    return 2 * (kNodeMax + 8) + 2 * kChCsrCap + 4 * kNodeMax + 4 * kChSlabFloats + cheb_region_bytes();
}

__global__ __launch_bounds__(kChThreads) void posemb_cheb_kernel(ChebArgs ca)
{
    DYN_SMEM(smem);
    __shared__ EigShared es;
    __shared__ float theta[kChP], resid[kChP], dsc[kChP];
    __shared__ double dscd[kChP];
    __shared__ int longrow[kChMaxLong], longfirst[kChMaxLong + 1];
    __shared__ int chunk_beg[kChMaxChunks + 1];
    __shared__ uint8_t chunk_row[kChMaxChunks], rowx[kNodeMax];      // long-row index of a chunk / of a row (0xFF: short)
    __shared__ int sh_clen, sh_long_t;
    __shared__ int wsum[kChThreads / 64 + 1];
    __shared__ int sh_item, sh_tot[3], sh_nlong, sh_nchunk, sh_fail;
    __shared__ int colsrc[64];
    constexpr int kCls = kClsCheb;
    const PosMulti &m = ca.m;
    const PosHead &hd = ca.hd;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int kNW = kChThreads / 64;
    uint16_t *crow = (uint16_t *)smem;                       // [kNodeMax + 1] row offsets of the deflated CSR
    uint16_t *ccol = crow + (kNodeMax + 8);                  // [kChCsrCap]
    float *scale = (float *)(ccol + kChCsrCap);              // [kNodeMax] M' = diag(scale) A' diag(scale)
    float *slab = scale + kNodeMax;                          // [chunk slots][P] partial sums of the long rows (kChSlabFloats)
    unsigned char *region = (unsigned char *)(slab + kChSlabFloats);
    for (;;) {                                               // items of this class
    __syncthreads();
    if (tid == 0) sh_item = atomicAdd(hd.next + kCls, 1);
    __syncthreads();
    if (sh_item >= hd.count[kCls]) return;
    const int gb = hd.list[(int64_t)kCls * hd.T + sh_item];
    PosArgs a;
    int b;
    item_args(m, gb, a, b);
    const int n0 = a.node_off[b], n = a.node_off[b + 1] - n0;
#ifdef GCC_POSEMB_ABLATE_CHEB                         // timing experiments only (tools/build_variant.sh)
    continue;
#endif
    const int k = min(min(n - 2, a.hidden), kMaxVec);
    long long tick_ = m.ticks ? device_ticks() : 0;
    if (m.ticks && tid == 0) atomicAdd((unsigned long long *)&m.ticks[kCls * 16 + 15], 1ull);
    const int32_t *rp = a.row_ptr + n0;
    // three block buffers in the workspace (L2-resident): X, W = M' X or filter scratch, rotation target
    float *XA0 = ca.xws + (int64_t)blockIdx.x * 3 * kNodeMax * kChP, *XB0 = XA0 + (int64_t)kNodeMax * kChP, *XC0 = XB0 + (int64_t)kNodeMax * kChP;

    // ---- twin-leaf and stalk groups (as posemb_direct_kernel)
    Defl d;
    defl_bind(d, region, kNodeMax);
    for (int i = tid; i < n; i += kChThreads) defl_init_node(d, i, rp, a.col_idx, n0);
    if (tid == 0) { sh_fail = 0; sh_nlong = 0; sh_nchunk = 0; }
    __syncthreads();
    for (int i = tid; i < n; i += kChThreads) defl_count_node(d, i, rp, a.col_idx, n0, hd.use_stalks != 0);
    __syncthreads();
    for (int p = tid; p < n; p += kChThreads) defl_order_node(d, p, rp, a.col_idx, n0);
    __syncthreads();
    defl_prefix_block<kChThreads>(d, n, sh_tot, (int *)slab);
    const int nr = sh_tot[0], z = sh_tot[1], zp = sh_tot[2];
    // ---- sparse M' = diag(scale) A' diag(scale): rows of the kept nodes, columns ascending; a super-leaf standing for t
    //      twins carries sqrt(t) (its coupling to the parent is sqrt(t / d_p), data_util.py:273-277 on the quotient); the
    //      stalk standing for s stalks carries sqrt(s / 2) on its middle node and 1 / sqrt(s) on its leaf (couplings
    //      sqrt(s / (2 d_h)) to the hub and 1 / sqrt(2) inside)
    for (int i = tid; i < n; i += kChThreads) {
        if (d.ridx[i] == kNone) continue;
        const int di = rp[i + 1] - rp[i];
        const int pi = d.par[i];
        float sc = 1.0f / sqrtf((float)(di < 1 ? 1 : di));
        if (pi != (int)kNone && d.tcnt[pi] >= 2) {
            sc = sqrtf((float)d.tcnt[pi]);
        } else {
            const int h = d.phub[pi != (int)kNone ? pi : i];
            if (h != (int)kNone && d.pcnt[h] >= 2) sc = pi != (int)kNone ? 1.0f / sqrtf((float)d.pcnt[h]) : sqrtf(0.5f * (float)d.pcnt[h]);
        }
        scale[d.ridx[i]] = sc;
    }
    if (tid <= nr) crow[tid] = 0;                            // nr <= kNodeMax = kChThreads
    __syncthreads();
    for (int i = wv; i < n; i += kNW) {                      // kept neighbours per kept row
        if (d.ridx[i] == kNone) continue;                    // wave-uniform
        int c = 0;
        for (int e0 = rp[i]; e0 < rp[i + 1]; e0 += 64) {
            const int e = e0 + lane;
            const bool keep = e < rp[i + 1] && d.ridx[a.col_idx[e] - n0] != kNone;
            c += __popcll(wave_ballot(keep));
        }
        if (lane == 0) crow[d.ridx[i] + 1] = (uint16_t)(c > 65535 ? 65535 : c);
    }
    __syncthreads();
    {                                                        // exclusive prefix: one row per thread
        const int c = tid < nr ? (int)crow[tid + 1] : 0;
        int incl = wave_scan_incl(c);
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int base = 0;
        for (int q = 0; q < wv; ++q) base += wsum[q];
        incl += base;
        __syncthreads();
        if (tid < nr) {
            if (incl > kChCsrCap) sh_fail = 1;               // does not fit: dense classes
            crow[tid + 1] = (uint16_t)(incl > 65535 ? 65535 : incl);
        }
    }
    __syncthreads();
    if (!sh_fail) {
        for (int i = wv; i < n; i += kNW) {
            if (d.ridx[i] == kNone) continue;
            int at = (int)crow[d.ridx[i]];
            for (int e0 = rp[i]; e0 < rp[i + 1]; e0 += 64) {
                const int e = e0 + lane;
                const int rj = e < rp[i + 1] ? (int)d.ridx[a.col_idx[e] - n0] : (int)kNone;
                const unsigned long long mk = wave_ballot(rj != (int)kNone);
                if (rj != (int)kNone) ccol[at + __popcll(mk & lanemask_lt())] = (uint16_t)rj;
                at += __popcll(mk);
            }
        }
    }
    __syncthreads();
    // what the expansion at the end needs of the tables -> workspace, 8 bytes per node (the dense matrices overlay the tables)
    uint16_t *xrec = (uint16_t *)(ca.tabs + (int64_t)blockIdx.x * kNodeMax * 4);
    for (int v = tid; v < n; v += kChThreads) defl_record(d, v, rp, a.col_idx, n0, xrec + 4 * v);
    __syncthreads();
    PHASE_TICK(0);                                           // deflation + sparse matrix
    bool failed = sh_fail != 0;
    // Long rows -> chunk tables for `slots` chunk slots (all threads; ends with a barrier).  Row r = thread r.
    auto build_chunks = [&](int slots) {
        const int len = (tid < nr && !failed) ? (int)crow[tid + 1] - (int)crow[tid] : 0;
        auto block_scan = [&](int v, int &total) -> int {                // exclusive prefix of v over the workgroup
            int incl = wave_scan_incl(v);
            if (lane == 63) wsum[wv] = incl;
            __syncthreads();
            int base = 0, tot = 0;
            for (int q = 0; q < kNW; ++q) { if (q < wv) base += wsum[q]; tot += wsum[q]; }
            __syncthreads();
            total = tot;
            return incl - v + base;
        };
        int T = kChLongDeg, nl = 0, x = 0;
        for (;;) {                                                       // block-uniform: at most half of the slots are rows
            x = block_scan(len > T ? 1 : 0, nl);
            if (nl <= slots / 2) break;
            T *= 2;
        }
        const bool lg = len > T;
        int total = 0;
        (void)block_scan(lg ? len : 0, total);
        int clen = nl ? (total + (slots - nl) - 1) / (slots - nl) : 8;
        clen = clen < 8 ? 8 : (clen + 3) & ~3;
        const int nch = lg ? (len + clen - 1) / clen : 0;
        int nc = 0;
        const int fx = block_scan(nch, nc);                              // nc <= total / clen + nl <= slots
        if (tid < nr) rowx[tid] = lg ? (uint8_t)x : (uint8_t)0xFF;
        if (lg) {
            longrow[x] = tid;
            longfirst[x] = fx;
            for (int j = 0; j < nch; ++j) { chunk_beg[fx + j] = (int)crow[tid] + j * clen; chunk_row[fx + j] = (uint8_t)x; }
        }
        if (tid == 0) { longfirst[nl] = nc; sh_nlong = nl; sh_nchunk = nc; sh_clen = clen; sh_long_t = T; }
        __syncthreads();
    };
    const int kq = min(k, nr);
    const bool build_failed = failed;
    // The solve itself, for a block of P columns (P = 64: k <= 32 wanted + guards, rounds 2-4; P = 32: round 5).  Of the
    // top k of the MERGED spectrum the quotient only has to deliver what the zp stalk contrasts at 1/sqrt(2) leave room
    // for -- k - zp pairs, 17 on average and at most 20 for 72 % of the C2 workload's items (hub ego-nets carry 12-30
    // stalk contrasts) -- so a 32-column block holds them with >= 12 guard columns at half the cost per sparse product, a
    // quarter per Gram matrix / rotation and an eighth per Ritz problem.  Returns 0: done (or handed to the dense classes),
    // 2: (P = 32 only) the first Ritz values show more wanted pairs than a narrow block converges: redo with P = 64.
    auto solve = [&](auto pc) -> int {
    constexpr int P = decltype(pc)::value;
    constexpr int Ldy = P + 1;
    constexpr int E = P * P / kChThreads;                    // entries of a P x P matrix per thread: 4 (P = 64) / 1 (P = 32)
    constexpr int TPM = P / E;                               // threads per matrix row: 16 / 32
    const int mi = tid / TPM, mj = E * (tid % TPM);          // this thread's strip of a P x P matrix: row mi, columns mj .. mj + E - 1
    constexpr int TPR = P / 8;                               // threads per block row in the products (8 columns each)
    constexpr int TPQ = P / 4;                               // ... in the rotation / tile loads (4 columns each)
    bool failed = build_failed;
    float *XA = XA0, *XB = XB0, *XC = XC0;
    build_chunks(kChSlabFloats / P);
    const int nchunk = sh_nchunk, nlong = sh_nlong, clen = sh_clen, longT = sh_long_t;
#ifdef GCC_AMD_HIPEMU
    if (getenv("GCC_POSEMB_DEBUG") && tid == 0) fprintf(stderr, "cheb item b=%d n=%d nr=%d nnz=%d nlong=%d nchunk=%d fail=%d P=%d clen=%d T=%d\n", b, n, nr, (int)crow[nr], nlong, nchunk, sh_fail, P, clen, longT);
#endif

    // dst = alpha * (M' src - center * src) - gamma * dst   (rows of 64 floats; 16 threads x float4 per row)
    // 8 threads per row (8 block vectors = 2 x float4 each), 128 rows at a time, four gathers per vector pair in flight.
    // With the block in L2 a product is bound by the latency of its gathers (~29 us at n' ~ 300; 4 threads per row with 16
    // vectors each was measured slower: 1.27 against 0.86 ms of products per item).  Blocks of at most kChLdsRows rows are
    // therefore copied into LDS first (coalesced, once per product: the region of the dense matrices is free while the
    // filter runs) and gathered from there; the result still goes to the L2-resident buffer, coalesced.
    constexpr int kChLdsRows = cheb_region_bytes() / (P * (int)sizeof(float));
    const bool lds_x = nr <= kChLdsRows;                     // block-uniform
    auto gather = [&](auto in_lds, const float *src, int e0, int e1, float *acc) {
        const float *base = decltype(in_lds)::value ? (const float *)region : src;
        const int q8 = 8 * (tid % TPR);
        for (int e = e0; e < e1; e += 4) {
            int cj[4];
            float sc[4];
            float4 xa[4], xb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool in = e + u < e1;
                cj[u] = (int)ccol[in ? e + u : e0];
                sc[u] = in ? scale[cj[u]] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xa[u] = *(const float4 *)(base + (int64_t)cj[u] * P + q8);
                xb[u] = *(const float4 *)(base + (int64_t)cj[u] * P + q8 + 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[0] = fmaf(sc[u], xa[u].x, acc[0]); acc[1] = fmaf(sc[u], xa[u].y, acc[1]);
                acc[2] = fmaf(sc[u], xa[u].z, acc[2]); acc[3] = fmaf(sc[u], xa[u].w, acc[3]);
                acc[4] = fmaf(sc[u], xb[u].x, acc[4]); acc[5] = fmaf(sc[u], xb[u].y, acc[5]);
                acc[6] = fmaf(sc[u], xb[u].z, acc[6]); acc[7] = fmaf(sc[u], xb[u].w, acc[7]);
            }
        }
    };
    auto spmm_body = [&](auto in_lds, const float *src, float *dst, float alpha, float center, float gamma) {
        const float *base = decltype(in_lds)::value ? (const float *)region : src;
        const int q8 = 8 * (tid % TPR), g8 = tid / TPR;
        auto finish = [&](int r, const float *acc) {
            const float sr = scale[r];
            const float4 oa = *(const float4 *)(base + (int64_t)r * P + q8), ob = *(const float4 *)(base + (int64_t)r * P + q8 + 4);
            float o[8] = {alpha * (sr * acc[0] - center * oa.x), alpha * (sr * acc[1] - center * oa.y),
                          alpha * (sr * acc[2] - center * oa.z), alpha * (sr * acc[3] - center * oa.w),
                          alpha * (sr * acc[4] - center * ob.x), alpha * (sr * acc[5] - center * ob.y),
                          alpha * (sr * acc[6] - center * ob.z), alpha * (sr * acc[7] - center * ob.w)};
            if (gamma != 0.f) {
                const float4 da = *(const float4 *)(dst + (int64_t)r * P + q8), db = *(const float4 *)(dst + (int64_t)r * P + q8 + 4);
                o[0] -= gamma * da.x; o[1] -= gamma * da.y; o[2] -= gamma * da.z; o[3] -= gamma * da.w;
                o[4] -= gamma * db.x; o[5] -= gamma * db.y; o[6] -= gamma * db.z; o[7] -= gamma * db.w;
            }
            *(float4 *)(dst + (int64_t)r * P + q8) = make_float4(o[0], o[1], o[2], o[3]);
            *(float4 *)(dst + (int64_t)r * P + q8 + 4) = make_float4(o[4], o[5], o[6], o[7]);
        };
        for (int c = g8; c < nchunk; c += kChThreads / TPR) {            // chunks of the long rows -> slab ...
            const int r = longrow[chunk_row[c]];
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            gather(in_lds, src, chunk_beg[c], min(chunk_beg[c] + clen, (int)crow[r + 1]), acc);
            *(float4 *)(slab + c * P + q8) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *(float4 *)(slab + c * P + q8 + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
        for (int r = g8; r < nr; r += kChThreads / TPR) {                // ... and the short rows, in the same pass
            const int e0 = (int)crow[r], e1 = (int)crow[r + 1];
            if (e1 - e0 > longT) continue;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            gather(in_lds, src, e0, e1, acc);
            finish(r, acc);
        }
        if (nchunk) {
            __syncthreads();
            for (int x = g8; x < nlong; x += kChThreads / TPR) {         // the long rows: sums of their chunks, in chunk order
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int c = longfirst[x]; c < longfirst[x + 1]; ++c) {
                    const float4 pa = *(const float4 *)(slab + c * P + q8), pb = *(const float4 *)(slab + c * P + q8 + 4);
                    acc[0] += pa.x; acc[1] += pa.y; acc[2] += pa.z; acc[3] += pa.w;
                    acc[4] += pb.x; acc[5] += pb.y; acc[6] += pb.z; acc[7] += pb.w;
                }
                finish(longrow[x], acc);
            }
        }
        __syncthreads();
    };
    // One filter step with the block RESIDENT in LDS (lds_x: nr * P floats fit the region; round 5).  Y_{i-1} is gathered
    // from LDS, the new rows wait in registers until every gather of the step is done and then replace it; Y_{i-2} is
    // thread-private -- a thread computes the same (row, 8 columns) strips in every step -- and lives in `prev` (the
    // workspace), read and overwritten by its owner only.  What a product still moves through L2 is off its chain: the
    // per-product copy of the block into LDS, the read-modify-write of the target block and the wait for its stores
    // (a third of a product's time at n' ~ 300) are gone.
    constexpr int RPP = kChThreads / TPR;                    // rows per pass of the workgroup: 128 (P = 64) / 256 (P = 32)
    constexpr int kPasses = 3;
    static_assert(kChLdsRows == kPasses * RPP, "an LDS-resident block is covered in kPasses passes");
    auto filter_step_lds = [&](float *prev, float alpha, float center, float gamma) {
        float *xs = (float *)region;
        const int q8 = 8 * (tid % TPR), g8 = tid / TPR;
        float nw[kPasses][8];
        auto finish = [&](int ps, int r, const float *acc) {             // -> nw[ps]; Y_{i-1}'s row becomes the next step's Y_{i-2}
            float4 da = make_float4(0.f, 0.f, 0.f, 0.f), db = da;
            float *pr_ = prev + (int64_t)r * P + q8;
            if (gamma != 0.f) { da = *(const float4 *)pr_; db = *(const float4 *)(pr_ + 4); }
            const float sr = scale[r];
            const float4 oa = *(const float4 *)(xs + r * P + q8), ob = *(const float4 *)(xs + r * P + q8 + 4);
            nw[ps][0] = alpha * (sr * acc[0] - center * oa.x) - gamma * da.x; nw[ps][1] = alpha * (sr * acc[1] - center * oa.y) - gamma * da.y;
            nw[ps][2] = alpha * (sr * acc[2] - center * oa.z) - gamma * da.z; nw[ps][3] = alpha * (sr * acc[3] - center * oa.w) - gamma * da.w;
            nw[ps][4] = alpha * (sr * acc[4] - center * ob.x) - gamma * db.x; nw[ps][5] = alpha * (sr * acc[5] - center * ob.y) - gamma * db.y;
            nw[ps][6] = alpha * (sr * acc[6] - center * ob.z) - gamma * db.z; nw[ps][7] = alpha * (sr * acc[7] - center * ob.w) - gamma * db.w;
            *(float4 *)pr_ = oa;
            *(float4 *)(pr_ + 4) = ob;
        };
        for (int c = g8; c < nchunk; c += RPP) {                         // chunks of the long rows -> slab ...
            const int r = longrow[chunk_row[c]];
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            gather(LdsYes(), xs, chunk_beg[c], min(chunk_beg[c] + clen, (int)crow[r + 1]), acc);
            *(float4 *)(slab + c * P + q8) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *(float4 *)(slab + c * P + q8 + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
#pragma unroll
        for (int ps = 0; ps < kPasses; ++ps) {                           // ... and this thread's short rows, in the same pass
            const int r = g8 + ps * RPP;
            if (r < nr) {
                const int e0 = (int)crow[r], e1 = (int)crow[r + 1];
                if (e1 - e0 <= longT) {
                    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    gather(LdsYes(), xs, e0, e1, acc);
                    finish(ps, r, acc);
                }
            }
        }
        if (nchunk) {
            __syncthreads();
#pragma unroll
            for (int ps = 0; ps < kPasses; ++ps) {                       // this thread's long rows: sums of their chunks, in chunk order
                const int r = g8 + ps * RPP;
                if (r < nr && rowx[r] != 0xFF) {
                    const int x = rowx[r];
                    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    for (int c = longfirst[x]; c < longfirst[x + 1]; ++c) {
                        const float4 pa = *(const float4 *)(slab + c * P + q8), pb = *(const float4 *)(slab + c * P + q8 + 4);
                        acc[0] += pa.x; acc[1] += pa.y; acc[2] += pa.z; acc[3] += pa.w;
                        acc[4] += pb.x; acc[5] += pb.y; acc[6] += pb.z; acc[7] += pb.w;
                    }
                    finish(ps, r, acc);
                }
            }
        }
        __syncthreads();                                                 // every gather of Y_{i-1} is done
#pragma unroll
        for (int ps = 0; ps < kPasses; ++ps) {
            const int r = g8 + ps * RPP;
            if (r < nr) {
                *(float4 *)(xs + r * P + q8) = make_float4(nw[ps][0], nw[ps][1], nw[ps][2], nw[ps][3]);
                *(float4 *)(xs + r * P + q8 + 4) = make_float4(nw[ps][4], nw[ps][5], nw[ps][6], nw[ps][7]);
            }
        }
        __syncthreads();
    };
    // dst = alpha * (M' src - center * src) - gamma * dst   (rows of 64 floats)
    auto spmm = [&](const float *src, float *dst, float alpha, float center, float gamma) {
        if (lds_x) {
            float4 *xs = (float4 *)region;
            const float4 *gs = (const float4 *)src;
            for (int i = tid; i < nr * (P / 4); i += kChThreads) xs[i] = gs[i];
            __syncthreads();
            spmm_body(LdsYes(), src, dst, alpha, center, gamma);
        } else {
            spmm_body(LdsNo(), src, dst, alpha, center, gamma);
        }
    };

    double *G = (double *)region, *K = G + P * P;            // fp64 [P][P] each
    float cut = 0.2f;
    int deg = 6, remaining = 6, round = 0, nrr = 0;
    unsigned long long flops = 0;                            // executed FLOPs of this item (diagnostics, ticks[class][14])
    const unsigned long long nnz_ = (unsigned long long)crow[nr], nr_ = (unsigned long long)nr;
    bool converged = false;
    if (!failed) {
        for (int i = tid; i < nr * P; i += kChThreads)       // start block: U(-1, 1), np.random.rand's role
            XA[i] = hash_unit((uint32_t)a.seed ^ ((uint32_t)gb * 0x9E3779B1u), (uint32_t)(i & (P - 1)), (uint32_t)(i / P));
        __syncthreads();
    }
    // A round = filter of degree `deg`, then either a cheap re-orthonormalisation (X <- X R^-1) or, when the degrees
    // planned after the last Ritz step are spent, a Rayleigh-Ritz step with the convergence test.
    for (; round < kChRounds && !failed; ++round) {
        remaining -= deg;
#ifdef GCC_AMD_HIPEMU
        if (getenv("GCC_POSEMB_RR_ALWAYS")) remaining = 0;
#endif
        const bool rr = remaining <= 0;                      // block-uniform
        // the first Ritz step only needs the Ritz VALUES (cut-off and rate of the filter rounds to come): tridiagonalisation
        // and bisection of the projected matrix, no vectors -- the block is just re-orthonormalised
        const bool vals_only = rr && nrr == 0, fullrr = rr && !vals_only;
        // ---- filter: scaled Chebyshev polynomial of degree `deg` (even) for [-1, cut], p(1) = 1
        if (lds_x && !(hd.use_cheb & 4)) {
            const float e = 0.5f * (cut + 1.0f), cen = 0.5f * (cut - 1.0f);
            float sigma = e / (1.0f - cen);
            const float tau = 2.0f / sigma;
            float4 *xs = (float4 *)region;
            for (int i = tid; i < nr * (P / 4); i += kChThreads) xs[i] = ((const float4 *)XA)[i];
            __syncthreads();
            filter_step_lds(XB, sigma / e, cen, 0.f);                    // Y_1 (XB keeps Y_0's rows, thread-private)
            for (int i = 2; i <= deg; ++i) {
                const float sn = 1.0f / (tau - sigma);
                filter_step_lds(XB, 2.0f * sn / e, cen, sigma * sn);
                sigma = sn;
            }
            for (int i = tid; i < nr * (P / 4); i += kChThreads) ((float4 *)XA)[i] = xs[i];     // the filtered block, for the Gram / rotation phases
            if (rr) spmm_body(LdsYes(), XA, XB, 1.0f, 0.f, 0.f);         // W = M' X, gathered from the block still in LDS
            else __syncthreads();
        } else {
            const float e = 0.5f * (cut + 1.0f), cen = 0.5f * (cut - 1.0f);
            float sigma = e / (1.0f - cen);
            const float tau = 2.0f / sigma;
            spmm(XA, XB, sigma / e, cen, 0.f);                           // Y_1 -> B
            float *prev = XA, *cur = XB;
            for (int i = 2; i <= deg; ++i) {
                const float sn = 1.0f / (tau - sigma);
                spmm(cur, prev, 2.0f * sn / e, cen, sigma * sn);          // Y_i overwrites Y_{i-2}
                float *t = prev; prev = cur; cur = t;
                sigma = sn;
            }                                                            // deg even: the result is in XA
            if (rr) spmm(XA, XB, 1.0f, 0.f, 0.f);                        // W = M' X
        }
        // products: 2 nnz 64 + 5 n 64 each; Gram matrices 2 n 64^2 (x2 with K); Cholesky + inverse + projected matrix ~ 64^3 x 2;
        // a Ritz problem 2 64^3 + 2 64^3 (tridiagonalisation, back-transformation of 64 vectors); rotation 2 n 64^2 (x2 with W)
        flops += (unsigned long long)(deg + (rr ? 1 : 0)) * (2ull * nnz_ * P + 5ull * nr_ * P)
                 + (rr ? 2ull : 1ull) * 2ull * nr_ * P * P + 2ull * P * P * P
                 + (rr ? 4ull * P * P * P : 0ull) + (fullrr ? 2ull : 1ull) * 2ull * nr_ * P * P;
        PHASE_TICK(1);                                                   // sparse products
        // ---- G = X^T X (and K = X^T W), fp64 products and accumulation ON THE MATRIX CORES (round 6): a 16 x 16 block of G (and
        //      the same block of K) per wave, v_mfma_f64_16x16x4_f64 over four rows of the block per instruction.  Both operands
        //      of a step are 16-float segments of the same four rows of X (W): lane (j, q) loads X[r0 + q][16 bi + j] and
        //      X[r0 + q][16 bj + j] straight from the L2-resident block -- no LDS tile, no barrier inside the loop.  The products
        //      of two fp32 values are exact in fp64, so this is the sum the vector loop formed (sum order aside).  (Rounds 2-5:
        //      tiles through LDS, per thread a strip of E entries, one LDS read per 4-8 fp64 FMAs: 113 us of a 1,236 us item.)
        {
            constexpr int NB = P / 16;                            // blocks per side: 4 (P = 64) / 2 (P = 32)
            constexpr int kUn = 4;                                // steps (of four rows) requested together
            if (wv < NB * NB) {                                   // (P = 32: four of the sixteen waves)
                const int bi = wv / NB, bj = wv % NB, j = lane & 15, q = lane >> 4;
                const f64x4 z4 = {0.0, 0.0, 0.0, 0.0};
                f64x4 ga[2] = {z4, z4}, ka[2] = {z4, z4};         // two accumulators per product: consecutive steps do not wait for each other
                const float *xa_ = XA + 16 * bi + j, *xb_ = XA + 16 * bj + j, *wb_ = XB + 16 * bj + j;
                for (int r0 = 0; r0 < nr; r0 += 4 * kUn) {
                    float av[kUn], bv[kUn], wv_[kUn];
#pragma unroll
                    for (int u = 0; u < kUn; ++u) {               // (rows past the end read row nr - 1 and are zeroed below)
                        const int64_t ro = (int64_t)min(r0 + 4 * u + q, nr - 1) * P;
                        av[u] = xa_[ro];
                        bv[u] = xb_[ro];
                        wv_[u] = rr ? wb_[ro] : 0.f;              // (block-uniform)
                    }
#pragma unroll
                    for (int u = 0; u < kUn; ++u) {
                        const bool in = r0 + 4 * u + q < nr;
                        const double a_ = in ? (double)av[u] : 0.0;
                        ga[u & 1] = mfma_16x16x4_f64(a_, (double)bv[u], ga[u & 1]);
                        if (rr) ka[u & 1] = mfma_16x16x4_f64(a_, (double)wv_[u], ka[u & 1]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {                     // c/d[r] = C[q + 4 r][j]
                    const int row = 16 * bi + q + 4 * r, col = 16 * bj + j;
                    G[row * P + col] = ga[0][r] + ga[1][r];
                    K[row * P + col] = ka[0][r] + ka[1][r];
                }
            }
        }
        __syncthreads();
        PHASE_TICK(5);                                           // Gram matrices
        // ---- small dense algebra, all 16 waves, fp64 in LDS: Jacobi scaling G^ = D G D; shifted Cholesky G^ = L L^T
        //      (right-looking, two barriers per column); Linv = L^-1 by recursive doubling over the diagonal blocks
        //      (inv [A 0; B C] = [A^-1 0; -C^-1 B A^-1  C^-1]: 6 levels); with a Ritz step H = Linv K^ Linv^T
        double *Li = K + P * P;                                  // [P][P] L^-1 (lower); upper triangle = scratch
        if (tid < P) {
            const double dd = G[tid * P + tid];
            const double di = dd > 1e-300 ? 1.0 / sqrt(dd) : 0.0;
            if (!(dd > 1e-300)) sh_fail = 1;
            dsc[tid] = (float)di;
            dscd[tid] = di;                                      // (an fp32 copy here would break the congruence D G D by 1e-7)
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < E; ++u) {
            const int i = mi, j = mj + u;
            const double sc = dscd[i] * dscd[j];
            // shifted Cholesky: guard columns that have collapsed onto the span of the others are damped instead of
            // breaking the factorisation; a converged block (unit pivots) is not affected
            G[i * P + j] = G[i * P + j] * sc + (i == j ? kChShift : 0.0);
            if (j <= i) {                                        // K^ lower part, symmetrised
                const double kv = 0.5 * (K[i * P + j] + K[j * P + i]) * sc;
                Li[i * P + j] = kv;                              // (parked in Li until K's upper part has been read)
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < E; ++u) {
            const int i = mi, j = mj + u;
            K[i * P + j] = j <= i ? Li[i * P + j] : Li[j * P + i];
        }
        __syncthreads();
        // Cholesky of the 64 x 64 matrix by ONE wave with no barrier inside (the panel version it replaces -- panels of 16
        // columns by one wave, rank-16 updates by all 16 -- spent 64 us per factorisation in its 12 barriers and fp64 LDS
        // traffic): lane = row, half a row (32 columns) in fp64 REGISTERS at a time (1024-thread workgroups leave 128
        // registers per lane), the column l of step k broadcast with v_readlane, the next column captured while the rows
        // are updated (as wave_tridiagonalize).  Columns 0..31 first; then columns 32..63 are loaded, the 32 finished
        // steps are replayed on them, and the factorisation continues.  L^T is written row-contiguous into Li (free
        // here) and transposed into G's lower triangle by all waves afterwards.
        if (wv == 0) {
            double *Lt = Li;                                     // [k][i] = L[i][k]
            double ar[32];
            bool bad = false;
            const bool live = lane < P;                          // (P = 32: the upper half of the wave has no row)
#pragma unroll 1
            for (int half = 0; half < P / 32 && !bad; ++half) {
                const int cb = 32 * half;
#pragma unroll
                for (int c = 0; c < 32; ++c) ar[c] = live ? G[lane * P + cb + c] : 0.0;
                if (half == 1) {                                 // replay steps 0..31 on the right half
#pragma unroll 1
                    for (int k = 0; k < 32; ++k) {
                        const double l = Lt[k * P + lane];       // 0 above the diagonal (written below)
#pragma unroll
                        for (int u = 0; u < 32; ++u) ar[u] = fma(-l, wave_readlane(l, 32 + u), ar[u]);
                    }
                }
                double cap = ar[0];
#pragma unroll 1
                for (int k = cb; k < cb + 32; ++k) {
                    const double piv = wave_readlane(cap, k);
                    if (!(piv > 0.1 * kChShift)) {               // wave-uniform: the block lost its numerical rank
#ifdef GCC_AMD_HIPEMU
                        if (getenv("GCC_POSEMB_DEBUG") && lane == 0) fprintf(stderr, "cheb chol fail round=%d col=%d piv=%g\n", round, k, piv);
#endif
                        bad = true;
                        break;
                    }
                    const double l = (live && lane >= k) ? cap * (1.0 / sqrt(piv)) : 0.0;  // L[lane][k]
                    if (live) Lt[k * P + lane] = l;
#pragma unroll
                    for (int jb = 0; jb < 4; ++jb) {
                        if (cb + 8 * jb + 7 > k) {               // wave-uniform: some column of the block is right of k
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const int j = 8 * jb + u;
                                ar[j] = fma(-l, wave_readlane(l, cb + j), ar[j]);
                                if (cb + j == k + 1) cap = ar[j];
                            }
                        }
                    }
                }
            }
            if (bad && lane == 0) sh_fail = 1;
        }
        __syncthreads();
        if (!sh_fail) {                                          // L -> lower triangle of G (what the inverse below reads)
#pragma unroll
            for (int u = 0; u < E; ++u)
                if (mj + u <= mi) G[mi * P + mj + u] = Li[(mj + u) * P + mi];
        }
        __syncthreads();
        if (sh_fail) { failed = true; break; }
        PHASE_TICK(6);                                           // Cholesky
        {   // Li = L with inverted diagonal; then the doubling levels
            const int i = mi, j4 = mj;
#pragma unroll
            for (int u = 0; u < E; ++u) {
                const int j = j4 + u;
                Li[i * P + j] = j < i ? G[i * P + j] : (j == i ? 1.0 / G[i * P + i] : 0.0);
            }
            __syncthreads();
            for (int sz = 1; sz < P; sz <<= 1) {
                const int per = sz * sz, pairs = P / (2 * sz);
                // T = B A, stored transposed in the (free) upper triangle
                for (int e = tid; e < pairs * per; e += kChThreads) {
                    const int pr = e / per, ii = (e - pr * per) / sz, jj = e % sz, r0 = 2 * sz * pr;
                    double acc = 0.0;
                    for (int q = jj; q < sz; ++q) acc += Li[(r0 + sz + ii) * P + r0 + q] * Li[(r0 + q) * P + r0 + jj];
                    Li[(r0 + jj) * P + r0 + sz + ii] = acc;
                }
                __syncthreads();
                // B' = -C T
                for (int e = tid; e < pairs * per; e += kChThreads) {
                    const int pr = e / per, ii = (e - pr * per) / sz, jj = e % sz, r0 = 2 * sz * pr;
                    double acc = 0.0;
                    for (int q = 0; q <= ii; ++q) acc += Li[(r0 + sz + ii) * P + r0 + sz + q] * Li[(r0 + jj) * P + r0 + sz + q];
                    Li[(r0 + sz + ii) * P + r0 + jj] = -acc;          // (B itself is not read in this phase)
                }
                __syncthreads();
            }
            if (rr) {
                // T1 = Linv K^ (into G), H = T1 Linv^T (into K): two P x P x P fp64 products on the matrix cores (round 6), a
                // 16 x 16 block per wave, operands from the fp64 matrices in LDS.  Linv is lower triangular and its upper triangle
                // holds scratch of the doubling levels: entries above the diagonal are read as zero, blocks above it are skipped.
                // (Per thread E entries with a dot product of up to P terms each before: ~10 us per product.)
                constexpr int NB = P / 16;
                const int bi = wv / NB, bj = wv % NB, jl = lane & 15, ql = lane >> 4;
                const f64x4 z4 = {0.0, 0.0, 0.0, 0.0};
                if (wv < NB * NB) {
                    f64x4 acc = z4;
                    for (int k0 = 0; k0 < 16 * (bi + 1); k0 += 4) {      // T1[i][j] = sum_{q <= i} Linv[i][q] K[q][j]
                        const int ii = 16 * bi + jl, kk = k0 + ql;
                        const double a_ = kk <= ii ? Li[ii * P + kk] : 0.0;
                        acc = mfma_16x16x4_f64(a_, K[kk * P + 16 * bj + jl], acc);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) G[(16 * bi + ql + 4 * r) * P + 16 * bj + jl] = acc[r];
                }
                __syncthreads();
                if (wv < NB * NB) {
                    f64x4 acc = z4;
                    for (int k0 = 0; k0 < 16 * (bj + 1); k0 += 4) {      // H[i][j] = sum_{q <= j} T1[i][q] Linv[j][q]
                        const int jj = 16 * bj + jl, kk = k0 + ql;
                        const double b_ = kk <= jj ? Li[jj * P + kk] : 0.0;
                        acc = mfma_16x16x4_f64(G[(16 * bi + jl) * P + kk], b_, acc);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) K[(16 * bi + ql + 4 * r) * P + 16 * bj + jl] = acc[r];
                }
                __syncthreads();
            }
        }
        // ---- LDS region from here on: Lf (Linv, fp32) | Af (H, later C with stride P) | Y | solver arrays.  G, K and Li
        //      are dead once Linv and H have been copied out (through registers: the fp32 copies overlay them)
        float *Lf = (float *)region, *Af = Lf + P * Ldy;
        {
            float lrow[E], hrow[E];
            const int i = mi, j4 = mj;
#pragma unroll
            for (int u = 0; u < E; ++u) {
                lrow[u] = j4 + u <= i ? (float)Li[i * P + j4 + u] : 0.f;
                hrow[u] = (float)(0.5 * (K[i * P + j4 + u] + K[(j4 + u) * P + i]));
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < E; ++u) {
                Lf[i * Ldy + j4 + u] = lrow[u];
                Af[i * Ldy + j4 + u] = hrow[u];
            }
        }
        PHASE_TICK(7);                                           // triangular inverse, projected matrix
        TriLds tw;
        tw.Y = Af + P * Ldy;
        tw.ldy = Ldy;
        tw.dg = tw.Y + P * Ldy;
        tw.of = tw.dg + P;
        tw.of2 = tw.of + P;
        tw.tau = tw.of2 + P;
        tw.pbuf = tw.tau + P;
        tw.vbuf = tw.pbuf + P;
        tw.coef = tw.vbuf + P;
        tw.cnt = (int *)(tw.coef + P * Ldy);
        tw.bw = kChBw;
        tw.ldu = kChBw + 1;
        tw.Ud = (float *)(tw.cnt + kChThreads);
        tw.Us = tw.Ud + P * tw.ldu;
        tw.Uf = (uint8_t *)(tw.Us + P * tw.ldu);
        __syncthreads();
        if (rr) {
            // Ritz problem: all P pairs of H by the dense solver core
            tridiagonalize<1, kChThreads, 2>(Af, Ldy, P, tw);
            eig_top_values<kChThreads, kVecCap>(tw, P, P, es);
            const bool ritz_failed = vals_only ? false
                : eig_top_vectors<1, kChThreads>(Af, Ldy, P, P, tw, es,
                                                 (uint32_t)a.seed ^ ((uint32_t)gb * 0x9E3779B1u) ^ (uint32_t)(round + 77), nullptr, tick_);
#ifdef GCC_AMD_HIPEMU
            if (getenv("GCC_POSEMB_DEBUG") && tid == 0 && ritz_failed) fprintf(stderr, "cheb ritz failed round=%d\n", round);
#endif
            if (ritz_failed) { failed = true; break; }           // block-uniform
            ++nrr;
            if (tid < P) theta[tid] = es.lamv[tid];
        }
        PHASE_TICK(2);                                           // Ritz problem
        // ---- C = D Linv^T Y (Y = I without a Ritz step), stride P, over H
        {
            float c[E];
#pragma unroll
            for (int u = 0; u < E; ++u) c[u] = 0.f;
            if (fullrr) {
                for (int q = mi; q < P; ++q) {                   // Linv^T[i][q] = Linv[q][i], q >= i
                    const float l = Lf[q * Ldy + mi];
#pragma unroll
                    for (int u = 0; u < E; ++u) c[u] = fmaf(l, tw.Y[q * Ldy + mj + u], c[u]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < E; ++u) c[u] = mj + u >= mi ? Lf[(mj + u) * Ldy + mi] : 0.f;
            }
            const float di = dsc[mi];
            __syncthreads();                                     // H (Af) and the reflectors in it are dead: C goes there
#pragma unroll
            for (int u = 0; u < E; ++u) Af[mi * P + mj + u] = c[u] * di;
        }
        __syncthreads();
        PHASE_TICK(8);                                           // C matrix
        // ---- X <- X C into the free buffer (W C stays in registers); residuals ||W_i - theta_i X_i||^2 accumulated per
        //      thread, then over the 64 row groups
        {
            const int q4 = 4 * (tid % TPQ), g16 = tid / TPQ;
            float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
            const float t0 = theta[q4], t1 = theta[q4 + 1], t2 = theta[q4 + 2], t3 = theta[q4 + 3];
            for (int r = g16; r < nr; r += kChThreads / TPQ) {
                float xn[4] = {0.f, 0.f, 0.f, 0.f}, wn[4] = {0.f, 0.f, 0.f, 0.f};
                for (int which = 0; which < (fullrr ? 2 : 1); ++which) {
                    const float *src = (which ? XB : XA) + (int64_t)r * P;
                    float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
                    // 16 columns (4 x 16 bytes) of the row in flight at once.  (Rounds 2-4 kept 32: with the 128 registers a
                    // 1024-thread workgroup leaves per lane that loop body spilled 38 values per iteration -- the rotation of a
                    // 32-column block took twice as long as that of a 64-column one.)
#pragma unroll 1
                    for (int part = 0; part < P / 16; ++part) {
                        float4 rowv[4];
#pragma unroll
                        for (int p = 0; p < 4; ++p) rowv[p] = *(const float4 *)(src + 16 * part + 4 * p);
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const float4 rv = rowv[p];
                            const float xs[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const float4 cv = *(const float4 *)(Af + (16 * part + 4 * p + u) * P + q4);
                                o0 = fmaf(xs[u], cv.x, o0); o1 = fmaf(xs[u], cv.y, o1);
                                o2 = fmaf(xs[u], cv.z, o2); o3 = fmaf(xs[u], cv.w, o3);
                            }
                        }
                    }
                    if (which) {                                         // the rotated W is only needed for the residual
                        wn[0] = o0; wn[1] = o1; wn[2] = o2; wn[3] = o3;
                    } else {
                        *(float4 *)(XC + (int64_t)r * P + q4) = make_float4(o0, o1, o2, o3);
                        xn[0] = o0; xn[1] = o1; xn[2] = o2; xn[3] = o3;
                    }
                }
                const float d0 = wn[0] - t0 * xn[0], d1 = wn[1] - t1 * xn[1], d2 = wn[2] - t2 * xn[2], d3 = wn[3] - t3 * xn[3];
                r0 = fmaf(d0, d0, r0); r1 = fmaf(d1, d1, r1); r2 = fmaf(d2, d2, r2); r3 = fmaf(d3, d3, r3);
            }
            __syncthreads();
            { float *t = XA; XA = XC; XC = t; }                          // block-uniform: the rotated block is X now
            if (fullrr) {
                *(float4 *)(slab + g16 * P + q4) = make_float4(r0, r1, r2, r3);     // (1024 / TPQ) groups x P columns = 4096 floats
                __syncthreads();
                if (tid < P) {
                    float sres = 0.f;
                    for (int g = 0; g < kChThreads / TPQ; ++g) sres += slab[g * P + tid];
                    resid[tid] = sqrtf(sres);
                }
                __syncthreads();
            }
        }
#ifdef GCC_AMD_HIPEMU
        if (getenv("GCC_POSEMB_DEBUG") && tid == 0) {
            double worst_off = 0, dmin = 1e9, dmax2 = 0, wmax = 0;
            for (int i = 0; i < P; ++i)
                for (int j = 0; j <= i; ++j) {
                    double sdot = 0;
                    for (int r = 0; r < nr; ++r) sdot += (double)XA[(int64_t)r * P + i] * XA[(int64_t)r * P + j];
                    if (i == j) { dmin = sdot < dmin ? sdot : dmin; dmax2 = sdot > dmax2 ? sdot : dmax2; }
                    else worst_off = fabs(sdot) > worst_off ? fabs(sdot) : worst_off;
                }
            for (int i = 0; i < P; ++i) { double sw = 0; for (int r = 0; r < nr; ++r) sw += (double)XB[(int64_t)r * P + i] * XB[(int64_t)r * P + i]; wmax = sw > wmax ? sw : wmax; }
            fprintf(stderr, "cheb gram-after-rotate round=%d rr=%d P=%d deg=%d: diag [%g, %g] offdiag %g  max|W col|^2 %g\n", round, (int)rr, P, deg, dmin, dmax2, worst_off, wmax);
        }
        __syncthreads();
#endif
        PHASE_TICK(3);                                           // rotation + residuals
        // the degree of one filter is bounded: fp32 keeps ~1e-7 of the dominant eigenvectors in every column and an
        // unconverged guard column much more; the filter magnifies that by T_deg(x(1)) relative to a column at the cut,
        // so beyond ~1e3 the block loses its numerical rank.  T_2m = 2 T_m^2 - 1: consecutive rounds with a
        // re-orthonormalisation in between multiply up like one long filter.
        if (rr) {
            // quotient pairs that can reach the merged top k: everything above the stalk eigenvalue (generous window: the
            // Ritz values still move) and, below it, what the zp stalk contrasts leave room for
            int kw = kq;
            if (zp > 0) {
                int nge = 0;
                for (int j = 0; j < kq; ++j) if (theta[j] >= kStalkEig - 1e-3f) nge = j + 1;
                kw = max(max(nge, min(kq, k - zp)), 1);
            }
            if (P < kChPWide && kw > P - kChNarrowGuards) return 2;   // block-uniform: more wanted pairs than the narrow block's guards allow
            float worst = 0.1f;                                  // (values only: nothing measured yet)
            if (fullrr) {
                worst = 0.f;
                for (int i = 0; i < kw; ++i) worst = fmaxf(worst, resid[i]);
            }
#ifdef GCC_AMD_HIPEMU
            if (getenv("GCC_POSEMB_DEBUG") && tid == 0)
                fprintf(stderr, "cheb b=%d n=%d nr=%d zp=%d round=%d deg=%d cut=%.4f worst=%.2e kw=%d theta[kw-1]=%.5f theta[P-1]=%.5f\n", b, n, nr, zp, round,
                        deg, cut, worst, kw, theta[kw - 1], theta[P - 1]);
#endif
            if (fullrr && worst <= kChTol) { converged = true; ++round; break; }
            if (nrr >= kChMaxRitz) break;
            cut = fminf(fmaxf(theta[P - 1] - 0.02f, -0.9f), 0.9f);
            const float e = 0.5f * (cut + 1.0f), cen = 0.5f * (cut - 1.0f);
            const float xk = fmaxf((theta[kw - 1] - cen) / e, 1.0001f);
            const float rate = logf(xk + sqrtf(xk * xk - 1.0f));        // the wanted end grows by exp(rate) per degree
            int need = (int)(2.3f * logf(fmaxf(worst, 1e-3f) / (0.5f * kChTol)) / rate) + 4;   // a Ritz step costs ~30 degrees: overshoot
            remaining = need < 4 ? 4 : (need > 60 ? 60 : need);
        }
        {
            const float x1 = (1.0f - 0.5f * (cut - 1.0f)) / (0.5f * (cut + 1.0f));
            const float ach = logf(x1 + sqrtf(x1 * x1 - 1.0f));
            int dmax = (int)(kChAmpLog / ach) & ~1;                     // T_dmax(x(1)) <= exp(kChAmpLog)
            dmax = dmax < 4 ? 4 : (dmax > 24 ? 24 : dmax);
            deg = ((remaining + 1) & ~1) < dmax ? ((remaining + 1) & ~1) : dmax;
            if (deg < 2) deg = 2;
        }
    }
    // ranks of the merged spectrum.  The top k must be strictly positive: the null space needs the direct solver (zeros of
    // M' are not resolved by the filter)
    if (converged) {
        __syncthreads();
        if (tid == 0) {
            const int na = rank_columns(theta, kq, k, z, zp, colsrc, nullptr);
            bool ok = na == 0 || theta[na - 1] > 10.f * kZeroEig;
            for (int t = 0; t < k; ++t) {
                const int src = colsrc[t];
                if (src < 0 && (src > -kStalkSrc || ((-src - kStalkSrc) & 1))) ok = false;    // a twin contrast or -1/sqrt(2)
            }
            sh_fail = ok ? 0 : 1;
        }
        __syncthreads();
        if (sh_fail) converged = false;
    }
    if (!converged) {
        // ---- hand the item to the dense classes (they run after this kernel in the stream)
        if (tid == 0) {
            int cls = nr <= kGMax ? kClsSlot : (nr <= kBMax ? kClsBig : kClsKrylov);
            if (n > kNodeMax) cls = kClsKrylov;
            if (cls == kClsKrylov && n >= hd.ldv) {
                atomicOr(a.status, (int32_t)GCC_STATUS_POSEMB_TOO_LARGE);
                cls = -1;
            }
            if (cls >= 0) hd.list[(int64_t)cls * hd.T + atomicAdd(hd.count + cls, 1)] = gb;
            atomicAdd(a.status + 3, 1);                          // diagnostics: items that left their first-choice solver
            sh_fail = cls < 0 ? 2 : 1;
        }
        __syncthreads();
        if (sh_fail == 2) {                                      // no room anywhere: zeros + flag, as the classify kernel does
            for (int i = tid; i < n * a.hidden; i += kChThreads) a.pos[(int64_t)n0 * a.hidden + i] = 0.f;
            if (a.evals) for (int i = tid; i < a.hidden; i += kChThreads) a.evals[(int64_t)b * a.hidden + i] = 0.f;
        }
        return 0;
    }
    // ---- eigenvalues ascending like eigsh(which="LA") (data_util.py:251); expand to the n original nodes;
    //      x = normalize(u, "l2") row-wise, zero padded (data_util.py:260-262)
    if (a.evals) {
        for (int i = tid; i < a.hidden; i += kChThreads)
            a.evals[(int64_t)b * a.hidden + i] = i < k ? (colsrc[i] >= 0 ? theta[colsrc[i]] : kStalkEig) : 0.f;
    }
    for (int v = wv; v < n; v += kNW) {
        const int rsrc = xrec[4 * v], o = xrec[4 * v + 1], g = xrec[4 * v + 2], cb = xrec[4 * v + 3];
        const float val = lane < k ? defl_expand(rsrc, o, g, cb, colsrc[lane], XA, P) : 0.f;
        const float s2 = wave_sum(val * val);
        const float inv = s2 > 0.f ? 1.0f / sqrtf(s2) : 1.0f;
        if (lane < a.hidden) {
            a.pos[(int64_t)(n0 + v) * a.hidden + lane] = val * inv;
            if (a.raw) a.raw[(int64_t)(n0 + v) * a.hidden + lane] = val;
        }
    }
    PHASE_TICK(4);                                               // expansion
    if (m.ticks && tid == 0) atomicAdd((unsigned long long *)&m.ticks[kCls * 16 + 14], flops);
    if (tid == 0) atomicMax(a.status + 1, round);                // diagnostics: most filter rounds of an item
    return 0;
    };   // solve
    // the narrow block where the quotient has to deliver few pairs (k - zp: known once the stalks are counted)
    const int want = max(min(kq, k - zp), 1);
    int rc = 2;
    if (!(hd.use_cheb & 2) && want <= kChNarrowWant) rc = solve(std::integral_constant<int, kChPNarrow>());
    if (rc == 2) {
        __syncthreads();
        solve(std::integral_constant<int, kChPWide>());
    }
    }   // next item
}

}  // namespace

extern "C" {

static long long *g_posemb_ticks = nullptr;
struct PosGrids { int32_t small, mid, slot, kry, big, cheb, w48, w64, pair; };
static PosGrids posemb_grids(int64_t T, bool gated = false)
{
    // fixed grids: enough workgroups for the typical class sizes (~73 % / 19 % / 6 % / 2 % of a batch at rw_hops 256);
    // larger classes loop.  No class may cover more than half of the 256 CUs (small: 2 workgroups per CU): a solver
    // workgroup holds most of a CU's LDS for milliseconds, and when every CU has one the training step's kernels whose
    // workgroups do not fit beside it wait for the whole launch to drain (3-5 ms stalls in the kernel trace).
    static int caps[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (caps[0] == 0) {                              // tuning knob: GCC_POSEMB_GRID_CAPS="small,mid,slot,krylov,big,cheb,w48,w64,pair"
        int c[9] = {256, 64, 128, 64, 64, 96, 128, 64, 192};     // (pair 128 -> 192 once the work lists were sorted, profiles/r6_bench_sorted_lists.txt: the driver's window 0.752 -> 0.740, sustained 0.735 -> 0.742) round 6: the one-wave teams 512 / 128 -> 128 / 64 (their workgroups hold 74 / 111 KiB of LDS each: at two per CU
                                                                 // no CU had room for a 49 KiB gin_in_kernel workgroup while they ran): sustained 0.776 -> 0.743 ms per step, the
                                                                 // driver's 20-step window unchanged (0.785 vs 0.788); smaller caps still (64 / 32, or cheb 64) win another 1-2 %
                                                                 // sustained and LOSE 5-10 % in the 20-step window (profiles/r6_bench_grid_caps.txt)    // the defaults: cheb 96, one-wave teams 512 / 128, pair 128 (scripts/gpu/r3_call4.sh: bench by caps; pair: r5_call10.sh 64: 0.837, 128: 0.836, 256: 0.856 ms per step; the cheb 64 / teams 256, 64 setting of r5_call12.sh measured 0.819 once and inside the spread afterwards: not adopted)
        const char *e = getenv("GCC_POSEMB_GRID_CAPS");
        if (e) (void)sscanf(e, "%d,%d,%d,%d,%d,%d,%d,%d,%d", &c[0], &c[1], &c[2], &c[3], &c[4], &c[5], &c[6], &c[7], &c[8]);
        for (int i = 0; i < 9; ++i) caps[i] = c[i] < 1 ? 1 : c[i];
    }
    PosGrids g;
    g.small = (int32_t)(T < caps[0] ? T : caps[0]);
    // behind a gate the heavy phases of concurrent calls take turns: one call's heavy workgroups may then hold half of
    // the CUs (GCC_POSEMB_GATED_CAPS="mid,cheb")
    static int gcaps[2] = {0, 0};
    if (gcaps[0] == 0) {
        int c[2] = {128, 128};
        const char *e = getenv("GCC_POSEMB_GATED_CAPS");
        if (e) (void)sscanf(e, "%d,%d", &c[0], &c[1]);
        gcaps[0] = c[0] < 1 ? 1 : c[0]; gcaps[1] = c[1] < 1 ? 1 : c[1];
    }
    const int cap_mid = gated ? gcaps[0] : caps[1], cap_cheb = gated ? gcaps[1] : caps[5];
    g.mid = (int32_t)((T + 3) / 4 < cap_mid ? (T + 3) / 4 : cap_mid);
    g.slot = (int32_t)((T + 7) / 8 < caps[2] ? (T + 7) / 8 : caps[2]);
    g.kry = (int32_t)((T + 15) / 16 < caps[3] ? (T + 15) / 16 : caps[3]);
    g.big = (int32_t)((T + 15) / 16 < caps[4] ? (T + 15) / 16 : caps[4]);
    g.cheb = (int32_t)((T + 7) / 8 < cap_cheb ? (T + 7) / 8 : cap_cheb);
    // one-wave teams: kWaveTeams items in flight per workgroup
    const int64_t wg = (T + kWaveTeams - 1) / kWaveTeams;
    g.w48 = (int32_t)(wg < caps[6] ? wg : caps[6]);
    g.w64 = (int32_t)((wg + 1) / 2 < caps[7] ? (wg + 1) / 2 : caps[7]);
    const int cap_pair = gated ? 2 * caps[8] : caps[8];
    g.pair = (int32_t)((T + 1) / 2 < cap_pair ? (T + 1) / 2 : cap_pair);
    return g;
}
// workspace sizing: the larger of the two grid sets, so that one workspace serves gated and ungated calls
static PosGrids posemb_grids_for_sizing(int64_t T)
{
    PosGrids a = posemb_grids(T, false);
    const PosGrids b = posemb_grids(T, true);
    a.mid = a.mid > b.mid ? a.mid : b.mid;
    a.cheb = a.cheb > b.cheb ? a.cheb : b.cheb;
    a.pair = a.pair > b.pair ? a.pair : b.pair;
    return a;
}
static int64_t posemb_head_bytes(int64_t T) { return ((16 + kNumCls * T) * 4 + 255) / 256 * 256; }
static int64_t posemb_slot_floats(void) { return (int64_t)kGMax * kGMax + (int64_t)kNodeMax * 4; }
static int64_t posemb_bslot_floats(void) { return (int64_t)kBMax * kBMax + (int64_t)kNodeMax * 4; }
static int64_t posemb_ldv(int32_t batch_size, int64_t node_cap) { return ((node_cap / batch_size + 63) / 64) * 64 + 64; }

// The solver classes of one call are independent of each other once the classify kernel has written their lists (only the block class
// hands items on: to the Krylov / workspace classes behind it).  On ONE in-order stream each class's launch waits for the previous one
// to drain -- the block class's last long item kept the other classes' workgroups off the machine --, so the call forks: the one-wave
// teams and the 65..128 class run on two side streams of the caller's stream (same priority, created once per caller stream) and join
// it at the end.  OFF by default (gcc_posemb_set_fork / GCC_POSEMB_FORK = 1): an isolated call of 16 views takes 4.4 instead of 8.7 ms and a
// pipeline with nothing else on the GPU gains accordingly, but next to the training step the extra concurrency costs the step more
// than the call gains (scripts/gpu/r5_call15.sh: 0.909 against 0.858 ms per step sustained) -- one in-order stream is the throttle
// the step's latency-bound kernels need.
static int g_posemb_fork = -1;               // gcc_posemb_set_fork: -1 = GCC_POSEMB_FORK decides (default 0)
#ifndef GCC_AMD_HIPEMU
struct PosSide { hipStream_t owner, side[2]; hipEvent_t fork, join[2]; };
static PosSide *posemb_side_streams(hipStream_t s)
{
    static std::mutex mu;
    static PosSide tab[16];
    static int ntab = 0;
    std::lock_guard<std::mutex> lk(mu);
    for (int i = 0; i < ntab; ++i)
        if (tab[i].owner == s) return &tab[i];
    if (ntab == 16) return nullptr;                  // (more caller streams than anyone uses: those calls stay on one stream)
    PosSide &t = tab[ntab];
    int prio = 0;
    if (hipStreamGetPriority(s, &prio) != hipSuccess) { (void)hipGetLastError(); prio = 0; }
    for (int i = 0; i < 2; ++i) {
        if (hipStreamCreateWithPriority(&t.side[i], hipStreamNonBlocking, prio) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipEventCreateWithFlags(&t.join[i], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    if (hipEventCreateWithFlags(&t.fork, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    t.owner = s;
    return &tab[ntab++];
}
#endif

int64_t gcc_posemb_multi_workspace_bytes(int32_t num_views, int32_t batch_size, int64_t node_cap, int32_t hidden)
{
    if (num_views < 1 || num_views > kMaxViews || batch_size < 1 || node_cap < 1 || hidden < 2 || hidden > kMaxVec) {
        snprintf(g_err, kErrLen, "gcc_posemb_multi_workspace_bytes: bad argument");
        return -1;
    }
    const int64_t T = (int64_t)num_views * batch_size;
    const PosGrids g = posemb_grids_for_sizing(T);
    return posemb_head_bytes(T) + g.slot * posemb_slot_floats() * (int64_t)sizeof(float)
           + g.big * posemb_bslot_floats() * (int64_t)sizeof(float)
           + (int64_t)(g.mid + g.small) * kNodeMax * 16
           + (int64_t)g.kry * 2 * (kM + 1) * posemb_ldv(batch_size, node_cap) * (int64_t)sizeof(float)
           + (int64_t)g.cheb * (3 * (int64_t)kNodeMax * kChP * (int64_t)sizeof(float) + (int64_t)kNodeMax * 16) + 512
           + (int64_t)g.pair * kPairSlotFloats * (int64_t)sizeof(float) + 256;
}

int64_t gcc_posemb_workspace_bytes(int32_t batch_size, int64_t node_cap, int32_t hidden)
{
    return gcc_posemb_multi_workspace_bytes(1, batch_size, node_cap, hidden);
}

int32_t gcc_posemb_multi(const gcc_posemb_view *views, int32_t num_views, int32_t batch_size, int64_t node_cap,
                         int32_t hidden, uint64_t seed, void *workspace, int64_t workspace_bytes, int32_t *status,
                         gcc_prof *prof, void *stream)
{
    return gcc_posemb_multi_gated(views, num_views, batch_size, node_cap, hidden, seed, workspace, workspace_bytes, status,
                                  prof, nullptr, nullptr, stream);
}

int32_t gcc_posemb_multi_gated(const gcc_posemb_view *views, int32_t num_views, int32_t batch_size, int64_t node_cap,
                               int32_t hidden, uint64_t seed, void *workspace, int64_t workspace_bytes, int32_t *status,
                               gcc_prof *prof, void *heavy_wait, void *heavy_record, void *stream)
{
    if (!views || !status || num_views < 1 || num_views > kMaxViews || batch_size < 1 || hidden < 2 || hidden > kMaxVec) {
        snprintf(g_err, kErrLen, "gcc_posemb_multi: bad argument");
        return -1;
    }
    const int64_t need = gcc_posemb_multi_workspace_bytes(num_views, batch_size, node_cap, hidden);
    if (!workspace || workspace_bytes < need) {
        snprintf(g_err, kErrLen, "gcc_posemb: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
        return -3;
    }
    PosMulti m;
    memset(&m, 0, sizeof(m));
    for (int i = 0; i < num_views; ++i) {
        if (!views[i].g || !views[i].pos) { snprintf(g_err, kErrLen, "gcc_posemb_multi: view %d is incomplete", i); return -1; }
        m.v[i] = {views[i].g->node_off, views[i].g->row_ptr, views[i].g->col_idx, views[i].pos, views[i].evals, views[i].raw};
    }
    m.nviews = num_views; m.B = batch_size; m.hidden = hidden; m.seed = seed; m.status = status;
    m.ticks = g_posemb_ticks;
    const int64_t T = (int64_t)num_views * batch_size;
    const PosGrids g = posemb_grids(T, heavy_wait != nullptr || heavy_record != nullptr);
    const PosGrids gs = posemb_grids_for_sizing(T);     // the workspace is carved up by the sizing grids
    hipStream_t s = (hipStream_t)stream;
    PosHead hd;
    hd.count = (int32_t *)workspace;
    hd.next = hd.count + 8;
    hd.list = hd.count + 16;
    hd.slots = (float *)((char *)workspace + posemb_head_bytes(T));
    hd.bslots = hd.slots + gs.slot * posemb_slot_floats();
    hd.bslot_floats = posemb_bslot_floats();
    hd.tabs = hd.bslots + gs.big * posemb_bslot_floats();
    hd.tabs_small_off = gs.mid;
    hd.T = (int32_t)T;
    hd.slot_floats = posemb_slot_floats();
    hd.ldv = (int32_t)posemb_ldv(batch_size, node_cap);
    {
        const char *e = getenv("GCC_POSEMB_CHEB");
        hd.use_cheb = e ? atoi(e) : 1;             // 0: dense classes only, 1: on; A/B knobs, OR-ed in: 2 = the wide block only, 4 = the filter through L2 (7 = rounds 2-4)
        const char *ew = getenv("GCC_POSEMB_WAVE");   // 0: the 256-thread small class takes every n' <= 64 (A/B runs)
        hd.use_wave = ew ? atoi(ew) != 0 : 1;
        const char *es = getenv("GCC_POSEMB_STALKS");  // 0: twin leaves only (A/B runs)
        hd.use_stalks = es ? atoi(es) != 0 : 1;
        const char *ep = getenv("GCC_POSEMB_PAIR");    // 0: the 65..128 class on 1,024-thread workgroups with the matrix in LDS (A/B runs)
        hd.use_pair = ep ? atoi(ep) != 0 : 1;
    }
    {   // the register-resident class's slots: the tail of the workspace
        char *end = (char *)workspace + need;
        hd.pslots = (float *)(((uintptr_t)(end - (int64_t)gs.pair * kPairSlotFloats * (int64_t)sizeof(float))) & ~(uintptr_t)255);
    }
    constexpr int lds_small = direct_lds_bytes<kJSmall, kSmallT, false>();
    constexpr int lds_big = direct_lds_bytes<kJMax, kMidT, false>();
#ifndef GCC_AMD_HIPEMU
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)posemb_direct_kernel<kClsSmall, 0, kJSmall, kSmallT, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_small);
        (void)hipFuncSetAttribute((const void *)posemb_direct_kernel<kClsMid, kJSmall + 1, kJMax, kMidT, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_big);
        (void)hipFuncSetAttribute((const void *)posemb_direct_kernel<kClsSlot, kJMax + 1, kGMax, 1024, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kGLds);
        (void)hipFuncSetAttribute((const void *)posemb_direct_kernel<kClsBig, kGMax + 1, kBMax, 1024, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kBLds);
        (void)hipFuncSetAttribute((const void *)posemb_wave_kernel<kClsW48, 48>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kWaveTeams * wave_team_bytes<48>());
        (void)hipFuncSetAttribute((const void *)posemb_wave_kernel<kClsW64, 64>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kWaveTeams * wave_team_bytes<64>());
        attr_set = true;
    }
#endif
    prof_mark(prof, 0, s);
    (void)hipMemsetAsync(workspace, 0, 64, s);       // class counts + work counters
    hipLaunchKernelGGL(posemb_classify_kernel, dim3((unsigned)((T + 3) / 4)), dim3(kClsThreads), 0, s, m, hd);
    static const bool sort_lists = [] { const char *e = getenv("GCC_POSEMB_SORT"); return !e || atoi(e) != 0; }();
    if (sort_lists) hipLaunchKernelGGL(posemb_sort_kernel, dim3(kNumCls), dim3(kSortThreads), 0, s, m, hd);
    KryArgs ka;
    ka.m = m;
    ka.hd = hd;
    ka.vws = hd.tabs + (int64_t)(gs.mid + gs.small) * kNodeMax * 4;
    ka.ldv = (int32_t)posemb_ldv(batch_size, node_cap);
    // longest items first
    const size_t lds_kry = (size_t)3 * ka.ldv * sizeof(float)
                           + sizeof(float) * (6 * kM + kVecCap * (kVecCap + 1) + kKThreads + 2 * kM * (kVecCap + 1)) + kM * kVecCap
                           + sizeof(uint16_t) * ((size_t)kCsrCap + ka.ldv);
#ifndef GCC_AMD_HIPEMU
    static size_t kry_lds_opt_in = 0;
    if (lds_kry > kry_lds_opt_in) {                  // more than 64 KiB of dynamic LDS has to be opted into
        (void)hipFuncSetAttribute((const void *)posemb_krylov_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kry);
        kry_lds_opt_in = lds_kry;
    }
#endif
    // the light classes (one-wave teams, the 256-thread small class) and the four-wave 65..128 class on the side streams, the block class and
    // what it may hand items on to on the caller's stream, behind the caller's gate (see gcc_posemb_multi_gated in the header)
    hipStream_t s1 = s, s2 = s;
#ifndef GCC_AMD_HIPEMU
    PosSide *side = nullptr;
    int fork_mode = 0;
    {
        const char *ef = getenv("GCC_POSEMB_FORK");
        fork_mode = g_posemb_fork >= 0 ? g_posemb_fork : (ef ? atoi(ef) : 0);
        if (fork_mode) side = posemb_side_streams(s);
    }
    if (side) {
        s1 = side->side[0]; s2 = fork_mode == 2 ? s1 : side->side[1];    // 2: ONE side stream for everything but the block class
        (void)hipEventRecord(side->fork, s);
        (void)hipStreamWaitEvent(s1, side->fork, 0);
        (void)hipStreamWaitEvent(s2, side->fork, 0);
    }
#endif
    if (heavy_wait) (void)hipStreamWaitEvent(s, (hipEvent_t)heavy_wait, 0);
    if (hd.use_cheb) {
        ChebArgs ca;
        ca.m = m;
        ca.hd = hd;
        char *after_kry = (char *)(ka.vws + (int64_t)gs.kry * 2 * (kM + 1) * ka.ldv);
        after_kry = (char *)(((uintptr_t)after_kry + 255) & ~(uintptr_t)255);
        ca.xws = (float *)after_kry;
        ca.tabs = (uint32_t *)(ca.xws + (int64_t)gs.cheb * 3 * kNodeMax * kChP);
#ifndef GCC_AMD_HIPEMU
        static bool cheb_attr = false;
        if (!cheb_attr) {
            (void)hipFuncSetAttribute((const void *)posemb_cheb_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, cheb_lds_bytes());
            cheb_attr = true;
        }
#endif
        hipLaunchKernelGGL(posemb_cheb_kernel, dim3(g.cheb), dim3(kChThreads), cheb_lds_bytes(), s, ca);
    }
    hipLaunchKernelGGL((posemb_direct_kernel<kClsSmall, 0, kJSmall, kSmallT, false>), dim3(g.small), dim3(kSmallT), lds_small, s1, m, hd);
    if (hd.use_wave) {
        hipLaunchKernelGGL((posemb_wave_kernel<kClsW64, 64>), dim3(g.w64), dim3(kWaveTeams * 64), kWaveTeams * wave_team_bytes<64>(), s1, m, hd);
        hipLaunchKernelGGL((posemb_wave_kernel<kClsW48, 48>), dim3(g.w48), dim3(kWaveTeams * 64), kWaveTeams * wave_team_bytes<48>(), s1, m, hd);
    }
    if (hd.use_pair)
        hipLaunchKernelGGL((posemb_direct_kernel<kClsMid, kJSmall + 1, kJMax, kPairT, true, true>), dim3(g.pair), dim3(kPairT), kPairLds, s2, m, hd);
    else
        hipLaunchKernelGGL((posemb_direct_kernel<kClsMid, kJSmall + 1, kJMax, kMidT, false>), dim3(g.mid), dim3(kMidT), lds_big, s2, m, hd);
    hipLaunchKernelGGL(posemb_krylov_kernel, dim3(g.kry), dim3(kKThreads), lds_kry, s, ka);
    hipLaunchKernelGGL((posemb_direct_kernel<kClsBig, kGMax + 1, kBMax, 1024, true>), dim3(g.big), dim3(1024), kBLds, s, m, hd);
    hipLaunchKernelGGL((posemb_direct_kernel<kClsSlot, kJMax + 1, kGMax, 1024, true>), dim3(g.slot), dim3(1024), kGLds, s, m, hd);
#ifndef GCC_AMD_HIPEMU
    if (side) {
        (void)hipEventRecord(side->join[0], s1);
        (void)hipEventRecord(side->join[1], s2);
        (void)hipStreamWaitEvent(s, side->join[0], 0);
        (void)hipStreamWaitEvent(s, side->join[1], 0);
    }
#endif
    if (heavy_record) (void)hipEventRecord((hipEvent_t)heavy_record, s);
    prof_mark(prof, 1, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_err, kErrLen, "gcc_posemb: %s", hipGetErrorString(e)); return -10; }
    return 0;
}

void gcc_posemb_set_fork(int32_t mode)   /* 0: one in-order stream (default), 1: three-way fork, 2: the block class beside the rest, -1: environment */
{
    g_posemb_fork = mode;
}

void gcc_posemb_debug_ticks(long long *device_ticks64)   /* device int64[GCC_POSEMB_TICK_CLASSES][16] or NULL; diagnostics only */
{
    g_posemb_ticks = device_ticks64;
}

int32_t gcc_posemb(const gcc_batch_out *g, int32_t batch_size, int32_t hidden, float *pos, float *evals,
                   float *raw, uint64_t seed, void *workspace, int64_t workspace_bytes, int32_t *status, gcc_prof *prof,
                   void *stream)
{
    if (!g) { snprintf(g_err, kErrLen, "gcc_posemb: bad argument"); return -1; }
    gcc_posemb_view v = {g, pos, evals, raw};
    return gcc_posemb_multi(&v, 1, batch_size, g->node_cap, hidden, seed, workspace, workspace_bytes, status, prof, stream);
}

}  // extern "C"
