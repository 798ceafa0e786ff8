/*
 * oracle/sampler_oracle.c -- CPU restatement of the GCC pre-training sampler.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gcc_amd/ may call into this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY STATUS: "parity unpinned" at the DGL boundary.  The arithmetic this
 * file restates lives in an un-vendored dependency of the reference:
 *   dgl (PyPI), constraint 0.5 > dgl >= 0.4.3  (/root/reference/README.md:45)
 *   - dgl.contrib.sampling.random_walk_with_restart
 *       call site /root/reference/gcc/datasets/graph_dataset.py:125-130
 *   - DGLGraph.subgraph        call site gcc/datasets/data_util.py:230
 *   - dgl.batch                call site gcc/datasets/data_util.py:26-32
 * The reference holds no golden vector for any of them, and
 * DGL's own std::mt19937 stream is not reproduced.  What IS pinned: the Philox
 * generator against the Random123 known-answer vectors, and this file against
 * an independent pure-Python restatement plus hand-computed tiny graphs
 * (tests/test_oracle_sampler.py).
 *
 * Published DGL 0.4.x algorithm being restated (GenericRandomWalkWithRestart):
 *   for each seed: repeat { cur = seed; for (t = 0;; ++t) {
 *       if (t > 0 && uniform() < restart_prob) break;          // restart
 *       cur = uniform random successor of cur; append cur; ++total;
 *       if (total == max_nodes_per_seed) stop everything; } }
 * i.e. the seed is never appended, the first step of every walk is
 * unconditional, and exactly L = max_nodes_per_seed entries are produced.
 *
 * Our RNG spec (all-integer, so CPU and GPU agree bit for bit):
 *   Philox4x32-10.  Walk stream: key = (run_seed_lo, run_seed_hi),
 *   counter = (walk_id, block, g_lo, g_hi) with g = sample_id * 2 + view.
 *   The walk's word stream is x[4*block + i].  For step t of a walk:
 *     t >= 1: restart  <=>  x[2t-1] < restart_u32  (restart_u32 = floor(p * 2^32))
 *     next = col_idx[row_ptr[cur] + ((uint64)x[2t] * deg(cur) >> 32)]
 *   Seed stream: key = (run_seed_lo ^ 0x5EED5EED, run_seed_hi ^ 0x00A11CE5),
 *   counter = (sample_id_lo, sample_id_hi, 0, 0); u = ((x0 << 21) | (x1 >> 11))
 *   * 2^-53; seed = first index with cdf[index] > u  (numpy's
 *   Generator-independent legacy choice(): cdf.searchsorted(u, side="right"),
 *   graph_dataset.py:85-92).
 *
 * Node set (data_util.py:221-226): [seed] + sorted(unique(trace) \ {seed}).
 * Induced subgraph (DGL VertexSubgraph): row i = node i of that list, entries
 * in parent-row order, relabelled to positions in the list.
 * Batch (dgl.batch): block-diagonal union, ids offset by the node prefix sum.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* graph_dataset.py:85-92 -- seeds drawn with probability ~ in_degree^0.75 */
void oracle_draw_seeds_sharded(const double *cdf, int64_t num_nodes, const int64_t *shard_off, int32_t num_shards,
                               int32_t batch_size, uint64_t run_seed, int64_t first_sample_id, int32_t count,
                               int32_t *seeds);

void oracle_draw_seeds(const double *cdf, int64_t num_nodes, uint64_t run_seed,
                       int64_t first_sample_id, int32_t count, int32_t *seeds)
{
    oracle_draw_seeds_sharded(cdf, num_nodes, 0, 0, count > 0 ? count : 1, run_seed, first_sample_id, count, seeds);
}

/* ... out of the worker shard its DataLoader batch belongs to (graph_dataset.py:23-30: worker w holds the graphs
 * jobs[w]; :63-76: jobs repeat with period num_shards; an IterableDataset worker yields whole batches, batch i comes
 * from worker i % num_workers).  cdf = every shard's own cdf over its node range [shard_off[s], shard_off[s+1]). */
void oracle_draw_seeds_sharded(const double *cdf, int64_t num_nodes, const int64_t *shard_off, int32_t num_shards,
                               int32_t batch_size, uint64_t run_seed, int64_t first_sample_id, int32_t count,
                               int32_t *seeds)
{
    uint32_t key[2] = { (uint32_t)run_seed ^ 0x5EED5EEDu, (uint32_t)(run_seed >> 32) ^ 0x00A11CE5u };
    for (int32_t b = 0; b < count; ++b) {
        uint64_t sid = (uint64_t)(first_sample_id + b);
        uint32_t ctr[4] = { (uint32_t)sid, (uint32_t)(sid >> 32), 0u, 0u }, x[4];
        oracle_philox4x32_10(ctr, key, x);
        uint64_t u53 = ((uint64_t)x[0] << 21) | (x[1] >> 11);
        double u = (double)u53 * (1.0 / 9007199254740992.0);
        int64_t lo = 0, hi = num_nodes;            /* first index with cdf[i] > u */
        if (num_shards > 1) {
            int64_t sh = (int64_t)((sid / (uint64_t)batch_size) % (uint64_t)num_shards);
            lo = shard_off[sh];
            hi = shard_off[sh + 1];
        }
        int64_t last = hi - 1;
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if (cdf[mid] > u) hi = mid; else lo = mid + 1;
        }
        seeds[b] = (int32_t)(lo <= last ? lo : last);
    }
}

static int cmp_i32(const void *a, const void *b)
{
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

/* graph_dataset.py:125-130 (DGL random_walk_with_restart), one seed, one view.
 * trace must hold L entries.  Returns the number of walks started. */
int32_t oracle_rwr_trace(const int32_t *row_ptr, const int32_t *col_idx, int32_t seed, int32_t L,
                         uint64_t run_seed, uint64_t g, uint32_t restart_u32, int32_t *trace)
{
    uint32_t key[2] = { (uint32_t)run_seed, (uint32_t)(run_seed >> 32) };
    int32_t total = 0, walk = 0;
    while (total < L) {
        int32_t cur = seed;
        uint32_t x[4];
        uint32_t blk = 0xFFFFFFFFu;
        for (int32_t t = 0;; ++t) {
            /* word index 2t-1 = restart test, 2t = neighbour draw */
            if (t > 0) {
                uint32_t wi = (uint32_t)(2 * t - 1);
                if ((wi >> 2) != blk) {
                    blk = wi >> 2;
                    uint32_t ctr[4] = { (uint32_t)walk, blk, (uint32_t)g, (uint32_t)(g >> 32) };
                    oracle_philox4x32_10(ctr, key, x);
                }
                if (x[wi & 3] < restart_u32) break;
            }
            uint32_t wi = (uint32_t)(2 * t);
            if ((wi >> 2) != blk) {
                blk = wi >> 2;
                uint32_t ctr[4] = { (uint32_t)walk, blk, (uint32_t)g, (uint32_t)(g >> 32) };
                oracle_philox4x32_10(ctr, key, x);
            }
            int32_t beg = row_ptr[cur];
            uint32_t deg = (uint32_t)(row_ptr[cur + 1] - beg);
            cur = col_idx[beg + (int32_t)(((uint64_t)x[wi & 3] * deg) >> 32)];
            trace[total++] = cur;
            if (total == L) break;
        }
        ++walk;
    }
    return walk;
}

/* data_util.py:221-226 -- [seed] + sorted(unique(trace) \ {seed}).
 * nodes must hold L + 1 entries; returns n. */
int32_t oracle_node_set(const int32_t *trace, int32_t L, int32_t seed, int32_t *nodes)
{
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(L > 0 ? L : 1));
    memcpy(tmp, trace, sizeof(int32_t) * (size_t)L);
    qsort(tmp, (size_t)L, sizeof(int32_t), cmp_i32);
    int32_t n = 0;
    nodes[n++] = seed;
    for (int32_t i = 0; i < L; ++i) {
        if (tmp[i] == seed) continue;
        if (i > 0 && tmp[i] == tmp[i - 1]) continue;
        nodes[n++] = tmp[i];
    }
    free(tmp);
    return n;
}

/* DGL VertexSubgraph (data_util.py:230).  map is a caller-provided int32[V]
 * scratch filled with -1 on entry and restored to -1 on exit.  sub_row_ptr holds
 * n + 1 entries; sub_col holds up to col_cap entries.  Returns nnz, or
 * -(needed) when col_cap is too small. */
int64_t oracle_induce(const int32_t *row_ptr, const int32_t *col_idx, const int32_t *nodes, int32_t n,
                      int32_t *map, int32_t *sub_row_ptr, int32_t *sub_col, int64_t col_cap)
{
    for (int32_t i = 0; i < n; ++i) map[nodes[i]] = i;
    int64_t nnz = 0;
    for (int32_t i = 0; i < n; ++i) {
        sub_row_ptr[i] = (int32_t)nnz;
        for (int32_t e = row_ptr[nodes[i]]; e < row_ptr[nodes[i] + 1]; ++e) {
            int32_t l = map[col_idx[e]];
            if (l >= 0) {
                if (nnz < col_cap) sub_col[nnz] = l;
                ++nnz;
            }
        }
    }
    sub_row_ptr[n] = (int32_t)nnz;
    for (int32_t i = 0; i < n; ++i) map[nodes[i]] = -1;
    return nnz <= col_cap ? nnz : -nnz;
}

/*
 * One batch, one view: graph_dataset.py:94-179 + data_util.py:26-32, minus the
 * positional embedding (oracle/posemb_oracle.py).
 *   seeds[B], L[B]               per-sample seed and max_nodes_per_seed
 *   node_off[B+1], edge_off[B+1] prefix sums (out)
 *   parent_nid[node_cap]         parent id per batched node (out)
 *   out_row_ptr[node_cap+1], out_col[edge_cap]  batched CSR with GLOBAL ids (out)
 *   stats[2]: [0] += walk steps taken, [1] += parent edges scanned by induction
 * clear_visit_counts != 0 additionally performs DGL 0.4.x's O(|V|) per-seed
 * std::fill of visit_counts (recalled behaviour, reported separately).
 * threads <= 1: serial.  Returns 0, or -1 if a capacity was too small.
 */
int32_t oracle_sample_batch(const int32_t *row_ptr, const int32_t *col_idx, int64_t num_nodes,
                            const int32_t *seeds, const int32_t *L, int32_t B, int32_t view,
                            uint64_t run_seed, int64_t first_sample_id, uint32_t restart_u32,
                            int32_t clear_visit_counts, int32_t threads,
                            int64_t node_cap, int64_t edge_cap,
                            int32_t *node_off, int64_t *edge_off, int32_t *parent_nid,
                            int32_t *out_row_ptr, int32_t *out_col, int64_t *stats)
{
    int32_t **s_nodes = (int32_t **)calloc((size_t)B, sizeof(int32_t *));
    int32_t **s_rp = (int32_t **)calloc((size_t)B, sizeof(int32_t *));
    int32_t **s_col = (int32_t **)calloc((size_t)B, sizeof(int32_t *));
    int32_t *s_n = (int32_t *)calloc((size_t)B, sizeof(int32_t));
    int64_t *s_nnz = (int64_t *)calloc((size_t)B, sizeof(int64_t));
    int64_t scanned = 0, steps = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads) reduction(+ : scanned, steps)
    {
        int32_t *map = (int32_t *)malloc(sizeof(int32_t) * (size_t)num_nodes);
        int32_t *visit = clear_visit_counts ? (int32_t *)malloc(sizeof(int32_t) * (size_t)num_nodes) : NULL;
        memset(map, 0xFF, sizeof(int32_t) * (size_t)num_nodes);
#pragma omp for schedule(dynamic, 1)
        for (int32_t b = 0; b < B; ++b) {
            int32_t l = L[b];
            if (visit) memset(visit, 0, sizeof(int32_t) * (size_t)num_nodes);
            int32_t *trace = (int32_t *)malloc(sizeof(int32_t) * (size_t)l);
            uint64_t g = (uint64_t)(first_sample_id + b) * 2u + (uint64_t)view;
            oracle_rwr_trace(row_ptr, col_idx, seeds[b], l, run_seed, g, restart_u32, trace);
            if (visit) for (int32_t i = 0; i < l; ++i) visit[trace[i]]++;
            steps += l;
            s_nodes[b] = (int32_t *)malloc(sizeof(int32_t) * (size_t)(l + 1));
            int32_t n = oracle_node_set(trace, l, seeds[b], s_nodes[b]);
            free(trace);
            s_n[b] = n;
            s_rp[b] = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
            int64_t cap = 4096;
            for (;;) {
                s_col[b] = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
                int64_t r = oracle_induce(row_ptr, col_idx, s_nodes[b], n, map, s_rp[b], s_col[b], cap);
                if (r >= 0) { s_nnz[b] = r; break; }
                free(s_col[b]);
                cap = -r;
            }
            for (int32_t i = 0; i < n; ++i)
                scanned += row_ptr[s_nodes[b][i] + 1] - row_ptr[s_nodes[b][i]];
        }
        free(map);
        free(visit);
    }
    stats[0] += steps;
    stats[1] += scanned;
    /* dgl.batch: data_util.py:26-32 */
    int32_t rc = 0;
    node_off[0] = 0; edge_off[0] = 0;
    for (int32_t b = 0; b < B; ++b) {
        node_off[b + 1] = node_off[b] + s_n[b];
        edge_off[b + 1] = edge_off[b] + s_nnz[b];
    }
    if (node_off[B] > node_cap || edge_off[B] > edge_cap) rc = -1;
    if (rc == 0) {
        for (int32_t b = 0; b < B; ++b) {
            int32_t no = node_off[b];
            int64_t eo = edge_off[b];
            for (int32_t i = 0; i < s_n[b]; ++i) {
                parent_nid[no + i] = s_nodes[b][i];
                out_row_ptr[no + i] = (int32_t)(eo + s_rp[b][i]);
            }
            for (int64_t e = 0; e < s_nnz[b]; ++e) out_col[eo + e] = no + s_col[b][e];
        }
        out_row_ptr[node_off[B]] = (int32_t)edge_off[B];
    }
    for (int32_t b = 0; b < B; ++b) { free(s_nodes[b]); free(s_rp[b]); free(s_col[b]); }
    free(s_nodes); free(s_rp); free(s_col); free(s_n); free(s_nnz);
    return rc;
}

int32_t oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
