/*
 * include/gcc_amd.h -- C ABI of libgcc_amd.so, the MI355X-native GCC
 * pre-training hot path (RWR ego-net sampler -> batcher -> GIN encoder ->
 * MoCo/InfoNCE head).
 *
 * The reference (THUDM/GCC) is pure Python and has no FFI of its own
 * (SURVEY.md §8b), so there is no existing binding to match; each entry point
 * below names the reference call site whose native work it replaces.  The
 * reference-side glue is the ctypes stub in gcc_amd/_cabi.py (shown in
 * INTEGRATION.md).
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer marked "device" is HBM memory
 *     owned by the caller (the Python host allocates it with torch and passes
 *     data_ptr()); the library allocates nothing and keeps no state.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*;
 *     NULL = the null stream) and performs no host synchronisation.
 *   - return value 0 = launched; <0 = argument error (gcc_last_error()).
 *     Capacity overflows detected on the device are reported through the
 *     caller's `status` words (GCC_STATUS_*), never by writing out of bounds.
 */
#ifndef GCC_AMD_H
#define GCC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCC_AMD_ABI_VERSION 1

/* bits of the device status word */
#define GCC_STATUS_SCRATCH_OVERFLOW 1  /* induction scratch too small            */
#define GCC_STATUS_NODE_OVERFLOW    2  /* a view needs more than node_cap nodes  */
#define GCC_STATUS_EDGE_OVERFLOW    4  /* a view needs more than edge_cap edges  */

int32_t gcc_abi_version(void);
const char *gcc_last_error(void);

/* -------------------------------------------------------------- profiling ---
 * A ring of hipEvents the kernels' launch functions record between kernels
 * when a call's `prof` argument is non-NULL (bench.py's live per-kernel
 * durations; no counterpart in the reference, whose only timers are the
 * BT/DT wall-clock meters of train.py:367-368). */
typedef struct gcc_prof gcc_prof;
gcc_prof *gcc_prof_create(int32_t num_marks);
void gcc_prof_destroy(gcc_prof *p);
/* milliseconds between two recorded marks; synchronises on `to_mark`. */
int32_t gcc_prof_elapsed_ms(gcc_prof *p, int32_t from_mark, int32_t to_mark, float *ms);

/* ---------------------------------------------------------------- graph ---
 * The parent graph in the layout x2dgl.py:39-62 guarantees (symmetric, no self
 * loops, no duplicates, no zero-degree nodes, rows sorted), resident in HBM.
 * Replaces the per-worker dgl.data.utils.load_graphs copy of
 * gcc/datasets/graph_dataset.py:23-30. */
typedef struct gcc_graph {
    const int32_t *row_ptr;   /* device [num_nodes + 1]                               */
    const int32_t *col_idx;   /* device [num_edges]                                   */
    const double  *seed_cdf;  /* device [num_nodes]: cumsum(deg^0.75)/sum, float64    */
                              /*   (graph_dataset.py:86-90)                           */
    const int32_t *ltab;      /* device [ltab_len]: max_nodes_per_seed by in-degree,  */
                              /*   index clamped to ltab_len-1 (graph_dataset.py:113-124) */
    int64_t num_nodes;
    int64_t num_edges;
    int32_t ltab_len;
    int32_t lmax;             /* max(ltab): sizes LDS and per-subgraph capacities     */
} gcc_graph;

/* --------------------------------------------------------------- sampler ---
 * One call = one DataLoader batch of LoadBalanceGraphDataset
 * (graph_dataset.py:85-179 + data_util.py:26-32,218-239, minus the positional
 * embedding which is gcc_posemb_*): draw batch_size seeds, run two independent
 * random walks with restart per seed (views q and k), build both induced
 * subgraphs and emit two batched CSRs.  Bit-exact against
 * oracle/sampler_oracle.c for the RNG spec written there. */
typedef struct gcc_sample_params {
    uint64_t run_seed;         /* Philox key                                          */
    int64_t  first_sample_id;  /* global index of sample 0 of this batch              */
    int32_t  batch_size;       /* B samples -> 2B subgraphs                           */
    uint32_t restart_u32;      /* floor(restart_prob * 2^32)                          */
    const int32_t *seeds;      /* device [B] or NULL; non-NULL overrides the seed draw */
    gcc_prof *prof;            /* NULL, or marks 0..3 recorded around walk/induce/pack */
} gcc_sample_params;

/* One view's batched graph = dgl.batch(list of subgraphs), data_util.py:26-32.
 * Node i of subgraph b is global node node_off[b] + i; node_off[b] itself is
 * the seed (data_util.py:226,238). */
typedef struct gcc_batch_out {
    int32_t *node_off;    /* device [B + 1]                                           */
    int32_t *edge_off;    /* device [B + 1]                                           */
    int32_t *parent_nid;  /* device [node_cap]: id in the parent graph                */
    int32_t *graph_id;    /* device [node_cap]: subgraph index of each node           */
    int32_t *row_ptr;     /* device [node_cap + 1]                                    */
    int32_t *col_idx;     /* device [edge_cap]: global (batched) node ids             */
    int64_t  node_cap;
    int64_t  edge_cap;
} gcc_batch_out;

/* bytes of caller-provided workspace needed for batch_size samples with
 * scratch_entries int32 slots of induction scratch (>= sum over subgraphs of
 * sum_i min(deg_i, n); 64 * batch_size * (lmax+1) is generous). */
int64_t gcc_sampler_workspace_bytes(const gcc_graph *g, int32_t batch_size, int64_t scratch_entries);

/* status: device int32[1], OR-ed with GCC_STATUS_* bits (caller zeroes it). */
int32_t gcc_sample_batch(const gcc_graph *g, const gcc_sample_params *p,
                         const gcc_batch_out *out_q, const gcc_batch_out *out_k,
                         void *workspace, int64_t workspace_bytes, int64_t scratch_entries,
                         int32_t *status, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GCC_AMD_H */
