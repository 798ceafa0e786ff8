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

/* 3: gcc_gin_pass.{node_cap,rows_hint}.
 * 2: gcc_sample_params.{hub_degree,max_hubs}, gcc_gin_weights.hidden, gcc_gin_pass.scalars, gcc_ginw_args.{scratch,
 * scratch_bytes,num_nodes}, gcc_graph.{flags,hub_index,hub_adj,num_hubs,hub_words,hub_table_degree}.  A caller built against another version must not pass its structs:
 * compare gcc_abi_version() with the header's constant after loading (gcc_amd/_cabi.py does). */
#define GCC_AMD_ABI_VERSION 3

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

/* -------------------------------------------------------------- streams ---
 * A HIP stream whose kernels may only run on the compute units set in
 * `cu_mask` (bit i of word i / 32; `words` 32-bit words, 256 CUs = 8 words).
 * The host layer gives the (few) data-pipeline streams a mask that leaves some
 * compute units to the training step alone (the reference gets this isolation
 * for free: its pipeline runs on CPU workers, train.py:577-586).
 * Returns 0 and the hipStream_t in *stream. */
int32_t gcc_stream_create_cu_mask(const uint32_t *cu_mask, int32_t words, void **stream);
/* diagnostics (tools/load_probe.py): a synthetic co-tenant -- `workgroups` workgroups of `threads` threads and `lds_bytes` of LDS that
 * spend `ticks` of the 100 MHz clock (at most max_iters rounds) on kind 0: barriers + LDS, 1: float4 reads streamed over buf, 2: FMA chains */
int32_t gcc_debug_load(int32_t kind, int32_t workgroups, int32_t threads, int32_t lds_bytes, int64_t ticks, int32_t max_iters,
                       const float *buf, int64_t buf_floats, float *sink, void *stream);
int32_t gcc_stream_destroy(void *stream);

/* ---------------------------------------------------------------- graph ---
 * The parent graph in the layout x2dgl.py:39-62 guarantees (symmetric, no self
 * loops, no duplicates, no zero-degree nodes, rows sorted), resident in HBM.
 * Replaces the per-worker dgl.data.utils.load_graphs copy of
 * gcc/datasets/graph_dataset.py:23-30. */
typedef struct gcc_graph {
    const int32_t *row_ptr;   /* device [num_nodes + 1]                               */
    const int32_t *col_idx;   /* device [num_edges]                                   */
    const double  *seed_cdf;  /* device [num_nodes]: cumsum(deg^0.75)/sum, float64    */
                              /*   (graph_dataset.py:86-90); with worker shards: each */
                              /*   shard's own cdf over its node range (ends at 1.0)  */
    const int32_t *ltab;      /* device [ltab_len]: max_nodes_per_seed by in-degree,  */
                              /*   index clamped to ltab_len-1 (graph_dataset.py:113-124) */
    int64_t num_nodes;
    int64_t num_edges;
    int32_t ltab_len;
    int32_t lmax;             /* max(ltab): sizes LDS and per-subgraph capacities     */
    /* Worker shards of LoadBalanceGraphDataset (graph_dataset.py:23-30,63-76): the corpus' graphs are laid out shard
     * by shard, shard s = nodes [shard_off[s], shard_off[s+1]); DataLoader batch i (= sample ids [i*bsz, (i+1)*bsz))
     * draws its seeds from shard i % num_shards only.  num_shards <= 1 (shard_off may be NULL): one shard = all. */
    const int64_t *shard_off; /* device [num_shards + 1] or NULL                      */
    int32_t num_shards;
    int32_t flags;            /* GCC_GRAPH_* bits                                     */
    /* Optional (NULL / 0: off): adjacency among the parent's high-degree rows, built once at upload.  The induction does not
     * scan a subgraph's hub rows; the edges BETWEEN two of them used to cost one search per pair in the shorter row (8-14
     * dependent loads; 55 of the walk launch's 120 us on the 1M-node graph) and are one bit probe with this table.  Used when
     * the call's effective hub_degree is >= hub_table_degree (every hub is in the table then). */
    const int32_t  *hub_index;        /* device [num_nodes]: index among the rows of degree >= hub_table_degree, -1 otherwise */
    const uint32_t *hub_adj;          /* device [num_hubs][hub_words]: bit (c & 31) of word c >> 5 of row a: hubs a, c adjacent */
    int32_t num_hubs, hub_words;      /* hub_words = (num_hubs + 31) / 32                                                      */
    int32_t hub_table_degree, reserved_;
} gcc_graph;
/* The caller has verified the contract above (symmetric, rows sorted ascending, no self loops, no duplicates).  Only
 * then may the induction skip hub rows (gcc_sample_params.hub_degree >= 0): their induced rows are rebuilt as mirror
 * images of the other rows' hits, which is the DGL-consistent subgraph on such a graph only.  Without the bit every
 * member row is scanned whatever hub_degree says (the result is then the induced subgraph of ANY sorted-row CSR). */
#define GCC_GRAPH_CONTRACT_CHECKED 1

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
    int32_t hub_degree;        /* member rows of at least this parent degree are NOT scanned by the induction: the graph is
                                * symmetric, so their induced rows are the mirror images of the other rows' hits + one search
                                * per pair of hubs -- same result bit for bit, fewer bytes scanned on power-law graphs.
                                * 0 = default (512), < 0 = scan every row */
    int32_t max_hubs;          /* most such rows per subgraph (1..32; 0 = default, 32): with more rows over hub_degree the
                                * subgraph's own threshold rises to the power of two that leaves at most this many */
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
 * scratch_entries int32 slots of induction scratch (one 1024-entry slot per unit of 256 aligned quads of the members'
 * parent rows: about the sum of the members' parent degrees over all subgraphs of the call). */
int64_t gcc_sampler_workspace_bytes(const gcc_graph *g, int32_t batch_size, int64_t scratch_entries);
/* ... for calls that cover num_steps consecutive DataLoader batches (gcc_sample_multi) */
#define GCC_SAMPLE_MAX_STEPS 16
int64_t gcc_sampler_workspace_bytes_multi(const gcc_graph *g, int32_t batch_size, int32_t num_steps, int64_t scratch_entries);

/* diagnostics: subsequent gcc_sample_batch calls add wall-clock ticks (100 MHz) of induce_kernel's phases into device
 * int64[16] ([0] prefix sums over the subgraphs, [1] hash map + row prefix sums, [2] segment scans, [15] workgroups). */
void gcc_sampler_debug_ticks(long long *device_ticks64);
/* tests only: cap the static grids of the two induce classes and of the big walk class (0 = the defaults), so that small
 * test batches exercise workgroups that walk through several virtual workgroups, subgraphs and list entries */
void gcc_sampler_debug_grids(int32_t small_grid, int32_t big_grid, int32_t walk_big_grid);

/* status: device int32[1], OR-ed with GCC_STATUS_* bits (caller zeroes it). */
int32_t gcc_sample_batch(const gcc_graph *g, const gcc_sample_params *p,
                         const gcc_batch_out *out_q, const gcc_batch_out *out_k,
                         void *workspace, int64_t workspace_bytes, int64_t scratch_entries,
                         int32_t *status, void *stream);

/* The batches of num_steps consecutive DataLoader steps in ONE launch set: step t samples the ids
 * p->first_sample_id + t * sample_id_stride + [0, batch_size) (stride = world * batch_size for a rank of a data-parallel
 * job) into outs[2 t] (view q) and outs[2 t + 1] (view k); p->seeds, if given, holds num_steps * batch_size seeds.
 * Every subgraph is the one gcc_sample_batch would produce for the same sample id (bit for bit); what changes is the
 * cost: the five kernels of a call are latency chains that a single step's 2 * batch_size subgraphs cannot fill the
 * GPU with (the reference's DataLoader prefetches whole batches ahead the same way, train.py:577-586).
 * num_steps <= GCC_SAMPLE_MAX_STEPS and 2 * batch_size * num_steps <= 16383; scratch_entries covers the whole call. */
int32_t gcc_sample_multi(const gcc_graph *g, const gcc_sample_params *p, int32_t num_steps, int64_t sample_id_stride,
                         const gcc_batch_out *outs, void *workspace, int64_t workspace_bytes, int64_t scratch_entries,
                         int32_t *status, void *stream);

/* --------------------------------------------------- positional embedding ---
 * _add_undirected_graph_positional_embedding + eigen_decomposision of
 * gcc/datasets/data_util.py:242-281 for every subgraph of a batched graph:
 * the k = min(n - 2, hidden) largest-algebraic eigenpairs of D^-1/2 A D^-1/2
 * (D = in-degree clipped at 1), eigenvalues ascending like scipy eigsh(which="LA"),
 * rows L2-normalised, zero-padded to `hidden` columns; k <= 0 gives zeros.
 * Twin leaves are deflated exactly first (their contrasts are null vectors).
 * Deflated size n' <= GCC_POSEMB_DIRECT_MAX: direct symmetric eigensolver
 * (tridiagonalisation, bisection, inverse iteration: exact multiplicities), the
 * matrix in LDS up to GCC_POSEMB_LDS_MAX and in a workspace slot above (a second
 * workspace class reaches GCC_POSEMB_BIG_MAX);
 * larger ones a thick-restart Krylov-Schur iteration
 * (single start vector, like ARPACK).  Eigenvectors are defined up to sign /
 * rotation inside degenerate eigenspaces; the reference's own output depends on
 * np.random.rand (data_util.py:248).  hidden <= 32. */
#define GCC_POSEMB_LDS_MAX 128
#define GCC_POSEMB_DIRECT_MAX 384
#define GCC_POSEMB_BIG_MAX 704      /* direct solver with the whole LDS of a CU up to this deflated size; Krylov-Schur above */
#define GCC_STATUS_POSEMB_NOT_CONVERGED 8
#define GCC_STATUS_POSEMB_TOO_LARGE 16   /* a subgraph with deflated size > GCC_POSEMB_DIRECT_MAX has more than
                                          * node_cap / batch_size nodes: zeros written (size node_cap accordingly) */
int64_t gcc_posemb_workspace_bytes(int32_t batch_size, int64_t node_cap, int32_t hidden);
/* pos: device [node_cap, hidden] out (rows >= node_off[B] untouched);
 * evals: device [B, hidden] out or NULL (eigenvalues, ascending, zero-padded);
 * raw:   device [node_cap, hidden] out or NULL (the eigenvectors before row normalisation);
 * seed: start vectors of the Krylov path are Philox(seed, subgraph) uniforms.
 * status: device int32[16]: [0] |= GCC_STATUS_*; diagnostics: [1] = most Krylov restart cycles / block filter rounds of
 *         an item, [2] += Arnoldi steps, [3] += items that left their first-choice solver (Krylov restart cap, block
 *         class handing on to the dense classes), [4] += items that set GCC_STATUS_POSEMB_NOT_CONVERGED, [5..9] = the
 *         first of them: item id (view * batch_size + subgraph), solver class, deflated size, reason, nodes. */
int32_t gcc_posemb(const gcc_batch_out *g, int32_t batch_size, int32_t hidden, float *pos, float *evals,
                   float *raw, uint64_t seed, void *workspace, int64_t workspace_bytes, int32_t *status, gcc_prof *prof,
                   void *stream);
/* The same for several batched graphs (views of several future steps) in ONE set of kernel launches: the
 * eigensolver kernels pull (view, subgraph) items from per-size-class work lists, so one call keeps the whole
 * GPU busy and its latency (bound by the slowest subgraph) is paid once.  All views share batch_size and node_cap. */
#define GCC_POSEMB_MAX_VIEWS 32
typedef struct gcc_posemb_view {
    const gcc_batch_out *g;
    float *pos, *evals, *raw;      /* as for gcc_posemb; evals/raw may be NULL */
} gcc_posemb_view;
int64_t gcc_posemb_multi_workspace_bytes(int32_t num_views, int32_t batch_size, int64_t node_cap, int32_t hidden);
int32_t gcc_posemb_multi(const gcc_posemb_view *views, int32_t num_views, int32_t batch_size, int64_t node_cap,
                         int32_t hidden, uint64_t seed, void *workspace, int64_t workspace_bytes, int32_t *status,
                         gcc_prof *prof, void *stream);
/* The same call with a GATE around its LDS-heavy launches (sparse block, Krylov and the 1024-thread dense classes:
 * workgroups that own most of a CU's LDS for milliseconds).  Several producer streams run such calls concurrently; with a
 * shared gate their heavy phases take turns, so that at most one call's heavy workgroups hold CUs at any time and the
 * rest of the GPU stays open to the training step's short kernels, while the light launches (classification, one-wave
 * teams) of all calls overlap freely.  heavy_wait: hipEvent_t the stream waits for before the first heavy launch (or
 * NULL); heavy_record: hipEvent_t recorded after the last one (or NULL). */
int32_t gcc_posemb_multi_gated(const gcc_posemb_view *views, int32_t num_views, int32_t batch_size, int64_t node_cap,
                               int32_t hidden, uint64_t seed, void *workspace, int64_t workspace_bytes, int32_t *status,
                               gcc_prof *prof, void *heavy_wait, void *heavy_record, void *stream);

/* How the solver classes of one gcc_posemb* call are issued.  0 (default): one after the other on the caller's stream -- next to a
 * training step that is what keeps the step's short kernels fed.  1: the one-wave teams and the 65..128 class on two side streams of the
 * caller's stream (same priority, created once per caller stream, joined before the call's end mark): half the latency of a call when
 * the pipeline has the GPU to itself (bench.py --mode sample-ready).  2: one side stream for everything but the block class.
 * -1: the environment variable GCC_POSEMB_FORK decides.  Process-wide; results do not depend on it. */
void gcc_posemb_set_fork(int32_t mode);

/* diagnostics: subsequent gcc_posemb* calls add wall-clock ticks (100 MHz) per solver class and phase into
 * device int64[GCC_POSEMB_TICK_CLASSES][16] -- EIGHT classes: small, mid, slot, Krylov, big, sparse block (Chebyshev),
 * one-wave teams n' <= 48, one-wave teams n' <= 64 (their ticks are WAVE time: 4 teams share a workgroup; the 'mid' class
 * runs on four-wave workgroups of 256 threads, three of which share a CU: its ticks are the time of one such workgroup);
 * phases of the dense classes 0..6 = matrix, tridiagonalise, bisect, inverse iteration, Gram-Schmidt, back-transform,
 * expand; [14] = executed f32 FLOPs; [15] = items; NULL switches it off.  A buffer sized for fewer classes is written
 * out of bounds. */
#define GCC_POSEMB_TICK_CLASSES 8
void gcc_posemb_debug_ticks(long long *device_ticks64);
/* the same for gin_in_kernel / gin_mid_kernel: device int64[3][16][2048] -- [kind 0 = gin_in first layer, 1 = gin_in other layers,
 * 2 = gin_mid][phase; 15 = tiles][workgroup = pass * 1024 + blockIdx.x]: every workgroup adds to its own slots (no shared counter) */
void gcc_gin_debug_ticks(long long *device_ticks64);

/* ------------------------------------------------------------ GIN encoder ---
 * GraphEncoder(gnn_model="gin", degree_input=True).forward of
 * gcc/models/graph_encoder.py:132-200 -> UnsupervisedGIN.forward gcc/models/gin.py:213-232
 * (DGL GINConv(sum, eps=0) + ApplyNodeFunc/MLP + BatchNorm1d + SumPooling +
 * linears_prediction + Dropout + F.normalize), and its backward.  The kernels
 * compute 64 channels (train.py:93 default); narrower models (gcc_gin_weights.hidden)
 * run zero-padded, exactly; d_in = pos_dim + deg_emb_dim + 1 <= 64.
 * All arithmetic is fp32 (f32 MFMA), BatchNorm / pooling sums accumulate in fp64. */
#define GCC_GIN_MAX_LAYERS 8     /* GIN message-passing layers = num_layers - 1 (train.py:79 -> 4) */
#define GCC_GIN_HIDDEN 64
#define GCC_GIN_STAT_REPLICAS 16  /* atomically accumulated rows are spread over this many copies (32 until ABI 3) */

typedef struct gcc_bn {          /* torch.nn.BatchNorm1d(64) */
    const float *weight, *bias;  /* device [64]                                           */
    float *running_mean, *running_var;   /* device [64], updated when update_running_stats */
    int64_t *num_batches_tracked;        /* device [1] or NULL                            */
} gcc_bn;

typedef struct gcc_gin_weights { /* state_dict of the reference GraphEncoder */
    int32_t num_gin_layers;      /* len(gnn.ginlayers)                                    */
    int32_t pos_dim, deg_emb_dim, max_degree;
    const float *degree_embedding;               /* [max_degree + 1, deg_emb_dim]          */
    const float *lin0_w[GCC_GIN_MAX_LAYERS];     /* ginlayers.i.apply_func.mlp.linears.0.weight [64, d_in|64] */
    const float *lin0_b[GCC_GIN_MAX_LAYERS];
    const float *lin1_w[GCC_GIN_MAX_LAYERS];     /* ...mlp.linears.1.weight [64, 64]       */
    const float *lin1_b[GCC_GIN_MAX_LAYERS];
    gcc_bn bn_a[GCC_GIN_MAX_LAYERS];             /* ...mlp.batch_norms.0                   */
    gcc_bn bn_b[GCC_GIN_MAX_LAYERS];             /* ginlayers.i.apply_func.bn              */
    gcc_bn bn_c[GCC_GIN_MAX_LAYERS];             /* gnn.batch_norms.i                      */
    const float *pred_w[GCC_GIN_MAX_LAYERS + 1]; /* gnn.linears_prediction.i.weight [64, d_in|64] */
    const float *pred_b[GCC_GIN_MAX_LAYERS + 1];
    float bn_eps, bn_momentum;   /* 1e-5, 0.1 (torch defaults, gin.py:51,104,189)          */
    float dropout_p;             /* 0.5 (graph_encoder.py:99)                              */
    float norm_eps;              /* 1e-5 (graph_encoder.py:196)                            */
    int32_t hidden;              /* --hidden-size (train.py:93) when it is below 64, 0 = 64: the COLUMN count of every weight
                                  * that reads a hidden representation (lin0_w of layers > 0, lin1_w, pred_w[i > 0]); the
                                  * kernels always compute 64 channels, so every weight keeps 64 ROWS and every per-channel
                                  * array 64 entries -- rows / entries >= hidden are zero padding owned by the caller (a zero
                                  * channel stays exactly zero through Linear, BatchNorm, ReLU and their backward, so the
                                  * padded model IS the narrow model).  Gradients come back in the same padded shapes. */
} gcc_gin_weights;

/* DEVICE-resident per-step scalars of a REPLAYED step.  A training step captured in a hipGraph replays the same
 * kernel arguments every time; what changes from step to step besides the data -- the learning rate (train.py:411-416),
 * Adam's bias corrections, the queue's ring pointer (memory_moco.py:55-61), the dropout key -- is read from this struct
 * by the kernels that need it (pass it where the `scalars` arguments below say; NULL = the by-value arguments), and
 * written by gcc_step_scalars_set, a one-thread launch issued in front of the replay. */
typedef struct gcc_step_scalars {
    float lr, bias_corr1, bias_corr2_sqrt;   /* Adam: lr, 1 - beta1^step, sqrt(1 - beta2^step) */
    int32_t enqueue_index;                   /* queue rows [index, index + nkeys) mod K are overwritten */
    uint64_t dropout_seed;                   /* Philox key of the pass's dropout masks (gcc_gin_pass.dropout_seed) */
} gcc_step_scalars;
int32_t gcc_step_scalars_set(gcc_step_scalars *dev, float lr, float beta1, float beta2, int32_t adam_step,
                             int32_t enqueue_index, uint64_t dropout_seed, void *stream);
/* The same without a launch between two replays: the HOST fills entry (n mod ring_len) of a ring in pinned, device-visible
 * host memory (gcc_step_scalars_fill: plain stores, no device work) before it launches the n-th step that uses the ring,
 * and the step's FIRST launch (gcc_step_scalars_fetch: one thread, part of the captured graph) copies entry
 * (*counter mod ring_len) into the device struct and increments the device-resident counter.  Host and device count the
 * same steps, so entry n is read by step n; the host must not run ring_len steps ahead of the device. */
void gcc_step_scalars_fill(gcc_step_scalars *host_entry, float lr, float beta1, float beta2, int32_t adam_step,
                           int32_t enqueue_index, uint64_t dropout_seed);
int32_t gcc_step_scalars_fetch(gcc_step_scalars *dev, const gcc_step_scalars *ring, int32_t ring_len,
                               unsigned long long *counter, void *stream);

typedef struct gcc_gin_pass {    /* one encoder invocation on one batched graph            */
    const int32_t *node_off, *row_ptr, *col_idx, *graph_id;   /* gcc_batch_out of the view */
    const float *pos;            /* device [node_cap, pos_dim]: ndata["pos_undirected"]    */
    int32_t batch_size;
    int32_t training;            /* 1: BatchNorm uses batch statistics (train.py:357-365)  */
    int32_t update_running_stats;/* 1: momentum update of running_mean/var                 */
    int32_t normalize;           /* graph_encoder.py:195 (train.py:83 default True)        */
    const float *dropout_keep;   /* device [num_gin_layers+1, B, 64] 0/1 keep masks, or NULL                    */
    uint64_t dropout_seed;       /* used when dropout_keep == NULL and dropout_philox != 0: keep(i, b, o) =     */
    int32_t dropout_philox;      /*   Philox4x32-10(key = seed, ctr = (b*64+o, i, 0xD50F, 0)).x >> 8 >= p * 2^24 */
    gcc_gin_weights w;
    /* activations, caller-allocated, kept for backward: */
    float *x0;                   /* [node_cap, 64] assembled input features (cols >= d_in are 0) */
    float *agg[GCC_GIN_MAX_LAYERS];   /* [node_cap, 64] h + sum_{u->v} h_u                */
    float *z1[GCC_GIN_MAX_LAYERS];    /* [node_cap, 64] linears.0 output                   */
    float *z2[GCC_GIN_MAX_LAYERS];    /* [node_cap, 64] linears.1 output                   */
    double *stats;               /* [num_gin_layers, 3, GCC_GIN_STAT_REPLICAS, 2, 64] column sum / sum of squares */
    double *pooled;              /* [num_gin_layers+1, B, 64] SumPooling of hidden_rep     */
    float *score;                /* [B, 64] score_over_layer before normalisation          */
    float *feat;                 /* [B, 64] output                                         */
    int32_t edge_multiplicity;   /* every edge of the CSR counts this many times (0 = 1): in-degree feature and
                                  * neighbour sum.  The reference's NodeClassificationDataset builds its DGL graph
                                  * with every undirected edge twice per direction (data_util.py:84-85 +
                                  * graph_dataset.py:301-302); forward only (backward requires 1). */
    double *bn_totals;           /* device [num_gin_layers][3][2][64] doubles, or NULL.  Training passes: gcc_gin_forward's
                                  * last kernel adds the 32 replicas of every BatchNorm's statistics up (in replica order:
                                  * the value each forward consumer computed for itself) and gcc_gin_backward's ~16 kernels
                                  * read these 2 numbers per channel instead of 64 before they can start. */
    const int32_t *seed_local;   /* device [B] or NULL: local index of the seed node of every graph (NULL: node 0, as the
                                  * sampler emits; graph classification marks g.out_degrees().argmax(),
                                  * data_util.py:236-237 with entire_graph=True) */
    const gcc_step_scalars *scalars;   /* device or NULL: with dropout_philox, the key is scalars->dropout_seed + dropout_seed
                                        * (read by the readout kernels of the forward AND the backward pass): dropout_seed is
                                        * then the pass's fixed offset -- 0, or the second pass's of an E2E step */
    int64_t node_cap;            /* (ABI 3) rows every per-node buffer of this pass holds (row_ptr: node_cap + 1 entries; graph_id, pos,
                                  * x0, agg, z1, z2: node_cap rows).  The tile kernels request a workgroup's first tile TOGETHER with the
                                  * live node count node_off[B] (one memory round trip instead of two dependent ones per kernel), clamped
                                  * to this capacity; rows past the live count are read and discarded.  Required (> 0). */
    int64_t rows_hint;           /* (ABI 3) 0, or an upper estimate of the live row count node_off[B] (e.g. 1.1 x the largest batch seen):
                                  * the tile kernels are launched with ceil(min(rows_hint, node_cap) / 64) workgroups per pass instead of
                                  * one per 64 rows of CAPACITY.  Any value is correct (workgroups walk on when there are more tiles);
                                  * workgroups without a tile cost what their speculative requests cost (0.63 vs 0.57 ms per step). */
} gcc_gin_pass;

/* Runs `npass` independent passes (e.g. query with model, key with model_ema)
 * in the same launches.  prof marks: 0 before, 1 after. */
int32_t gcc_gin_forward(const gcc_gin_pass *passes, int32_t npass, gcc_prof *prof, void *stream);

/* The eval-mode forward as ONE call (generate.py:33-53: model.eval(); feat_q = model(graph_q); feat_k = model(graph_k);
 * emb = (feat_q + feat_k) / 2): a workgroup carries a subgraph -- or a run of up to four subgraphs of at most 64 nodes --
 * through feature assembly, every GIN layer, the pooled readout and F.normalize with the hidden representation resident in
 * LDS (up to 320 nodes; larger ego-nets go through a second kernel that keeps up to 256 rows in LDS and gathers from the
 * L2-resident global copy above).  Every pass must have training = 0 (running statistics) and x0, z1[0], z2[0] ([node_cap,
 * 64] scratch: x0 holds the call's work list), score and feat; pooled is written when not NULL.  mean_out: device [B, 64]
 * or NULL -- receives the mean of the passes' feat (npass = 2: generate.py:52).  Same results as gcc_gin_forward in eval
 * mode to ~1e-6 (the neighbour sums run in another order). */
int32_t gcc_gin_eval_fused(const gcc_gin_pass *passes, int32_t npass, float *mean_out, void *stream);
/* diagnostics: device int64[2][16] -- per kernel of the call (subgraphs and runs of up to 320 nodes; the rest) 100 MHz ticks per
 * phase (features, pooling, weights, own rows / one wave's neighbour sums, gather / one wave's products, Linears / write-back,
 * mirror / barrier wait, readout; [8..11] = first start, last start, last end, longest stay of a workgroup; [15] = workgroups),
 * summed over every 4th (8th) workgroup of the following calls; NULL switches it off */
void gcc_gin_eval_debug_ticks(long long *device_ticks64);

typedef struct gcc_gin_grads {   /* same shapes as the weights; written (not accumulated)  */
    float *degree_embedding;
    float *lin0_w[GCC_GIN_MAX_LAYERS], *lin0_b[GCC_GIN_MAX_LAYERS];
    float *lin1_w[GCC_GIN_MAX_LAYERS], *lin1_b[GCC_GIN_MAX_LAYERS];
    float *bn_a_w[GCC_GIN_MAX_LAYERS], *bn_a_b[GCC_GIN_MAX_LAYERS];
    float *bn_b_w[GCC_GIN_MAX_LAYERS], *bn_b_b[GCC_GIN_MAX_LAYERS];
    float *bn_c_w[GCC_GIN_MAX_LAYERS], *bn_c_b[GCC_GIN_MAX_LAYERS];
    float *pred_w[GCC_GIN_MAX_LAYERS + 1], *pred_b[GCC_GIN_MAX_LAYERS + 1];
} gcc_gin_grads;

/* bytes of workspace gcc_gin_backward needs for a pass with this node capacity */
int64_t gcc_gin_backward_workspace_bytes(int64_t node_cap, int32_t batch_size, int32_t num_gin_layers);

/* Backward of one training-mode pass: dfeat [B, 64] -> grads (overwritten).
 * If `accumulate` != 0 the results are added to `grads` instead (E2E mode runs
 * two passes through the same weights, train.py:397-398). */
int32_t gcc_gin_backward(const gcc_gin_pass *pass, const float *dfeat, const gcc_gin_grads *grads,
                         int32_t accumulate, void *workspace, int64_t workspace_bytes, int64_t node_cap,
                         gcc_prof *prof, void *stream);

/* ------------------------------------- GIN encoder at any width (training) ---
 * The same encoder -- GraphEncoder(gnn_model="gin").forward, graph_encoder.py:132-200 -> gin.py:213-232 -- for hidden /
 * output sizes the 64-channel kernels above do not serve (`--hidden-size` above 64, train.py:93), forward in training
 * or eval mode and the full backward, fp32 on the matrix cores.  Not fused: one launch per operator (CSR gather, strided
 * MFMA GEMM, fp64 column statistics, BatchNorm + ReLU passes, per-graph pooling); every activation is kept in the
 * caller's workspace for the backward pass.  Weights / gradients use gcc_gin_weights / gcc_gin_grads with the tensors'
 * own (unpadded) shapes: lin0_w[0] [hidden, d_in], lin0_w[i > 0] and lin1_w [hidden, hidden], pred_w[0] [out_dim, d_in],
 * pred_w[i > 0] [out_dim, hidden], per-channel arrays [hidden]; gcc_gin_weights.hidden is ignored.  Dropout: explicit
 * keep masks only (the host draws them, as torch.nn.Dropout does: gin.py:202,230).  The batched graph must be symmetric
 * (the sampler's output is): the gather is its own transpose in the backward pass. */
typedef struct gcc_ginx_pass {
    const int32_t *node_off, *row_ptr, *col_idx, *graph_id;   /* gcc_batch_out of the view                              */
    const float *pos;            /* device [node_cap, pos_dim]                                                          */
    const int32_t *seed_local;   /* device [B] or NULL (gcc_gin_pass.seed_local)                                        */
    int32_t batch_size;
    int32_t training;            /* 1: batch statistics; 0: running statistics (no backward)                            */
    int32_t update_running_stats;
    int32_t normalize;           /* graph_encoder.py:195                                                                */
    const float *dropout_keep;   /* device [num_gin_layers + 1, B, out_dim] 0 / 1 keep masks, or NULL: no dropout       */
    int32_t hidden, out_dim;     /* node_hidden_dim, output_dim: any positive size                                      */
    int32_t edge_multiplicity;   /* gcc_gin_pass.edge_multiplicity (forward; the backward pass requires 0 / 1)          */
    int32_t reserved_;
    int64_t node_cap;            /* rows the launches are sized for (node_off[B] <= node_cap, read on the device)       */
    gcc_gin_weights w;
    void *workspace;             /* device, gcc_ginx_workspace_bytes(): activations of the forward pass (kept for the   */
    int64_t workspace_bytes;     /*   backward pass of the SAME struct) + backward scratch                              */
    float *feat;                 /* device [B, out_dim] out                                                             */
    float *pooled_out;           /* device [num_gin_layers, B, hidden] out or NULL: SumPooling of hidden_rep[1..]       */
} gcc_ginx_pass;
int64_t gcc_ginx_workspace_bytes(int64_t node_cap, int32_t batch_size, int32_t num_gin_layers, int32_t d_in, int32_t hidden,
                                 int32_t out_dim);
int32_t gcc_ginx_forward(const gcc_ginx_pass *p, void *stream);
/* dfeat: device [B, out_dim]; grads: written (not accumulated).  After gcc_ginx_forward of the same pass (training = 1). */
int32_t gcc_ginx_backward(const gcc_ginx_pass *p, const float *dfeat, const gcc_gin_grads *grads, void *stream);

/* MemoryMoCo.forward + NCESoftmaxLoss (mode 0: memory_moco.py:26-63, criterions.py:5-17) / the in-batch E2E head (mode 1:
 * out = rows mem^T / T with mem = the other view's features, K = B, labels on the diagonal: train.py:400, criterions.py:20-33)
 * at any feature size D, dense: out [B, K + 1] (mode 0: column 0 = <q, k> / T) or [B, B]; dlog: same shape, softmax - onehot;
 * grad_rows [B, D] = d loss / d rows and (mode 1) grad_mem [B, D] = d loss / d mem, both for a unit upstream gradient and
 * taken against the queue as it is NOW -- the caller enqueues afterwards; loss, prob: device scalars (mean CE; mean label
 * logit, train.py:394,401); acc: device double[2 + B * D] scratch (ABI 3: the B x D tail accumulates grad_rows, whose reduction runs
 * over the K queue rows and is split over workgroups). */
int32_t gcc_ncex_forward(const float *q, const float *k, const float *mem, int32_t B, int32_t K, int32_t D, float inv_T, int32_t mode,
                         float *out, float *dlog, float *grad_rows, float *grad_mem, float *loss, float *prob, double *acc, void *stream);
/* memory.index_copy_(0, (arange(nkeys) + index) % K, keys) (memory_moco.py:55-61) for rows of D floats.  nkeys <= K is required
 * (rc -1 otherwise), as gcc_queue_enqueue requires it: with more keys than queue rows the indices collide and the reference's
 * index_copy_ result depends on the order its kernel happens to write in -- nothing a replacement could be bit-equal to. */
int32_t gcc_queue_enqueue_x(float *mem, int32_t K, int32_t D, const float *keys, int32_t nkeys, int32_t index, void *stream);

/* ------------------------------------------ wide GIN layers, bf16 (config 5) ---
 * BASELINE.json configs[4]: "GIN hid=256 layers=8 deg=32 bf16, SpMM+MFMA-MLP roofline run on batched
 * subgraphs".  The layer stack of UnsupervisedGIN.forward (gcc/models/gin.py:213-221) with hidden 256 and
 * BatchNorm in eval mode (generate.py:71 model.eval()), i.e. per layer
 *     agg = h + sum_{u in row v} h_u                                   (GINConv sum, eps = 0; gin.py:179-185)
 *     z1  = relu(s0 * (agg W0^T) + t0)                                 (mlp.linears.0 + mlp.batch_norms.0; gin.py:113-116)
 *     h'  = relu(s2 * relu(s1 * (z1 W1^T) + t1) + t2)                  (linears.1, apply_func.bn, gin.batch_norms; gin.py:55-57,219-220)
 * with the Linear biases and BatchNorm running statistics folded into the per-channel scale/shift pairs by
 * the caller.  Activations and weights are bf16 (stored as their 16 bits), every product accumulates in
 * f32 on the matrix cores, activations are rounded to bf16 (nearest even) where they are stored: agg, z1, h'.
 * One workgroup keeps one subgraph (<= 128 nodes) in LDS across all `num_layers` layers; pooled[b][i] is the
 * SumPooling of hidden_rep[i] (gin.py:205,228), i = 0 being the input. */
#define GCC_GINW_HIDDEN 256
#define GCC_GINW_MAX_NODES 128
#define GCC_STATUS_GINW_TOO_LARGE 32     /* a subgraph has more than GCC_GINW_MAX_NODES nodes and no scratch was given: its outputs are 0 */
#define GCC_STATUS_GINW_BAD_EDGE 64      /* a neighbour id outside its own subgraph was skipped                 */
typedef struct gcc_ginw_layer {
    const uint16_t *w0, *w1;             /* device [256, 256] bf16, torch Linear layout [out, in]               */
    const float *s0, *t0, *s1, *t1, *s2, *t2;   /* device [256] folded scale / shift                             */
    const uint16_t *w0_frag, *w1_frag;   /* device [256 * 256] bf16: the same weights re-laid by gcc_ginw_pack_weights
                                          * (which = 0 / 1), or NULL.  When every layer of a call has both, the kernel
                                          * requests 1 KiB-contiguous fragments instead of 16 rows x 64 B each (a wave
                                          * request over 16 rows costs ~3x the issue time; 13 % of the fused launch)  */
} gcc_ginw_layer;
typedef struct gcc_ginw_args {
    const int32_t *node_off;             /* device [B + 1]                                                       */
    const int32_t *row_ptr, *col_idx;    /* device batched CSR, global node ids; row v lists the in-neighbours   */
    const uint16_t *x_in;                /* device [N, 256] bf16                                                 */
    uint16_t *x_out;                     /* device [N, 256] bf16 output of the last layer, or NULL               */
    float *pooled;                       /* device [B, num_layers + 1, 256] or NULL                              */
    int32_t batch_size, num_layers;      /* 1 <= num_layers <= GCC_GIN_MAX_LAYERS                                */
    gcc_ginw_layer layers[GCC_GIN_MAX_LAYERS];
    /* Subgraphs over GCC_GINW_MAX_NODES nodes (ego-nets of the pre-training workload reach ~900): with scratch of
     * gcc_ginw_scratch_bytes(num_nodes, batch_size) bytes (device, 16-byte aligned) they run block by block -- one launch
     * per layer, one (subgraph, 128-row block) per workgroup, the block's adjacency strip taken 128 columns at a time with
     * the products accumulating in registers, rows through two global ping-pong buffers between layers -- with the same
     * rounding points as small subgraphs.  scratch == NULL: they are refused (GCC_STATUS_GINW_TOO_LARGE), as before. */
    void *scratch;
    int64_t scratch_bytes;
    int64_t num_nodes;                   /* rows of x_in (node_off[B] <= num_nodes); used with scratch only              */
} gcc_ginw_args;
int64_t gcc_ginw_scratch_bytes(int64_t num_nodes, int32_t batch_size);
/* status: device int32[1], OR of GCC_STATUS_GINW_* (zeroed by the caller).  prof marks: 0 before, 1 after. */
int32_t gcc_ginw_forward(const gcc_ginw_args *a, int32_t *status, gcc_prof *prof, void *stream);
/* Re-lays a [256, 256] bf16 Linear weight (torch layout) in the order gcc_ginw_forward's waves request it: fragment
 * (output block w < 4, fragment m < 4, k-step ks < 8) is 1 KiB contiguous at ((w * 4 + m) * 8 + ks) * 512 elements, lane
 * (16 lg + lr) holding W[row][32 ks + 8 lg .. + 7]; which = 0 (first Linear of a layer): row = 64 w + 32 (m / 2) +
 * 2 lr + m % 2 (the rows of two adjacent fragments interleaved), which = 1 (second Linear): row = 64 w + 16 m + lr.
 * Done once per model. */
int32_t gcc_ginw_pack_weights(const uint16_t *w, uint16_t *w_frag, int32_t which, void *stream);
/* diagnostics, as gcc_posemb_debug_ticks: device int64[16] (rows in, neighbour counts, fragments, aggregation, first
 * Linear, second Linear, rows out; [15] = subgraphs); NULL switches it off. */
void gcc_ginw_debug_ticks(long long *device_ticks64);

/* ------------------------------------------------------- MoCo / InfoNCE head ---
 * MemoryMoCo.forward (gcc/contrastive/memory_moco.py:26-63, use_softmax=True) fused
 * with NCESoftmaxLoss (gcc/contrastive/criterions.py:12-17), and the E2E variant
 * out = feat_k feat_q^T / T with NCESoftmaxLossNS (train.py:400, criterions.py:27-33).
 * logits[b][j] = q_b . mem_j * inv_T; the positive is an extra column q_b . k_b
 * (pos_mode 0) or the diagonal column j == b (pos_mode 1).  The [B, K+1] logits are
 * only written when out_dense != NULL; loss/backward use an online row-softmax. */
#define GCC_NCE_DIM 64
typedef struct gcc_nce_args {
    const float *q;          /* device [B, 64] rows whose softmax is taken                      */
    const float *k;          /* device [B, 64] positives (pos_mode 0) or NULL                   */
    const float *mem;        /* device [K, 64] negatives: the queue, or the other view (E2E)    */
    const float *patch;      /* device [patch_rows, 64] or NULL: rows patch_index.. (mod K) of   */
    int32_t patch_index;     /*   mem are read from here (queue rows already overwritten by the */
    int32_t patch_rows;      /*   enqueue that follows the forward, memory_moco.py:55-61)       */
    int32_t B, K;
    int32_t pos_mode;        /* 0: MoCo, 1: E2E diagonal                                        */
    float inv_T;             /* 1 / nce_t (train.py:87)                                         */
    float *lse, *pos;        /* device [B] out: row log-sum-exp and positive logit               */
    float *loss, *prob;      /* device [1] out: mean(lse - pos) and mean(pos) (train.py:394,407) */
    float *out_dense;        /* device [B, K + (pos_mode == 0)] or NULL                         */
    int32_t dtype;           /* GCC_NCE_F32: exact fp32 MFMA (parity mode, 1e-3 of the reference)  */
                             /* GCC_NCE_BF16: q, k and the queue rounded to bf16 on load, fp32     */
                             /*   accumulation and softmax (throughput mode of north_star; the     */
                             /*   queue itself stays fp32 in HBM / in the checkpoint)              */
} gcc_nce_args;
#define GCC_NCE_F32 0
#define GCC_NCE_BF16 1

int64_t gcc_nce_workspace_bytes(int32_t B, int32_t K);
int32_t gcc_nce_forward(const gcc_nce_args *a, void *workspace, int64_t workspace_bytes, gcc_prof *prof,
                        void *stream);
/* dq[b] = dloss / (B T) * (sum_j p_bj mem_j + [pos_mode 0] (p_b,pos - 1) k_b - [pos_mode 1] mem_b).
 * by_mem_row != 0 computes the gradient w.r.t. the OTHER operand of the E2E product instead:
 * call it with q/mem swapped; p is then normalised with lse[mem row] (args->lse has K entries). */
int32_t gcc_nce_backward(const gcc_nce_args *a, const float *dloss, int32_t by_mem_row, float *dq,
                         void *workspace, int64_t workspace_bytes, gcc_prof *prof, void *stream);

/* memory.index_copy_(0, (arange(n) + index) % K, keys) of memory_moco.py:55-61; when saved != NULL
 * the overwritten rows are first copied there (the `patch` of gcc_nce_args). */
int32_t gcc_queue_enqueue(float *mem, int32_t K, const float *keys, int32_t nkeys, int32_t index, float *saved,
                          void *stream);

/* clip_grad_norm_(max_norm) + Adam.step() of train.py:409,417 over one flat buffer (torch.optim.Adam
 * semantics: L2 weight decay added to the gradient, bias correction with `step` >= 1, eps outside the
 * sqrt).  grad_norm: device [1] out (the pre-clip norm); max_norm <= 0 disables clipping.
 * grad_scale > 0 multiplies the gradient first (1 / world for a gradient that was SUMMED over ranks:
 * norm, clipping and the stored clipped gradient all see grad * grad_scale); 1 on a single GPU.
 * scratch: device double[64], zeroed once by the caller (partial sums and an arrival counter that resets itself). */
int32_t gcc_adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                      float beta1, float beta2, float eps, float weight_decay, int32_t step, float max_norm,
                      float grad_scale, float *grad_norm, double *scratch, void *stream);

/* The meters train.py:418-428 updates every step (loss, prob, gnorm, graph size; max nodes / edges of a q view),
 * accumulated on the device so that a step never synchronises with the host (the reference does, train.py:433):
 * acc: device double[5] += {loss, prob, grad_norm, nodes(q) + nodes(k), 1}; mx: device int32[2] = max with
 * {nodes(q), edges(q)}.  The caller reads and zeroes them when a log line is due. */
int32_t gcc_step_meters(double *acc, int32_t *mx, const float *loss, const float *prob, const float *grad_norm,
                        const int32_t *node_off_q, const int32_t *edge_off_q, const int32_t *node_off_k,
                        int32_t batch_size, void *stream);

/* moment_update of train.py:169-172 over one flat parameter buffer: ema = m * ema + (1 - m) * p */
int32_t gcc_ema_update(float *ema, const float *p, int64_t n, float m, void *stream);

/* The tail of a MoCo step (train.py:409,417-431) in the two launches of gcc_adam_step: clip + Adam over param[0, n),
 * then -- inside the Adam launch -- moment_update of ema[0, n_ema) from param[0, n_ema) (n_ema >= n: the live
 * parameters are a prefix of the flat buffer; the rest is only averaged, as the reference averages its unused
 * set2set / lin_readout weights) and one step of gcc_step_meters with this step's gradient norm.  ema == NULL and/or
 * meters == NULL leave that part out; with both NULL this is gcc_adam_step.  Same results as the three calls. */
typedef struct {
    double *acc;                 /* device double[5], as gcc_step_meters */
    int32_t *mx;                 /* device int32[2] */
    const float *loss, *prob;    /* device [1] each */
    const int32_t *node_off_q, *edge_off_q, *node_off_k;
    int32_t batch_size;
} gcc_step_meters_args;
int32_t gcc_adam_ema_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                          float beta1, float beta2, float eps, float weight_decay, int32_t step, float max_norm,
                          float grad_scale, float *grad_norm, double *scratch, float *ema, int64_t n_ema, float ema_m,
                          const gcc_step_meters_args *meters, void *stream);
/* The same two launches and gcc_queue_enqueue with lr / bias corrections / ring pointer taken from a device-resident
 * gcc_step_scalars (a step captured in a hipGraph: see above).  Same results as the by-value calls. */
int32_t gcc_adam_ema_step_scalars(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                                  float beta1, float beta2, float eps, float weight_decay, float max_norm,
                                  float grad_scale, float *grad_norm, double *scratch, float *ema, int64_t n_ema, float ema_m,
                                  const gcc_step_meters_args *meters, const gcc_step_scalars *scalars, void *stream);
int32_t gcc_queue_enqueue_scalars(float *mem, int32_t K, const float *keys, int32_t nkeys,
                                  const gcc_step_scalars *scalars, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GCC_AMD_H */
