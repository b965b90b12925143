/* gbnns.h -- C ABI of libgbnns_hip.so: the MI355X (gfx950) implementation of the two-stage
 * graph-based ANN search of Shekhale/gbnns_dim_red.
 *
 * The reference has no FFI: its "interface" for this path is the set of free functions in
 * search/search_function.h that search/final_test.cpp calls.  Each entry point below names the
 * reference code it replaces (paths relative to the reference repository root).  The C++ drop-in
 * headers in gbnns_dim_red_amd/search/ keep the reference's names and signatures and call these
 * functions; INTEGRATION.md shows the binding.
 *
 * Conventions: plain pointers and sizes only; every function returns a gbnns_status (0 = OK) and
 * never throws; gbnns_last_error() gives a thread-local message for the last failure.  A handle
 * may be used by one host thread at a time; distinct handles are independent.  All vectors are
 * IEEE binary32, row-major; ids are uint32 (n < 2^31).  There is NO CPU fallback: without a
 * usable gfx950 device gbnns_index_create fails with GBNNS_ERR_NO_DEVICE.
 *
 * Streams: the calls of one handle share its workspace and are ordered by stream order.  A call that names
 * another stream than the handle's previous call first makes the new stream wait for what that call left in
 * flight (the previous stream must still exist, or the device is synchronised) -- results never depend on
 * which stream a call was given.  To overlap batches use one handle per stream (handles may share borrowed
 * device tensors).
 *
 * Device memory per handle besides the index data, per workspace ("lane": one for plain calls, one more per batch in
 * flight with GBNNS_FLAG_DEFER_JOIN, at most four): per-batch buffers (n_q x (d_low + 2 d_hidden + ef + 6) x 4 bytes at
 * most) plus the exact fall-back walk's 64 slots of two n-bit sets and an ef-entry list: 16 n + 512 ef bytes
 * (16 MB at n = 10^6, 160 MB at 10^7).  The large-ef first pass (ef >= 385, deep batches) adds one n-bit set
 * per resident wavefront, capped at 8 GiB.
 *
 * Ids in DEVICE buffers are not validated on the host.  An entry id >= n is never dereferenced: that query
 * gets answer 0xFFFFFFFF, an all-0xFFFFFFFF candidate row and zero counters.  Candidate ids >= n passed to
 * gbnns_rerank read row 0 instead.  HOST buffers are validated (GBNNS_ERR_INVALID).
 */
#ifndef GBNNS_H_
#define GBNNS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GBNNS_VERSION 100 /* 0.1.0 */

typedef enum {
    GBNNS_OK = 0,
    GBNNS_ERR_INVALID = 1,     /* bad argument / inconsistent sizes / id out of range */
    GBNNS_ERR_NO_DEVICE = 2,   /* no HIP device, or device ordinal out of range */
    GBNNS_ERR_HIP = 3,         /* a HIP runtime call failed (message in gbnns_last_error) */
    GBNNS_ERR_OOM = 4,         /* host or device allocation failed */
    GBNNS_ERR_UNSUPPORTED = 5  /* valid request outside what this build implements */
} gbnns_status;

/* support_func.h:87-163: the Metric* slot.  L2 = L2Metric::Dist (:107-128, drops d%4 tail
 * dims); NEG_DOT = Angular::Dist (:131-163, negative dot product). */
typedef enum { GBNNS_METRIC_L2 = 0, GBNNS_METRIC_NEG_DOT = 1 } gbnns_metric;

typedef enum { GBNNS_MEM_HOST = 0, GBNNS_MEM_DEVICE = 1 } gbnns_mem_kind;

/* Which per-query body of the reference harness one batch call reproduces:
 *   NET   search_function.h:353-362  MLP projection -> walk(ef,k=ef) in low-dim space -> re-rank
 *   LOWQ  search_function.h:158-164  same with caller-supplied low-dim queries (performTest)
 *   PLAIN search_function.h:174-181  walk(ef,k) directly in the space of `db`, answer = top of the heap trimmed
 *         to k, i.e. the k-th best (the best for k = 1, which is what the reference's drivers pass) */
typedef enum { GBNNS_MODE_NET = 0, GBNNS_MODE_LOWQ = 1, GBNNS_MODE_PLAIN = 2 } gbnns_mode;

typedef struct gbnns_index gbnns_index;

/* Everything final_test.cpp:50-76 loads for one dataset.  The callee copies HOST buffers to HBM
 * (caller keeps ownership and may free them after create returns); DEVICE buffers (mem_kind =
 * GBNNS_MEM_DEVICE, pointers valid on `device`) are borrowed and must outlive the index.  The
 * graph is always given in host memory as CSR (neighbour order preserved: it drives tie
 * behaviour through visit order); it is converted once to the device layout. */
typedef struct {
    uint32_t struct_size;          /* = sizeof(gbnns_index_desc) */
    int32_t device;                /* HIP device ordinal */
    int32_t metric;                /* gbnns_metric, used by walk and re-rank */
    int32_t mem_kind;              /* gbnns_mem_kind of db, db_low, net_* */
    uint64_t n;                    /* base vectors */
    uint32_t d;                    /* original dimension */
    uint32_t d_low;                /* low dimension (0: PLAIN mode only) */
    uint32_t d_hidden;             /* MLP width (0: no net -> NET mode unavailable) */
    uint32_t reserved0;
    const float* db;               /* [n x d]          final_test.cpp:50 */
    const float* db_low;           /* [n x d_low]      final_test.cpp:56; may be NULL */
    const uint64_t* graph_offsets; /* [n + 1] host     final_test.cpp:61-63 (loadEdges) */
    const uint32_t* graph_nbrs;    /* [offsets[n]] host */
    const float* net_l1;           /* [d_hidden x (d+1)]        rows = [W | b], final_test.cpp:73-76 */
    const float* net_l2;           /* [d_hidden x (d_hidden+1)] */
    const float* net_l3;           /* [d_low x (d_hidden+1)]    */
} gbnns_index_desc;

/* Replaces the data/graph/net set-up half of final_test.cpp main (:50-76) + the per-call
 * VisitedListPool allocation (search_function.h:331). */
int gbnns_index_create(const gbnns_index_desc* desc, gbnns_index** out);
int gbnns_index_destroy(gbnns_index* index);
/* what the handle was created with */
uint64_t gbnns_index_n(const gbnns_index* index);
uint32_t gbnns_index_d(const gbnns_index* index);
uint32_t gbnns_index_d_low(const gbnns_index* index);
int gbnns_index_device(const gbnns_index* index);

/* The `auxiliary_graph` argument of getOneSearchResults (search_function.h:44; naive_test.cpp:103-105
 * passes the KL graph): host CSR over the same n nodes, neighbour order preserved, copied to the device
 * layout.  NULL, NULL removes it.  Used by searches that set GBNNS_FLAG_AUX_GRAPH.  Synchronises the
 * device. */
int gbnns_index_set_aux_graph(gbnns_index* index, const uint64_t* offsets, const uint32_t* nbrs);

typedef struct {
    uint32_t struct_size;      /* = sizeof(gbnns_search_args) */
    int32_t mode;              /* gbnns_mode */
    int32_t ef;                /* beam width (= recheck_size in NET/LOWQ, search_function.h:432-434) */
    int32_t k;                 /* PLAIN: heap is trimmed to k, answer = its top = the k-th best (reference uses 1) */
    int32_t mem_kind;          /* gbnns_mem_kind of every buffer below */
    int32_t hash_capacity;     /* 0 = auto (sized from the LDS budget and earlier batches); else entries
                                  (>= 128) of the per-query LDS visited set */
    uint64_t n_q;
    const float* queries;      /* [n_q x d] */
    const float* queries_low;  /* LOWQ: [n_q x d_low], else NULL */
    const uint32_t* entry_ids; /* [n_q x max(n_entries, 1)] entry node(s) per query, NULL = node 0
                                  (search_function.h:417-427) */
    uint32_t* out_ids;         /* [n_q] answers (ans[i], search_function.h:361) */
    int32_t* out_hops;         /* optional [n_q]  TripleResult.hops */
    int32_t* out_dist_calc;    /* optional [n_q]  TripleResult.dist_calc (walk only, without the
                                  "+ recheck_size" the harness adds at :362) */
    uint32_t* out_cand;        /* optional [n_q x min(k,ef)] result heap in POP order (worst->best),
                                  0xFFFFFFFF padded; k = ef in NET/LOWQ */
    float* out_cand_dist;      /* optional, same shape: low-dim distances of out_cand (+inf pad) */
    float* out_q_low;          /* optional NET: [n_q x d_low] projected queries */
    int32_t* out_edges;        /* optional [n_q]: neighbour ids read by the walk (sum of the degrees of
                                  the expanded nodes) -- feeds the algorithmic-bytes figure */
    void* stream;              /* hipStream_t to enqueue on (NULL = default stream) */
    uint32_t flags;            /* GBNNS_FLAG_* */
    uint32_t hops_bound;       /* with GBNNS_FLAG_AUX_GRAPH: auxiliary rows are expanded while hops < hops_bound
                                  (search_function.h:73; the harness passes 50, :273/:338) */
    uint32_t n_entries;        /* entry points per query (inter_points[i].size(), search_function.h:54); 0 or 1 =
                                  one.  More than one: one walk per entry point over a shared result heap, exactly
                                  as :54-93 -- served by the general kernel (exact, not tuned: no driver of the
                                  reference uses it) */
    uint32_t defer_depth;      /* with GBNNS_FLAG_DEFER_JOIN: batches in flight, 2 .. 4 (0 = 3, the measured optimum) */
} gbnns_search_args;

/* The throughput option (opt-in; never a default, never what bench.py reports as `value`): the projection's three layers
 * as GEMMs on the matrix cores (v_mfma_f32_32x32x2_f32, csrc/mlp.hip: mlp_layer_mfma_kernel).  An output is then ONE
 * k-ordered fma chain instead of the reference's eight separately rounded running sums (support_func.h:131-163): q_low
 * differs from the exact kernels' in its last bits (a few 1e-7), and a few answers of a batch may differ from the
 * reference's -- bench.py states how many (throughput_option.id_mismatches_vs_reference).  (Rounds 1-2 had this option on
 * flag bit 1; removed in round 3, rebuilt in round 5 on bit 8.) */
#define GBNNS_FLAG_MFMA_PROJECTION 256u
/* Diagnostic: keep the re-rank (getRealNearest, search_function.h:105-125) in its own kernel launch even
 * where the walk kernels could re-rank each query at the end of its walk.  Results are identical either
 * way; the flag exists for A/B measurements and for timing the two stages separately. */
#define GBNNS_FLAG_NO_FUSED_RERANK 2u

/* use_second_graph = true (search_function.h:73-89): while a query has made fewer than `hops_bound` hops,
 * the auxiliary row of the expanded node (gbnns_index_set_aux_graph) is offered before its main row.  With
 * GBNNS_FLAG_LLF (the reference's `llf`) the main row is skipped on a hop whose auxiliary step inserted
 * something (:82).  Bit-identical to the reference. */
#define GBNNS_FLAG_AUX_GRAPH 4u
#define GBNNS_FLAG_LLF 8u

/* Diagnostic: run the kernels an index of n >= 2^24 nodes or tables >= 4 GiB would get (64-bit offsets, 4-byte
 * visited-set slots) on a small index, so that the tests can reach them.  Same results. */
#define GBNNS_FLAG_WIDE_INDEX 16u

/* Diagnostic: take the first pass with HBM visited bitmaps (persistent wavefronts, LDS-list kernel) whatever ef and
 * batch size -- by default it serves ef >= 385 on batches deeper than 1.5 rounds of the wavefronts an LDS visited
 * table would allow.  Same results. */
#define GBNNS_FLAG_BITMAP_PASS 32u

/* Batches in flight.  A handle owns up to four workspaces with an internal HIP stream each ("lanes"); plain calls use
 * the first one on the caller's stream.  Every query is independent (search_function.h:348), so the answers never
 * depend on how a call is laid out.
 *   GBNNS_FLAG_DEFER_JOIN  DEVICE buffers: the batch runs on the next of `defer_depth` lanes (consecutive calls rotate)
 *                          after what was enqueued on args->stream before the call, and the call returns WITHOUT
 *                          making args->stream wait for it.  That wait is enqueued by the (defer_depth - 1)-th following
 *                          call on this handle -- after that call's own batch has been released, so `defer_depth`
 *                          batches are in flight: the half-empty tail of batch i's walk kernel (a 10 k batch is < 2
 *                          "rounds" of resident wavefronts) runs beside the projection and the head of batch i+1, and
 *                          batches too small to fill the machine (the reference's 1 000 GIST queries are one
 *                          wavefront per SIMD) run side by side -- or by gbnns_index_join, or by any call without the
 *                          flag.  Until then the outputs of the call must not be read and its inputs / outputs must
 *                          not be reused: a serving loop rotates `defer_depth` sets of buffers.  args->stream must
 *                          stay alive until the call has been joined.
 *   GBNNS_FLAG_SERIAL      the caller's stream, kernels back to back, whatever else is asked (what per-kernel timing
 *                          needs; gbnns_profile_enable(..., 1) implies it).  HOST-buffer calls run this way and
 *                          synchronously unless they ask for GBNNS_FLAG_DEFER_JOIN with page-locked buffers (below). */
#define GBNNS_FLAG_SERIAL 64u
#define GBNNS_FLAG_DEFER_JOIN 128u

/* Replaces the timed query loop of performNetTest (search_function.h:346-387) / performTest
 * (:151-188): one call = the whole batch.  With HOST buffers the call copies in, runs and
 * copies out synchronously (what the drop-in harness times).  With DEVICE buffers everything is
 * enqueued on args->stream and the call returns without synchronising.
 * Queries are independent (the reference's loop is an OpenMP parallel for over them): the order in which the device
 * serves the queries of a batch is the library's business -- batches of >= 32 768 queries are walked in a locality
 * order -- and every output row i always belongs to query i. */
int gbnns_search_ex(gbnns_index* index, const gbnns_search_args* args);

/* Enqueues, on the streams of the GBNNS_FLAG_DEFER_JOIN calls not yet joined, the waits for their batches (no-op when
 * nothing is owed).  Everything enqueued on those streams afterwards sees the calls' outputs. */
int gbnns_index_join(gbnns_index* index);

/* Host batches in flight.  GBNNS_FLAG_DEFER_JOIN also takes HOST buffers when every one of them is page-locked
 * (hipHostMalloc / hipHostRegister; with a pageable buffer among them the flag is ignored and the call is the plain
 * synchronous one -- a caller that follows the protocol below is correct either way): the copy of the queries to the device, the kernels and
 * the copies out all go to the lane's stream and the call returns at once, so the wire time of batch i+1 (0.10 ms of
 * a 0.54 ms synchronous call for 10 000 x 128 floats) passes under the kernels of batches i and i-1.  The ids are
 * stored by the kernel straight into the page-locked out_ids (so are they in a synchronous HOST call whose out_ids
 * happens to be page-locked).  gbnns_index_wait blocks the calling thread until every deferred batch but the `keep`
 * most recent has finished: their host buffers then hold the results and may be refilled (keep = depth - 1 after
 * each call keeps the pipeline full; keep = 0 drains it).  Defined in terms of the reference: each call is still one
 * run of the timed loop of search_function.h:346-387 over its own batch. */
int gbnns_index_wait(gbnns_index* index, uint32_t keep);

/* Page-locks (hipHostRegister) / releases a host buffer, for callers that do not link the HIP runtime themselves: HOST
 * buffers that are page-locked are copied in by DMA without a staging pass, take ids, hop counts, dist_calc and edge
 * counts straight from the kernels (no copies out), and may be used with GBNNS_FLAG_DEFER_JOIN.  The drop-in's
 * perform*Test functions pin the query / answer vectors before their timed region (the reference builds its
 * VisitedListPool there, search_function.h:333) and release them after it.  Pinning an already page-locked buffer
 * and releasing one that was not pinned here are no-ops.  Registrations are made in whole pages and never overlap:
 * small buffers that share a page share its registration (counted), which is released with its last user. */
int gbnns_host_pin(void* ptr, size_t bytes);
int gbnns_host_unpin(void* ptr);

/* Convenience form of the above: NET mode, host buffers, synchronous. */
int gbnns_search_batch(gbnns_index* index, const float* queries, size_t n_q, int ef,
                       const uint32_t* entry_ids, uint32_t* out_ids, int32_t* out_hops,
                       int32_t* out_dist_calc, uint32_t* out_cand);

/* GetLowQueryFromNet (support_func.h:645-658) over a batch: x [n_x x d] -> out [n_x x d_low].
 * Also what produces `<name>_base_angular_optimal.fvecs` from the base set. */
int gbnns_project(gbnns_index* index, const float* x, uint64_t n_x, float* out, int mem_kind,
                  void* stream);

/* getRealNearest (search_function.h:105-125) over a batch: for query i, cand[i*stride ..] holds
 * count[i] (NULL: stride) candidate ids in POP order (worst -> best in the low-dim space); the
 * answer is the strict minimum of the exact distance in the space of `db`, earlier entries
 * winning ties. */
int gbnns_rerank(gbnns_index* index, const float* queries, uint64_t n_q, const uint32_t* cand,
                 uint32_t cand_stride, const int32_t* count, uint32_t* out_ids, int mem_kind,
                 void* stream);

/* Per-kernel device timing (hipEvent pairs on the launch stream), accumulated since the last
 * reset.  Reading synchronises the recorded events. */
typedef struct {
    uint32_t struct_size;         /* in: the caller's sizeof(gbnns_profile) (0 = the 160-byte layout of rounds 1-4, which ends before
                                     project_kernel); out: the bytes gbnns_profile_read filled in -- never more than the caller said */
    uint32_t calls;               /* search calls accumulated */
    double project_ms;            /* MLP layers + normalise */
    double walk_ms;               /* LDS-resident beam-walk kernel */
    double walk_general_ms;       /* exact general-case kernel (queries the LDS kernel handed over) */
    double rerank_ms;             /* original-space re-rank kernel (~0 when the walk kernels re-rank: then it is part of walk_ms) */
    double total_ms;              /* first launch .. last launch of each call */
    uint64_t queries;             /* queries processed */
    uint64_t general_queries;     /* of which were (re)run by the general kernel */
    char walk_kernel[96];         /* first-pass walk kernel of the last profiled call, template arguments included */
    char project_kernel[32];      /* kernel family of the handle's last projection: "mlp_net_kernel" (one launch) or "mlp_layer_kernels" */
} gbnns_profile;

int gbnns_profile_enable(gbnns_index* index, int on);
/* Diagnostic knobs (tests, A/B runs); results never depend on them.  Since round 6 every knob that steers a handle's searches
 * belongs to the handle: gbnns_index_knob(index, name, value) sets it for that handle alone, and gbnns_debug_knob(name, value)
 * only changes the process-wide DEFAULT a handle created afterwards starts from (initial defaults: the environment variables
 * named below).  Two handles of one process can therefore run with different settings side by side.  The three "knn_*" knobs
 * steer gbnns_exact_knn, which has no handle, and stay process-wide (gbnns_debug_knob only).
 * "coop": the two-wavefront walk for small batches (one query per workgroup of two wavefronts: one keeps the result lists, one
 * expands the predicted next node) -- -1 (default) where the shape has it and the batch leaves the room, 0 never,
 * 1 wherever the shape has it, 2 the same with three wavefronts per query (the expanding wavefront split in two: measured slower,
 * kept for tests and A/B runs) (GBNNS_COOP).  "coop_pack": 1 launches that three-wavefront form with its own LDS only (default 0:
 * a batch of at most c workgroups per CU asks for 1 / c of a CU's LDS each, which spreads them evenly; GBNNS_COOP_PACK).
 * "quotient": 0 keeps the walk_hot*
 * kernels' visited set in its packed form (default 1: the denser quotient form where it fits; initial value from the
 * environment variable GBNNS_QUOTIENT).  "vs_disp": probe number at which a probe sequence of the quotient form gives
 * up and the id goes to the stash / the query is handed over, 1 .. 15 (default 15; <= 0 restores it; GBNNS_DEBUG_VS_DISP).
 * "max_waves": most first-pass wavefronts per CU the LDS shares are cut for (0 = default; GBNNS_MAX_WAVES).
 * "spec_min_nq": smallest batch whose ef <= 64 first pass requests a hop's rows before its visited test
 * (walk_hot_spec_kernel; default 32 768, 0 = never; GBNNS_SPEC_MIN_NQ) -- on indexes whose visited sets are not in the
 * quotient form (more than 2^21 rows or so), unless "spec_any_form" is 1 (tests; GBNNS_SPEC_ANY_FORM).
 * "spec_tail": a synchronous call whose last round of wavefronts fills at most this many percent of the device's 8 192
 * slots runs that round with the rows requested before the visited test, and a synchronous batch of at most 60 % of the
 * slots runs so throughout (default 50, 0 = neither; GBNNS_SPEC_TAIL).
 * "mlp_small": smallest batch in flight (GBNNS_FLAG_DEFER_JOIN) whose hidden projection layers run on the small-footprint
 * kernel (mlp_layer_sw_kernel; default 4 096, up to 32 times that, 0 = never; GBNNS_MLP_SMALL).
 * "mlp_net": 1 (default) = batches of 2 048 queries and more whose net has d % 8 == 0 and d_hidden % 8 == 0 are projected
 * by the one-launch kernel (mlp_net_kernel: the three layers of a strip of queries in one workgroup, activations in LDS),
 * 0 = always the per-layer kernels; identical outputs either way (GBNNS_MLP_NET).
 * "mlp_slab": 1 (default) = a projection layer that is one round of the machine for mlp_slab_kernel (small batches; the GIST
 * shape's 1 000 queries through 960 -> 1 024 -> 1 024 -> 64) takes it, 0 = never; identical outputs either way (GBNNS_MLP_SLAB).
 * "late_rows": the wide-row instances that have both forms (walk_reg_wide_kernel: 192- / 256-byte rows at ef <= 64; the two-list kernel
 * over 576-byte rows) request a hop's rows before its visited test (0) or after it, for the new ids only (1); -1 (default) = by
 * shape and residency (576-byte rows with at least five wavefronts per CU, 192-byte rows at ef <= 64: after; GBNNS_LATE_ROWS).
 * "knn_chunk": most base rows per filtered chunk of gbnns_exact_knn (default 32 768; the pool path takes four times that;
 * GBNNS_KNN_CHUNK).
 * "knn_pool_min_k": shortest list gbnns_exact_knn's filter path keeps as an unordered pool with a radix select (one
 * wavefront per query) instead of a binary heap (default 64; GBNNS_KNN_POOL_MIN_K).
 * "knn_filter": gbnns_exact_knn's matrix-core filter -- 0 never, 1 by size (default), 2 whenever the shape allows
 * (GBNNS_KNN_FILTER). */
int gbnns_debug_knob(const char* name, int value);
int gbnns_index_knob(gbnns_index* index, const char* name, int value);   /* GBNNS_ERR_INVALID: not a handle knob */
int gbnns_index_knob_get(gbnns_index* index, const char* name, int* out_value);   /* the handle's current value */
int gbnns_profile_read(gbnns_index* index, gbnns_profile* out, int reset);

/* hnswlikeGD (support_func.h:521-575, need_const_degree = false) + addReverseEdgesForGD
 * (:402-445): prunes a kNN graph (CSR, host) over `ds` [n x d] (host) into the search graph, as
 * prepare_graph.cpp:70 does with M = 30, reverse = true.  Host code (OpenMP); returns malloc'ed
 * CSR arrays the caller releases with gbnns_free. */
int gbnns_build_graph_gd(const uint64_t* knn_offsets, const uint32_t* knn_nbrs, const float* ds,
                         uint64_t n, uint32_t d, int M, int metric, int reverse, int threads,
                         uint64_t** out_offsets, uint32_t** out_nbrs);
void gbnns_free(void* p);

/* The same builder with the per-node pruning (support_func.h:528-563) on the device, one node per wavefront.
 * The reference sorts a node's candidates with std::sort on the distance alone, which leaves the order of EQUAL
 * distances to the library's algorithm; nodes whose list contains equal distances (and lists longer than 1024)
 * are therefore finished on the host by that very call, so the result is always the host builder's, bit for bit.
 * The reverse-edge pass (:402-445) is serial and order dependent and stays on the host.  *out_host_nodes
 * (optional) = nodes that went to the host.  M > 64 or d > 128: everything runs on the host. */
int gbnns_build_graph_gd_device(int device, const uint64_t* knn_offsets, const uint32_t* knn_nbrs, const float* ds,
                                uint64_t n, uint32_t d, int M, int metric, int reverse, int threads,
                                uint64_t** out_offsets, uint32_t** out_nbrs, uint64_t* out_host_nodes);

/* Exact brute-force nearest neighbours on the device, in the reference's distance arithmetic.
 *   k = 1: getTruth (support_func.h:270-290) -- the strict minimum of Dist(base_j, q_i) over ascending j;
 *   k > 1: the exact kNN lists that feed the graph builder (dim_red/support_func.py:374-384 writes
 *          `<name>_knn_1k_<style>.ivecs`, prepare_graph.cpp:66 reads it).
 * out_ids [n_q x k] (and optional out_dist): the k smallest (distance, id) pairs of each query in ascending
 * pair order, ties towards the lower id; 0xFFFFFFFF / +inf where fewer than k rows qualify.
 * self_offset >= 0 says query i IS base row i + self_offset and must not be reported as its own neighbour
 * (kNN graph of a set over itself, possibly computed in slices of queries); -1 turns that off.
 * Large L2 problems (d % 4 == 0, d <= 128, n >= 2^17, n_q >= 2048, k <= 4096) go through a matrix-core filter first
 * (v_mfma_f32_32x32x16_bf16 on hi / lo bf16 halves, error-bounded) and only the rows it cannot rule out get their exact
 * distance: same output, byte for byte, several times faster.  Lists of 64 entries and more are kept as
 * unordered pools with a radix select per chunk (one wavefront per query) instead of binary heaps: 1 000-NN lists of
 * 10^6 x 32 in 1.4 s (7.6 s).
 * d <= 8192 (d > 128 runs a kernel that streams the query through in chunks); GBNNS_METRIC_NEG_DOT needs
 * d % 8 == 0 (else GBNNS_ERR_UNSUPPORTED).  Buffers are all host or all
 * device (mem_kind); the work runs on `stream` and the call returns when it has finished (a k x n_q x 8-byte
 * workspace lives for the duration of the call). */
int gbnns_exact_knn(int device, const float* base, uint64_t n, const float* queries, uint64_t n_q,
                    uint32_t d, int k, int metric, int64_t self_offset, uint32_t* out_ids, float* out_dist,
                    int mem_kind, void* stream);

/* ---- several devices of one node: query-sharded replicas ----------------------------------------------------
 * The reference's only parallelism on this path is `#pragma omp parallel for` over the queries of a batch
 * (search_function.h:152; the loop body :348-385 is pure per query).  Here that loop is cut into contiguous blocks,
 * one per replica of the index: one gbnns_index per entry of `devices` (NULL / 0 = every visible device; an ordinal
 * may repeat: the replicas then share that device), each driven by its own host thread on its own HIP stream.
 * Nothing is exchanged while searching; results are combined as described per call.  desc as for
 * gbnns_index_create (HOST buffers; desc->device is ignored).  What the drop-in uses when GBNNS_DEVICES=0,1,...
 * names more than one device. */
typedef struct gbnns_multi gbnns_multi;
int gbnns_multi_create(const gbnns_index_desc* desc, const int32_t* devices, int32_t n_devices, gbnns_multi** out);
int gbnns_multi_destroy(gbnns_multi* multi);
int gbnns_multi_size(const gbnns_multi* multi);                    /* replicas */
gbnns_index* gbnns_multi_replica(gbnns_multi* multi, int32_t i);   /* borrowed */
int gbnns_multi_device_of(const gbnns_multi* multi, int32_t i);
void* gbnns_multi_stream(gbnns_multi* multi, int32_t i);           /* hipStream_t of replica i */
int gbnns_multi_set_aux_graph(gbnns_multi* multi, const uint64_t* offsets, const uint32_t* nbrs);
/* Block of part `part` of `parts` parts of a batch: [*lo, *hi), contiguous, sizes differ by at most one. */
void gbnns_shard_bounds(uint64_t n_q, int32_t parts, int32_t part, uint64_t* lo, uint64_t* hi);
/* One batch over all replicas, HOST buffers holding the WHOLE batch (args as gbnns_search_ex; args->stream is
 * ignored): replica r searches rows gbnns_shard_bounds(n_q, R, r) and writes its answers, counters and candidate
 * rows straight into the caller's arrays -- identical to what one gbnns_search_ex call would have written.
 * Returns when every replica has finished. */
int gbnns_multi_search_ex(gbnns_multi* multi, const gbnns_search_args* args);
/* Device-resident form (NET / PLAIN): query_blocks[r] = rows gbnns_shard_bounds(n_q, R, r) of the batch, resident on
 * replica r's device ([rows x d]); entry_blocks likewise or NULL; out_ids_all[r] = [n_q] uint32 on replica r's
 * device.  Every replica searches its block, then the answer ids are all-gathered (ONE ncclAllGather of uint32
 * per batch over RCCL / xGMI, sends padded to the largest block; librccl is loaded on first use; replicas must then
 * sit on distinct devices) so that every out_ids_all[r] holds the whole answer vector.  `args` supplies mode, ef,
 * k, flags, hops_bound, hash_capacity; its buffers and n_q are ignored.  Enqueued on the replicas' streams:
 * gbnns_multi_synchronize waits for them. */
int gbnns_multi_search_device(gbnns_multi* multi, const gbnns_search_args* args, uint64_t n_q,
                              const float* const* query_blocks, const uint32_t* const* entry_blocks,
                              uint32_t* const* out_ids_all);
int gbnns_multi_synchronize(gbnns_multi* multi);
/* A handle with ONE replica copies its answers instead of gathering them.  on != 0: it goes through the exchange leg as
 * well -- librccl loaded, a one-rank communicator (ncclCommInitAll over its device), ncclAllGather of the single block --
 * so that symbol resolution, stream order and the gather buffer's layout can be exercised on a one-GPU machine. */
int gbnns_multi_rccl_single_rank(gbnns_multi* multi, int on);
/* ncclGetVersion() of the librccl this handle has loaded (e.g. 22707), 0 while it has not loaded one. */
int gbnns_multi_rccl_version(gbnns_multi* multi);
const char* gbnns_multi_last_error(void);

int gbnns_device_count(void);
int gbnns_version(void);
const char* gbnns_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* GBNNS_H_ */
