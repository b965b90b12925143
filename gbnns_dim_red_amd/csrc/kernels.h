// kernels.h -- launch interface between the C-ABI host layer (handle.cpp, lanes.cpp, search_core.cpp, sizing.cpp, graph_api.cpp, pin.cpp) and the gfx950 kernels
// (the .hip units beside it: walk_hot, walk_l2 / walk_dot / walk_wide, walk_bitmap, walk_general, rerank, mlp, gd_order, knn).  Plain structs of device pointers and sizes; no HIP types besides hipStream_t.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gbnns {

constexpr uint32_t kInvalidId = 0xFFFFFFFFu;
constexpr int kWave = 64;
constexpr int kTieCap = 64;            // LDS tie list of the fast walk kernel (one slot per lane)
constexpr int kGeneralSlots = 64;      // persistent waves of the general walk kernel
constexpr int kRetrySlots = 256;       // persistent waves of the retry pass (one per CU: it takes all the LDS)

// Beam walk (search_function.h:43-102 + :15-40).  One query per wavefront.
struct WalkParams {
    const float* q;          // [nq x qstride] query vectors in the walked space
    uint32_t qstride;        // floats
    const float* db;         // [n x dstride] base vectors in the walked space (rows 16-B aligned)
    uint32_t dstride;        // floats, multiple of 4
    uint32_t dim;            // true dimension of the walked space
    const uint32_t* ell;     // [n x ell_stride] padded adjacency, kInvalidId-terminated rows
    uint32_t ell_stride;     // multiple of 16
    uint32_t n;
    uint32_t nq;
    int32_t ef;
    int32_t k;               // results kept (<= ef); cand_stride = min(k, ef)
    const uint32_t* entries; // [nq x n_entries] or nullptr (= node 0)
    uint32_t n_entries;      // entry points per query (0 / 1: one); more than one runs on the general kernel only
    // LDS visited set of the fast kernel
    uint32_t hash_cap;       // entries of the LDS visited set (any size >= 128)
    uint32_t hash_limit;     // max entries before a query is handed to the general kernel
    // outputs
    uint32_t* cand;          // [nq x cand_stride] pop order (worst -> best), kInvalidId pad
    float* cand_dist;        // optional, same shape
    uint32_t zero_dist_bits; // bit pattern of a zero distance on output: 0 (L2), 0x80000000 (negative dot: -0)
    uint32_t cand_stride;
    int32_t* count;          // [nq] valid entries in cand
    int32_t* hops;           // [nq]
    int32_t* dist_calc;      // [nq]
    int32_t* edges;          // optional [nq]: neighbour ids read (sum of degrees of expanded nodes)
    uint32_t* best;          // optional [nq]: id of the best result (PLAIN mode answer)
    // hand-over lists: first pass -> A -> retry pass (largest LDS visited set) -> B -> general kernel
    uint32_t* ovf_count;     // [1]   list A
    uint32_t* ovf_list;      // [nq]
    uint32_t* ovf2_count;    // [1]   list B
    uint32_t* ovf2_list;     // [nq]
    uint32_t* r_cursor;      // [1]   work-queue head of the retry pass
    // general kernel workspace: per slot [visited bits | tie bits: 2 x bitmap_words][keys ef]
    // first pass with the visited set in HBM (large ef): per-slot bitmaps and the work-queue head
    uint32_t* fp_bitmap;     // [slots x bitmap_words]
    uint32_t* fp_cursor;     // [1]
    uint32_t rr_reserve;     // bytes the bitmap first pass of the two-list kernel sets aside for the re-rank query (0: none)
    uint32_t* g_cursor;      // [1] work-queue head
    uint32_t* g_total;       // [1] running count of queries the general kernel processed
    uint32_t* max_dc;        // [1] max dist_calc over the batch (feeds the host's visited-set sizing)
    uint32_t* next_ctrl;     // [5] control words of the NEXT call: the general kernel clears them
    uint32_t* g_bitmap;      // [slots x 2 x bitmap_words]: visited bits, then the tie set (one bit per node)
    uint64_t* g_keys;        // [slots x (ef + n_entries - 1)]
    uint32_t bitmap_words;
    int32_t all_general;     // 1: the general kernel takes every query (fast kernel skipped)
    // fused re-rank (rr_db != nullptr): every walk kernel re-ranks its own query at the end of the walk
    // (pair form: rr_dim % 8 == 0; rr_dstride * 4 bytes of LDS available) and writes the answer to rr_out
    const float* rr_q;       // [nq x rr_qstride] original-space queries
    uint32_t rr_qstride;
    const float* rr_db;      // [n x rr_dstride] original-space vectors
    uint32_t rr_dstride, rr_dim, rr_n;
    int32_t rr_metric;       // gbnns_metric of the re-rank
    uint32_t* rr_out;        // [nq]
    // auxiliary graph (search_function.h:73-89, use_second_graph): while hops < hops_bound a node's auxiliary
    // row is expanded before its main row; with llf the main row is skipped when that step inserted anything
    const uint32_t* aux_ell; // [n x aux_stride] or nullptr (= use_second_graph false)
    uint32_t aux_stride;
    uint32_t hops_bound;
    int32_t llf;
    // visited-set form of the walk_hot* first pass: vs_shr == 0 -> five 24-bit ids per 16-byte bucket; else the quotient
    // form (seven 16-bit entries per bucket; n <= 2^W): vs_shr = (32 - W + floor(log2 buckets)) | (32 - W) << 8 |
    // (13 remainder bits ? 1 << 16 : 0) | limit << 28 (top four bits: a probe sequence gives up -- stash, hand-over -- when
    // its probe number reaches 15, or 7 << 1 with 13 remainder bits; less in test runs)
    uint32_t vs_shr;
    int32_t spec_rows;       // ef <= 64 hot first pass: 1 = request the rows before the visited test (big launches, see sizing.cpp)
    int32_t late_rows;       // generic two-list kernels over 192- / 256- / 576-byte rows: 1 = request a hop's rows after its visited test (new ids only)
    uint32_t spec_from;      // ... and in the tested-first instances, for the wavefronts from this work item on (the last, partial round
                             // of a launch walks a draining machine: the shorter hop wins there); 0xFFFFFFFF = none
    const uint32_t* order;   // optional [nq]: work item b of a first pass runs query order[b] (a permutation: locality order of a deep batch)
    int32_t coop;            // 1: the first pass is the two-wavefront walk (walk_coop.hip: kCoopExtraLds more bytes of LDS per query), 2: its three-wavefront form (+ kCoop3MoreLds)
    uint32_t coop_lds_floor; // ... launched with at least this much LDS per workgroup: a batch of at most c workgroups per CU asks for 1 / c of the CU's LDS,
                             // so that the dispatcher cannot stack more than c on one CU and leave others short (0: none)
    int32_t force_wide;      // diagnostic: treat the index as a large one (64-bit offsets, 4-byte visited-set slots)
    unsigned long long* stamps;  // diagnostic builds only (GBNNS_STAMPS): [32] segment cycle sums / histograms
    int32_t stamps_on;           // 1 in diagnostic builds: use the instrumented generic kernel
};

bool walk_uses_hot(const WalkParams& p, int metric);           // first pass runs walk_hot_kernel
bool walk_uses_lds_list(const WalkParams& p);                   // result list in LDS (walk_fast_kernel) instead of registers
size_t walk_fast_lds_bytes(const WalkParams& p, bool hot);
bool walk_uses_packed(const WalkParams& p);                     // visited set of 24-bit ids, five per 16-byte bucket
// `form` of a visited set: 0 = 4-byte slots, 1 = five 24-bit ids per 16-byte bucket, 2 = quotient form (seven 16-bit entries)
size_t walk_hash_bytes(uint32_t entries, int form);             // LDS bytes of a visited set of `entries` ids
uint32_t walk_hash_entries(size_t bytes, int form);             // ids that fit into `bytes` (whole buckets)
bool walk_knows_quotient(const WalkParams& p, int metric);      // the first-pass kernel of this shape reads p.vs_shr
int walk_hash_form(const WalkParams& p, bool hot);              // the form the first pass uses (p.vs_shr chooses 2 for the hot kernels)
size_t walk_fast_lds_fixed_bytes(int ef, uint32_t dstride, bool hot, bool lds_list = false, int coop = 0);  // everything but the visited set
                                                                                             // (lds_list: walk_uses_lds_list; coop: WalkParams::coop)
constexpr size_t kCoop3MoreLds = 768;   // three wavefronts: the ranger's own three slots of adjacency words
constexpr size_t kCoopExtraLds = 1280;  // the two-wavefront walk's two 64-word result buffers + three 64-word slots of adjacency words requested ahead (its mailbox lives in the query area)
bool walk_coop_serves(const WalkParams& p, int metric);   // shape the two-wavefront walk has an instance for (walk_coop.hip)
hipError_t launch_walk_coop(const WalkParams& p, hipStream_t s);
hipError_t launch_walk_fast(const WalkParams& p, int metric, hipStream_t s);
hipError_t launch_walk_retry(const WalkParams& p, int metric, hipStream_t s);
hipError_t launch_walk_general(const WalkParams& p, int metric, hipStream_t s);
bool walk_bitmap_uses_reg(const WalkParams& p, int metric);
size_t walk_bitmap_lds_bytes(const WalkParams& p, int metric);
const char* walk_first_pass_name(hipStream_t s);  // (mangled) name of the first-pass kernel this thread launched last
size_t walk_rr_room(const WalkParams& p, int metric, bool hot, bool bitmap_pass);  // LDS the fused re-rank may stage its query in
hipError_t launch_walk_bitmap(const WalkParams& p, int metric, unsigned slots, hipStream_t s);  // persistent first pass, HBM bitmaps

// Re-rank (search_function.h:105-125).  One query per wavefront, one candidate per lane.
struct RerankParams {
    const float* q;          // [nq x qstride] original-space queries
    uint32_t qstride;
    const float* db;         // [n x dstride]
    uint32_t dstride;        // floats, multiple of 4
    uint32_t dim;
    const uint32_t* cand;    // [nq x cand_stride] pop order
    uint32_t cand_stride;
    const int32_t* count;    // [nq]
    uint32_t nq;
    uint32_t n;              // rows of db (ids are clamped against it)
    uint32_t* out;           // [nq]
};
hipError_t launch_rerank(const RerankParams& p, int metric, hipStream_t s);
// diagnostic (tests): one batch merge of a sorted list [size] with up to 64 survivor keys (~0 = none)
hipError_t launch_debug_merge(int regs, const uint64_t* entries, int size, const uint64_t* surv, int ef, uint64_t* out,
                              int* out_size, hipStream_t s);

// MLP projection (support_func.h:624-658).  W is repacked by the host: [dout x wstride] weights
// (zero padded, wstride multiple of 8) + separate bias[dout].
struct LayerParams {
    const float* x;          // [nq x xstride]
    uint32_t xstride;
    const float* w;          // [dout x wstride]
    uint32_t wstride;
    const float* bias;       // [dout]
    float* out;              // [nq x ostride]
    uint32_t ostride;
    uint32_t nq, din, dout;
    int32_t relu;
    int32_t normalize;       // 1: follow the layer by normalizeVector over its dout outputs (fused when dout <= 64)
    int32_t small_footprint; // 1: hidden layers on the 60-register / 9 KB kernel (batches in flight: its blocks fit beside walk wavefronts)
};
hipError_t launch_mlp_layer(const LayerParams& p, hipStream_t s);
// the throughput option (GBNNS_FLAG_MFMA_PROJECTION): the same layer as a v_mfma_f32_32x32x2_f32 GEMM -- NOT bit-exact
hipError_t launch_mlp_layer_mfma(const LayerParams& p, hipStream_t s);
// The whole three-layer net in one launch (mlp_net.hip): activations stay in LDS, one block per 16 .. 40 queries.  Same
// arithmetic as three launch_mlp_layer calls (relu, relu, normalize), bit for bit.
struct NetLaunch {
    const float* x;          // [nq x xstride]
    uint32_t xstride, nq;
    const float* w[3];       // [dout x wstride], rows zero padded
    uint32_t wstride[3];
    const float* bias[3];
    uint32_t din[3], dout[3];
    float* out;              // [nq x ostride]; columns [dout[2], ostride) are written as zero
    uint32_t ostride;
    int32_t cus;             // compute units of the device (0: 256)
    int32_t force_a;         // diagnostic: queries per lane group (2 .. 5; 0 = chosen by the launcher)
    unsigned long long* stamps;  // diagnostic builds (-DGBNNS_NET_STAMPS): [blocks x 8] 100 MHz timestamps of the phase ends
};
bool mlp_net_serves(const NetLaunch& n);   // shape, alignment and batch size fit the one-launch kernel
hipError_t launch_mlp_net(const NetLaunch& n, hipStream_t s);
// The throughput option as ONE launch (mlp_mfma_net.hip): v_mfma_f32_16x16x4_f32, one workgroup per CU over its share of the batch, weights
// repacked once per handle into the instruction's B-operand order, activations in LDS.  NOT bit-exact.
bool mlp_mfma_net_serves(uint32_t d, uint32_t d_hidden, uint32_t d_low);
size_t mlp_mfma_net_packed_floats(uint32_t d, uint32_t d_hidden, uint32_t d_low);
hipError_t launch_mlp_mfma_pack(const NetLaunch& n, float* packed, hipStream_t s);
hipError_t launch_mlp_mfma_net(const NetLaunch& n, const float* packed, hipStream_t s);
// y [nq x stride]: y /= sqrt(L2(y, 0)) over dim (4-lane order, d%4 tail ignored in the norm),
// pad columns [dim, stride) are written as zero.
hipError_t launch_normalize(float* y, uint32_t stride, uint32_t dim, uint32_t nq, hipStream_t s);
// One wide layer of a small batch with the one-launch kernel's inner loop (mlp_net.hip, mlp_slab_kernel): 8 A queries x 128 neurons
// per workgroup, inputs a slab of 256 at a time.
bool mlp_slab_serves(const LayerParams& p);
bool mlp_slab_wins(const LayerParams& p, int cus);  // ... and the layer is one round of the machine in its workgroup shape
hipError_t launch_mlp_slab(const LayerParams& p, int cus, hipStream_t s, int force_a = 0);

// Exact brute-force kNN scan (knn.hip): getTruth (support_func.h:270-290) generalised to k results per query.
struct KnnParams {
    const float* base;       // [n x bstride]
    uint32_t bstride;        // floats
    uint64_t n;
    const float* q;          // [nq x qstride]
    uint32_t qstride;
    uint32_t nq;
    uint32_t dim;            // <= 128
    int32_t k;
    int64_t self_offset;     // >= 0: query i is base row i + self_offset and is not its own neighbour; -1: off
    uint64_t* heap;          // workspace [k x heap_stride], pre-filled with all-ones
    size_t heap_stride;      // >= nq
    uint32_t* out_ids;       // [nq x k] ascending (distance, id); 0xFFFFFFFF where fewer than k rows exist
    float* out_dist;         // optional [nq x k]
    uint64_t row0;           // id of `base` row 0 (scans of a slice of the base set; 0 = the whole set)
    int32_t keep_heap;       // 1: leave the heaps as they are (no sort, no outputs): more rows follow
};
hipError_t launch_knn_scan(const KnnParams& p, int metric, hipStream_t s);

// Matrix-core filter in front of the exact scan (knn.hip, round 4; L2 metric, d % 4 == 0, d <= 128).
// Rows packed for v_mfma_f32_32x32x16_bf16: per row, per group of 8 dims, 8 bf16 "hi" parts then 8 bf16 "lo" parts
// (x = hi + lo + r, |r| <= 2^-16 |x|), dims zero-padded to a multiple of 16; and the row's squared norm.
hipError_t launch_knn_pack(const float* x, uint32_t stride, uint32_t dim, uint64_t rows, uint16_t* packed, float* norms, hipStream_t s);
struct KnnFilterParams {
    const uint16_t* qpack;   // [nq x 2 dp] packed queries
    const uint16_t* bpack;   // [rows x 2 dp] packed base rows of this chunk
    const float* bnorm;      // [rows] squared norms of the chunk's rows
    const float* rhs;        // [nq] 0.5 ((1 - c) |q|^2 - T_q): a row passes when  q.x - 0.5 (1 - c) |x|^2 >= rhs  (-inf: everything passes)
    uint32_t nq, dp;         // dp = padded dimension (multiple of 16)
    uint32_t rows;           // rows of the chunk
    uint32_t row0;           // id of the chunk's first row
    uint32_t cap;            // candidate slots per query
    uint32_t* cand;          // [nq x cap] ids that passed
    uint32_t* count;         // [nq] how many passed (may exceed cap: then *overflow is set and the chunk is scanned exactly)
    uint32_t* overflow;      // [1]
};
hipError_t launch_knn_filter(const KnnFilterParams& p, hipStream_t s);
// thresholds from the heaps: rhs[q] = 0.5 ((1 - c) |q|^2 - d_k(q)), -inf while the heap is not full
hipError_t launch_knn_thresholds(const uint64_t* heap, size_t heap_stride, int k, const float* qnorm, uint32_t nq, float* rhs, hipStream_t s);
// exact distances (the reference's arithmetic) of the candidates, offered to the heaps; then the new thresholds
struct KnnRescoreParams {
    KnnParams k;             // base = the WHOLE base set, heap, q, ...
    const uint32_t* cand;
    const uint32_t* count;
    uint32_t cap;
    const float* qnorm;
    float* rhs;
};
hipError_t launch_knn_rescore(const KnnRescoreParams& p, hipStream_t s);
hipError_t launch_knn_finalize(const KnnParams& p, hipStream_t s);   // heap sort + outputs
// long lists (64 <= k <= 4 096): the k smallest keys of a query as an unordered pool, one wavefront per query and chunk (knn.hip)
struct KnnPoolParams {
    KnnParams k;             // base = the WHOLE base set, q, dims, self_offset, outputs (heap unused)
    uint64_t* pool;          // [nq x k] keys, all-ones = no entry
    uint64_t* root;          // [nq] largest key of the pool (all-ones while it is not full)
    const uint32_t* cand;    // [nq x cap] rows the filter kept
    const uint32_t* count;   // [nq]
    uint32_t cap;
    const float* qnorm;
    float* rhs;              // [nq] the filter's thresholds, rewritten from `root`
};
hipError_t launch_knn_pool_update(const KnnPoolParams& p, hipStream_t s);
hipError_t launch_knn_pool_finalize(const KnnPoolParams& p, hipStream_t s);   // ascending order + outputs
constexpr float kKnnFilterSlack = 1.0f / 4096.0f;  // c: |approximate - reference distance| <= c (|q|^2 + |x|^2) with room to spare (knn.hip)

// GD pruning of a kNN graph, one node per wavefront (support_func.h:521-563).  deg[i] = 0xFFFFFFFF marks a node
// left to the host (equal distances in its list, list longer than 1024, id out of range).
struct GdParams {
    const float* ds;         // [n x dstride], rows zero-padded to a multiple of 4 floats
    uint32_t dstride, dim;
    uint32_t n;
    int32_t M;               // 2 .. 64
    const uint64_t* knn_off; // [n + 1]
    const uint32_t* knn_nbr;
    uint32_t* adj;           // [n x 2M]
    uint32_t* deg;           // [n]
};
hipError_t launch_gd_prune(const GdParams& p, int metric, hipStream_t s);

// Locality order of a batch (counting sort on the sign bits of the first `bits` = 10 .. 16 walked-space coordinates;
// dim >= bits): hist [2^bits] u32 scratch, order [nq] u32 out.
hipError_t launch_query_order(const float* q, uint32_t qstride, uint32_t dim, uint32_t nq, uint32_t bits, uint32_t* hist, uint32_t* order,
                              hipStream_t s);

// helpers
hipError_t launch_fill_u32(uint32_t* p, uint32_t v, size_t count, hipStream_t s);

}  // namespace gbnns
