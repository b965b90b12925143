// api.cpp -- host side of libgbnns_hip.so: the C ABI declared in include/gbnns.h.
//
// Owns device memory (index data in HBM, growable per-index workspaces: "lanes"), converts the reference's
// host-side data structures to the device layouts, and sequences the kernels of one batch call on
// one HIP stream -- the caller's, or with GBNNS_FLAG_DEFER_JOIN a lane's own, several batches in flight:
//     [MLP layer x3 + normalise] -> walk (first pass, fused re-rank) [-> retry pass] -> walk (general kernel,
//     hand-over list) [-> re-rank]
// There is no CPU fallback anywhere in this file: if HIP is unusable every entry point fails.

#include "../../include/gbnns.h"
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cxxabi.h>
#include <deque>
#include <map>
#include <new>
#include <mutex>
#include <set>
#include <string>
#include <vector>

using namespace gbnns;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(e_ == hipErrorOutOfMemory ? GBNNS_ERR_OOM : GBNNS_ERR_HIP,           \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,     \
                        __LINE__);                                                           \
    } while (0)

inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return GBNNS_OK;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            bytes = 0;
            if (e != hipSuccess) return fail(GBNNS_ERR_HIP, "hipFree: %s", hipGetErrorString(e));
        }
        const size_t want = need + need / 8;  // slack so slightly larger batches do not realloc
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(GBNNS_ERR_OOM, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
        }
        bytes = want;
        return GBNNS_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T>
    T* as() const {
        return static_cast<T*>(p);
    }
};

struct ProfCall {
    hipEvent_t ev[5];  // begin, after project, after walk, after general, after rerank
    uint64_t queries;
    bool has_project, has_rerank;
};

constexpr int kMaxLanes = 4;

// GBNNS_SLOW_US=<microseconds>: a search call whose HOST side (enqueueing; no waiting in a deferred call) takes longer
// is reported on stderr with the time spent before each checkpoint.  Diagnostic; off by default.
struct SlowLog {
    using clock = std::chrono::steady_clock;
    long limit_us;
    clock::time_point t0;
    int n = 0;
    const char* name[24];
    long us[24];
    SlowLog() : limit_us(getenv("GBNNS_SLOW_US") ? atol(getenv("GBNNS_SLOW_US")) : 0) {}
    void start() { if (limit_us) { t0 = clock::now(); n = 0; } }
    void mark(const char* what) {
        if (limit_us && n < 24) { name[n] = what; us[n++] = (long)std::chrono::duration_cast<std::chrono::microseconds>(clock::now() - t0).count(); }
    }
    void finish() {
        if (!limit_us || n == 0 || us[n - 1] < limit_us) return;
        std::fprintf(stderr, "gbnns slow call:");
        for (int i = 0; i < n; ++i) std::fprintf(stderr, " %s@%ld", name[i], us[i]);
        std::fprintf(stderr, " us\n");
    }
};
thread_local SlowLog g_slow;

// One workspace of per-batch buffers + control words.  A handle has several so that consecutive batches can be
// in flight side by side on internal streams (the tail of one batch's walk -- a 10 k batch is < 2 "rounds" of
// resident wavefronts -- then runs beside the projection and the first round of the next one).
struct Lane {
    hipStream_t stream = nullptr;      // internal stream (created on the lane's first deferred call)
    hipEvent_t done_ev = nullptr;      // recorded after the lane's batch of a deferred call
    hipEvent_t prev_ev = nullptr;      // ... and the one of the lane's batch before (the two alternate)
    uint64_t ticket = 0, prev_ticket = 0;  // serial numbers of those two batches (0 = none), for gbnns_index_wait
    DevBuf q_in, q_low, h1, h2, cand, cand_dist, cnt, hops, dc, edges, out, entries, ovf_list, ovf2_list, ctrl;
    DevBuf g_bitmap, g_keys, fp_bitmap, order, order_hist;
    // visited-set sizing feedback: stats of an earlier call arrive asynchronously in pinned memory
    uint32_t* h_stats = nullptr;       // [4] copy of ctrl after the walk kernels
    hipEvent_t stats_ev = nullptr;
    bool stats_pending = false;
    int stats_ef = 0;
    uint32_t stats_cap = 0;
    // which of the two control-word blocks the next call uses, and whether each is known to be zero
    int ctrl_phase = 0;
    bool ctrl_clean[2] = {true, true};
    bool ctrl_ready = false;
    uint32_t last_general = 0;
    DevBuf* bufs(int i) {
        DevBuf* b[] = {&q_in, &q_low, &h1, &h2, &cand, &cand_dist, &cnt, &hops, &dc, &edges, &out, &entries,
                       &ovf_list, &ovf2_list, &ctrl, &g_bitmap, &g_keys, &fp_bitmap, &order, &order_hist};
        return i < (int)(sizeof b / sizeof b[0]) ? b[i] : nullptr;
    }
};

}  // namespace

struct gbnns_index {
    int device = 0;
    int metric = 0;
    uint64_t n = 0;
    uint32_t d = 0, d_low = 0, d_hidden = 0;
    uint32_t d_pad = 0, dl_pad = 0;
    const float* db = nullptr;      // [n x d_pad]
    const float* db_low = nullptr;  // [n x dl_pad]
    DevBuf db_own, db_low_own, ell, net, aux_ell;
    uint32_t ell_stride = 0, aux_stride = 0;
    bool has_aux = false;
    bool has_net = false;
    float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr, *w3 = nullptr, *b3 = nullptr;
    uint32_t ws1 = 0, ws2 = 0, ws3 = 0;
    int cus = 0;                    // compute units of the device (sizes the one-launch projection's query strips)
    // workspaces: lane 0 serves plain calls on the caller's stream; the batches of deferred calls rotate over
    // lanes 0 .. n_lanes-1, each on its own internal stream (see gbnns_search_ex)
    Lane lanes[kMaxLanes];
    hipEvent_t fork_ev = nullptr;      // caller's stream -> lanes
    // profiling
    bool profiling = false;
    std::vector<ProfCall> pending;
    gbnns_profile acc{};
    // visited-set sizing feedback, shared by the lanes (host-side bookkeeping; the statistics of a call arrive
    // asynchronously in the lane's pinned block)
    std::map<int, uint32_t> cap_for_ef;
    std::map<int, uint32_t> maxdc_for_ef;  // largest dist_calc seen per (ef, mode, aux, wide): the raw figure behind cap_for_ef
    std::map<int, int> calm_streak;   // per (ef, mode): consecutive observed batches without hand-over / resize
    uint32_t stats_tick = 0;
    std::deque<std::pair<hipEvent_t, hipStream_t>> joins;  // (batch's event, caller's stream) of the deferred calls not yet joined, oldest first
    int next_lane = 0;
    uint64_t issued = 0;               // deferred calls so far
    // stream of the last call that left work in flight (the workspace and the control words are ordered by
    // stream order only: a call on another stream first waits for that work, see enter_stream)
    hipStream_t last_stream = nullptr;
    bool in_flight = false;
    hipEvent_t order_ev = nullptr;
};

namespace {

// Host -> device copy of a caller's (pageable) array through the library's own page-locked staging buffer, in pieces.
// A plain hipMemcpy from pageable memory lets the runtime page-lock the caller's pages on the fly and remember the
// mapping; a caller that frees such an array and gets the same addresses back from its allocator for another one (numpy,
// std::vector) then has the next copy fault on the stale mapping now and then (round 4: "an illegal memory access" inside
// the first upload of gbnns_index_create, about one full test-suite run in three; never in a short run).  One-time
// uploads of an index do not miss the extra pass over host memory.
int h2d_staged(void* dst, const void* src, size_t bytes) {
    static std::mutex mu;
    static void* stage = nullptr;
    constexpr size_t kPiece = 8u << 20;
    std::lock_guard<std::mutex> lk(mu);
    if (!stage) HIP_TRY(hipHostMalloc(&stage, 2 * kPiece, hipHostMallocDefault));
    hipEvent_t ev[2] = {nullptr, nullptr};
    HIP_TRY(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    int rc = GBNNS_OK;
    size_t done = 0;
    for (int i = 0; done < bytes && rc == GBNNS_OK; ++i, done += kPiece) {
        const size_t nb = std::min(kPiece, bytes - done);
        char* half = static_cast<char*>(stage) + (size_t)(i & 1) * kPiece;
        hipError_t e = i >= 2 ? hipEventSynchronize(ev[i & 1]) : hipSuccess;  // the piece that used this half has left it
        if (e == hipSuccess) {
            std::memcpy(half, static_cast<const char*>(src) + done, nb);
            e = hipMemcpyAsync(static_cast<char*>(dst) + done, half, nb, hipMemcpyHostToDevice, nullptr);
        }
        if (e == hipSuccess) e = hipEventRecord(ev[i & 1], nullptr);
        if (e != hipSuccess) rc = fail(GBNNS_ERR_HIP, "staged upload: %s", hipGetErrorString(e));
    }
    if (rc == GBNNS_OK && hipStreamSynchronize(nullptr) != hipSuccess) rc = fail(GBNNS_ERR_HIP, "staged upload: %s", hipGetErrorString(hipGetLastError()));
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
    return rc;
}

int upload(DevBuf& dst, const void* src, size_t rows, size_t row_floats, size_t pad_floats,
           int mem_kind) {
    // copies a [rows x row_floats] f32 matrix into a zero-padded [rows x pad_floats] device matrix
    const size_t bytes = rows * pad_floats * sizeof(float);
    int rc = dst.ensure(bytes ? bytes : 4);
    if (rc) return rc;
    if (mem_kind == GBNNS_MEM_DEVICE) {
        if (pad_floats == row_floats) {
            HIP_TRY(hipMemcpy(dst.p, src, bytes, hipMemcpyDeviceToDevice));
        } else {
            HIP_TRY(hipMemset(dst.p, 0, bytes));
            HIP_TRY(hipMemcpy2D(dst.p, pad_floats * 4, src, row_floats * 4, row_floats * 4, rows, hipMemcpyDeviceToDevice));
        }
        return GBNNS_OK;
    }
    if (pad_floats == row_floats) return h2d_staged(dst.p, src, bytes);
    // padded rows: the rows are packed into a temporary device matrix first, then spread on the device
    DevBuf packed;
    if ((rc = packed.ensure(rows * row_floats * 4 ? rows * row_floats * 4 : 4))) return rc;
    rc = h2d_staged(packed.p, src, rows * row_floats * 4);
    hipError_t e = rc ? hipSuccess : hipMemset(dst.p, 0, bytes);
    if (!rc && e == hipSuccess) e = hipMemcpy2D(dst.p, pad_floats * 4, packed.p, row_floats * 4, row_floats * 4, rows, hipMemcpyDeviceToDevice);
    if (!rc && e == hipSuccess) e = hipDeviceSynchronize();
    packed.release();
    if (!rc && e != hipSuccess) rc = fail(GBNNS_ERR_HIP, "padded upload: %s", hipGetErrorString(e));
    return rc;
}

// [dout x (din+1)] rows = [W | b]  ->  W [dout x wstride] (zero padded) followed by bias [dout]
void repack_layer(const float* layer, uint32_t din, uint32_t dout, uint32_t wstride,
                  std::vector<float>& out) {
    const size_t base = out.size();
    out.resize(base + (size_t)dout * wstride + round_up(dout, 4), 0.f);  // keeps the next layer 16-B aligned
    float* w = out.data() + base;
    float* b = w + (size_t)dout * wstride;
    for (uint32_t o = 0; o < dout; ++o) {
        const float* row = layer + (size_t)o * (din + 1);
        std::memcpy(w + (size_t)o * wstride, row, (size_t)din * sizeof(float));
        b[o] = row[din];
    }
}

// CSR (host) -> padded adjacency.  Order inside each list is preserved.  A neighbour id that
// repeats inside one list is dropped after its first occurrence: the reference would find it
// already visited (search_function.h:25), so this changes nothing observable.
int build_ell(const uint64_t* off, const uint32_t* nbr, uint64_t n, std::vector<uint32_t>& ell,
              uint32_t& stride) {
    uint64_t maxdeg = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (off[i + 1] < off[i]) return fail(GBNNS_ERR_INVALID, "graph_offsets not monotone at %llu",
                                             (unsigned long long)i);
        maxdeg = std::max<uint64_t>(maxdeg, off[i + 1] - off[i]);
    }
    if (maxdeg > (1u << 20)) return fail(GBNNS_ERR_UNSUPPORTED, "max degree %llu too large",
                                         (unsigned long long)maxdeg);
    stride = round_up((uint32_t)std::max<uint64_t>(maxdeg, 1), 16);
    if ((double)n * stride * 4.0 > 200e9)
        return fail(GBNNS_ERR_UNSUPPORTED, "padded adjacency would need %.1f GB", n * stride * 4e-9);
    ell.assign((size_t)n * stride, kInvalidId);
    std::vector<uint32_t> tmp;
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t* src = nbr + off[i];
        const uint32_t deg = (uint32_t)(off[i + 1] - off[i]);
        uint32_t* dst = ell.data() + (size_t)i * stride;
        bool dup = false;
        for (uint32_t j = 0; j < deg; ++j) {
            if (src[j] >= n) return fail(GBNNS_ERR_INVALID, "node %llu: neighbour id %u >= n",
                                         (unsigned long long)i, src[j]);
        }
        if (deg > 1) {
            tmp.assign(src, src + deg);
            std::sort(tmp.begin(), tmp.end());
            dup = std::adjacent_find(tmp.begin(), tmp.end()) != tmp.end();
        }
        if (!dup) {
            std::memcpy(dst, src, (size_t)deg * 4);
        } else {
            uint32_t m = 0;
            for (uint32_t j = 0; j < deg; ++j) {
                bool seen = false;
                for (uint32_t l = 0; l < m && !seen; ++l) seen = dst[l] == src[j];
                if (!seen) dst[m++] = src[j];
            }
        }
    }
    return GBNNS_OK;
}

constexpr size_t kMaxLds = 160 * 1024;
// LDS is handed out in granules of 1 280 bytes (measured, tools/ubench/occupancy_census.hip: one-wavefront workgroups of
// 5 120 B -> 32 per CU, 5 121 .. 6 400 B -> 25, 6 401 .. 7 680 B -> 21): a wavefront's share is a multiple of it.
#ifndef GBNNS_LDS_GRAN
#define GBNNS_LDS_GRAN 1280
#endif
constexpr size_t kLdsGran = GBNNS_LDS_GRAN;

// Diagnostic knobs (gbnns_debug_knob; the environment gives their initial values, read once when the library loads):
// "quotient" 0 = never the quotient form of the visited set (GBNNS_QUOTIENT); "vs_disp" = probe number at which a probe
// sequence of that form gives up, 1 .. 15 (GBNNS_DEBUG_VS_DISP; 15 = the product's).
int knob_env(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
std::atomic<int> g_knob_quotient{knob_env("GBNNS_QUOTIENT", 1)};
std::atomic<int> g_knob_vs_disp{[] { const int v = knob_env("GBNNS_DEBUG_VS_DISP", 15); return v <= 0 ? 15 : v; }()};
// "max_waves" = most first-pass wavefronts per CU the LDS shares are cut for (GBNNS_MAX_WAVES; 0 = the per-kernel defaults)
std::atomic<int> g_knob_max_waves{std::max(0, std::min(32, knob_env("GBNNS_MAX_WAVES", 0)))};
// "spec_min_nq" = smallest batch whose ef <= 64 first pass requests the rows before the visited test (walk_hot_spec_kernel;
// GBNNS_SPEC_MIN_NQ; 0 = never)
std::atomic<int> g_knob_spec_min_nq{std::max(0, knob_env("GBNNS_SPEC_MIN_NQ", 32768))};
// "spec_any_form" = 1: ... whatever the form of the visited set (tests; default: only tables NOT in the quotient form)
std::atomic<int> g_knob_spec_any_form{knob_env("GBNNS_SPEC_ANY_FORM", 0)};
// "knn_pool_min_k" = shortest list gbnns_exact_knn keeps as an unordered pool (one wavefront per query and chunk) instead of a heap
// (measured on 10^6 x 32: k = 48 0.327 against 0.333 s, k = 100 0.425 against 0.536 s, k = 1 000 2.4 against 7.6 s)
// "mlp_small" = smallest batch IN FLIGHT whose hidden projection layers run on the small-footprint kernel (0 = never)
std::atomic<int> g_knob_mlp_small{std::max(0, knob_env("GBNNS_MLP_SMALL", 4096))};
// "mlp_net" 0 = never the one-launch projection (mlp_net.hip), 1 = for the shapes and batch sizes it serves (GBNNS_MLP_NET)
std::atomic<int> g_knob_mlp_net{knob_env("GBNNS_MLP_NET", 1)};
// "spec_tail" = largest partial last round of a lone launch, in percent of the device's wavefront slots, whose wavefronts
// request their rows before the visited test (0 = off)
std::atomic<int> g_knob_spec_tail{std::max(0, std::min(100, knob_env("GBNNS_SPEC_TAIL", 50)))};
// "knn_chunk" = most rows per filtered chunk (a multiple of 64)
std::atomic<int> g_knob_knn_chunk{std::max(64, knob_env("GBNNS_KNN_CHUNK", 1 << 15) & ~63)};  // (a multiple of 64, never 0: the chunk loops step by it)
std::atomic<int> g_knob_knn_pool_min_k{std::max(1, knob_env("GBNNS_KNN_POOL_MIN_K", 64))};
// "knn_filter" 0 = gbnns_exact_knn without the matrix-core filter (GBNNS_KNN_FILTER; tests compare the two paths)
std::atomic<int> g_knob_knn_filter{knob_env("GBNNS_KNN_FILTER", 1)};

// A handle's workspace (projected queries, candidate lists, hand-over lists, control words) is shared by its
// calls and ordered by stream order.  When a call names another stream than the last one that left work in
// flight, the new stream first waits for that work (an event recorded on the old stream now covers everything
// enqueued there so far).  Should the old stream be gone, the device is synchronised instead.
int flush_join(gbnns_index* ix);

int enter_stream(gbnns_index* ix, hipStream_t s) {
    int rc = flush_join(ix);  // deferred calls not yet joined: their streams first wait for their lanes
    if (rc) return rc;
    if (ix->in_flight && ix->last_stream != s) {
        hipError_t e = hipSuccess;
        if (!ix->order_ev) e = hipEventCreateWithFlags(&ix->order_ev, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(ix->order_ev, ix->last_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(s, ix->order_ev, 0);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            HIP_TRY(hipDeviceSynchronize());
        }
    }
    ix->last_stream = s;
    ix->in_flight = true;
    return GBNNS_OK;
}

}  // namespace

extern "C" {

int gbnns_version(void) { return GBNNS_VERSION; }

const char* gbnns_last_error(void) { return g_err.c_str(); }

int gbnns_exact_knn(int device, const float* base, uint64_t n, const float* queries, uint64_t n_q,
                    uint32_t d, int k, int metric, int64_t self_offset, uint32_t* out_ids, float* out_dist,
                    int mem_kind, void* stream) {
    if (!base || !queries || !out_ids) return fail(GBNNS_ERR_INVALID, "null argument");
    if (n == 0 || n >= (1ull << 32) - 1) return fail(GBNNS_ERR_INVALID, "n must be in [1, 2^32 - 1)");
    if (n_q >= (1ull << 31)) return fail(GBNNS_ERR_INVALID, "n_q too large");
    if (k < 1 || k > (1 << 20)) return fail(GBNNS_ERR_INVALID, "k must be in [1, 2^20]");
    if (d == 0) return fail(GBNNS_ERR_INVALID, "d must be >= 1");
    if (metric != GBNNS_METRIC_L2 && metric != GBNNS_METRIC_NEG_DOT) return fail(GBNNS_ERR_INVALID, "unknown metric %d", metric);
    if (mem_kind != GBNNS_MEM_HOST && mem_kind != GBNNS_MEM_DEVICE) return fail(GBNNS_ERR_INVALID, "unknown mem_kind %d", mem_kind);
    if (d > 8192) return fail(GBNNS_ERR_UNSUPPORTED, "gbnns_exact_knn: d <= 8192 (a tile of base rows is staged in LDS; got %u)", d);
    if (metric == GBNNS_METRIC_NEG_DOT && d % 8 != 0)
        return fail(GBNNS_ERR_UNSUPPORTED, "gbnns_exact_knn: the negative-dot form needs d %% 8 == 0 (got %u)", d);
    if (self_offset < -1) return fail(GBNNS_ERR_INVALID, "self_offset must be >= -1");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(GBNNS_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= count) return fail(GBNNS_ERR_NO_DEVICE, "device %d out of range (%d devices)", device, count);
    if (n_q == 0) return GBNNS_OK;
    HIP_TRY(hipSetDevice(device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool host = mem_kind == GBNNS_MEM_HOST;
    const uint32_t nq = (uint32_t)n_q;
    DevBuf b_dev, q_dev, ids_dev, dist_dev, heap;
    struct Release {  // DevBuf has no destructor (index members are released by gbnns_index_destroy)
        DevBuf* b[5];
        ~Release() { for (DevBuf* x : b) x->release(); }
    } release{{&b_dev, &q_dev, &ids_dev, &dist_dev, &heap}};
    int rc;
    KnnParams p{};
    p.base = base; p.q = queries; p.out_ids = out_ids; p.out_dist = out_dist;
    if (host) {
        if ((rc = b_dev.ensure((size_t)n * d * 4))) return rc;
        if ((rc = q_dev.ensure((size_t)nq * d * 4))) return rc;
        if ((rc = ids_dev.ensure((size_t)nq * k * 4))) return rc;
        if (out_dist && (rc = dist_dev.ensure((size_t)nq * k * 4))) return rc;
        HIP_TRY(hipMemcpyAsync(b_dev.p, base, (size_t)n * d * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(q_dev.p, queries, (size_t)nq * d * 4, hipMemcpyHostToDevice, s));
        p.base = b_dev.as<float>(); p.q = q_dev.as<float>(); p.out_ids = ids_dev.as<uint32_t>();
        p.out_dist = out_dist ? dist_dev.as<float>() : nullptr;
    }
    p.bstride = d; p.qstride = d; p.n = n; p.nq = nq; p.dim = d; p.k = k; p.self_offset = self_offset;
    p.heap_stride = ((size_t)nq + 63) & ~(size_t)63;
    // Matrix-core filter in front of the exact distances (knn.hip): the L2 metric on rows of whole 16-byte steps, sets
    // large enough to amortise its passes.  Same output, byte for byte (tests/test_gpu_parity.py); "knn_filter" 0 = off.
    const int knob_filter = g_knob_knn_filter.load(std::memory_order_relaxed);  // 0 = never, 1 = by size, 2 = whenever the shape allows (tests)
    const bool filter_shape = metric == GBNNS_METRIC_L2 && d % 4 == 0 && d <= 128 && n > (uint64_t)4 * k &&
                              (knob_filter == 2 || (knob_filter == 1 && n >= (1u << 17) && nq >= 2048));
    // long lists keep a query's keys as an unordered pool instead of a heap (knn.hip, knn_pool_update_kernel)
    const bool pool_path = filter_shape && k >= g_knob_knn_pool_min_k.load(std::memory_order_relaxed) && k <= 4096;
    const bool filter = filter_shape && k <= 512 && !pool_path;
    if (!pool_path) {  // (the pool path keeps its keys in slabs of its own: at k = 1 000 and 10^6 queries the heaps would be 9 GB)
        if ((rc = heap.ensure(p.heap_stride * (size_t)k * 8))) return rc;
        p.heap = heap.as<uint64_t>();
        HIP_TRY(hipMemsetAsync(p.heap, 0xFF, p.heap_stride * (size_t)k * 8, s));
    }
    if (pool_path) {
        const uint32_t dp = (d + 15u) & ~15u;
        const uint32_t cap = (uint32_t)(4 * k + 64);
        const uint64_t rows_first = cap & ~63u;  // a chunk of at most `cap` rows cannot overflow a query's list: the first chunk (no thresholds yet) and the fallback
        // (chunks four times the heap path's: every chunk costs a query a pass over its whole pool; measured at k = 1 000: 32 K rows 1.98 s, 64 K 1.86 s, 128 K 1.89 s)
        const uint64_t max_chunk = std::min<uint64_t>(4ull * (uint64_t)g_knob_knn_chunk.load(std::memory_order_relaxed), std::max<uint64_t>(64, (n / 16 + 63) & ~(uint64_t)63));
        // queries in slabs whose candidate lists take at most 4 GiB
        const uint32_t slab = (uint32_t)std::min<uint64_t>(nq, std::max<uint64_t>(1024, ((4ull << 30) / ((uint64_t)cap * 4)) & ~(uint64_t)127));
        DevBuf bpack, bnorm, qpack, qnorm, rhs, cand, count, flag, pool, root;
        struct Release3 {
            DevBuf* b[10];
            ~Release3() { for (DevBuf* x : b) x->release(); }
        } release3{{&bpack, &bnorm, &qpack, &qnorm, &rhs, &cand, &count, &flag, &pool, &root}};
        if ((rc = bpack.ensure((size_t)n * dp * 4))) return rc;
        if ((rc = bnorm.ensure(((size_t)n + 64) * 4))) return rc;
        if ((rc = qpack.ensure((size_t)slab * dp * 4))) return rc;
        if ((rc = qnorm.ensure((size_t)slab * 4))) return rc;
        if ((rc = rhs.ensure((size_t)slab * 4))) return rc;
        if ((rc = cand.ensure((size_t)slab * cap * 4))) return rc;
        if ((rc = count.ensure((size_t)slab * 4))) return rc;
        if ((rc = flag.ensure(4))) return rc;
        if ((rc = pool.ensure((size_t)slab * (size_t)k * 8))) return rc;
        if ((rc = root.ensure((size_t)slab * 8))) return rc;
        HIP_TRY(launch_fill_u32(bnorm.as<uint32_t>() + n, 0x7F800000u, 64, s));
        HIP_TRY(launch_knn_pack(p.base, d, d, n, bpack.as<uint16_t>(), bnorm.as<float>(), s));
        for (uint32_t q0 = 0; q0 < nq; q0 += slab) {
            const uint32_t qn = std::min(slab, nq - q0);
            KnnPoolParams pp{};
            pp.k = p;
            pp.k.q = p.q + (size_t)q0 * d; pp.k.nq = qn; pp.k.out_ids = p.out_ids + (size_t)q0 * k;
            pp.k.out_dist = p.out_dist ? p.out_dist + (size_t)q0 * k : nullptr;
            pp.k.self_offset = self_offset >= 0 ? self_offset + (int64_t)q0 : -1;
            pp.pool = pool.as<uint64_t>(); pp.root = root.as<uint64_t>(); pp.cand = cand.as<uint32_t>(); pp.count = count.as<uint32_t>();
            pp.cap = cap; pp.qnorm = qnorm.as<float>(); pp.rhs = rhs.as<float>();
            HIP_TRY(launch_knn_pack(pp.k.q, d, d, qn, qpack.as<uint16_t>(), qnorm.as<float>(), s));
            HIP_TRY(hipMemsetAsync(pool.p, 0xFF, (size_t)qn * (size_t)k * 8, s));
            HIP_TRY(hipMemsetAsync(root.p, 0xFF, (size_t)qn * 8, s));
            HIP_TRY(launch_fill_u32(rhs.as<uint32_t>(), 0xFF800000u, qn, s));  // -inf: every row passes
            uint32_t h_flag = 0;
            auto filter_rows = [&](uint64_t r0, uint32_t rows) -> int {
                HIP_TRY(hipMemsetAsync(count.p, 0, (size_t)qn * 4, s));
                HIP_TRY(hipMemsetAsync(flag.p, 0, 4, s));
                KnnFilterParams f{};
                f.qpack = qpack.as<uint16_t>(); f.bpack = bpack.as<uint16_t>() + (size_t)r0 * dp * 2; f.bnorm = bnorm.as<float>() + r0;
                f.rhs = rhs.as<float>(); f.nq = qn; f.dp = dp; f.rows = rows; f.row0 = (uint32_t)r0; f.cap = cap;
                f.cand = cand.as<uint32_t>(); f.count = count.as<uint32_t>(); f.overflow = flag.as<uint32_t>();
                HIP_TRY(launch_knn_filter(f, s));
                return GBNNS_OK;
            };
            for (uint64_t r0 = 0, chunk = 0; r0 < n; r0 += chunk) {
                chunk = r0 == 0 ? std::min<uint64_t>(n, rows_first) : std::min<uint64_t>(max_chunk, r0);
                const uint32_t rows = (uint32_t)std::min<uint64_t>(chunk, n - r0);
                if ((rc = filter_rows(r0, rows))) return rc;
                HIP_TRY(hipMemcpyAsync(&h_flag, flag.p, 4, hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                if (!h_flag) {
                    HIP_TRY(launch_knn_pool_update(pp, s));
                    continue;
                }
                // some query kept more rows of this chunk than its list holds: the chunk again in pieces that cannot overflow
                for (uint64_t r1 = r0; r1 < r0 + rows; r1 += rows_first) {
                    if ((rc = filter_rows(r1, (uint32_t)std::min<uint64_t>(rows_first, r0 + rows - r1)))) return rc;
                    HIP_TRY(launch_knn_pool_update(pp, s));
                }
            }
            HIP_TRY(launch_knn_pool_finalize(pp, s));
        }
    } else if (filter) {
        const uint32_t dp = (d + 15u) & ~15u;
        const uint32_t cap = (uint32_t)std::max(256, 4 * k + 64);
        // The first rows are scanned exactly (they fill the heaps: thresholds exist afterwards); then filtered chunks, each
        // at most as long as everything before it -- a query is then expected to keep about k rows of a chunk, whatever
        // the chunk -- and at most 32 K rows (4 MB of packed rows at d = 32: L2 / Infinity-Cache resident while swept).
        // Chunks start on multiples of 64 rows (the filter reads the norms in aligned groups of four).
        const uint64_t max_chunk = std::min<uint64_t>((uint64_t)g_knob_knn_chunk.load(std::memory_order_relaxed), std::max<uint64_t>(64, (n / 16 + 63) & ~(uint64_t)63));  // (small sets, tests: a sixteenth)
        const uint64_t first = std::min<uint64_t>(n, (std::max<uint64_t>(std::min<uint64_t>(max_chunk, 8192), 4ull * (uint64_t)k) + 63) & ~(uint64_t)63);
        DevBuf bpack, bnorm, qpack, qnorm, rhs, cand, count, flag;
        struct Release2 {
            DevBuf* b[8];
            ~Release2() { for (DevBuf* x : b) x->release(); }
        } release2{{&bpack, &bnorm, &qpack, &qnorm, &rhs, &cand, &count, &flag}};
        if ((rc = bpack.ensure((size_t)n * dp * 4))) return rc;       // 2 dp bf16 per row
        if ((rc = bnorm.ensure(((size_t)n + 64) * 4))) return rc;  // (+inf behind the last row: the filter reads whole blocks of 32)
        if ((rc = qpack.ensure((size_t)nq * dp * 4))) return rc;
        if ((rc = qnorm.ensure((size_t)nq * 4))) return rc;
        if ((rc = rhs.ensure((size_t)nq * 4))) return rc;
        if ((rc = cand.ensure((size_t)nq * cap * 4))) return rc;
        if ((rc = count.ensure((size_t)nq * 4))) return rc;
        if ((rc = flag.ensure(4))) return rc;
        HIP_TRY(launch_fill_u32(bnorm.as<uint32_t>() + n, 0x7F800000u, 64, s));
        HIP_TRY(launch_knn_pack(p.base, d, d, n, bpack.as<uint16_t>(), bnorm.as<float>(), s));
        HIP_TRY(launch_knn_pack(p.q, d, d, nq, qpack.as<uint16_t>(), qnorm.as<float>(), s));
        KnnParams ps = p;          // the first rows: the exact scan, heaps kept
        ps.n = first; ps.row0 = 0; ps.keep_heap = 1;
        HIP_TRY(launch_knn_scan(ps, metric, s));
        HIP_TRY(launch_knn_thresholds(p.heap, p.heap_stride, k, qnorm.as<float>(), nq, rhs.as<float>(), s));
        uint32_t h_flag = 0;
        for (uint64_t r0 = first, chunk = 0; r0 < n; r0 += chunk) {
            chunk = std::min<uint64_t>(max_chunk, r0);
            const uint32_t rows = (uint32_t)std::min<uint64_t>(chunk, n - r0);
            HIP_TRY(hipMemsetAsync(count.p, 0, (size_t)nq * 4, s));
            HIP_TRY(hipMemsetAsync(flag.p, 0, 4, s));
            KnnFilterParams f{};
            f.qpack = qpack.as<uint16_t>(); f.bpack = bpack.as<uint16_t>() + (size_t)r0 * dp * 2; f.bnorm = bnorm.as<float>() + r0;
            f.rhs = rhs.as<float>(); f.nq = nq; f.dp = dp; f.rows = rows; f.row0 = (uint32_t)r0; f.cap = cap;
            f.cand = cand.as<uint32_t>(); f.count = count.as<uint32_t>(); f.overflow = flag.as<uint32_t>();
            HIP_TRY(launch_knn_filter(f, s));
            HIP_TRY(hipMemcpyAsync(&h_flag, flag.p, 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            if (h_flag) {
                // some query kept more rows of this chunk than its list holds (adversarial order of the rows, or thresholds
                // not yet tight): the chunk is scanned exactly for everyone instead -- same heaps, nothing offered twice
                KnnParams pc = p;
                pc.base = p.base + (size_t)r0 * d; pc.n = rows; pc.row0 = r0; pc.keep_heap = 1;
                HIP_TRY(launch_knn_scan(pc, metric, s));
                HIP_TRY(launch_knn_thresholds(p.heap, p.heap_stride, k, qnorm.as<float>(), nq, rhs.as<float>(), s));
            } else {
                KnnRescoreParams rp{};
                rp.k = p; rp.cand = cand.as<uint32_t>(); rp.count = count.as<uint32_t>(); rp.cap = cap;
                rp.qnorm = qnorm.as<float>(); rp.rhs = rhs.as<float>();
                HIP_TRY(launch_knn_rescore(rp, s));
            }
        }
        HIP_TRY(launch_knn_finalize(p, s));
    } else {
        HIP_TRY(launch_knn_scan(p, metric, s));
    }
    if (host) {
        HIP_TRY(hipMemcpyAsync(out_ids, p.out_ids, (size_t)nq * k * 4, hipMemcpyDeviceToHost, s));
        if (out_dist) HIP_TRY(hipMemcpyAsync(out_dist, p.out_dist, (size_t)nq * k * 4, hipMemcpyDeviceToHost, s));
    }
    // the temporaries are released below: wait for the work that uses them
    HIP_TRY(hipStreamSynchronize(s));
    return GBNNS_OK;
}

// graph_build.cpp
extern "C" int gbnns_internal_gd_finish(const uint64_t* knn_offsets, const uint32_t* knn_nbrs, const float* ds,
                                        uint64_t n, uint32_t d, int M, int metric, int reverse, int threads,
                                        uint32_t* adj, uint32_t* deg, uint64_t* host_nodes, uint64_t** out_offsets,
                                        uint32_t** out_nbrs);

int gbnns_build_graph_gd_device(int device, const uint64_t* knn_offsets, const uint32_t* knn_nbrs, const float* ds,
                                uint64_t n, uint32_t d, int M, int metric, int reverse, int threads,
                                uint64_t** out_offsets, uint32_t** out_nbrs, uint64_t* out_host_nodes) {
    if (!knn_offsets || !knn_nbrs || !ds || !out_offsets || !out_nbrs || M < 2 || n == 0)
        return fail(GBNNS_ERR_INVALID, "gbnns_build_graph_gd_device: bad argument");
    if (n >= (1ull << 31)) return fail(GBNNS_ERR_INVALID, "n must be < 2^31");
    if (metric != GBNNS_METRIC_L2 && metric != GBNNS_METRIC_NEG_DOT) return fail(GBNNS_ERR_INVALID, "unknown metric %d", metric);
    *out_offsets = nullptr;
    *out_nbrs = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(GBNNS_ERR_NO_DEVICE, "no HIP device available (use gbnns_build_graph_gd for the host builder)");
    if (device < 0 || device >= count) return fail(GBNNS_ERR_NO_DEVICE, "device %d out of range (%d devices)", device, count);
    const uint32_t cap = 2u * (uint32_t)M;
    std::vector<uint32_t> adj, deg;
    try {
        adj.resize((size_t)n * cap);
        deg.assign(n, 0xFFFFFFFFu);
    } catch (...) {
        return fail(GBNNS_ERR_OOM, "host allocation failed");
    }
    // shapes the kernel does not take (kept neighbours sit one per lane; the node's vector is staged in LDS) stay
    // on the host entirely -- same result, the host path is the reference's own algorithm
    const bool on_device = M <= 64 && d <= 128;
    if (on_device) {
        HIP_TRY(hipSetDevice(device));
        DevBuf ds_dev, off_dev, nbr_dev, adj_dev, deg_dev;
        struct Release {
            DevBuf* b[5];
            ~Release() { for (DevBuf* x : b) x->release(); }
        } release{{&ds_dev, &off_dev, &nbr_dev, &adj_dev, &deg_dev}};
        const uint64_t total = knn_offsets[n];
        const uint32_t dpad = round_up(d, 4);
        int rc;
        if ((rc = upload(ds_dev, ds, n, d, dpad, GBNNS_MEM_HOST))) return rc;
        if ((rc = off_dev.ensure((n + 1) * 8))) return rc;
        if ((rc = nbr_dev.ensure(std::max<uint64_t>(total, 1) * 4))) return rc;
        if ((rc = adj_dev.ensure((size_t)n * cap * 4))) return rc;
        if ((rc = deg_dev.ensure((size_t)n * 4))) return rc;
        if (int rc2 = h2d_staged(off_dev.p, knn_offsets, (n + 1) * 8)) return rc2;
        if (int rc2 = h2d_staged(nbr_dev.p, knn_nbrs, total * 4)) return rc2;
        GdParams p{};
        p.ds = ds_dev.as<float>(); p.dstride = dpad; p.dim = d; p.n = (uint32_t)n; p.M = M;
        p.knn_off = off_dev.as<uint64_t>(); p.knn_nbr = nbr_dev.as<uint32_t>();
        p.adj = adj_dev.as<uint32_t>(); p.deg = deg_dev.as<uint32_t>();
        HIP_TRY(launch_gd_prune(p, metric, nullptr));
        HIP_TRY(hipMemcpy(adj.data(), adj_dev.p, (size_t)n * cap * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(deg.data(), deg_dev.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    }
    const int rc = gbnns_internal_gd_finish(knn_offsets, knn_nbrs, ds, n, d, M, metric, reverse, threads, adj.data(),
                                            deg.data(), out_host_nodes, out_offsets, out_nbrs);
    if (rc == GBNNS_ERR_INVALID) return fail(rc, "gbnns_build_graph_gd_device: neighbour id out of range");
    if (rc) return fail(rc, "gbnns_build_graph_gd_device: host allocation failed");
    return GBNNS_OK;
}

int gbnns_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

void gbnns_free(void* p) { std::free(p); }

int gbnns_index_create(const gbnns_index_desc* desc, gbnns_index** out) {
    if (!desc || !out) return fail(GBNNS_ERR_INVALID, "null argument");
    *out = nullptr;
    if (desc->struct_size != sizeof(gbnns_index_desc))
        return fail(GBNNS_ERR_INVALID, "gbnns_index_desc.struct_size mismatch (%u != %zu)",
                    desc->struct_size, sizeof(gbnns_index_desc));
    if (desc->n == 0 || desc->n >= (1ull << 31)) return fail(GBNNS_ERR_INVALID, "n must be in [1, 2^31)");
    if (desc->d == 0 || !desc->db) return fail(GBNNS_ERR_INVALID, "db / d missing");
    if (!desc->graph_offsets || !desc->graph_nbrs) return fail(GBNNS_ERR_INVALID, "graph missing");
    if (desc->metric != GBNNS_METRIC_L2 && desc->metric != GBNNS_METRIC_NEG_DOT)
        return fail(GBNNS_ERR_INVALID, "unknown metric %d", desc->metric);
    if (desc->mem_kind != GBNNS_MEM_HOST && desc->mem_kind != GBNNS_MEM_DEVICE)
        return fail(GBNNS_ERR_INVALID, "unknown mem_kind %d", desc->mem_kind);
    if ((desc->d_low != 0) != (desc->db_low != nullptr))
        return fail(GBNNS_ERR_INVALID, "d_low and db_low must be given together");
    const bool has_net = desc->net_l1 || desc->net_l2 || desc->net_l3;
    if (has_net && !(desc->net_l1 && desc->net_l2 && desc->net_l3 && desc->d_hidden && desc->d_low))
        return fail(GBNNS_ERR_INVALID, "net needs all three layers, d_hidden and d_low");

    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(GBNNS_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (desc->device < 0 || desc->device >= count)
        return fail(GBNNS_ERR_NO_DEVICE, "device %d out of range (%d devices)", desc->device, count);
    HIP_TRY(hipSetDevice(desc->device));

    gbnns_index* ix = new (std::nothrow) gbnns_index;
    if (!ix) return fail(GBNNS_ERR_OOM, "host allocation failed");
    ix->device = desc->device;
    ix->metric = desc->metric;
    ix->n = desc->n;
    ix->d = desc->d;
    ix->d_low = desc->d_low;
    ix->d_hidden = desc->d_hidden;
    (void)hipDeviceGetAttribute(&ix->cus, hipDeviceAttributeMultiprocessorCount, ix->device);
    ix->d_pad = round_up(desc->d, 4);
    ix->dl_pad = round_up(desc->d_low, 4);
    int rc = GBNNS_OK;

    auto take = [&](const float* src, uint32_t dim, uint32_t pad, DevBuf& own, const float*& dst) -> int {
        if (desc->mem_kind == GBNNS_MEM_DEVICE && pad == dim) {
            dst = src;  // borrowed: rows already 16-B aligned
            return GBNNS_OK;
        }
        int r = upload(own, src, ix->n, dim, pad, desc->mem_kind);
        dst = own.as<float>();
        return r;
    };
    rc = take(desc->db, ix->d, ix->d_pad, ix->db_own, ix->db);
    if (!rc && desc->db_low) rc = take(desc->db_low, ix->d_low, ix->dl_pad, ix->db_low_own, ix->db_low);

    if (!rc) {
        std::vector<uint32_t> ell;
        rc = build_ell(desc->graph_offsets, desc->graph_nbrs, ix->n, ell, ix->ell_stride);
        if (!rc) rc = ix->ell.ensure(ell.size() * 4);
        if (!rc) {
            rc = h2d_staged(ix->ell.p, ell.data(), ell.size() * 4);
        }
    }

    if (!rc && has_net) {
        const uint32_t d = ix->d, dh = ix->d_hidden, dl = ix->d_low;
        std::vector<float> l1, l2, l3;
        const float *p1 = desc->net_l1, *p2 = desc->net_l2, *p3 = desc->net_l3;
        if (desc->mem_kind == GBNNS_MEM_DEVICE) {
            l1.resize((size_t)dh * (d + 1));
            l2.resize((size_t)dh * (dh + 1));
            l3.resize((size_t)dl * (dh + 1));
            hipError_t e = hipMemcpy(l1.data(), p1, l1.size() * 4, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(l2.data(), p2, l2.size() * 4, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(l3.data(), p3, l3.size() * 4, hipMemcpyDeviceToHost);
            if (e != hipSuccess) rc = fail(GBNNS_ERR_HIP, "net download: %s", hipGetErrorString(e));
            p1 = l1.data();
            p2 = l2.data();
            p3 = l3.data();
        }
        if (!rc) {
            // (rows padded with zeros to 16 floats: the one-launch projection reads whole 16-input blocks)
            ix->ws1 = round_up(d, 16);
            ix->ws2 = round_up(dh, 16);
            ix->ws3 = round_up(dh, 16);

            std::vector<float> packed;
            repack_layer(p1, d, dh, ix->ws1, packed);
            const size_t o2 = packed.size();
            repack_layer(p2, dh, dh, ix->ws2, packed);
            const size_t o3 = packed.size();
            repack_layer(p3, dh, dl, ix->ws3, packed);
            rc = ix->net.ensure(packed.size() * 4);
            if (!rc) {
                hipError_t e = hipMemcpy(ix->net.p, packed.data(), packed.size() * 4, hipMemcpyHostToDevice);
                if (e != hipSuccess) rc = fail(GBNNS_ERR_HIP, "net upload: %s", hipGetErrorString(e));
            }
            float* base = ix->net.as<float>();
            ix->w1 = base;
            ix->b1 = ix->w1 + (size_t)dh * ix->ws1;
            ix->w2 = base + o2;
            ix->b2 = ix->w2 + (size_t)dh * ix->ws2;
            ix->w3 = base + o3;
            ix->b3 = ix->w3 + (size_t)dl * ix->ws3;
            ix->has_net = true;
        }
    }
    if (!rc) rc = ix->lanes[0].ctrl.ensure(512);
    if (!rc) {
        // (hipMemset on device memory may return before the fill has run: callers' streams may be non-blocking
        // ones that do not order themselves after the null stream, so wait for it here)
        hipError_t e = hipMemset(ix->lanes[0].ctrl.p, 0, 512);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) rc = fail(GBNNS_ERR_HIP, "ctrl init: %s", hipGetErrorString(e));
        ix->lanes[0].ctrl_ready = true;
    }
    if (rc) {
        gbnns_index_destroy(ix);
        return rc;
    }
    *out = ix;
    return GBNNS_OK;
}

int gbnns_index_set_aux_graph(gbnns_index* ix, const uint64_t* offsets, const uint32_t* nbrs) {
    if (!ix) return fail(GBNNS_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipDeviceSynchronize());  // no search may still be reading the old table
    ix->has_aux = false;
    if (!offsets && !nbrs) return GBNNS_OK;
    if (!offsets || !nbrs) return fail(GBNNS_ERR_INVALID, "auxiliary graph: offsets / nbrs missing");
    std::vector<uint32_t> ell;
    int rc = build_ell(offsets, nbrs, ix->n, ell, ix->aux_stride);
    if (rc) return rc;
    if ((rc = ix->aux_ell.ensure(ell.size() * 4))) return rc;
    if (int rc2 = h2d_staged(ix->aux_ell.p, ell.data(), ell.size() * 4)) return rc2;
    ix->has_aux = true;
    return GBNNS_OK;
}

uint32_t gbnns_index_d_low(const gbnns_index* ix) { return ix ? ix->d_low : 0u; }
uint64_t gbnns_index_n(const gbnns_index* ix) { return ix ? ix->n : 0u; }
uint32_t gbnns_index_d(const gbnns_index* ix) { return ix ? ix->d : 0u; }
int gbnns_index_device(const gbnns_index* ix) { return ix ? ix->device : -1; }

int gbnns_index_destroy(gbnns_index* ix) {
    if (!ix) return GBNNS_OK;
    (void)hipSetDevice(ix->device);
    for (auto& pc : ix->pending)
        for (auto& e : pc.ev) (void)hipEventDestroy(e);
    (void)hipDeviceSynchronize();  // lanes may still be running a call whose join was deferred
    if (ix->order_ev) (void)hipEventDestroy(ix->order_ev);
    if (ix->fork_ev) (void)hipEventDestroy(ix->fork_ev);
    for (Lane& L : ix->lanes) {
        if (L.stats_ev) (void)hipEventDestroy(L.stats_ev);
        if (L.done_ev) (void)hipEventDestroy(L.done_ev);
        if (L.prev_ev) (void)hipEventDestroy(L.prev_ev);
        if (L.h_stats) (void)hipHostFree(L.h_stats);
        if (L.stream) (void)hipStreamDestroy(L.stream);
        for (int i = 0; DevBuf* b = L.bufs(i); ++i) b->release();
    }
    DevBuf* bufs[] = {&ix->db_own, &ix->db_low_own, &ix->ell, &ix->aux_ell, &ix->net};
    for (DevBuf* b : bufs) b->release();
    delete ix;
    return GBNNS_OK;
}

}  // extern "C"

namespace {

// MLP over rows x [nx x xstride] (device) -> out [nx x dl_pad] (device); h1/h2 are scratch.
int run_project(gbnns_index* ix, Lane& L, const float* x, uint32_t xstride, uint32_t nx, float* out,
                hipStream_t s, bool in_flight = false, bool mfma = false) {
    // the whole net in one launch where it serves (round 5, mlp_net.hip: 0.048 against 0.075 ms on the SIFT shape; the
    // round-2 one-launch form -- csrc/project.hip, deleted in round 4 -- was slower than the three launches)
    if (!mfma && g_knob_mlp_net.load(std::memory_order_relaxed)) {
        NetLaunch n{};
        n.x = x; n.xstride = xstride; n.nq = nx; n.out = out; n.ostride = ix->dl_pad; n.cus = ix->cus;
        n.w[0] = ix->w1; n.w[1] = ix->w2; n.w[2] = ix->w3;
        n.wstride[0] = ix->ws1; n.wstride[1] = ix->ws2; n.wstride[2] = ix->ws3;
        n.bias[0] = ix->b1; n.bias[1] = ix->b2; n.bias[2] = ix->b3;
        n.din[0] = ix->d; n.din[1] = n.din[2] = ix->d_hidden;
        n.dout[0] = n.dout[1] = ix->d_hidden; n.dout[2] = ix->d_low;
        if (mlp_net_serves(n)) {
            HIP_TRY(launch_mlp_net(n, s));
            std::snprintf(ix->acc.project_kernel, sizeof(ix->acc.project_kernel), "mlp_net_kernel");
            return GBNNS_OK;
        }
    }
    std::snprintf(ix->acc.project_kernel, sizeof(ix->acc.project_kernel), mfma ? "mlp_layer_mfma_kernel" : "mlp_layer_kernels");
    int rc = L.h1.ensure((size_t)nx * ix->d_hidden * 4);
    if (!rc) rc = L.h2.ensure((size_t)nx * ix->d_hidden * 4);
    if (rc) return rc;
    LayerParams p{};
    // batches in flight whose walks fill the machine: the small-footprint kernel for the hidden layers (mlp.hip; sift-like
    // +2.5 % at ef 64, +3.6 % at ef 36; a 1 000-query GIST batch -- one walk wavefront per SIMD, nothing to squeeze in
    // beside -- and every batch that runs alone are faster on the big-tile kernel)
    const int small_min = g_knob_mlp_small.load(std::memory_order_relaxed);
    // (... up to 32 times that: a 1 M-query DEEP batch is twenty rounds of the machine on its own, its projection is not
    // waiting for room, and the big-tile kernel's 12 % matter again: 43.1 against 41.5 M queries/s)
    p.small_footprint = (in_flight && small_min > 0 && nx >= (uint32_t)small_min && (uint64_t)nx <= 32ull * (uint64_t)small_min) ? 1 : 0;
    auto layer = [&](const LayerParams& lp) { return mfma ? launch_mlp_layer_mfma(lp, s) : launch_mlp_layer(lp, s); };
    p.x = x; p.xstride = xstride; p.w = ix->w1; p.wstride = ix->ws1; p.bias = ix->b1;
    p.out = L.h1.as<float>(); p.ostride = ix->d_hidden; p.nq = nx; p.din = ix->d;
    p.dout = ix->d_hidden; p.relu = 1;
    HIP_TRY(layer(p));
    p.x = L.h1.as<float>(); p.xstride = ix->d_hidden; p.w = ix->w2; p.wstride = ix->ws2;
    p.bias = ix->b2; p.out = L.h2.as<float>(); p.din = ix->d_hidden;
    HIP_TRY(layer(p));
    p.x = L.h2.as<float>(); p.w = ix->w3; p.wstride = ix->ws3; p.bias = ix->b3; p.out = out;
    p.ostride = ix->dl_pad; p.dout = ix->d_low; p.relu = 0; p.normalize = 1;
    HIP_TRY(layer(p));
    return GBNNS_OK;
}

int prof_flush(gbnns_index* ix) {
    for (auto& pc : ix->pending) {
        HIP_TRY(hipEventSynchronize(pc.ev[4]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, pc.ev[0], pc.ev[1]));
        ix->acc.project_ms += ms;
        HIP_TRY(hipEventElapsedTime(&ms, pc.ev[1], pc.ev[2]));
        ix->acc.walk_ms += ms;
        HIP_TRY(hipEventElapsedTime(&ms, pc.ev[2], pc.ev[3]));
        ix->acc.walk_general_ms += ms;
        HIP_TRY(hipEventElapsedTime(&ms, pc.ev[3], pc.ev[4]));
        ix->acc.rerank_ms += ms;
        HIP_TRY(hipEventElapsedTime(&ms, pc.ev[0], pc.ev[4]));
        ix->acc.total_ms += ms;
        ix->acc.calls += 1;
        ix->acc.queries += pc.queries;
        for (int i = 0; i < 5; ++i) (void)hipEventDestroy(pc.ev[i]);
    }
    ix->pending.clear();
    return GBNNS_OK;
}

}  // namespace

extern "C" {

// Diagnostic (not in gbnns.h): copies the 32 stamp/histogram sums of a GBNNS_STAMPS build and clears them.
int gbnns_debug_read_stamps(gbnns_index* ix, unsigned long long* out32) {
    if (!ix || !out32) return fail(GBNNS_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out32, ix->lanes[0].ctrl.as<uint32_t>() + 8, 256, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(ix->lanes[0].ctrl.as<uint32_t>() + 8, 0, 256));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return GBNNS_OK;
}

// Diagnostic (not in gbnns.h): one batch merge of the register-list walk kernels on host-supplied keys.
// entries: sorted u64 keys [size]; surv: 64 keys, ~0 = no survivor in that lane; out: 64*regs keys;
// out_info: {new size, merged (0 = boundary tie, list untouched), new worst}.
int gbnns_debug_merge(int regs, const unsigned long long* entries, int size, const unsigned long long* surv, int ef,
                      unsigned long long* out, int* out_info) {
    if (!(regs == 1 || regs == 2 || regs == 4) || size < 1 || size > ef || ef > 64 * regs)
        return fail(GBNNS_ERR_INVALID, "bad debug_merge arguments");
    unsigned long long *d_e = nullptr, *d_s = nullptr, *d_o = nullptr;
    int* d_i = nullptr;
    HIP_TRY(hipMalloc(&d_e, 256 * 8));
    HIP_TRY(hipMalloc(&d_s, 64 * 8));
    HIP_TRY(hipMalloc(&d_o, 256 * 8));
    HIP_TRY(hipMalloc(&d_i, 16));
    HIP_TRY(hipMemcpy(d_e, entries, (size_t)size * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_s, surv, 64 * 8, hipMemcpyHostToDevice));
    HIP_TRY(launch_debug_merge(regs, reinterpret_cast<const uint64_t*>(d_e), size, reinterpret_cast<const uint64_t*>(d_s), ef,
                               reinterpret_cast<uint64_t*>(d_o), d_i, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d_o, (size_t)64 * regs * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_info, d_i, 12, hipMemcpyDeviceToHost));
    (void)hipFree(d_e); (void)hipFree(d_s); (void)hipFree(d_o); (void)hipFree(d_i);
    return GBNNS_OK;
}

int gbnns_debug_knob(const char* name, int value) {
    if (!name) return fail(GBNNS_ERR_INVALID, "gbnns_debug_knob: null name");
    if (!std::strcmp(name, "quotient")) g_knob_quotient.store(value, std::memory_order_relaxed);
    else if (!std::strcmp(name, "vs_disp")) g_knob_vs_disp.store(value <= 0 ? 15 : value, std::memory_order_relaxed);
    else if (!std::strcmp(name, "max_waves")) g_knob_max_waves.store(std::max(0, std::min(32, value)), std::memory_order_relaxed);
    else if (!std::strcmp(name, "spec_any_form")) g_knob_spec_any_form.store(value != 0, std::memory_order_relaxed);
    else if (!std::strcmp(name, "spec_min_nq")) g_knob_spec_min_nq.store(std::max(0, value), std::memory_order_relaxed);
    else if (!std::strcmp(name, "mlp_small")) g_knob_mlp_small.store(std::max(0, value), std::memory_order_relaxed);
    else if (!std::strcmp(name, "mlp_net")) g_knob_mlp_net.store(value != 0, std::memory_order_relaxed);
    else if (!std::strcmp(name, "spec_tail")) g_knob_spec_tail.store(std::max(0, std::min(100, value)), std::memory_order_relaxed);
    else if (!std::strcmp(name, "knn_chunk")) g_knob_knn_chunk.store(std::max(64, value & ~63), std::memory_order_relaxed);
    else if (!std::strcmp(name, "knn_pool_min_k")) g_knob_knn_pool_min_k.store(std::max(1, value), std::memory_order_relaxed);
    else if (!std::strcmp(name, "knn_filter")) g_knob_knn_filter.store(value, std::memory_order_relaxed);
    else return fail(GBNNS_ERR_INVALID, "gbnns_debug_knob: unknown knob '%s'", name);
    return GBNNS_OK;
}

int gbnns_profile_enable(gbnns_index* ix, int on) {
    if (!ix) return fail(GBNNS_ERR_INVALID, "null index");
    ix->profiling = on != 0;
    return GBNNS_OK;
}

int gbnns_profile_read(gbnns_index* ix, gbnns_profile* out, int reset) {
    if (!ix || !out) return fail(GBNNS_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ix->device));
    int rc = prof_flush(ix);
    if (rc) return rc;
    uint32_t total = 0;  // ctrl[5]: queries the general kernel has processed since the last reset
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&total, ix->lanes[0].ctrl.as<uint32_t>() + 5, 4, hipMemcpyDeviceToHost));
    ix->acc.general_queries = total;
    ix->acc.struct_size = sizeof(gbnns_profile);
    *out = ix->acc;
    if (reset) {
        ix->acc = gbnns_profile{};
        HIP_TRY(hipMemset(ix->lanes[0].ctrl.as<uint32_t>() + 5, 0, 4));
        HIP_TRY(hipStreamSynchronize(nullptr));  // callers' streams need not order themselves after the null stream
    }
    return GBNNS_OK;
}

int gbnns_project(gbnns_index* ix, const float* x, uint64_t n_x, float* out, int mem_kind,
                  void* stream) {
    if (!ix || !x || !out) return fail(GBNNS_ERR_INVALID, "null argument");
    if (!ix->has_net) return fail(GBNNS_ERR_INVALID, "index has no net");
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint64_t chunk = 1u << 16;
    int rc = enter_stream(ix, s);
    if (rc) return rc;
    Lane& L = ix->lanes[0];
    rc = L.q_low.ensure((size_t)std::min<uint64_t>(chunk, n_x) * ix->dl_pad * 4);
    if (rc) return rc;
    if (mem_kind == GBNNS_MEM_HOST) {
        rc = L.q_in.ensure((size_t)std::min<uint64_t>(chunk, n_x) * ix->d * 4);
        if (rc) return rc;
    }
    for (uint64_t b = 0; b < n_x; b += chunk) {
        const uint32_t m = (uint32_t)std::min<uint64_t>(chunk, n_x - b);
        const float* xin = x + b * ix->d;
        if (mem_kind == GBNNS_MEM_HOST) {
            HIP_TRY(hipMemcpyAsync(L.q_in.p, xin, (size_t)m * ix->d * 4, hipMemcpyHostToDevice, s));
            xin = L.q_in.as<float>();
        }
        float* dst = L.q_low.as<float>();
        const bool direct = mem_kind == GBNNS_MEM_DEVICE && ix->dl_pad == ix->d_low;
        if (direct) dst = out + b * ix->d_low;
        rc = run_project(ix, L, xin, ix->d, m, dst, s);
        if (rc) return rc;
        if (!direct) {
            const hipMemcpyKind kind = mem_kind == GBNNS_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
            HIP_TRY(hipMemcpy2DAsync(out + b * ix->d_low, (size_t)ix->d_low * 4, dst, (size_t)ix->dl_pad * 4,
                                     (size_t)ix->d_low * 4, m, kind, s));
        }
        if (mem_kind == GBNNS_MEM_HOST) HIP_TRY(hipStreamSynchronize(s));
    }
    return GBNNS_OK;
}

int gbnns_rerank(gbnns_index* ix, const float* queries, uint64_t n_q, const uint32_t* cand,
                 uint32_t cand_stride, const int32_t* count, uint32_t* out_ids, int mem_kind,
                 void* stream) {
    if (!ix || !queries || !cand || !out_ids) return fail(GBNNS_ERR_INVALID, "null argument");
    if (n_q == 0) return GBNNS_OK;
    if (cand_stride == 0 || n_q >= (1ull << 31)) return fail(GBNNS_ERR_INVALID, "bad sizes");
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t nq = (uint32_t)n_q;
    const bool host = mem_kind == GBNNS_MEM_HOST;
    int rc;
    if ((rc = enter_stream(ix, s))) return rc;
    Lane& L = ix->lanes[0];
    RerankParams r{};
    r.db = ix->db; r.dstride = ix->d_pad; r.dim = ix->d; r.qstride = ix->d; r.cand_stride = cand_stride;
    r.nq = nq; r.n = (uint32_t)ix->n;
    if ((rc = L.cnt.ensure((size_t)nq * 4))) return rc;
    int32_t* cnt_dev = L.cnt.as<int32_t>();
    if (host) {
        if ((rc = L.q_in.ensure((size_t)nq * ix->d * 4))) return rc;
        if ((rc = L.cand.ensure((size_t)nq * cand_stride * 4))) return rc;
        if ((rc = L.out.ensure((size_t)nq * 4))) return rc;
        for (uint64_t i = 0; i < n_q; ++i) {
            const uint32_t c = count ? (uint32_t)std::max(count[i], 0) : cand_stride;
            if (c > cand_stride) return fail(GBNNS_ERR_INVALID, "count[%llu] > stride", (unsigned long long)i);
            for (uint32_t j = 0; j < c; ++j)
                if (cand[i * cand_stride + j] >= ix->n)
                    return fail(GBNNS_ERR_INVALID, "candidate id %u >= n", cand[i * cand_stride + j]);
        }
        HIP_TRY(hipMemcpyAsync(L.q_in.p, queries, (size_t)nq * ix->d * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(L.cand.p, cand, (size_t)nq * cand_stride * 4, hipMemcpyHostToDevice, s));
        if (count) HIP_TRY(hipMemcpyAsync(cnt_dev, count, (size_t)nq * 4, hipMemcpyHostToDevice, s));
        r.q = L.q_in.as<float>(); r.cand = L.cand.as<uint32_t>(); r.out = L.out.as<uint32_t>();
    } else {
        r.q = queries; r.cand = cand; r.out = out_ids;
        if (count) cnt_dev = const_cast<int32_t*>(count);
    }
    if (!count) HIP_TRY(launch_fill_u32(reinterpret_cast<uint32_t*>(cnt_dev), cand_stride, nq, s));
    r.count = cnt_dev;
    HIP_TRY(launch_rerank(r, ix->metric, s));
    if (host) {
        HIP_TRY(hipMemcpyAsync(out_ids, r.out, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        ix->in_flight = false;
    }
    return GBNNS_OK;
}

}  // extern "C"

namespace {

// Makes lane i usable for deferred calls: its internal stream, its "done" event and its zeroed control words.
int ensure_lane(gbnns_index* ix, int i) {
    Lane& L = ix->lanes[i];
    if (!L.stream) HIP_TRY(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
    if (!L.done_ev) HIP_TRY(hipEventCreateWithFlags(&L.done_ev, hipEventDisableTiming));
    if (!L.prev_ev) HIP_TRY(hipEventCreateWithFlags(&L.prev_ev, hipEventDisableTiming));
    if (!L.ctrl_ready) {
        int rc = L.ctrl.ensure(512);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(L.ctrl.p, 0, 512, L.stream));  // in the lane's stream order: first use follows it
        L.ctrl_ready = true;
    }
    return GBNNS_OK;
}

// How one call is laid out over the handle's lanes (workspace + internal stream each).
//   * default: lane 0 in the caller's stream -- kernels back to back.
//   * GBNNS_FLAG_DEFER_JOIN (DEVICE buffers, or HOST buffers that are all page-locked): the whole batch on the next of
//     `defer_depth` lanes, consecutive calls rotating, so that the projection (and, for HOST buffers, the copy-in) of
//     batch i+1 runs in the half-empty tail of batch i's walk kernel.
// Measured and NOT done (profiles/r03_split_timelines.txt): cutting one batch into sub-batches on two streams.  A walk
// kernel of 2 500 queries lasts 0.14 ms -- the latency chain of its longest walk -- where 10 000 queries take 0.33 ms,
// and the projection blocks of the next piece (4 wavefronts, 14 KB of LDS) are not scheduled while a walk kernel still
// has workgroups to dispatch (its single wavefronts take the LDS as it frees up), so the pieces queue behind one
// another: 0.43 ms (halves) ... 0.52 ms (quarters) against 0.40 ms undivided.  The same holds with page-locked HOST
// buffers, where the halves' copies do overlap: 0.55 against 0.52 ms.
// The device-visible alias of a page-locked host buffer of `bytes` bytes (hipHostMalloc / hipHostRegister memory); nullptr
// for pageable memory -- and for a buffer whose page-locked range ends before its last byte (a partly registered array, an
// interior pointer near the end of a registration): both ends must be page-locked and map to one contiguous device range.
// Stores through the alias are visible to the host once the storing stream's work has completed.
template <class T>
T* pinned_alias(const T* host_ptr, size_t bytes) {
    if (!host_ptr || bytes == 0) return nullptr;
    auto probe = [](const void* p) -> void* {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, p) != hipSuccess) {
            (void)hipGetLastError();  // (unregistered memory is an error for some runtimes, a type for others)
            return nullptr;
        }
        return at.type == hipMemoryTypeHost ? at.devicePointer : nullptr;
    };
    char* const first = static_cast<char*>(probe(host_ptr));
    if (!first) return nullptr;
    if (bytes > 1) {
        char* const last = static_cast<char*>(probe(reinterpret_cast<const char*>(host_ptr) + (bytes - 1)));
        if (last != first + (bytes - 1)) return nullptr;
        // both ends page-locked and contiguous on the device side; two separate registrations with an unregistered hole
        // between them would pass that too: where the runtime reports the extent of the mapping the first byte belongs to,
        // the whole buffer has to lie inside it
        hipDeviceptr_t rbase = nullptr;
        size_t rsize = 0;
        if (hipMemGetAddressRange(&rbase, &rsize, first) == hipSuccess && rbase && rsize) {
            if (first < static_cast<char*>(rbase) || first + bytes > static_cast<char*>(rbase) + rsize) return nullptr;
        } else {
            (void)hipGetLastError();
        }
    }
    return reinterpret_cast<T*>(first);
}

void plan_call(gbnns_index* ix, const gbnns_search_args* a, int& lanes, int& lane) {
    lanes = 1;
    lane = 0;
#ifdef GBNNS_STAMPS
    return;
#endif
    if ((a->flags & GBNNS_FLAG_SERIAL) || ix->profiling) return;
    if (!(a->flags & GBNNS_FLAG_DEFER_JOIN)) return;  // (HOST buffers: page-locked, checked by gbnns_search_ex)
    lanes = a->defer_depth ? (int)std::min<uint32_t>(std::max<uint32_t>(a->defer_depth, 2u), (uint32_t)kMaxLanes) : 3;  // measured best: 3
    lane = ix->next_lane % lanes;
    ix->next_lane = (lane + 1) % lanes;
}

// The callers' streams wait for deferred calls, oldest first, until at most `keep` of them remain unjoined.
int flush_joins(gbnns_index* ix, size_t keep) {
    while (ix->joins.size() > keep) {
        const std::pair<hipEvent_t, hipStream_t> j = ix->joins.front();
        ix->joins.pop_front();
        HIP_TRY(hipStreamWaitEvent(j.second, j.first, 0));
    }
    return GBNNS_OK;
}

int flush_join(gbnns_index* ix) { return flush_joins(ix, 0); }

// One (sub-)batch on one lane's workspace, enqueued on stream s; arguments validated by gbnns_search_ex.  With HOST
// buffers the copies in and out are enqueued on s too and, when sync_host, waited for.
int search_core(gbnns_index* ix, Lane& L, const gbnns_search_args* a, hipStream_t s, bool sync_host) {
    int rc;
    const uint32_t n_ent = a->n_entries ? a->n_entries : 1u;
    const bool host = a->mem_kind == GBNNS_MEM_HOST;
    const uint32_t nq = (uint32_t)a->n_q;
    const int ef = a->ef;
    const bool plain = a->mode == GBNNS_MODE_PLAIN;
    const int k = plain ? std::max(1, std::min(a->k > 0 ? a->k : 1, ef)) : ef;
    const uint32_t cstride = (uint32_t)k;

    // ---- workspace ----------------------------------------------------------------------
    if ((rc = L.cnt.ensure((size_t)nq * 4))) return rc;
    if ((rc = L.hops.ensure((size_t)nq * 4))) return rc;
    if ((rc = L.dc.ensure((size_t)nq * 4))) return rc;
    if ((rc = L.ovf_list.ensure((size_t)nq * 4))) return rc;
    if ((rc = L.ovf2_list.ensure((size_t)nq * 4))) return rc;
    if (host || !a->out_cand)
        if ((rc = L.cand.ensure((size_t)nq * cstride * 4))) return rc;
    if (a->out_cand_dist && host)
        if ((rc = L.cand_dist.ensure((size_t)nq * cstride * 4))) return rc;
    // ids into HOST memory: page-locked memory takes the kernels' stores directly (40 KB of a 10 000-query batch: no
    // copy launch behind the walk), pageable memory gets a copy out of the lane's buffer
    uint32_t* const ids_alias = host ? pinned_alias(a->out_ids, (size_t)nq * 4) : nullptr;
    if (host && !ids_alias)
        if ((rc = L.out.ensure((size_t)nq * 4))) return rc;
    if (host && a->out_edges)
        if ((rc = L.edges.ensure((size_t)nq * 4))) return rc;
    // general-kernel slots: visited bits + tie bits (n / 4 bytes per slot) and the result list -- 16 n bytes + 512 ef
    // per handle in all (see gbnns.h, "Device memory")
    const uint32_t bitmap_words = ((uint32_t)((ix->n + 31) / 32) + 3u) & ~3u;  // per slot; a multiple of 4 words: slots stay 16-B aligned (the bitmap pass clears with 16-B stores)
    {
        const size_t before = L.g_bitmap.bytes;  // (re)allocation always changes the size
        if ((rc = L.g_bitmap.ensure((size_t)kGeneralSlots * 2 * bitmap_words * 4))) return rc;
        // the tie bits must start out all zero (the kernel keeps them so); the visited bits are cleared per query
        if (L.g_bitmap.bytes != before) HIP_TRY(hipMemsetAsync(L.g_bitmap.p, 0, L.g_bitmap.bytes, s));
    }
    if ((rc = L.g_keys.ensure((size_t)kGeneralSlots * ((size_t)ef + n_ent - 1) * 8))) return rc;

    g_slow.mark("workspace");
    // ---- inputs -------------------------------------------------------------------------
    const float* q_dev = a->queries;
    if (host) {
        if ((rc = L.q_in.ensure((size_t)nq * ix->d * 4))) return rc;
        HIP_TRY(hipMemcpyAsync(L.q_in.p, a->queries, (size_t)nq * ix->d * 4, hipMemcpyHostToDevice, s));
        q_dev = L.q_in.as<float>();
    }
    const uint32_t* entries_dev = a->entry_ids;
    if (a->entry_ids && host) {
        if ((rc = L.entries.ensure((size_t)nq * n_ent * 4))) return rc;
        HIP_TRY(hipMemcpyAsync(L.entries.p, a->entry_ids, (size_t)nq * n_ent * 4, hipMemcpyHostToDevice, s));
        entries_dev = L.entries.as<uint32_t>();
    }
    if (a->entry_ids && host) {
        for (size_t i = 0; i < (size_t)nq * n_ent; ++i)
            if (a->entry_ids[i] >= ix->n) return fail(GBNNS_ERR_INVALID, "entry id %u >= n", a->entry_ids[i]);
    }

    g_slow.mark("copy_in");
    ProfCall pc{};
    const bool prof = ix->profiling;
    if (prof) {
        for (int i = 0; i < 5; ++i) HIP_TRY(hipEventCreate(&pc.ev[i]));
        pc.queries = nq;
        HIP_TRY(hipEventRecord(pc.ev[0], s));
    }

    // ---- stage 1: queries in the walked space ------------------------------------------
    WalkParams w{};
    if (plain) {
        w.q = q_dev; w.qstride = ix->d; w.db = ix->db; w.dstride = ix->d_pad; w.dim = ix->d;
    } else {
        if ((rc = L.q_low.ensure((size_t)nq * ix->dl_pad * 4))) return rc;
        float* ql = L.q_low.as<float>();
        if (a->mode == GBNNS_MODE_NET) {
            if ((rc = run_project(ix, L, q_dev, ix->d, nq, ql, s, s == L.stream && L.stream != nullptr,
                                  (a->flags & GBNNS_FLAG_MFMA_PROJECTION) != 0)))
                return rc;
            w.q = ql; w.qstride = ix->dl_pad;
        } else if (host) {
            HIP_TRY(hipMemcpyAsync(ql, a->queries_low, (size_t)nq * ix->d_low * 4, hipMemcpyHostToDevice, s));
            w.q = ql; w.qstride = ix->d_low;
        } else {
            w.q = a->queries_low; w.qstride = ix->d_low;
        }
        w.db = ix->db_low; w.dstride = ix->dl_pad; w.dim = ix->d_low;
        if (a->out_q_low && a->mode == GBNNS_MODE_NET) {
            const hipMemcpyKind kind = host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
            HIP_TRY(hipMemcpy2DAsync(a->out_q_low, (size_t)ix->d_low * 4, ql, (size_t)ix->dl_pad * 4,
                                     (size_t)ix->d_low * 4, nq, kind, s));
        }
    }
    // Deep batches are walked in locality order (walk_common.h, walk_query_of): a counting sort on the sign bits of the
    // first 12 walked-space coordinates, three small launches.  Wavefronts resident together then walk neighbouring
    // regions and find each other's rows in the caches: -8 % kernel time on a 4 M-node index, -3 % on 1 M -- which the
    // sort's launches would eat on a 10 000-query batch, so only from GBNNS_ORDER_MIN queries on (default 32 768).
    static const uint32_t order_min = getenv("GBNNS_ORDER_MIN") ? (uint32_t)strtoul(getenv("GBNNS_ORDER_MIN"), nullptr, 10) : 32768u;
    static const uint32_t order_bits = getenv("GBNNS_ORDER_BITS") ? (uint32_t)atoi(getenv("GBNNS_ORDER_BITS")) : 12u;  // tuning: 10 .. 16
    if (!plain && nq >= order_min && order_min > 0 && w.dim >= 16u) {
        if ((rc = L.order.ensure((size_t)nq * 4))) return rc;
        if ((rc = L.order_hist.ensure((size_t)4 << 16))) return rc;
        HIP_TRY(launch_query_order(w.q, w.qstride, w.dim, nq, order_bits, L.order_hist.as<uint32_t>(), L.order.as<uint32_t>(), s));
        w.order = L.order.as<uint32_t>();
    }
    if (prof) HIP_TRY(hipEventRecord(pc.ev[1], s));

    // ---- stage 2: beam walk -----------------------------------------------------------
    w.ell = ix->ell.as<uint32_t>(); w.ell_stride = ix->ell_stride; w.n = (uint32_t)ix->n; w.nq = nq;
    w.ef = ef; w.k = k; w.entries = entries_dev; w.n_entries = n_ent;
    w.cand = (!host && a->out_cand) ? a->out_cand : L.cand.as<uint32_t>();
    w.cand_dist = a->out_cand_dist ? (host ? L.cand_dist.as<float>() : a->out_cand_dist) : nullptr;
    w.cand_stride = cstride;
    w.zero_dist_bits = ix->metric == GBNNS_METRIC_NEG_DOT ? 0x80000000u : 0u;
    w.count = L.cnt.as<int32_t>();
    // (per-query counters into page-locked HOST memory are stored there directly, like the ids: written once per
    // query by the kernel that finishes it)
    int32_t* const hops_alias = host ? pinned_alias(a->out_hops, (size_t)nq * 4) : nullptr;
    int32_t* const dc_alias = host ? pinned_alias(a->out_dist_calc, (size_t)nq * 4) : nullptr;
    int32_t* const edges_alias = host ? pinned_alias(a->out_edges, (size_t)nq * 4) : nullptr;
    w.hops = host ? (hops_alias ? hops_alias : L.hops.as<int32_t>()) : (a->out_hops ? a->out_hops : L.hops.as<int32_t>());
    w.dist_calc = host ? (dc_alias ? dc_alias : L.dc.as<int32_t>()) : (a->out_dist_calc ? a->out_dist_calc : L.dc.as<int32_t>());
    w.edges = a->out_edges ? (host ? (edges_alias ? edges_alias : L.edges.as<int32_t>()) : a->out_edges) : nullptr;
    uint32_t* out_dev = host ? (ids_alias ? ids_alias : L.out.as<uint32_t>()) : a->out_ids;
    w.best = plain ? out_dev : nullptr;
    // Control words, two per-call blocks used alternately: [0] list A count, [1] general cursor,
    // [2] max dist_calc, [3] list B count, [4] retry cursor, [6] bitmap-pass cursor.  A call works on one block while its
    // general kernel (the last walk launch) clears the other for the next call -- no per-call memset
    // launch.  Word 5 of block 0 = general-kernel query total (persistent); words 8..71 = diagnostics.
    uint32_t* ctrl_base = L.ctrl.as<uint32_t>();
    const int cur = L.ctrl_phase;
    uint32_t* ctrl = ctrl_base + (cur ? 72 : 0);
    uint32_t* ctrl_next = ctrl_base + (cur ? 0 : 72);
    if (!L.ctrl_clean[cur]) {  // after a failed call only (word 5 of block 0 is the persistent general-kernel total)
        HIP_TRY(hipMemsetAsync(ctrl, 0, 20, s));
        HIP_TRY(hipMemsetAsync(ctrl + 6, 0, 4, s));
    }
    L.ctrl_clean[cur] = false;
    L.ctrl_phase = cur ^ 1;
    w.next_ctrl = ctrl_next;
    w.ovf_count = ctrl; w.g_cursor = ctrl + 1; w.max_dc = ctrl + 2; w.ovf2_count = ctrl + 3; w.r_cursor = ctrl + 4;
    w.g_total = ix->lanes[0].ctrl.as<uint32_t>() + 5; w.ovf_list = L.ovf_list.as<uint32_t>(); w.ovf2_list = L.ovf2_list.as<uint32_t>();
    w.g_bitmap = L.g_bitmap.as<uint32_t>(); w.g_keys = L.g_keys.as<uint64_t>();
    w.bitmap_words = bitmap_words;

    // Visited-set capacity.  The walk kernel's occupancy is LDS-bound, and a 10k-query batch is only
    // a few "rounds" deep (queries / (256 CUs x resident wavefronts)), so the table is sized from
    // the LDS budget: take the number of entries the walks need (first guess 43*ef; afterwards
    // 17/15 x the largest dist_calc of earlier batches, doubled whenever a batch handed queries
    // over), find how many wavefronts per CU that allows, then give each wavefront the whole
    // 160 KB / wavefronts share (capacity need not be a power of two: slot = mulhi(hash, cap)).
    g_slow.mark("stage1");
    if (L.stats_pending && hipEventQuery(L.stats_ev) == hipSuccess) {
        L.stats_pending = false;
        const uint32_t ovf = L.h_stats[0], maxdc = L.h_stats[2];
        // entries so that the largest walk seen (+ 1/16 margin + one pass of new ids) stays under the
        // 15/16 fill limit
        uint32_t need = (maxdc + maxdc / 16 + 64) / 15 * 16 + 16;
        // max_dc covers the retry / general passes too, so a hand-over needs no extra sizing rule
        uint32_t& seen = ix->maxdc_for_ef[L.stats_ef];
        seen = std::max(seen, maxdc);
        uint32_t& slot = ix->cap_for_ef[L.stats_ef];  // stats_ef = skey of that call
        const bool grew = need > slot;
        slot = std::max(slot, need);  // never shrinks: batches with one long walk do not make it oscillate
        // calm = the last observed batch of this (ef, mode) handed nothing over and did not move the size
        const bool quiet = ovf + L.h_stats[3] == 0 && !grew;
        int& streak = ix->calm_streak[L.stats_ef];
        streak = quiet ? std::min(streak + 1, 1 << 20) : 0;
    }
    w.force_wide = (a->flags & GBNNS_FLAG_WIDE_INDEX) ? 1 : 0;
    const bool aux = (a->flags & GBNNS_FLAG_AUX_GRAPH) != 0;
    const int skey = (ef * 8 + a->mode * 2 + (aux ? 1 : 0)) * 2 + w.force_wide;  // sizing statistics are kept per (ef, mode, aux, wide)
    const int calm = ix->calm_streak.count(skey) ? ix->calm_streak[skey] : 0;
    if (aux) {
        w.aux_ell = ix->aux_ell.as<uint32_t>(); w.aux_stride = ix->aux_stride;
        w.hops_bound = a->hops_bound; w.llf = (a->flags & GBNNS_FLAG_LLF) ? 1 : 0;
    }
    w.stamps = reinterpret_cast<unsigned long long*>(ctrl_base + 8);  // words 8..71, diagnostic builds
#ifdef GBNNS_STAMPS
    w.stamps_on = 1;
#endif
    const bool hot = walk_uses_hot(w, ix->metric);
    const bool packed = walk_uses_packed(w);
    const size_t lds_fixed = walk_fast_lds_fixed_bytes(ef, w.dstride, hot, walk_uses_lds_list(w));
    // The hot first pass may keep its visited set in the quotient form (walk_hot.hip, GBNNS_VS_ASM: seven 16-bit entries
    // per bucket instead of five 24-bit ids): ids are told apart inside a home bucket by W - floor(log2 buckets) <= 13
    // bits (n <= 2^W), so the table needs at least 2^(W-13) buckets.
    uint32_t idbits = 1;
    while (idbits < 32 && (1ull << idbits) < ix->n) ++idbits;
    const bool quotient_on = g_knob_quotient.load(std::memory_order_relaxed) != 0;  // tuning / A-B runs, tests: gbnns_debug_knob
    const bool vs_ok = walk_knows_quotient(w, ix->metric);
    constexpr uint32_t kStashBuckets = 4;  // (walk_common.h: the table's last four "buckets" are the stash)
    const uint32_t quotient_min = 7u * ((idbits > 13 ? 1u << (idbits - 13) : 1u) + kStashBuckets + 8u);  // entries (>= 8 real buckets: probe steps of up to 8)
    uint32_t cap;
    int form = packed ? 1 : 0;
    const bool auto_cap = a->hash_capacity == 0;
    // (most wavefronts per CU worth cutting the LDS for: the register files' limit of the first-pass kernel -- 32 for the
    // one-register hot instances, 28 / 24 / 20 for the others -- or the diagnostic knob)
    const int knob_waves = g_knob_max_waves.load(std::memory_order_relaxed);
    const size_t wave_cap = knob_waves > 0 ? (size_t)knob_waves : 32;
    // visited-set capacity for `need` entries in the given form, and the wavefronts per CU it leaves (0: no fit)
    auto size_table = [&](int f, uint32_t need, size_t& slots) -> uint32_t {
        const uint32_t floor_entries = f == 2 ? quotient_min : 0u, extra = f == 2 ? 7u * kStashBuckets : 0u;  // (the stash's four "buckets" hold no slots)
        need = std::max(need + extra, floor_entries);
        const size_t gran = kLdsGran;
        const size_t want = (lds_fixed + walk_hash_bytes(need + 4, f) + gran - 1) / gran * gran;
        slots = std::min<size_t>(wave_cap, kMaxLds / want);
        if (slots == 0) return need;  // does not fit LDS at all: the general kernel takes the batch
        // One more wavefront per CU when it costs only part of the margin: `need` keeps 1/16 of headroom over the
        // largest walk seen; a share that still leaves 1/32 is taken (a later, longer walk is handed over once and
        // raises the requirement for good -- it never shrinks).
        if (slots < wave_cap && ix->maxdc_for_ef.count(skey)) {
            const uint32_t m = ix->maxdc_for_ef[skey];
            const uint32_t need_min = std::max((m + m / 32 + 64) / 15 * 16 + 16 + extra, floor_entries);
            const size_t share1 = kMaxLds / (slots + 1) / gran * gran;
            if (share1 > lds_fixed && walk_hash_entries(share1 - lds_fixed, f) >= need_min + 4) slots += 1;
        }
        const size_t share = kMaxLds / slots / gran * gran;
        return walk_hash_entries(share - lds_fixed, f);
    };
    if (!auto_cap) {
        cap = (uint32_t)a->hash_capacity;
        if (vs_ok && quotient_on && cap >= quotient_min) form = 2;  // (an explicit capacity is a number of entries, whatever the form)
    } else {
        uint32_t need;
        if (ix->cap_for_ef.count(skey)) {
            need = ix->cap_for_ef[skey];
        } else {
            const uint32_t target = std::max<uint32_t>(512u, 32u * (uint32_t)ef);
            need = target + target / 3 + 64;
        }
        size_t slots = 0;
        cap = size_table(form, need, slots);
        if (vs_ok && quotient_on) {
            // the quotient form when it leaves at least as many wavefronts per CU (its bucket test is the shorter one)
            size_t slots_q = 0;
            const uint32_t cap_q = size_table(2, need, slots_q);
            if (slots_q >= slots && slots_q > 0) {
                form = 2;
                cap = cap_q;
                slots = slots_q;
            }
        }
        static const bool dbg = getenv("GBNNS_DEBUG_SIZING") != nullptr;  // diagnostic: the sizing decision of every call
        if (dbg)
            std::fprintf(stderr, "[gbnns sizing] ef %d need %u maxdc %u fixed %zu form %d slots %zu cap %u\n", ef, need,
                         ix->maxdc_for_ef.count(skey) ? ix->maxdc_for_ef[skey] : 0u, lds_fixed, form, slots, cap);
    }
    cap = walk_hash_entries(walk_hash_bytes(cap, form), form);  // whole buckets
    w.vs_shr = 0;
    if (form == 2) {
        const uint32_t buckets = cap / 7u > kStashBuckets ? cap / 7u - kStashBuckets : 0u;
        uint32_t lg = 0;
        while ((2u << lg) <= buckets) ++lg;  // floor(log2 buckets)
        lg = std::min(lg, idbits - 1u);      // (more buckets than ids: a smaller shift only keeps more bits)
        if (buckets == 0 || idbits > lg + 13) {
            form = packed ? 1 : 0;  // (an explicit capacity too small for the form)
            cap = walk_hash_entries(walk_hash_bytes(cap, form), form);
        } else {
            // (tests: gbnns_debug_knob("vs_disp", 1..15) makes probe sequences give up that early, to exercise the hand-over)
            const uint32_t disp = (uint32_t)std::min(15, std::max(1, g_knob_vs_disp.load(std::memory_order_relaxed)));
            // twelve remainder bits and a 4-bit probe number when the table has 2^(W-12) buckets, else thirteen and 3 bits
            const bool r13 = idbits > lg + 12;
            w.vs_shr = (32u - idbits + lg) | (32u - idbits) << 8 | (r13 ? 1u << 16 | std::min(disp, 7u) << 29 : disp << 28);
        }
    }
    {
        // Big batches over an index too large for the quotient form (DEEP10M: 24-bit ids) request a hop's rows before its
        // visited test (walk_hot_spec_kernel: 21.6 against 23.1 ms per 1 M-query launch); with the quotient form testing
        // first wins at every batch size (SIFT-shaped 65 536-query launch 0.90 against 0.79 of the peak).  DESIGN.md 5.1.
        const int spec_min = g_knob_spec_min_nq.load(std::memory_order_relaxed);
        w.spec_rows = (spec_min > 0 && nq >= (uint32_t)spec_min && (w.vs_shr == 0 || g_knob_spec_any_form.load(std::memory_order_relaxed))) ? 1 : 0;
    }
    {
        // A launch's last round, when it is a partial one (10 000 queries on 8 192 wavefront slots: 1 808 of them), walks a
        // draining machine: those wavefronts request their rows BEFORE the visited test (the shorter hop; the rows of
        // already-visited ids cost nothing there) -- walk_hot_kernel 0.320 -> 0.307 ms, 0.68 -> 0.71 of the peak.  Only for a
        // batch that runs alone: with batches in flight the neighbours fill that tail and the extra rows cost 2 - 4 %.
        // wavefront slots of the device (ef <= 64 hot instances: 8 per SIMD, 32 per CU; the CU count is the device's, not a literal)
        const uint32_t slots = (uint32_t)(ix->cus > 0 ? ix->cus : 256) * 32u;
        const int knob = g_knob_spec_tail.load(std::memory_order_relaxed);
        w.spec_from = 0xFFFFFFFFu;
        if (sync_host && knob > 0 && nq > slots && nq % slots != 0 && nq % slots <= slots * (uint32_t)knob / 100u) w.spec_from = nq - nq % slots;
        // ... and a lone batch that never fills the machine runs that way from its first wavefront (2 000 / 4 096 / 6 000 queries:
        // 0.124 / 0.161 / 0.211 ms against 0.140 / 0.174 / 0.213; a full round of 8 192: 0.270 against 0.255 -- not there)
        if (sync_host && knob > 0 && nq <= slots * 6u / 10u) w.spec_from = 0u;
    }
    w.hash_cap = cap;
    w.hash_limit = cap - cap / 16;
    if (w.vs_shr) {  // quotient form: the last four of the cap / 7 buckets are the stash, not slots
        const uint32_t slots = cap - 7u * kStashBuckets;
        w.hash_limit = slots - slots / 16;
    }
    w.all_general = (walk_fast_lds_bytes(w, hot) > kMaxLds || n_ent > 1) ? 1 : 0;  // several entry points: general kernel only
    // Fused re-rank: with a register-list first pass (ef <= 512; and its retry / general successors) every
    // wavefront re-ranks its own query when its walk ends; no re-rank launch.  Needs the pair form
    // (d % 8 == 0) and room for the original-space query in the walk kernels' LDS.
    // Large ef: the visited table of such a walk would leave a handful of wavefronts per CU, so the first pass
    // keeps its visited sets as bitmaps in HBM and runs as many persistent wavefronts as the LDS holds result
    // lists (LDS-list kernel); taken when that at least doubles the resident wavefronts.
    size_t bitmap_per_cu = 0;
    const bool want_fuse = !plain && ix->d % 8 == 0 && !(a->flags & GBNNS_FLAG_NO_FUSED_RERANK);
    w.rr_reserve = want_fuse ? (uint32_t)ix->d_pad * 4u : 0u;
    {
        // measured crossover on the GloVe-like shape: ef = 300 is faster with the register list + LDS table
        // (4.3 vs 5.5 ms), ef = 400 with the bitmap pass (7.1 vs 9.4 ms); SIFT-like ef <= 180 clearly the former
        // (with the register-list variant of the pass: ef = 300 4.2 vs 4.3 ms, a tie; SIFT-like ef 140 .. 180 5.2 .. 4.3
        // against 8.5 .. 6.0 M queries/s on the hot instances -- clearing n/8 bytes per query is not free there)
        // with the quotient form of the table (round 3) the crossover moved up: GloVe-like ef = 400 3.85 ms (table) against
        // 4.85 (bitmap pass), ef = 500 5.61 / 5.51, ef = 600 8.86 / 6.83; SIFT-like ef = 450 4.41 / 4.68, ef = 500 5.30 / 5.29
        static const int min_ef_env = getenv("GBNNS_BITMAP_MIN_EF") ? atoi(getenv("GBNNS_BITMAP_MIN_EF")) : 0;  // (tuning runs)
        const int min_ef = min_ef_env ? min_ef_env : (form == 2 ? 480 : 385);
        const bool forced = (a->flags & GBNNS_FLAG_BITMAP_PASS) != 0;  // diagnostic: whatever ef and batch size
        if (!w.all_general && (ef >= min_ef || forced) && !(a->flags & GBNNS_FLAG_WIDE_INDEX) && (a->hash_capacity == 0 || forced)) {
            const size_t gran = kLdsGran;
            const size_t per_wave = (walk_bitmap_lds_bytes(w, ix->metric) + gran - 1) / gran * gran;
            const size_t per_cu = std::min<size_t>(32, kMaxLds / per_wave);
            const size_t table_waves = std::min<size_t>(32, kMaxLds / ((walk_fast_lds_bytes(w, hot) + gran - 1) / gran * gran));
            // ... and only when the batch is deeper than 1.5 rounds of the wavefronts the table would allow (a
            // 1 000-query batch is resident at once either way, and the register list is faster per hop)
            // ... or, short of that, when the table needs a second round and the bitmap pass holds the whole batch at once
            const bool one_round = (size_t)nq > table_waves * 256 && (size_t)nq <= per_cu * 256;
            if (((per_cu >= 2 * std::max<size_t>(table_waves, 1) && (2 * (size_t)nq > 3 * table_waves * 256 || one_round)) || forced) &&
                per_cu >= 1 && per_cu * 256 * (size_t)bitmap_words * 4 <= (8ull << 30))
                bitmap_per_cu = per_cu;
        }
    }
    // (the ef > 128 hot instance keeps its result list in LDS and stages the re-rank query in the visited-set area)
    const size_t rr_room = walk_rr_room(w, ix->metric, hot, bitmap_per_cu != 0);
    const bool fuse = !walk_uses_lds_list(w) && (!bitmap_per_cu || walk_bitmap_uses_reg(w, ix->metric)) && want_fuse && !w.all_general &&
                      (size_t)ix->d_pad * 4 <= rr_room;
    if (fuse) {
        w.rr_q = q_dev; w.rr_qstride = ix->d; w.rr_db = ix->db; w.rr_dstride = ix->d_pad; w.rr_dim = ix->d;
        w.rr_n = (uint32_t)ix->n; w.rr_out = out_dev; w.rr_metric = ix->metric;
    }

    // Once batches of this (ef, mode) have been calm (no hand-over, size settled), the retry launch is left
    // out: the first pass then appends what it cannot finish to list B directly and the general kernel --
    // always launched -- takes it.  Still exact; a surprise hand-over is just slower once, and un-calms.
    const bool skip_retry = auto_cap && calm >= 2 && !w.all_general;
    if (skip_retry) {
        w.ovf_count = w.ovf2_count;
        w.ovf_list = w.ovf2_list;
    }
    bool bitmap_pass = false;
    if (bitmap_per_cu) {
        if ((rc = L.fp_bitmap.ensure(bitmap_per_cu * 256 * (size_t)bitmap_words * 4))) return rc;
        w.fp_bitmap = L.fp_bitmap.as<uint32_t>();
        w.fp_cursor = ctrl + 6;
        HIP_TRY(launch_walk_bitmap(w, ix->metric, (unsigned)(bitmap_per_cu * 256), s));
        bitmap_pass = true;
    }
    if (!w.all_general) {
        if (!bitmap_pass) HIP_TRY(launch_walk_fast(w, ix->metric, s));
        // retry pass: hand-overs of the first pass, one wavefront per CU with all the LDS
        WalkParams w2 = w;
        w2.vs_shr = 0;  // (the retry kernels keep the packed form)
        const size_t gran = kLdsGran;
        w2.hash_cap = walk_hash_entries(kMaxLds / gran * gran - walk_fast_lds_fixed_bytes(ef, w.dstride, false, walk_uses_lds_list(w)), packed);
        w2.hash_limit = w2.hash_cap - w2.hash_cap / 16;
        if (skip_retry) {
            // nothing to launch
        } else if (w2.hash_cap > cap) {
            HIP_TRY(launch_walk_retry(w2, ix->metric, s));
        } else {
            HIP_TRY(hipMemcpyAsync(ctrl + 3, ctrl, 4, hipMemcpyDeviceToDevice, s));  // nothing to gain: A -> B
            HIP_TRY(hipMemcpyAsync(w.ovf2_list, w.ovf_list, (size_t)nq * 4, hipMemcpyDeviceToDevice, s));
        }
    }
    if (prof) {
        HIP_TRY(hipEventRecord(pc.ev[2], s));
        // name of the first-pass kernel of this call, template arguments included ("walk_general_kernel" when there was none)
        const char* mangled = w.all_general ? nullptr : walk_first_pass_name(s);
        std::string name = "walk_general_kernel";
        if (mangled) {
            int st = 0;
            char* dm = abi::__cxa_demangle(mangled, nullptr, nullptr, &st);
            name = (st == 0 && dm) ? dm : mangled;
            std::free(dm);
            size_t pos = name.find("walk_");  // drop "void gbnns::(anonymous namespace)::" and the parameter list
            if (pos != std::string::npos) name = name.substr(pos);
            pos = name.rfind("(gbnns::WalkParams)");
            if (pos != std::string::npos) name = name.substr(0, pos);
        }
        std::snprintf(ix->acc.walk_kernel, sizeof(ix->acc.walk_kernel), "%s", name.c_str());
    }
    g_slow.mark("walk");
    HIP_TRY(launch_walk_general(w, ix->metric, s));
    g_slow.mark("general");
    L.ctrl_clean[cur ^ 1] = true;  // cleared by that launch
    if (prof) HIP_TRY(hipEventRecord(pc.ev[3], s));
    // statistics of this call (hand-over counts, largest walk), read back asynchronously: every call until
    // things are calm, every 16th afterwards (each read is a small copy on the stream)
    ix->stats_tick += 1;
    if (auto_cap && !w.all_general && !L.stats_pending && (calm < 4 || (ix->stats_tick & 15u) == 0)) {
        if (!L.h_stats) {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&L.h_stats), 16, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&L.stats_ev, hipEventDisableTiming));
        }
        HIP_TRY(hipMemcpyAsync(L.h_stats, ctrl, 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(L.stats_ev, s));
        L.stats_pending = true;
        L.stats_ef = skey;
        L.stats_cap = cap;
    }

    g_slow.mark("stats");
    // ---- stage 3: re-rank in the original space ------------------------------------------
    if (!plain && !fuse) {
        RerankParams r{};
        r.q = q_dev; r.qstride = ix->d; r.db = ix->db; r.dstride = ix->d_pad; r.dim = ix->d;
        r.cand = w.cand; r.cand_stride = cstride; r.count = w.count; r.nq = nq; r.n = (uint32_t)ix->n; r.out = out_dev;
        HIP_TRY(launch_rerank(r, ix->metric, s));
    }
    if (prof) {
        HIP_TRY(hipEventRecord(pc.ev[4], s));
        ix->pending.push_back(pc);
    }

    // ---- outputs ----------------------------------------------------------------------
    if (host) {
        if (!ids_alias) HIP_TRY(hipMemcpyAsync(a->out_ids, out_dev, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
        if (a->out_hops && !hops_alias) HIP_TRY(hipMemcpyAsync(a->out_hops, w.hops, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
        if (a->out_dist_calc && !dc_alias)
            HIP_TRY(hipMemcpyAsync(a->out_dist_calc, w.dist_calc, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
        if (a->out_edges && !edges_alias) HIP_TRY(hipMemcpyAsync(a->out_edges, w.edges, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
        if (a->out_cand)
            HIP_TRY(hipMemcpyAsync(a->out_cand, w.cand, (size_t)nq * cstride * 4, hipMemcpyDeviceToHost, s));
        if (a->out_cand_dist)
            HIP_TRY(hipMemcpyAsync(a->out_cand_dist, w.cand_dist, (size_t)nq * cstride * 4, hipMemcpyDeviceToHost, s));
        g_slow.mark("copy_out");
        if (sync_host) {
            HIP_TRY(hipStreamSynchronize(s));
            ix->in_flight = false;
        }
    }
    return GBNNS_OK;
}


}  // namespace

extern "C" {

int gbnns_search_ex(gbnns_index* ix, const gbnns_search_args* a) {
    if (!ix || !a) return fail(GBNNS_ERR_INVALID, "null argument");
    if (a->struct_size != sizeof(gbnns_search_args))
        return fail(GBNNS_ERR_INVALID, "gbnns_search_args.struct_size mismatch (%u != %zu)",
                    a->struct_size, sizeof(gbnns_search_args));
    if (a->mode < GBNNS_MODE_NET || a->mode > GBNNS_MODE_PLAIN) return fail(GBNNS_ERR_INVALID, "bad mode");
    if (a->ef <= 0) return fail(GBNNS_ERR_INVALID, "ef must be >= 1");
    if (a->mem_kind != GBNNS_MEM_HOST && a->mem_kind != GBNNS_MEM_DEVICE)
        return fail(GBNNS_ERR_INVALID, "unknown mem_kind %d", a->mem_kind);
    if (a->n_q == 0) return GBNNS_OK;
    if (a->n_q >= (1ull << 31)) return fail(GBNNS_ERR_INVALID, "n_q too large");
    if (!a->queries || !a->out_ids) return fail(GBNNS_ERR_INVALID, "queries / out_ids missing");
    if (a->mode == GBNNS_MODE_NET && !ix->has_net) return fail(GBNNS_ERR_INVALID, "NET mode needs a net");
    if (a->mode != GBNNS_MODE_PLAIN && !ix->db_low) return fail(GBNNS_ERR_INVALID, "mode needs db_low");
    if (a->mode == GBNNS_MODE_LOWQ && !a->queries_low) return fail(GBNNS_ERR_INVALID, "queries_low missing");
    if ((a->flags & GBNNS_FLAG_LLF) && !(a->flags & GBNNS_FLAG_AUX_GRAPH))
        return fail(GBNNS_ERR_INVALID, "GBNNS_FLAG_LLF needs GBNNS_FLAG_AUX_GRAPH");
    if ((a->flags & GBNNS_FLAG_AUX_GRAPH) && !ix->has_aux)
        return fail(GBNNS_ERR_INVALID, "GBNNS_FLAG_AUX_GRAPH without gbnns_index_set_aux_graph");
    if (a->hash_capacity != 0 && a->hash_capacity < 128)
        return fail(GBNNS_ERR_INVALID, "hash_capacity must be 0 (auto) or >= 128");
    const uint32_t n_ent = a->n_entries ? a->n_entries : 1u;
    if (n_ent > 1 && !a->entry_ids) return fail(GBNNS_ERR_INVALID, "n_entries > 1 needs entry_ids");
    if (n_ent > 4096) return fail(GBNNS_ERR_INVALID, "n_entries too large");
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t s = static_cast<hipStream_t>(a->stream);
    int rc;
    int n_lanes = 1, lane = 0;
    g_slow.start();
    plan_call(ix, a, n_lanes, lane);
    if (n_lanes > 1 && a->mem_kind == GBNNS_MEM_HOST) {
        // a deferred call returns before its copies have run: every buffer has to be page-locked (a copy from or to
        // pageable memory is staged by the runtime, synchronously).  Pageable buffers: the flag is ignored, plain call.
        const size_t nq = (size_t)a->n_q, kk = (size_t)std::max(1, std::min(a->mode == GBNNS_MODE_PLAIN ? a->k : a->ef, a->ef));
        const struct { const void* p; size_t bytes; } bufs[] = {
            {a->queries, nq * ix->d * 4}, {a->queries_low, nq * ix->d_low * 4},
            {a->entry_ids, nq * std::max<size_t>(a->n_entries, 1) * 4}, {a->out_ids, nq * 4}, {a->out_hops, nq * 4},
            {a->out_dist_calc, nq * 4}, {a->out_edges, nq * 4}, {a->out_cand, nq * kk * 4}, {a->out_cand_dist, nq * kk * 4},
            {a->out_q_low, nq * ix->d_low * 4}};
        for (const auto& b : bufs)
            if (b.p && !pinned_alias(static_cast<const char*>(b.p), b.bytes)) n_lanes = 1;
        if (n_lanes == 1) ix->next_lane = lane;  // (the rotation did not advance)
    }
    if (n_lanes <= 1) {
        if ((rc = enter_stream(ix, s))) return rc;
        return search_core(ix, ix->lanes[0], a, s, true);
    }

    // ---- deferred join: the batch runs on lane `lane`'s internal stream ---------------------------------------
    if ((rc = ensure_lane(ix, lane))) return rc;
    Lane& L = ix->lanes[lane];
    if (!ix->fork_ev) HIP_TRY(hipEventCreateWithFlags(&ix->fork_ev, hipEventDisableTiming));
    bool same_stream = !ix->joins.empty() && ix->last_stream == s;
    for (const auto& j : ix->joins) same_stream = same_stream && j.second == s;
    if (same_stream) {
        // earlier calls' joins are still owed to this very stream: fork first, so that this batch is released beside
        // them, then let the stream wait for the oldest ones -- all but depth - 2, so that with this call at most
        // depth - 1 stay unjoined and `depth` batches are in flight
        HIP_TRY(hipEventRecord(ix->fork_ev, s));
        if ((rc = flush_joins(ix, (size_t)n_lanes - 2))) return rc;
    } else {
        if ((rc = enter_stream(ix, s))) return rc;
        HIP_TRY(hipEventRecord(ix->fork_ev, s));
    }
    ix->last_stream = s;
    ix->in_flight = true;
    HIP_TRY(hipStreamWaitEvent(L.stream, ix->fork_ev, 0));
    g_slow.mark("fork");
    if ((rc = search_core(ix, L, a, L.stream, false))) {
        (void)hipDeviceSynchronize();  // leave nothing in flight behind an error
        ix->in_flight = false;
        return rc;
    }
    // (the join of the lane's batch before last -- the previous user of prev_ev -- has been enqueued by now: at most
    // depth - 1 joins stay owed, and that batch is at least depth calls old)
    std::swap(L.done_ev, L.prev_ev);
    L.prev_ticket = L.ticket;
    HIP_TRY(hipEventRecord(L.done_ev, L.stream));
    L.ticket = ++ix->issued;
    ix->joins.emplace_back(L.done_ev, s);
    g_slow.mark("recorded");
    g_slow.finish();
    return GBNNS_OK;
}

int gbnns_index_wait(gbnns_index* ix, uint32_t keep) {
    if (!ix) return fail(GBNNS_ERR_INVALID, "null index");
    HIP_TRY(hipSetDevice(ix->device));
    const uint64_t upto = ix->issued > keep ? ix->issued - keep : 0;
    // a lane's stream runs its batches in order: its newest batch within the range covers the older ones
    for (Lane& L : ix->lanes) {
        if (L.ticket && L.ticket <= upto) HIP_TRY(hipEventSynchronize(L.done_ev));
        else if (L.prev_ticket && L.prev_ticket <= upto) HIP_TRY(hipEventSynchronize(L.prev_ev));
    }
    return GBNNS_OK;
}

int gbnns_index_join(gbnns_index* ix) {
    if (!ix) return fail(GBNNS_ERR_INVALID, "null index");
    HIP_TRY(hipSetDevice(ix->device));
    return flush_join(ix);
}

// The registrations made here: whole pages, never overlapping one another (two small heap buffers often share a page, and
// page-locking a page twice / releasing it under a neighbour is what the runtime's tables are not built for), counted per
// user buffer -- so that gbnns_host_unpin releases exactly what gbnns_host_pin registered and never a registration the
// caller made itself.
namespace {
constexpr uintptr_t kPage = 4096;
struct PinRange { uintptr_t hi; int users; };
std::mutex g_pin_mu;
std::map<uintptr_t, PinRange> g_pin_ranges;        // lo -> [lo, hi), page-aligned, disjoint
std::map<const void*, std::vector<uintptr_t>> g_pin_users;  // user pointer -> the ranges (by lo) it holds
}  // namespace

int gbnns_host_pin(void* ptr, size_t bytes) {
    if (!ptr || bytes == 0) return fail(GBNNS_ERR_INVALID, "gbnns_host_pin: empty buffer");
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (g_pin_users.count(ptr)) return GBNNS_OK;  // pinned here already
    const uintptr_t lo = reinterpret_cast<uintptr_t>(ptr) & ~(kPage - 1);
    const uintptr_t hi = (reinterpret_cast<uintptr_t>(ptr) + bytes + kPage - 1) & ~(kPage - 1);
    // pages of [lo, hi) that an earlier call registered are shared (counted); the gaps between them are registered now
    std::vector<uintptr_t> held;
    std::vector<std::pair<uintptr_t, uintptr_t>> gaps;
    uintptr_t at = lo;
    auto it = g_pin_ranges.upper_bound(lo);
    if (it != g_pin_ranges.begin()) {
        auto prev = std::prev(it);
        if (prev->second.hi > lo) it = prev;
    }
    for (; it != g_pin_ranges.end() && it->first < hi; ++it) {
        if (it->first > at) gaps.push_back({at, it->first});
        held.push_back(it->first);
        at = std::max(at, it->second.hi);
    }
    if (at < hi) gaps.push_back({at, hi});
    if (held.empty() && pinned_alias(static_cast<char*>(ptr), bytes)) return GBNNS_OK;  // page-locked by the caller: nothing to do, nothing to undo
    for (const auto& gp : gaps) {
        const hipError_t e = hipHostRegister(reinterpret_cast<void*>(gp.first), gp.second - gp.first, hipHostRegisterDefault);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            // (part of the range is page-locked by someone else, or the pages are not ours to lock: the buffer stays
            // pageable for the calls that probe it -- pinned_alias -- and what was registered so far is kept for its users)
            for (uintptr_t h : held) g_pin_ranges[h].users += 1;
            g_pin_users[ptr] = held;
            return fail(GBNNS_ERR_HIP, "hipHostRegister: %s", hipGetErrorString(e));
        }
        g_pin_ranges[gp.first] = PinRange{gp.second, 0};
        held.push_back(gp.first);
    }
    for (uintptr_t h : held) g_pin_ranges[h].users += 1;
    g_pin_users[ptr] = held;
    return GBNNS_OK;
}

int gbnns_host_unpin(void* ptr) {
    if (!ptr) return GBNNS_OK;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    auto u = g_pin_users.find(ptr);
    if (u == g_pin_users.end()) return GBNNS_OK;  // not registered by gbnns_host_pin: nothing to undo
    for (uintptr_t h : u->second) {
        auto r = g_pin_ranges.find(h);
        if (r == g_pin_ranges.end()) continue;
        if (--r->second.users <= 0) {
            if (hipHostUnregister(reinterpret_cast<void*>(h)) != hipSuccess) (void)hipGetLastError();
            g_pin_ranges.erase(r);
        }
    }
    g_pin_users.erase(u);
    return GBNNS_OK;
}

int gbnns_search_batch(gbnns_index* index, const float* queries, size_t n_q, int ef,
                       const uint32_t* entry_ids, uint32_t* out_ids, int32_t* out_hops,
                       int32_t* out_dist_calc, uint32_t* out_cand) {
    gbnns_search_args a{};
    a.struct_size = sizeof a;
    a.mode = GBNNS_MODE_NET;
    a.ef = ef;
    a.k = ef;
    a.mem_kind = GBNNS_MEM_HOST;
    a.n_q = n_q;
    a.queries = queries;
    a.entry_ids = entry_ids;
    a.out_ids = out_ids;
    a.out_hops = out_hops;
    a.out_dist_calc = out_dist_calc;
    a.out_cand = out_cand;
    return gbnns_search_ex(index, &a);
}

}  // extern "C"
