// api_internal.h -- what the translation units of the C-ABI host layer share (round 5: api.cpp, 1 760 lines, was cut into
// handle.cpp / graph_api.cpp / lanes.cpp / search_core.cpp / sizing.cpp / pin.cpp; no behaviour change).
//   handle.cpp      the index handle: creation (uploads, layouts), destruction, projection / re-rank entry points, profiling,
//                   the diagnostic knobs, error text
//   graph_api.cpp   graph preparation: gbnns_exact_knn (heaps / matrix-core filter / pools), gbnns_build_graph_gd_device
//   lanes.cpp       gbnns_search_ex and the batches-in-flight machinery (lanes, fork / join events), gbnns_index_wait / _join
//   search_core.cpp one batch on one lane: workspaces, copies, the kernel sequence
//   sizing.cpp      the visited-set sizing rule of the first pass (capacity, form, wavefronts per CU)
//   pin.cpp         gbnns_host_pin / gbnns_host_unpin and the page registry
#pragma once

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


namespace gbnns_api {

using namespace gbnns;

extern thread_local std::string g_err;
int fail(int code, const char* fmt, ...);

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
extern thread_local SlowLog g_slow;

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
    bool stats_hot2 = false;     // that call ran a hand-laid-out kernel over two-pass adjacency rows (sizing: knob vs_fill2)
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

// The diagnostic knobs (defined and documented in handle.cpp).  Round 6: the eleven that steer a handle's searches live IN the handle
// (gbnns_index::knob, set with gbnns_index_knob); the process-wide values (environment, gbnns_debug_knob) are only the defaults a handle
// starts from, so a test or an A/B run that flips one no longer changes every other handle of the process.  The three knobs of
// gbnns_exact_knn -- a function without a handle -- stay process-wide.
struct Knobs {
    int quotient, vs_disp, max_waves, spec_min_nq, spec_any_form, mlp_small, mlp_net, mlp_slab, late_rows, vs_fill2, spec_tail, coop, coop_pack;
};
Knobs knob_defaults();                                    // the process-wide defaults as they stand now
bool knob_set(Knobs& k, const char* name, int value);     // clamps like the environment does; false = no such handle knob
extern std::atomic<int> g_knob_knn_chunk, g_knob_knn_pool_min_k, g_knob_knn_filter;

}  // namespace gbnns_api

struct gbnns_index {
    using DevBuf = gbnns_api::DevBuf;
    using Lane = gbnns_api::Lane;
    using ProfCall = gbnns_api::ProfCall;
    int device = 0;
    int metric = 0;
    uint64_t n = 0;
    uint32_t d = 0, d_low = 0, d_hidden = 0;
    uint32_t d_pad = 0, dl_pad = 0;
    const float* db = nullptr;      // [n x d_pad]
    const float* db_low = nullptr;  // [n x dl_pad]
    DevBuf db_own, db_low_own, ell, net, aux_ell;
    DevBuf net_mfma;                // the net repacked for the one-launch matrix-core projection (filled on the option's first use)
    bool net_mfma_ready = false;
    uint32_t ell_stride = 0, aux_stride = 0;
    bool has_aux = false;
    bool has_net = false;
    float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr, *w3 = nullptr, *b3 = nullptr;
    uint32_t ws1 = 0, ws2 = 0, ws3 = 0;
    int cus = 0;                    // compute units of the device (sizes the one-launch projection's query strips)
    gbnns_api::Knobs knob{};        // this handle's diagnostic knobs (gbnns_index_knob; start: the process defaults at creation)
    // workspaces: lane 0 serves plain calls on the caller's stream; the batches of deferred calls rotate over
    // lanes 0 .. n_lanes-1, each on its own internal stream (see gbnns_search_ex)
    Lane lanes[gbnns_api::kMaxLanes];
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

namespace gbnns_api {

// handle.cpp
int h2d_staged(void* dst, const void* src, size_t bytes);
int upload(DevBuf& dst, const void* src, size_t rows, size_t row_floats, size_t pad_floats, int mem_kind);
int build_ell(const uint64_t* off, const uint32_t* nbr, uint64_t n, std::vector<uint32_t>& ell, uint32_t& stride);
int run_project(gbnns_index* ix, Lane& L, const float* x, uint32_t xstride, uint32_t nx, float* out, hipStream_t s, bool in_flight = false,
                bool mfma = false);
int prof_flush(gbnns_index* ix);

constexpr size_t kMaxLds = 160 * 1024;
// LDS is handed out in granules of 1 280 bytes (measured, tools/ubench/occupancy_census.hip: one-wavefront workgroups of
// 5 120 B -> 32 per CU, 5 121 .. 6 400 B -> 25, 6 401 .. 7 680 B -> 21): a wavefront's share is a multiple of it.
#ifndef GBNNS_LDS_GRAN
#define GBNNS_LDS_GRAN 1280
#endif
constexpr size_t kLdsGran = GBNNS_LDS_GRAN;

// lanes.cpp
int enter_stream(gbnns_index* ix, hipStream_t s);
int ensure_lane(gbnns_index* ix, int i);
void plan_call(gbnns_index* ix, const gbnns_search_args* a, int& lanes, int& lane);
int flush_joins(gbnns_index* ix, size_t keep);
int flush_join(gbnns_index* ix);

// sizing.cpp: the first pass's visited set for this batch -- sets w.vs_shr / hash_cap / hash_limit / spec_rows / spec_from
struct FirstPassSizing {
    bool hot, packed, auto_cap;   // hand-laid-out instance; 24-bit packed ids; capacity chosen by the library (hash_capacity == 0)
    int form;                     // 0 = 4-byte slots, 1 = packed, 2 = quotient
    uint32_t cap;                 // entries
};
FirstPassSizing size_first_pass(gbnns_index* ix, WalkParams& w, const gbnns_search_args* a, int ef, int skey, uint32_t nq, bool sync_host);

// search_core.cpp
int search_core(gbnns_index* ix, Lane& L, const gbnns_search_args* a, hipStream_t s, bool sync_host);


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

}  // namespace gbnns_api
