// lanes.cpp -- gbnns_search_ex and the batches-in-flight machinery: lanes (workspace + internal stream + done events), the fork /
// join events between the caller's stream and the lanes, gbnns_index_wait / _join, gbnns_search_batch.  Cut out of api.cpp in round 5.

#include "api_internal.h"

using namespace gbnns;
using namespace gbnns_api;

namespace gbnns_api {

// A handle's workspace (projected queries, candidate lists, hand-over lists, control words) is shared by its
// calls and ordered by stream order.  When a call names another stream than the last one that left work in
// flight, the new stream first waits for that work (an event recorded on the old stream now covers everything
// enqueued there so far).  Should the old stream be gone, the device is synchronised instead.
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

}  // namespace gbnns_api

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
