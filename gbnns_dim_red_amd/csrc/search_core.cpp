// search_core.cpp -- one batch on one lane: workspaces, copies in and out, the kernel sequence projection -> first-pass walk (+ retry)
// -> general kernel (-> re-rank), statistics feedback.  Cut out of api.cpp in round 5; the sizing rule it calls is in sizing.cpp.

#include "api_internal.h"

using namespace gbnns;
using namespace gbnns_api;

namespace gbnns_api {

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
        {   // diagnostic (GBNNS_DEBUG_SIZING): what the first pass of that call handed over -- hand-overs stay exact but cost a retry pass
            static const bool dbg = getenv("GBNNS_DEBUG_SIZING") != nullptr;
            if (dbg && (ovf || L.h_stats[3]))
                std::fprintf(stderr, "[gbnns stats] key %d: first pass handed over %u queries, %u went on to the general kernel (longest walk %u, capacity %u)\n",
                             L.stats_ef, ovf, L.h_stats[3], maxdc, L.stats_cap);
        }
        // entries so that the largest walk seen (+ 1/16 margin + one pass of new ids) stays under the
        // 15/16 fill limit
        uint32_t need = (maxdc + maxdc / 16 + 64) / 15 * 16 + 16;
        // (the hand-laid-out kernels over adjacency rows of two passes: a lower fill -- knob vs_fill2, handle.cpp)
        const int fill2 = ix->knob.vs_fill2;
        if (L.stats_hot2 && fill2 > 0) need = std::max<uint32_t>(need, (uint32_t)((uint64_t)maxdc * 100u / (uint32_t)fill2) + 64u);
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
    // A small batch that runs ALONE (the reference's gist row: 1 000 queries per synchronous call, final_test.cpp:87) on the shapes that
    // have it: the two-wavefront walk (walk_coop.hip) -- a keeper wavefront with the result lists, a scout wavefront that expands the
    // predicted next node ahead.  One wavefront per query leaves such a launch a chain of dependent latencies on a mostly idle machine
    // (at most four queries per CU); measured on the gist shape at ef 200: first-pass kernel 0.487 against 0.526 ms.  NOT with batches in
    // flight: those fill the machine by themselves and are bound by bytes, and the second wavefront's issue slots and speculative rows
    // cost them (1.8 against 2.5 M queries/s).  Knob "coop": 0 never, 1 whenever the shape allows (tests), -1 this rule.
    {
        const int knob = ix->knob.coop;
        const uint32_t cus = (uint32_t)(ix->cus > 0 ? ix->cus : 256);
        const bool in_flight = s == L.stream && L.stream != nullptr;
        w.coop = 0;
        if (knob != 0 && n_ent == 1 && !(a->flags & (GBNNS_FLAG_BITMAP_PASS | GBNNS_FLAG_WIDE_INDEX)) && walk_coop_serves(w, ix->metric) &&
            (knob > 0 || (nq <= 4u * cus && !in_flight)))
            w.coop = knob == 2 ? 2 : 1;   // (2: the three-wavefront form -- measured slower, 0.54 against 0.49 ms on the gist shape: tests and A/B runs only)
    }
    // the first pass's visited set: capacity, form (packed / quotient), the wavefronts per CU it leaves (sizing.cpp)
    FirstPassSizing fps = size_first_pass(ix, w, a, ef, skey, nq, sync_host);
    if (w.coop && ix->knob.coop < 0) {
        // (auto: only when every workgroup of the batch is resident at once -- LDS share per query, eight workgroups of two wavefronts per CU)
        const size_t per_wg = (walk_fast_lds_bytes(w, false) + kLdsGran - 1) / kLdsGran * kLdsGran;
        const size_t per_cu = std::min<size_t>(8, per_wg ? kMaxLds / per_wg : 0);
        if ((size_t)nq > per_cu * (size_t)(ix->cus > 0 ? ix->cus : 256)) {
            w.coop = 0;
            fps = size_first_pass(ix, w, a, ef, skey, nq, sync_host);
        }
    }
    w.coop_lds_floor = 0;
    if (w.coop) {
        // (three-wavefront form) every workgroup of such a batch is resident at once; with c = ceil(n_q / CUs) of them per CU the launch
        // asks for 1 / c of a CU's LDS per workgroup -- its registers and its own LDS would let the dispatcher stack more on one CU (three
        // wavefronts each: a SIMD with four or five of them is the launch's tail) while other CUs sit half empty
        const size_t cus = (size_t)(ix->cus > 0 ? ix->cus : 256);
        const size_t c = std::max<size_t>(1, ((size_t)nq + cus - 1) / cus);
        if (w.coop >= 2 && c <= 4 && ix->knob.coop_pack == 0) w.coop_lds_floor = (uint32_t)std::min<size_t>(64 * 1024, kMaxLds / c / kLdsGran * kLdsGran);
    }
    const bool hot = fps.hot, packed = fps.packed, auto_cap = fps.auto_cap;
    const int form = fps.form;
    const uint32_t cap = fps.cap;
    w.all_general = (walk_fast_lds_bytes(w, hot) > kMaxLds || n_ent > 1) ? 1 : 0;  // several entry points: general kernel only
    // Fused re-rank: with a register-list first pass (ef <= 512; and its retry / general successors) every
    // wavefront re-ranks its own query when its walk ends; no re-rank launch.  Needs the pair form
    // (d % 8 == 0) and room for the original-space query in the walk kernels' LDS.
    // Large ef: the visited table of such a walk would leave a handful of wavefronts per CU, so the first pass
    // keeps its visited sets as bitmaps in HBM and runs as many persistent wavefronts as the LDS holds result
    // lists (LDS-list kernel); taken when that at least doubles the resident wavefronts.
    size_t bitmap_per_cu = 0;
    // (L2: d % 8 == 4 too -- glove's 300 -- the pair form's last 16-byte step is the even lane's alone)
    const bool pair_form = ix->d % 8 == 0 || (ix->d % 4 == 0 && ix->metric == GBNNS_METRIC_L2);
    const bool want_fuse = !plain && pair_form && !(a->flags & GBNNS_FLAG_NO_FUSED_RERANK);
    w.rr_reserve = want_fuse ? (uint32_t)ix->d_pad * 4u : 0u;
    {
        // measured crossover on the GloVe-like shape: ef = 300 is faster with the register list + LDS table
        // (4.3 vs 5.5 ms), ef = 400 with the bitmap pass (7.1 vs 9.4 ms); SIFT-like ef <= 180 clearly the former
        // (with the register-list variant of the pass: ef = 300 4.2 vs 4.3 ms, a tie; SIFT-like ef 140 .. 180 5.2 .. 4.3
        // against 8.5 .. 6.0 M queries/s on the hot instances -- clearing n/8 bytes per query is not free there)
        // with the quotient form of the table (round 3) the crossover moved up: GloVe-like ef = 400 3.85 ms (table) against
        // 4.85 (bitmap pass), ef = 500 5.61 / 5.51, ef = 600 8.86 / 6.83; SIFT-like ef = 450 4.41 / 4.68, ef = 500 5.30 / 5.29
        static const int min_ef_env = getenv("GBNNS_BITMAP_MIN_EF") ? atoi(getenv("GBNNS_BITMAP_MIN_EF")) : 0;  // (tuning runs)
        // 576-byte rows (the reference's glove 300 -> 144; pair form of the two-list kernels, 8 wavefronts per CU by registers whatever
        // the table): walk + re-rank at ef 600 / 800 / 1 000 11.2 / 18.5 / 23.7 ms with the table against 12.5 / 16.3 / 20.1 with the
        // bitmap pass in the same form and its rows requested after the bit test -- the bitmap pass from ef = 700
        const bool rows576 = w.dim == 144u && w.dstride == 144u && ix->metric == GBNNS_METRIC_L2;
        // (second half of round 5: for 576-byte rows the beam alone does not decide -- on a GD(M = 30) graph ef = 600 computes 10 500
        // distances per query, four wavefronts per CU by LDS, 21.4 ms on the table against 1.75 us per distance on the bitmap pass; the rule
        // below -- the pass must at least double the resident wavefronts: 8 by registers against <= 4 by the table -- does from ef 450 on)
        const int min_ef = min_ef_env ? min_ef_env : (rows576 ? 450 : (form == 2 ? 480 : 385));
        const bool forced = (a->flags & GBNNS_FLAG_BITMAP_PASS) != 0;  // diagnostic: whatever ef and batch size
        if (!w.all_general && !w.coop && (ef >= min_ef || forced) && !(a->flags & GBNNS_FLAG_WIDE_INDEX) && (a->hash_capacity == 0 || forced)) {
            const size_t gran = kLdsGran;
            const size_t cus = ix->cus > 0 ? (size_t)ix->cus : 256;  // (the device's, as in sizing.cpp: the slot counts below follow it)
            const size_t per_wave = (walk_bitmap_lds_bytes(w, ix->metric) + gran - 1) / gran * gran;
            const size_t per_cu = std::min<size_t>(rows576 ? 8 : 32, kMaxLds / per_wave);  // (576-byte rows: 223 registers, two wavefronts per SIMD)
            const size_t table_waves = std::min<size_t>(32, kMaxLds / ((walk_fast_lds_bytes(w, hot) + gran - 1) / gran * gran));
            // ... and only when the batch is deeper than 1.5 rounds of the wavefronts the table would allow (a
            // 1 000-query batch is resident at once either way, and the register list is faster per hop)
            // ... or, short of that, when the table needs a second round and the bitmap pass holds the whole batch at once
            const bool one_round = (size_t)nq > table_waves * cus && (size_t)nq <= per_cu * cus;
            if (((per_cu >= 2 * std::max<size_t>(table_waves, 1) && (2 * (size_t)nq > 3 * table_waves * cus || one_round)) || forced) &&
                per_cu >= 1 && per_cu * cus * (size_t)bitmap_words * 4 <= (8ull << 30))
                bitmap_per_cu = per_cu;
        }
    }
    {
        // Two-list kernels over wide rows: rows after the visited test where the launch is bound by bandwidth -- 576-byte rows with at
        // least five wavefronts per CU by LDS (ef 300 / 400 / 600: 4.15 / 5.36 / 10.4 against 5.08 / 6.74 / 10.7 ms; ef 800 / 1 000, four and
        // fewer per CU: 18.2 / 23.8 against 17.4 / 22.3) on a batch that fills the machine
        const int knob = ix->knob.late_rows;
        const size_t per_wave = (walk_fast_lds_bytes(w, hot) + kLdsGran - 1) / kLdsGran * kLdsGran;
        const size_t lds_waves = per_wave ? kMaxLds / per_wave : 0;
        // ... and 192-byte rows at ef <= 64 (walk_reg_wide_kernel<12>, the reference's deep 96 -> 48: 2.24 GB moved for 1.62 GB of
        // algorithmic bytes at ef = 40, 6.5 TB/s; 0.351 -> 0.342 ms alone, 29.4 -> 30.9 M queries/s in flight; its longer beams: no gain)
        const bool l2 = ix->metric == GBNNS_METRIC_L2 && w.dim == w.dstride && nq >= 2048u;
        // ... and the 384- / 512-byte rows of PLAIN walks over deep / sift vectors at beams of more than 128 (2.15 -> 2.08 ms at d = 128,
        // ef = 140; 1.63 -> 1.52 ms at d = 96, ef = 160)
        const bool big_rows = w.dim == 144u || ((w.dim == 96u || w.dim == 128u) && ef > 128);
        const bool auto_late = l2 && ((big_rows && lds_waves >= 5) || (w.dim == 48u && ef <= 64));
        // (the bitmap pass over 576-byte rows: always -- 8 wavefronts per CU whatever the beam)
        const bool bitmap_late = bitmap_per_cu != 0 && w.dim == 144u && l2;
        w.late_rows = knob < 0 ? ((auto_late || bitmap_late) ? 1 : 0) : knob;
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
        const size_t cus = ix->cus > 0 ? (size_t)ix->cus : 256;
        if ((rc = L.fp_bitmap.ensure(bitmap_per_cu * cus * (size_t)bitmap_words * 4))) return rc;
        w.fp_bitmap = L.fp_bitmap.as<uint32_t>();
        w.fp_cursor = ctrl + 6;
        HIP_TRY(launch_walk_bitmap(w, ix->metric, (unsigned)(bitmap_per_cu * cus), s));
        bitmap_pass = true;
    }
    if (!w.all_general) {
        if (!bitmap_pass) HIP_TRY(launch_walk_fast(w, ix->metric, s));
        // retry pass: hand-overs of the first pass, one wavefront per CU with all the LDS
        WalkParams w2 = w;
        w2.vs_shr = 0;  // (the retry kernels keep the packed form)
        w2.coop = 0;    // (... and one wavefront per query)
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
        L.stats_hot2 = hot && ix->ell_stride > 32u;
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


}  // namespace gbnns_api
