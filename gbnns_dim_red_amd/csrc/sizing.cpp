// sizing.cpp -- the sizing rule of the first-pass walk kernel's exact visited set.  Given the batch's beam, the
// statistics of earlier batches of the same (ef, mode, aux, wide) class and the LDS budget it decides: the table's capacity, its form
// (4-byte slots / five 24-bit ids per 16-byte bucket / the quotient form of seven 16-bit entries), how many wavefronts per CU that
// leaves, the fill limit beyond which a query is handed to the retry pass, and whether rows are requested before the visited test.
// Cut out of api.cpp's search_core in round 5 (no behaviour change).
#include "api_internal.h"

using namespace gbnns;

namespace gbnns_api {

FirstPassSizing size_first_pass(gbnns_index* ix, WalkParams& w, const gbnns_search_args* a, int ef, int skey, uint32_t nq, bool sync_host) {
    const bool hot = walk_uses_hot(w, ix->metric);
    const bool packed = walk_uses_packed(w);
    const size_t lds_fixed = walk_fast_lds_fixed_bytes(ef, w.dstride, hot, walk_uses_lds_list(w), w.coop);
    // The hot first pass may keep its visited set in the quotient form (walk_hot.hip, GBNNS_VS_ASM: seven 16-bit entries
    // per bucket instead of five 24-bit ids): ids are told apart inside a home bucket by W - floor(log2 buckets) <= 13
    // bits (n <= 2^W), so the table needs at least 2^(W-13) buckets.
    uint32_t idbits = 1;
    while (idbits < 32 && (1ull << idbits) < ix->n) ++idbits;
    const bool quotient_on = ix->knob.quotient != 0;  // tuning / A-B runs, tests: gbnns_debug_knob
    const bool vs_ok = walk_knows_quotient(w, ix->metric);
    constexpr uint32_t kStashBuckets = 4;  // (walk_common.h: the table's last four "buckets" are the stash)
    const uint32_t quotient_min = 7u * ((idbits > 13 ? 1u << (idbits - 13) : 1u) + kStashBuckets + 8u);  // entries (>= 8 real buckets: probe steps of up to 8)
    uint32_t cap;
    int form = packed ? 1 : 0;
    const bool auto_cap = a->hash_capacity == 0;
    // (most wavefronts per CU worth cutting the LDS for: the register files' limit of the first-pass kernel -- 32 for the
    // one-register hot instances, 28 / 24 / 20 for the others -- or the diagnostic knob)
    const int knob_waves = ix->knob.max_waves;
    // (the two- / three-wavefront walk of a lone small batch: as many workgroups per CU as the batch puts there, the table takes the rest
    // of the LDS -- shorter probe sequences, gist shape ef 200 / 400: 0.486 / 0.933 against 0.490 / 0.945 ms)
    const size_t cus = (size_t)(ix->cus > 0 ? ix->cus : 256);
    const size_t coop_cap = std::min<size_t>(32, std::max<size_t>(1, ((size_t)nq + cus - 1) / cus));
    const size_t wave_cap = knob_waves > 0 ? (size_t)knob_waves : (w.coop ? coop_cap : 32);
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
            uint32_t need_min = std::max((m + m / 32 + 64) / 15 * 16 + 16 + extra, floor_entries);
            // (the hand-laid-out kernels over two-pass adjacency rows: the extra wavefront must leave the table at the fill the rule aims at + 4 points)
            const int fill2 = ix->knob.vs_fill2;
            if (hot && ix->ell_stride > 32u && fill2 > 0) need_min = std::max(need_min, (uint32_t)((uint64_t)m * 100u / (uint32_t)(fill2 + 4)) + extra);
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
            const uint32_t disp = (uint32_t)std::min(15, std::max(1, ix->knob.vs_disp));
            // twelve remainder bits and a 4-bit probe number when the table has 2^(W-12) buckets, else thirteen and 3 bits
            const bool r13 = idbits > lg + 12;
            w.vs_shr = (32u - idbits + lg) | (32u - idbits) << 8 | (r13 ? 1u << 16 | std::min(disp, 7u) << 29 : disp << 28);
        }
    }
    {
        // Big batches over an index too large for the quotient form (DEEP10M: 24-bit ids) request a hop's rows before its
        // visited test (walk_hot_spec_kernel: 21.6 against 23.1 ms per 1 M-query launch); with the quotient form testing
        // first wins at every batch size (SIFT-shaped 65 536-query launch 0.90 against 0.79 of the peak).  DESIGN.md 5.1.
        const int spec_min = ix->knob.spec_min_nq;
        w.spec_rows = (spec_min > 0 && nq >= (uint32_t)spec_min && (w.vs_shr == 0 || ix->knob.spec_any_form)) ? 1 : 0;
    }
    {
        // A launch's last round, when it is a partial one (10 000 queries on 8 192 wavefront slots: 1 808 of them), walks a
        // draining machine: those wavefronts request their rows BEFORE the visited test (the shorter hop; the rows of
        // already-visited ids cost nothing there) -- walk_hot_kernel 0.320 -> 0.307 ms, 0.68 -> 0.71 of the peak.  Only for a
        // batch that runs alone: with batches in flight the neighbours fill that tail and the extra rows cost 2 - 4 %.
        // wavefront slots of the device (ef <= 64 hot instances: 8 per SIMD, 32 per CU; the CU count is the device's, not a literal)
        const uint32_t slots = (uint32_t)(ix->cus > 0 ? ix->cus : 256) * 32u;
        const int knob = ix->knob.spec_tail;
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
    return FirstPassSizing{hot, packed, auto_cap, form, cap};
}

}  // namespace gbnns_api
