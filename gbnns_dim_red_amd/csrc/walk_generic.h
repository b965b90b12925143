// walk_generic.h -- the generic first-pass / retry walks as kernel templates: result list in LDS (walk_fast_kernel,
// ef > 1 024), in registers (walk_reg_kernel, ef <= 128), two-list (walk_reg_big_kernel, 128 < ef <= 1 024), and their
// HBM-bitmap variants (walk_bitmap_*_kernel).  Instantiated by walk_l2.hip / walk_dot.hip / walk_wide.hip / walk_bitmap.hip.
#pragma once

#include "walk_lists.h"

namespace gbnns {

namespace {

// rows of at least this many floats take the four-lanes-per-row distance in the run-time-length instances (l2_quad_rows, walk_lists.h)
constexpr uint32_t kQuadRowsMinDim = 128;  // (measured: d = 96 is faster a lane per row -- 1.56 against 1.86 ms at ef 120 --, d = 128 four lanes per row: 1.39 against 1.75)


// ---- fast kernel: result list, tie list, visited hash set and the query all live in LDS -------
//
// LDS layout (dynamic): [keys: ef_pad x u64][tie: kTieCap x u64][q: dstride x f32][hash: cap x u32]
// The visited set is an open-addressing hash set of node ids (exact: an id is "visited" iff it
// was inserted).  A query that would exceed hash_limit entries, or whose tie list overflows, is
// appended to the hand-over list and re-run from scratch by the general kernel.

// BITMAP: the visited set is one bit per node in HBM (`bitmap`, private to this wavefront's slot, cleared here per
// query) instead of the LDS table -- for large ef, where the table of a 10 000-distance walk would leave four
// wavefronts per CU: the LDS then holds the result list, the tie list and the query only.
template <int METRIC, int STEPS, bool PACKED, bool BITMAP = false>
__device__ __forceinline__ void walk_fast_one(const WalkParams& p, uint32_t qi, unsigned char* smem,
                                              uint32_t* ovf_count, uint32_t* ovf_list, uint32_t* bitmap = nullptr) {
    const int lane = lane_id();
    const int ef = p.ef;
    const int ef_pad = (ef + 63) & ~63;
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
    TieList tie{keys + ef_pad, kTieCap};
    float* qf = reinterpret_cast<float*>(tie.a + kTieCap);
    uint32_t* hash = reinterpret_cast<uint32_t*>(qf + p.dstride);
    const float4* qs = reinterpret_cast<const float4*>(qf);
    const uint32_t cap = p.hash_cap;  // any size: slot = mulhi(id * C, cap)
    // PACKED (n < 2^24): five 24-bit ids per 16-byte bucket (visited_claim_mask_packed), else 4-byte slots
    const uint32_t nbuckets = PACKED ? cap / 5u : cap >> 2;
    const uint32_t hash_lds = (uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(hash));

    if constexpr (BITMAP) {
        for (uint32_t i = lane; i < p.bitmap_words; i += 64) bitmap[i] = 0u;
    } else if constexpr (PACKED) packed_table_init(hash, nbuckets, 0u, lane);
    else for (uint32_t i = lane; i < cap; i += 64) hash[i] = kInvalidId;
    for (uint32_t i = lane; i < p.dstride; i += 64)
        qf[i] = (i < p.dim) ? p.q[(size_t)qi * p.qstride + i] : 0.f;
    wave_sync();

    WalkState st;
    st.size = 0; st.tsize = 0; st.first_un = 0; st.hops = 0; st.dist_calc = 1; st.edges = 0;

    const uint32_t entry = p.entries ? p.entries[qi] : 0u;
    if (entry >= p.n) { write_bad_entry(p, qi, lane); return; }
    {
        const float d0 = walk_dist<METRIC, STEPS>(qs, p.db + (size_t)entry * p.dstride, p.dim);
        if (lane == 0) {
            keys[0] = make_key(fkey(d0), entry);
            if constexpr (BITMAP) bitmap[entry >> 5] = 1u << (entry & 31u);
            else if constexpr (PACKED) packed_table_put_first(hash, nbuckets, entry);
            else hash[4u * __umulhi(entry * 0x9E3779B1u, nbuckets)] = entry;  // first slot of its bucket
        }
        st.size = 1;
        wave_sync();
    }

    bool handed_over = false;
    uint32_t node;
    // makeStep (search_function.h:15-40) over one adjacency row; `found` = something was inserted (:34)
    auto make_step = [&](const uint32_t* row, uint32_t stride, bool& found) {
        for (uint32_t c = 0; c < stride; c += 64) {
            const uint32_t nb = (c + lane < stride) ? row[c + lane] : kInvalidId;
            const bool valid = nb != kInvalidId;
            const uint64_t mv = __ballot(valid);
            if (!mv) break;
            if constexpr (!BITMAP)
                if ((uint32_t)st.dist_calc + 64u > p.hash_limit) { handed_over = true; break; }
            st.edges += __popcll(mv);
            bool fresh;
            if constexpr (BITMAP) {
                fresh = false;
                if (valid) {
                    const uint32_t bit = 1u << (nb & 31u);
                    fresh = !(atomicOr(&bitmap[nb >> 5], bit) & bit);
                }
            } else if constexpr (PACKED) fresh = __builtin_amdgcn_inverse_ballot_w64(visited_claim_mask_packed(hash_lds, nbuckets, nb, mv));
            else fresh = visited_claim(hash, nbuckets, nb, valid);
            uint32_t dk = 0xFFFFFFFFu;
            const uint64_t mf = __ballot(fresh);
            if constexpr (METRIC == 0 && STEPS == 0) {
                if (p.dim >= kQuadRowsMinDim) {
                    const float dq = l2_quad_rows<false>(mf, nb, p.db, p.dstride, p.dim, qf, lane);
                    dk = fresh ? fkey(dq) : 0xFFFFFFFFu;
                } else if (fresh) dk = fkey(walk_dist<METRIC, STEPS>(qs, p.db + (size_t)nb * p.dstride, p.dim));
            } else {
                if (fresh) dk = fkey(walk_dist<METRIC, STEPS>(qs, p.db + (size_t)nb * p.dstride, p.dim));
            }
            st.dist_calc += __popcll(mf);
            // reference order: neighbours are offered one by one in list order
            const uint32_t worst0 = key_hi(keys[st.size - 1]);
            uint64_t m = __ballot(fresh && (st.size < ef || dk < worst0));
            if (m) found = true;  // the first of them is inserted whatever the others do
            while (m) {
                const int l = __ffsll((unsigned long long)m) - 1;
                m &= m - 1;
                const uint32_t dl = (uint32_t)__shfl((int)dk, l);
                const uint32_t il = (uint32_t)__shfl((int)nb, l);
                if (!offer(keys, tie, st, ef, dl, il, lane)) { handed_over = true; break; }
            }
            if (handed_over) break;
        }
    };
    while (select_candidate(keys, tie, st, node, lane)) {
        bool found = false;
        if (p.aux_ell && (uint32_t)st.hops < p.hops_bound)  // search_function.h:73-80
            make_step(p.aux_ell + (size_t)node * p.aux_stride, p.aux_stride, found);
        if (!handed_over && !(found && p.llf))               // :82-89
            make_step(p.ell + (size_t)node * p.ell_stride, p.ell_stride, found);
        if (handed_over) break;
        st.hops += 1;
    }

    if (handed_over) {
        if (lane == 0) {
            const uint32_t slot = atomicAdd(ovf_count, 1u);
            ovf_list[slot] = qi;
        }
        return;
    }
    write_results(p, qi, keys, st, lane);
}

// First pass: one query per workgroup (= wavefront).  Retry pass: persistent wavefronts, one per CU
// with the largest visited set LDS allows, re-run the queries the first pass handed over; what
// still does not fit goes to the general kernel.
template <typename F>
__device__ __forceinline__ void retry_loop(const WalkParams& p, F&& run) {
    const uint32_t total = *p.ovf_count;
    while (true) {
        uint32_t w = 0;
        if (lane_id() == 0) w = atomicAdd(p.r_cursor, 1u);
        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
        if (w >= total) break;
        run(p.ovf_list[w]);
        wave_sync();
    }
}

template <int METRIC, int STEPS, bool RETRY, bool PACKED>
__global__ __launch_bounds__(64) void walk_fast_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (RETRY) {
        retry_loop(p, [&](uint32_t qi) { walk_fast_one<METRIC, STEPS, PACKED>(p, qi, smem, p.ovf2_count, p.ovf2_list); });
    } else {
        walk_fast_one<METRIC, STEPS, PACKED>(p, walk_query_of(p, blockIdx.x), smem, p.ovf_count, p.ovf_list);
    }
}

// First pass for large ef: persistent wavefronts (as many as the LDS holds result lists), each with its own
// visited bitmap in HBM, pulling query indices from a counter.
template <int METRIC, int STEPS>
__global__ __launch_bounds__(64) void walk_bitmap_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* bitmap = p.fp_bitmap + (size_t)blockIdx.x * p.bitmap_words;
    while (true) {
        uint32_t w = 0;
        if (lane_id() == 0) w = atomicAdd(p.fp_cursor, 1u);
        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
        if (w >= p.nq) break;
        walk_fast_one<METRIC, STEPS, false, true>(p, walk_query_of(p, w), smem, p.ovf_count, p.ovf_list, bitmap);
        wave_sync();
    }
}


// AUX: the auxiliary-graph walk (search_function.h:73-89): a hop expands the node's auxiliary row first (while
// hops < hops_bound), then -- unless llf and that step inserted something -- its main row.
// BITMAP: visited set = one bit per node in HBM (`bitmap`, this wavefront's slot), see walk_bitmap_kernel.
// LATE: rows requested after the visited test -- 1 always, 0 never, -1 by WalkParams::late_rows (both orders in the kernel)
template <int METRIC, int STEPS, bool OFF32, int R, bool ONE_CHUNK = false, bool AUX = false, bool BITMAP = false, bool QLDS_W = false, int LATE = -1>
__device__ __forceinline__ void walk_reg_one(const WalkParams& p, uint32_t qi, unsigned char* smem,
                                             uint32_t* ovf_count, uint32_t* ovf_list, uint32_t* bitmap = nullptr) {
    static_assert(!(AUX && ONE_CHUNK), "auxiliary rows have their own length");
#ifdef GBNNS_STAMPS
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned int probe_iters = 0;  // (the hand-scheduled probe does not count its iterations)
    unsigned int hist[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // [0..7] survivors per hop (0,1,2,3,4,5-8,9-16,17+), [8] merges, [9] merge fallbacks, [10] sequential offers, [11] fast selects
    STAMP(t_begin)
#endif
    constexpr bool kEarlyLoad = (STEPS > 0);  // speculative row loads (METRIC 1 is instantiated with STEPS 0 or 8 only)
    // 128-byte rows: two lanes per neighbour (lane = 2 * slot + half), 32 adjacency slots per pass
    // (STEPS == 24: the 384-byte rows of the reference's PLAIN walks over deep vectors, beams of up to 128 -- a lane per row ran them at
    // 0.61 - 0.65 of the HBM peak, the pair form of the two-list instance at ef = 160 at 0.84)
    constexpr bool kPair = (STEPS == 8) || ((STEPS == 12 || STEPS == 16 || STEPS == 24) && METRIC == 0);  // 128-byte rows; 192- / 256- / 384-byte rows with L2
    constexpr bool kAlt = (STEPS == 8 && METRIC == 1);        // dot metric: even / odd 16-B pieces instead of halves
    constexpr int kQSteps = kPair ? STEPS / 2 : STEPS;        // 16-B steps of the row one lane holds
    constexpr uint32_t kRowBytes = (uint32_t)STEPS * 16u;
    constexpr uint32_t kChunk = kPair ? 32u : 64u;           // adjacency slots per pass
    constexpr uint64_t kSlotLanes = kPair ? 0x5555555555555555ull : ~0ull;  // lanes that own a slot
    const int lane = lane_id();
    const uint32_t slot = kPair ? (uint32_t)lane >> 1 : (uint32_t)lane;    // adjacency slot of this lane
    const uint32_t half = kPair ? (uint32_t)lane & 1u : 0u;
    const int ef = p.ef;
    uint64_t* tie = reinterpret_cast<uint64_t*>(smem);
    uint64_t* stage = tie + kRegTieCap;  // scatter buffer of the batch merge
    float* qf = reinterpret_cast<float*>(stage + reg_stage_slots(R));
    uint32_t* hash = reinterpret_cast<uint32_t*>(qf + p.dstride);
    const float4* qs = reinterpret_cast<const float4*>(qf);
    const uint32_t cap = p.hash_cap;  // any size: slot = mulhi(id * C, cap)
    const uint32_t hash_lds = (uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(hash));

    // OFF32 instantiations serve "compact" indexes (tables < 4 GiB and n < 2^24, walk_off32): 32-bit byte
    // offsets and the packed visited set (24-bit ids, five per 16-byte bucket)
    constexpr bool packed = OFF32;
    // (first pass of a compact index: the host may ask for the quotient form of the table -- p.vs_shr, GBNNS_VS_ASM)
    const uint32_t vs_shr = (packed && !BITMAP && !AUX) ? p.vs_shr : 0u;
    const uint32_t nbuckets = vs_shr ? cap / 7u - kStashBuckets : (packed ? cap / 5u : cap >> 2);
    if constexpr (BITMAP) { for (uint32_t i = lane; i < p.bitmap_words; i += 64) bitmap[i] = 0u; }
    else if (vs_shr) quotient_table_init(hash, nbuckets, lane);
    else if constexpr (packed) packed_table_init(hash, nbuckets, 0u, lane);
    else for (uint32_t i = lane; i < cap; i += 64) hash[i] = kInvalidId;
    for (uint32_t i = lane; i < p.dstride; i += 64)
        qf[i] = (i < p.dim) ? p.q[(size_t)qi * p.qstride + i] : 0.f;
    wave_sync();

    // the query stays in registers (every lane holds all of it): the occupancy scan shows the walk is
    // issue-bound from ~14 wavefronts/CU, so the registers cost nothing and each hop saves 8 LDS reads
    // QLDS_W (walk_reg_wide_kernel: 192- / 256-byte rows, ef <= 64, one pass): the lane's query pieces are re-read from the
    // wavefront's LDS copy every hop, in the shadow of the row loads -- 24 / 32 registers less across the hop
    constexpr bool kWideQLds = QLDS_W;
    static_assert(!QLDS_W || (kPair && (STEPS == 12 || STEPS == 16) && R == 1 && !BITMAP), "QLDS_W: pair-form L2 instances over wide rows");
    RowRegs<kWideQLds ? 0 : kQSteps> qreg;
    if constexpr (kEarlyLoad && !kWideQLds) {
#pragma unroll
        for (int t = 0; t < kQSteps; ++t) qreg.v[t] = kAlt ? qs[2 * t + half] : qs[kQSteps * half + t];
    }

    RegList<R> L;  // this lane's R list entries
    L.clear();
    int size = 1, tsize = 0, hops = 0, dist_calc = 1, edges = 0;
    uint32_t worst;                                // hi of lane size-1 (wave-uniform)
    const uint32_t entry = p.entries ? p.entries[qi] : 0u;
    if (entry >= p.n) { write_bad_entry(p, qi, lane); return; }
    {
        const float d0 = walk_dist<METRIC, STEPS>(qs, row_ptr<OFF32>(p.db, entry, p.dstride), p.dim);
        worst = fkey(d0);
        if (lane == 0) {
            L.hi[0] = worst;
            L.lo[0] = entry << 1;
            if constexpr (BITMAP) bitmap[entry >> 5] = 1u << (entry & 31u);
            else if (vs_shr) quotient_table_put_first(hash, nbuckets, entry, vs_shr);
            else if constexpr (packed) packed_table_put_first(hash, nbuckets, entry);
            else hash[4u * __umulhi(entry * 0x9E3779B1u, nbuckets)] = entry;  // first slot of its bucket
        }
        wave_sync();
    }

    int status = 0;  // 0 = walking, 1 = finished, 2 = handed over to the general kernel
    // Adjacency prefetch: when a node is picked, the row of the entry that will be picked next IF
    // this expansion inserts nothing closer is requested too.  The load stays in flight behind this
    // hop's vector gathers (loads retire in order), so a correct guess removes one of the two
    // dependent memory round trips of the next hop; a wrong guess costs one 128-B row.
    uint32_t pf_node = kInvalidId, pf_val = kInvalidId;
    while (true) {
        STAMP(t0)
        // ---- next node to expand: closest unexpanded entry, ties -> largest id -------------
        uint64_t mu[R];
        int p1 = -1, p2 = -1;  // ranks of the two closest unexpanded entries
        uint32_t node = 0, pred = kInvalidId;
        bool picked = false;
        if constexpr (R == 1) {
            // one list register: straight branches to the rare path (the kernel is instruction-issue
            // bound -- a flag-and-merge formulation costs ~20 more scalar instructions per hop)
            mu[0] = __ballot(!(L.lo[0] & 1u)) & RegList<R>::lane_mask(0, ef);
            // one select + one branch; the empty asm keeps the compiler from folding it back into
            // `mu == 0 || tsize != 0`, which it evaluates with five 64-bit mask instructions
            uint64_t fastm = tsize == 0 ? mu[0] : 0ull;
            asm("" : "+s"(fastm));
            if (fastm == 0) goto slow_select;
            {
                const int q1 = __ffsll((unsigned long long)fastm) - 1;
                const uint64_t m2 = clear_bit64(fastm, q1);
                p1 = q1;
                if (m2) {
                    const int q2 = __ffsll((unsigned long long)m2) - 1;
                    if (readlane_u32(L.hi[0], q1) == readlane_u32(L.hi[0], q2)) goto slow_select;
                    pred = readlane_u32(L.lo[0], q2) >> 1;
                }
                node = readlane_u32(L.lo[0], q1) >> 1;
                if (lane == q1) L.lo[0] |= 1u;
#ifdef GBNNS_STAMPS
                hist[11] += 1;
#endif
                goto have_node;
            }
        slow_select:
            p1 = mu[0] ? __ffsll((unsigned long long)mu[0]) - 1 : -1;
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                mu[r] = __ballot(!(L.lo[r] & 1u)) & RegList<R>::lane_mask(r, ef);
                uint64_t m = mu[r];
                if (p1 < 0 && m) {
                    p1 = r * 64 + __ffsll((unsigned long long)m) - 1;
                    m &= m - 1;
                }
                if (p1 >= 0 && p2 < 0 && m) p2 = r * 64 + __ffsll((unsigned long long)m) - 1;
            }
            if (p1 >= 0 && tsize == 0) {
                // common case: the two closest unexpanded entries have different distances
                if (p2 >= 0) {
                    if (L.hi_at(p1) != L.hi_at(p2)) {
                        picked = true;
                        node = L.lo_at(p1) >> 1;
                        pred = L.lo_at(p2) >> 1;
                        L.mark_expanded(p1, lane);
                    }
                } else {
                    picked = true;
                    node = L.lo_at(p1) >> 1;
                    L.mark_expanded(p1, lane);
                }
            }
        }
        if (!picked) {
            // rare: equal-distance run among the unexpanded entries, a non-empty tie list, or the end
            int best = -1;
            uint32_t hi_p = 0;
            if (p1 >= 0) {
                hi_p = L.hi_at(p1);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint64_t ms = __ballot(!(L.lo[r] & 1u) && L.hi[r] == hi_p) & RegList<R>::lane_mask(r, ef);
                    if (ms) best = r * 64 + 63 - __clzll((long long)ms);
                }
            }
            bool from_tie = false;
            if (tsize > 0 && (best < 0 || hi_p == worst)) {
                // tie entries all sit at the worst distance: the largest id among them competes
                uint32_t v = (lane < tsize) ? key_id(tie[lane]) + 1u : 0u;
                int w = lane;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const uint32_t ov = (uint32_t)__shfl_xor((int)v, off);
                    const int ow = __shfl_xor(w, off);
                    if (ov > v) { v = ov; w = ow; }
                }
                // every lane now holds the same (v, w); tell the compiler so (keeps loop state scalar)
                v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
                w = __builtin_amdgcn_readfirstlane(w);
                const uint32_t lid = (best >= 0) ? (L.lo_at(best) >> 1) : 0u;
                if (best < 0 || v - 1u > lid) {
                    from_tie = true;
                    node = v - 1u;
                    if (lane == 0) tie[w] = tie[tsize - 1];
                    tsize -= 1;
                    wave_sync();
                }
            }
            if (!from_tie) {
                if (best < 0) { status = 1; break; }
                node = L.lo_at(best) >> 1;
                L.mark_expanded(best, lane);
            }
        }
    have_node:
        STAMP(t1)
        STAMP_ADD(0, t0, t1)

        // ---- adjacency row of `node` (prefetched or loaded now), then the prefetch for the next hop
        const uint32_t* row = reinterpret_cast<const uint32_t*>(
            row_ptr<OFF32>(reinterpret_cast<const float*>(p.ell), node, p.ell_stride));
        uint32_t nb0;
        if (node == pf_node) nb0 = pf_val;
        else nb0 = (slot < p.ell_stride) ? row[slot] : kInvalidId;
        // consume nb0 BEFORE issuing the prefetch: the wait for a (conditionally issued) row load
        // must not also cover the younger prefetch load
        const uint64_t mv0 = __ballot(nb0 != kInvalidId);
        STAMP(t2)
        STAMP_ADD(1, t1, t2)
        pf_node = pred;
        if (pred != kInvalidId)
            pf_val = (slot < p.ell_stride) ? reinterpret_cast<const uint32_t*>(row_ptr<OFF32>(reinterpret_cast<const float*>(p.ell), pred, p.ell_stride))[slot] : kInvalidId;
        STAMP(t3)
        STAMP_ADD(2, t2, t3)

        // ---- expand: neighbours in list order, 64 per pass --------------------------------------
        // ONE_CHUNK (rows of at most 64 slots): a single pass, the loop and its bookkeeping fold away
        // AUX: two rows per hop -- g = 0 the auxiliary row (makeStep :73-80), g = 1 the main row (:82-89)
        bool found = false;  // makeStep's flag (:34); only read when AUX
        for (int g = AUX ? ((uint32_t)hops < p.hops_bound ? 0 : 1) : 1; g < 2; ++g) {
        const bool is_aux = AUX && g == 0;
        if (AUX && g == 1 && found && p.llf) break;
        const uint32_t* grow = row;
        uint32_t gstride = p.ell_stride;
        if (is_aux) {
            grow = reinterpret_cast<const uint32_t*>(row_ptr<OFF32>(reinterpret_cast<const float*>(p.aux_ell), node, p.aux_stride));
            gstride = p.aux_stride;
        }
        for (uint32_t c = 0; c < (ONE_CHUNK ? kChunk : gstride); c += kChunk) {
            uint32_t nb = nb0;
            uint64_t mv = mv0;
            if (c || is_aux) {
                nb = (c + slot < gstride) ? grow[c + slot] : kInvalidId;
                mv = __ballot(nb != kInvalidId);
            }
            if (!mv) break;
            if constexpr (!BITMAP)
                if ((uint32_t)dist_calc + 64u > p.hash_limit) { status = 2; break; }
            const bool valid = nb != kInvalidId;
            edges += __popcll(mv & kSlotLanes);
            // row loads go out before the visited test: its LDS round trips overlap the memory latency (rows of already-visited
            // neighbours are fetched in vain) -- or, wide rows in the pair form with WalkParams::late_rows, after it for the new ids only:
            // the 192-byte-row launch at ef = 40 moved 2.24 GB for 1.62 GB of algorithmic bytes, 6.5 TB/s -- bandwidth-bound on those
            RowRegs<kQSteps> rr;
            uint32_t roff = 0;  // row byte offset, kept live past the loads (see below)
            constexpr bool kLateLoad = kPair && ONE_CHUNK && STEPS >= 12 && STEPS <= 16 && LATE != 0;
            const bool late = kLateLoad && (LATE > 0 || p.late_rows != 0);
            auto request_rows = [&](bool want) {
                // (see walk_reg_big_one: every lane loads, empty slots read row 0; measured: the pair form gains in the one-pass
                // hop only, 12- / 16-step rows one lane each wherever their 48 / 64 row registers would be carried around the loop)
                // (384-byte rows: every lane loads in the pass loop too, as in walk_reg_big_one)
                constexpr bool kAllLanes = (kPair && (ONE_CHUNK || STEPS == 24)) || (!kPair && kQSteps >= 12);
                const uint32_t nbl = kAllLanes ? (want ? nb : 0u) : nb;
                const bool ld = kAllLanes || want;
                if constexpr (OFF32) {
                    roff = kPair ? nbl * kRowBytes + half * (kAlt ? 16u : kRowBytes / 2u) : nbl * (p.dstride * 4u);
                    const float* rp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.db) + roff);
                    if constexpr (kAlt) { if (ld) load_row_alt(rr, rp); }
                    else { if (ld) load_row<kQSteps>(rr, rp); }
                } else {
                    const float* rp = row_ptr<OFF32>(p.db, nbl, p.dstride) + half * (kAlt ? 4u : kRowBytes / 8u);
                    if constexpr (kAlt) { if (ld) load_row_alt(rr, rp); }
                    else { if (ld) load_row<kQSteps>(rr, rp); }
                }
            };
            if constexpr (kEarlyLoad) {
                if (!late) request_rows(valid);
            }
            // pair form: the even lane of a pair tests / claims the id, the odd lane ends up with the distance
            uint64_t mclaimed;
            if constexpr (BITMAP) {
                bool fr = false;
                if (valid && (!kPair || half == 0u)) {  // the lane that owns the slot tests and sets the bit
                    const uint32_t bit = 1u << (nb & 31u);
                    fr = !(atomicOr(&bitmap[nb >> 5], bit) & bit);
                }
                mclaimed = __ballot(fr);
            } else if (vs_shr) {
                uint64_t movf;
                mclaimed = visited_claim_mask_quotient(hash_lds, nbuckets, nb, mv & kSlotLanes, vs_shr, movf);
                if (__builtin_expect(movf != 0, 0)) {
                    if (!stash_claim(hash_lds, nbuckets, movf, nb, mclaimed, lane)) { status = 2; break; }
                }
            } else if constexpr (packed) mclaimed = visited_claim_mask_packed(hash_lds, nbuckets, nb, mv & kSlotLanes);
            else mclaimed = visited_claim_mask(hash_lds, nbuckets, nb, mv & kSlotLanes);
            const uint64_t mfresh = kPair ? (mclaimed << 1) : mclaimed;
            const bool fresh = __builtin_amdgcn_inverse_ballot_w64(mfresh);
            STAMP(t4)
            STAMP_ADD(3, t3, t4)
            if constexpr (kEarlyLoad && kLateLoad) {
                if (late) request_rows(__builtin_amdgcn_inverse_ballot_w64(mclaimed | mfresh));  // (both lanes of a new id's pair)
            }
            uint32_t dk = 0xFFFFFFFFu;
            if constexpr (kEarlyLoad) {
                if constexpr (kAlt) {
                    const uint32_t kd = fkey(dot_pair_from_regs(rr, qreg.v));  // all lanes; odd lanes hold distances
                    dk = fresh ? kd : 0xFFFFFFFFu;
                } else if constexpr (kPair && STEPS == 8) {
                    const uint32_t kd = fkey_sumsq(l2_pair_from_regs(rr, qreg.v));  // all lanes; odd lanes hold distances
                    dk = fresh ? kd : 0xFFFFFFFFu;
                } else if constexpr (kPair && kWideQLds) {
                    const float4* ql = qs + kQSteps * half;
                    asm volatile("" : "+v"(ql));  // (not hoisted out of the hop loop)
                    const uint32_t kd = fkey_sumsq(l2_pair_from_regs_wide<kQSteps>(rr, ql));
                    dk = fresh ? kd : 0xFFFFFFFFu;
                } else if constexpr (kPair) {
                    const uint32_t kd = fkey_sumsq(l2_pair_from_regs_wide<kQSteps>(rr, qreg.v));
                    dk = fresh ? kd : 0xFFFFFFFFu;
                } else if constexpr (STEPS == 8) {
                    if (fresh) dk = fkey_sumsq(l2_from_regs8(rr, qreg.v));
                } else {
                    if (fresh) dk = fkey(l2_from_regs<STEPS>(rr, qreg.v));
                }
                // The address register must not double as a load destination: if it does, the next
                // hop's address computation has to wait for every load in flight (vmcnt(0)), which
                // serialises the adjacency prefetch with the gather (tools/check_isa.sh).
                asm volatile("" ::"v"(roff));
            } else {
                if constexpr (METRIC == 0 && STEPS == 0) {
                    // long rows (PLAIN walks over the original vectors): four lanes per row (l2_quad_rows); short ones: a lane per row
                    if (p.dim >= kQuadRowsMinDim) {
                        const float dq = l2_quad_rows<OFF32>(mfresh, nb, p.db, p.dstride, p.dim, qf, lane);
                        dk = fresh ? fkey(dq) : 0xFFFFFFFFu;
                    } else if (fresh) dk = fkey(walk_dist<METRIC, STEPS>(qs, row_ptr<OFF32>(p.db, nb, p.dstride), p.dim));
                } else {
                    if (fresh) dk = fkey(walk_dist<METRIC, STEPS>(qs, row_ptr<OFF32>(p.db, nb, p.dstride), p.dim));
                }
            }
            dist_calc += __popcll(mfresh);
            const bool offer_it = fresh && (size < ef || dk < worst);
            uint64_t m = __ballot(offer_it);
            if (AUX && m) found = true;  // the first of them is inserted whatever happens to the others
            STAMP(t5)
            STAMP_ADD(4, t4, t5)
            // several survivors: merge them in one pass (falls through to the sequential offers on a
            // boundary tie); reference order = one by one in list order (search_function.h:31-37)
#ifdef GBNNS_STAMPS
            {
                const int ns_ = __popcll(m);
                hist[ns_ <= 4 ? ns_ : (ns_ <= 8 ? 5 : (ns_ <= 16 ? 6 : 7))] += 1;
            }
#endif
            {
                if ((m & (m - 1)) != 0) {
                    bool merged;
                    if constexpr (R == 1) merged = reg_merge(m, offer_it, dk, nb, L, size, worst, tsize, stage, ef, lane);
                    else merged = reg_merge_multi<R>(m, offer_it, dk, nb, L, size, worst, tsize, stage, ef, lane);
                    if (merged) {
                        m = 0;
#ifdef GBNNS_STAMPS
                        hist[8] += 1;
                    } else {
                        hist[9] += 1;
#endif
                    }
                }
            }
#ifdef GBNNS_STAMPS
            hist[10] += __popcll(m);
#endif
            while (m) {
                const int l = __ffsll((unsigned long long)m) - 1;
                m &= m - 1;
                if (!reg_offer<R>(readlane_u32(dk, l), readlane_u32(nb, l) << 1, L, size, worst, tsize, tie, ef, lane)) {
                    status = 2;
                    break;
                }
            }
            STAMP(t6)
            STAMP_ADD(5, t5, t6)
            if (status) break;
        }
        if (status) break;
        }
        if (status) break;
        hops += 1;
    }
#ifdef GBNNS_STAMPS
    {
        STAMP(t_end)
        seg[6] = t_end - t_begin;
        if (lane == 0 && p.stamps)
        {
            for (int i = 0; i < 7; ++i) atomicAdd(p.stamps + i, seg[i]);
            for (int i = 0; i < 12; ++i) atomicAdd(p.stamps + 8 + i, (unsigned long long)hist[i]);
            atomicAdd(p.stamps + 20, (unsigned long long)probe_iters);
        }
    }
#endif

    if (status == 2) {
        if (lane == 0) {
            const uint32_t slot = atomicAdd(ovf_count, 1u);
            ovf_list[slot] = qi;
        }
        return;
    }
    reg_write_results<R>(p, qi, L, size, hops, dist_calc, edges, lane);
    if (p.rr_db) {
        const int kept = size < p.k ? size : p.k;
        fused_rerank(p, qi, kept, smem, lane, [&](int rank) { return reg_id_at_rank<R>(L, rank); });
    }
}

// ---- generic walk for 128 < ef <= 1024: walk_reg_one's hop around the two-list result structure (BigList) ---------------
//
// Every shape the hot instances do not take (256-byte rows, the dot metric, adjacency rows of more than one pass,
// auxiliary graphs, large indexes, the HBM-bitmap first pass): same expansion as walk_reg_one, but the result list is
// the base list in LDS + the front list in one register, so that selection and insertion cost what they cost at
// ef <= 64 whatever ef is (the R-register lists spent 28 % of a hop selecting and 30 % inserting at ef = 300).
// LDS: [BigList: big_list_fixed_bytes(ef)][query: dstride floats][visited set | (BITMAP) re-rank scratch].
// ONE_PASS: adjacency rows of one pass (the host checks ell_stride), no auxiliary graph -- the hop is straight-line code.
template <int METRIC, int STEPS, bool OFF32, bool AUX = false, bool BITMAP = false, bool ONE_PASS = false, bool LATE = false>
__device__ __forceinline__ void walk_reg_big_one(const WalkParams& p, uint32_t qi, unsigned char* smem,
                                                 uint32_t* ovf_count, uint32_t* ovf_list, uint32_t* bitmap = nullptr) {
    constexpr bool kEarlyLoad = (STEPS > 0);
    // 128-byte rows, and 192- / 256- / 384- / 512- / 576-byte rows with L2: two lanes per neighbour
    constexpr bool kPair = (STEPS == 8) || ((STEPS == 12 || STEPS == 16 || STEPS == 24 || STEPS == 32 || STEPS == 36) && METRIC == 0);
    // LATE (an instance of its own -- both orders in one kernel cost 30 registers): the rows are requested AFTER the visited test, for
    // the new ids only.  Requested before it -- one memory round trip less per hop, what a launch that is short of wavefronts wants
    // -- a 10 000-query launch over 576-byte rows at ef = 300 moved 35.8 GB for 23.5 GB of algorithmic bytes, 7 TB/s of HBM traffic:
    // there bandwidth, not latency, is what runs out (5.08 -> 4.15 ms with the rows requested late).  The host decides
    // (WalkParams::late_rows, search_core.cpp).
    constexpr bool kLateLoad = LATE;
    static_assert(!LATE || (kPair && STEPS >= 12), "LATE: pair-form instances over wide rows");
    constexpr bool late = LATE;
    constexpr bool kAlt = (STEPS == 8 && METRIC == 1);
    constexpr int kQSteps = kPair ? STEPS / 2 : STEPS;        // row steps (16 bytes) per lane
    constexpr uint32_t kRowBytes = (uint32_t)STEPS * 16u;
    constexpr uint32_t kChunk = kPair ? 32u : 64u;
    constexpr uint64_t kSlotLanes = kPair ? 0x5555555555555555ull : ~0ull;
    const int lane = lane_id();
#ifdef GBNNS_STAMPS  // diagnostic build: cycles per segment of the hop (tools/stamps.py)
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned st_pf1 = 0;  // hops whose node was the runner-up prediction
    STAMP(t_begin)
    unsigned long long t_prev = t_begin;
#endif
    const uint32_t slot = kPair ? (uint32_t)lane >> 1 : (uint32_t)lane;
    const uint32_t half = kPair ? (uint32_t)lane & 1u : 0u;
    const int ef = p.ef;
    BigList B;
    B.init(smem, ef);
    float* qf = reinterpret_cast<float*>(smem + big_list_fixed_bytes(ef));
    unsigned char* after_q = reinterpret_cast<unsigned char*>(qf + p.dstride);  // visited set, or re-rank scratch
    uint32_t* hash = reinterpret_cast<uint32_t*>(after_q);
    const float4* qs = reinterpret_cast<const float4*>(qf);
    const uint32_t cap = p.hash_cap;
    const uint32_t hash_lds = (uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(hash));
    constexpr bool packed = OFF32;
    // (first pass of a compact index: the host may ask for the quotient form of the table -- p.vs_shr, GBNNS_VS_ASM)
    const uint32_t vs_shr = (packed && !BITMAP && !AUX) ? p.vs_shr : 0u;
    const uint32_t nbuckets = vs_shr ? cap / 7u - kStashBuckets : (packed ? cap / 5u : cap >> 2);
    if constexpr (BITMAP) {
        // 16 bytes per lane and store (n / 8 bytes per query: 150 KB at n = 1.2 M); not unrolled: the unrolled form's address
        // registers were the kernel's register peak
        uint4* b4 = reinterpret_cast<uint4*>(bitmap);
        const uint32_t n4 = p.bitmap_words >> 2;
#pragma clang loop unroll(disable)
        for (uint32_t i = lane; i < n4; i += 64) b4[i] = make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t i = (n4 << 2) + lane; i < p.bitmap_words; i += 64) bitmap[i] = 0u;
    }
    else if (vs_shr) quotient_table_init(hash, nbuckets, lane);
    else if constexpr (packed) packed_table_init(hash, nbuckets, 0u, lane);
    else for (uint32_t i = lane; i < cap; i += 64) hash[i] = kInvalidId;
    for (uint32_t i = lane; i < p.dstride; i += 64)
        qf[i] = (i < p.dim) ? p.q[(size_t)qi * p.qstride + i] : 0.f;
    wave_sync();
    RowRegs<kQSteps> qreg;
    if constexpr (kEarlyLoad) {
#pragma unroll
        for (int t = 0; t < kQSteps; ++t) qreg.v[t] = kAlt ? qs[2 * t + half] : qs[kQSteps * half + t];
    }

    int hops = 0, dist_calc = 1, edges = 0;
    // (readfirstlane: every lane computes the same entry id / distance; the compiler must know they are wave-uniform)
    const uint32_t entry = (uint32_t)__builtin_amdgcn_readfirstlane((int)(p.entries ? p.entries[qi] : 0u));
    if (entry >= p.n) { write_bad_entry(p, qi, lane); return; }
    {
        uint32_t k0;
        if constexpr (kPair) {
            // the entry's distance in the pair form, on the query registers (lanes 0 / 1 would do; every pair computes it):
            // the one-lane form reads the whole row and the whole query into registers and was the kernel's register peak
            RowRegs<kQSteps> er;
            const float* rp = row_ptr<OFF32>(p.db, entry, p.dstride) + half * (kAlt ? 4u : kRowBytes / 8u);
            if constexpr (kAlt) load_row_alt(er, rp);
            else load_row<kQSteps>(er, rp);
            uint32_t kd;
            if constexpr (kAlt) kd = fkey(dot_pair_from_regs(er, qreg.v));
            else if constexpr (STEPS == 8) kd = fkey_sumsq(l2_pair_from_regs(er, qreg.v));
            else kd = fkey_sumsq(l2_pair_from_regs_wide<kQSteps>(er, qreg.v));
            k0 = readlane_u32(kd, 1);  // odd lanes hold the distance
        } else {
            k0 = fkey(walk_dist<METRIC, STEPS>(qs, row_ptr<OFF32>(p.db, entry, p.dstride), p.dim));
        }
        B.worst = B.fworst = (uint32_t)__builtin_amdgcn_readfirstlane((int)k0);
        B.F.hi[0] = lane == 0 ? B.worst : B.F.hi[0];
        B.F.lo[0] = lane == 0 ? entry << 1 : B.F.lo[0];
        if (lane == 0) {
            if constexpr (BITMAP) bitmap[entry >> 5] = 1u << (entry & 31u);
            else if (vs_shr) quotient_table_put_first(hash, nbuckets, entry, vs_shr);
            else if constexpr (packed) packed_table_put_first(hash, nbuckets, entry);
            else hash[4u * __umulhi(entry * 0x9E3779B1u, nbuckets)] = entry;
        }
        wave_sync();
    }

    int status = 0;  // 0 = walking, 1 = finished, 2 = handed over
    uint32_t pf_node = kInvalidId, pf_val = kInvalidId;
    // prefetch 2 (as in walk_hot_big): a unique survivor closer than the runner-up IS the next node -- its adjacency row is
    // requested before the insertion instead of after the next selection (one-pass instances; a quarter of the hops)
    constexpr bool kPf2 = ONE_PASS && !AUX && !BITMAP;
    uint32_t pf2_node = kInvalidId, pf2_val = kInvalidId;
    while (true) {
        uint32_t node, pred, h2;
        STAMP(t0)
        STAMP_ADD(7, t_prev, t0)
        if (!B.select(node, pred, h2, lane)) { status = 1; break; }
        STAMP(t1)
        STAMP_ADD(0, t0, t1)
        // ---- adjacency row of `node` (prefetched or loaded now), then the prefetch for the next hop
        const uint32_t* row = reinterpret_cast<const uint32_t*>(
            row_ptr<OFF32>(reinterpret_cast<const float*>(p.ell), node, p.ell_stride));
        uint32_t nb0;
#ifdef GBNNS_STAMPS
        if (node == pf_node) st_pf1 += 1;
#endif
        if (node == pf_node) nb0 = pf_val;
        else if (kPf2 && node == pf2_node) nb0 = pf2_val;
        else nb0 = (slot < p.ell_stride) ? row[slot] : kInvalidId;
        pf2_node = kInvalidId;
        const uint64_t mv0 = __ballot(nb0 != kInvalidId);  // consumed BEFORE the prefetch is issued
        STAMP(t2)
        STAMP_ADD(1, t1, t2)
        pf_node = pred;
        if (pred != kInvalidId)
            pf_val = (slot < p.ell_stride) ? reinterpret_cast<const uint32_t*>(row_ptr<OFF32>(reinterpret_cast<const float*>(p.ell), pred, p.ell_stride))[slot] : kInvalidId;

        bool found = false;  // makeStep's flag (:34); only read when AUX
        for (int g = AUX ? ((uint32_t)hops < p.hops_bound ? 0 : 1) : 1; g < 2; ++g) {
        const bool is_aux = AUX && g == 0;
        if (AUX && g == 1 && found && p.llf) break;
        const uint32_t* grow = row;
        uint32_t gstride = p.ell_stride;
        if (is_aux) {
            grow = reinterpret_cast<const uint32_t*>(row_ptr<OFF32>(reinterpret_cast<const float*>(p.aux_ell), node, p.aux_stride));
            gstride = p.aux_stride;
        }
        for (uint32_t c = 0; c < (ONE_PASS ? 1u : gstride); c += kChunk) {
            uint32_t nb = nb0;
            uint64_t mv = mv0;
            if (c || is_aux) {
                nb = (c + slot < gstride) ? grow[c + slot] : kInvalidId;
                mv = __ballot(nb != kInvalidId);
            }
            if (!mv) break;
            if constexpr (!BITMAP)
                if ((uint32_t)dist_calc + 64u > p.hash_limit) { status = 2; break; }
            const bool valid = nb != kInvalidId;
            STAMP(t3)
            if (c == 0 && !is_aux) { STAMP_ADD(2, t2, t3) }
            edges += __popcll(mv & kSlotLanes);
            RowRegs<kQSteps> rr;
            uint32_t roff = 0;
            // the row loads of this pass; `want` = lanes whose row is needed (the others read row 0, all of them the same lines)
            auto request_rows = [&](bool want) {
                // Rows of 12 / 16 steps (48 / 64 registers per lane): EVERY lane loads -- empty slots read row 0, all of them
                // the same lines -- so that the row registers are defined by this pass alone.  Loaded under `if (valid)`
                // the other lanes keep "the previous value", the compiler carries 64 registers around the hop loop and
                // copies them twice per hop (measured in the code object: 2 x 32 v_mov_b64 per hop on 256-byte rows).
                constexpr bool kAllLanes = kPair || kQSteps >= 12;
                const uint32_t nbl = kAllLanes ? (want ? nb : 0u) : nb;
                const bool ld = kAllLanes || want;
                if constexpr (OFF32) {
                    roff = kPair ? nbl * kRowBytes + half * (kAlt ? 16u : kRowBytes / 2u) : nbl * (p.dstride * 4u);
                    const float* rp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.db) + roff);
                    if constexpr (kAlt) { if (ld) load_row_alt(rr, rp); }
                    else { if (ld) load_row<kQSteps>(rr, rp); }
                } else {
                    const float* rp = row_ptr<OFF32>(p.db, nbl, p.dstride) + half * (kAlt ? 4u : kRowBytes / 8u);
                    if constexpr (kAlt) { if (ld) load_row_alt(rr, rp); }
                    else { if (ld) load_row<kQSteps>(rr, rp); }
                }
            };
            if constexpr (kEarlyLoad) {
                if (!late) request_rows(valid);
            }
            uint64_t mclaimed;
            if constexpr (BITMAP) {
                bool fr = false;
                if (valid && (!kPair || half == 0u)) {
                    const uint32_t bit = 1u << (nb & 31u);
                    fr = !(atomicOr(&bitmap[nb >> 5], bit) & bit);
                }
                mclaimed = __ballot(fr);
            } else if (vs_shr) {
                uint64_t movf;
                mclaimed = visited_claim_mask_quotient(hash_lds, nbuckets, nb, mv & kSlotLanes, vs_shr, movf);
                if (__builtin_expect(movf != 0, 0)) {
                    if (!stash_claim(hash_lds, nbuckets, movf, nb, mclaimed, lane)) { status = 2; break; }
                }
            } else if constexpr (packed) mclaimed = visited_claim_mask_packed(hash_lds, nbuckets, nb, mv & kSlotLanes);
            else mclaimed = visited_claim_mask(hash_lds, nbuckets, nb, mv & kSlotLanes);
            const uint64_t mfresh = kPair ? (mclaimed << 1) : mclaimed;
            STAMP(t4)
            STAMP_ADD(3, t3, t4)
            const bool fresh = __builtin_amdgcn_inverse_ballot_w64(mfresh);
            if constexpr (kEarlyLoad && kLateLoad) {
                if (late) request_rows(__builtin_amdgcn_inverse_ballot_w64(mclaimed | mfresh));  // (both lanes of a new id's pair)
            }
            uint32_t dk = 0xFFFFFFFFu;
            if constexpr (kEarlyLoad) {
                if constexpr (kAlt) {
                    const uint32_t kd = fkey(dot_pair_from_regs(rr, qreg.v));
                    dk = fresh ? kd : 0xFFFFFFFFu;
                } else if constexpr (kPair && STEPS == 8) {
                    const uint32_t kd = fkey_sumsq(l2_pair_from_regs(rr, qreg.v));
                    dk = fresh ? kd : 0xFFFFFFFFu;
                } else if constexpr (kPair) {
                    const uint32_t kd = fkey_sumsq(l2_pair_from_regs_wide<kQSteps>(rr, qreg.v));
                    dk = fresh ? kd : 0xFFFFFFFFu;
                } else {
                    if (fresh) dk = fkey(l2_from_regs<STEPS>(rr, qreg.v));
                }
                asm volatile("" ::"v"(roff));  // the address register must not double as a load destination
            } else {
                if constexpr (METRIC == 0 && STEPS == 0) {
                    // long rows (PLAIN walks over the original vectors): four lanes per row (l2_quad_rows); short ones: a lane per row
                    if (p.dim >= kQuadRowsMinDim) {
                        const float dq = l2_quad_rows<OFF32>(mfresh, nb, p.db, p.dstride, p.dim, qf, lane);
                        dk = fresh ? fkey(dq) : 0xFFFFFFFFu;
                    } else if (fresh) dk = fkey(walk_dist<METRIC, STEPS>(qs, row_ptr<OFF32>(p.db, nb, p.dstride), p.dim));
                } else {
                    if (fresh) dk = fkey(walk_dist<METRIC, STEPS>(qs, row_ptr<OFF32>(p.db, nb, p.dstride), p.dim));
                }
            }
            dist_calc += __popcll(mfresh);
            const uint64_t m = (B.l + B.f < ef) ? mfresh : __ballot(fresh && dk < B.worst);
            STAMP(t5)
            STAMP_ADD(4, t4, t5)
            if constexpr (kPf2) {
                if (m) {
                    uint32_t x = dk;  // all-ones outside the new ids
                    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xf, 0xf, false));   // quad_perm 1,0,3,2
                    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xf, 0xf, false));   // quad_perm 2,3,0,1
                    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xf, 0xf, false));  // row_half_mirror
                    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xf, 0xf, false));  // row_mirror
                    const uint32_t dmin = min(min(readlane_u32(x, 0), readlane_u32(x, 16)), min(readlane_u32(x, 32), readlane_u32(x, 48)));
                    if (dmin < h2) {
                        const uint64_t me = __ballot(dk == dmin) & m;
                        if (me != 0 && (me & (me - 1)) == 0) {
                            pf2_node = readlane_u32(nb, __ffsll((unsigned long long)me) - 1);
                            pf2_val = (slot < p.ell_stride) ? reinterpret_cast<const uint32_t*>(row_ptr<OFF32>(reinterpret_cast<const float*>(p.ell), pf2_node, p.ell_stride))[slot] : kInvalidId;
                        }
                    }
                }
            }
            if (m) {
                if (AUX) found = true;  // the first of them is inserted whatever happens to the others
                // BigList::insert takes up to 32 survivors (its eviction step is one lane per split): the two halves of
                // a 64-slot pass go in one after the other -- same union, same rule (boundary ties fall back to the
                // sequential offers either way)
                const uint64_t ma = kPair ? m : (m & 0xFFFFFFFFull), mb = kPair ? 0ull : (m & 0xFFFFFFFF00000000ull);
                if (ma && !B.insert(ma, dk, nb, lane)) { status = 2; break; }
                if (!kPair && mb) {
                    const uint64_t mb2 = (B.l + B.f < ef) ? mb : (mb & __ballot(dk < B.worst));  // the first half may have lowered the bar
                    if (mb2 && !B.insert(mb2, dk, nb, lane)) { status = 2; break; }
                }
            }
            STAMP(t6)
            STAMP_ADD(5, t5, t6)
#ifdef GBNNS_STAMPS
            t_prev = t6;
#endif
        }
        if (status) break;
        }
        if (status) break;
        hops += 1;
    }

#ifdef GBNNS_STAMPS
    {
        STAMP(t_end)
        seg[6] = t_end - t_begin;
        if (lane == 0 && p.stamps) {
            for (int i = 0; i < 7; ++i) atomicAdd(p.stamps + i, seg[i]);
            atomicAdd(p.stamps + 30, seg[7]);
            atomicAdd(p.stamps + 21, B.st_flush); atomicAdd(p.stamps + 22, B.st_refresh); atomicAdd(p.stamps + 23, B.st_evict);
            atomicAdd(p.stamps + 24, (unsigned long long)B.st_nflush); atomicAdd(p.stamps + 25, (unsigned long long)B.st_nrefresh);
            atomicAdd(p.stamps + 26, (unsigned long long)B.st_nbase); atomicAdd(p.stamps + 27, (unsigned long long)B.st_nseq);
            atomicAdd(p.stamps + 28, (unsigned long long)B.st_ninsert); atomicAdd(p.stamps + 29, (unsigned long long)B.st_slow);
            atomicAdd(p.stamps + 7, (unsigned long long)st_pf1);
        }
    }
#endif
    if (status == 2) {
        if (lane == 0) {
            const uint32_t s = atomicAdd(ovf_count, 1u);
            ovf_list[s] = qi;
        }
        return;
    }
    // (wide walked rows = the wide original rows of GIST: the re-rank keeps 24 loads in flight per lane, rerank_pairs_core)
    B.template finish<(STEPS >= 12 ? 24 : 8)>(p, qi, hops, dist_calc, edges, after_q, lane);
}

// (rows of 32 steps and more: at most two wavefronts per SIMD -- left to its occupancy heuristic the compiler squeezes the 512-byte-row
// instance into 165 registers for a third wavefront and splits the row loads into dependent groups: 2.50 against 2.08 ms at ef = 140)
template <int METRIC, int STEPS, bool OFF32, bool RETRY, bool AUX = false, bool ONE_PASS = false, bool LATE = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, STEPS >= 32 ? 2 : 8))) void walk_reg_big_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (RETRY) {
        retry_loop(p, [&](uint32_t qi) { walk_reg_big_one<METRIC, STEPS, OFF32, AUX>(p, qi, smem, p.ovf2_count, p.ovf2_list); });
    } else {
        walk_reg_big_one<METRIC, STEPS, OFF32, AUX, false, ONE_PASS, LATE>(p, walk_query_of(p, blockIdx.x), smem, p.ovf_count, p.ovf_list);
    }
}

template <int METRIC, int STEPS, bool OFF32, bool RETRY, int R, bool ONE_CHUNK = false, bool AUX = false>
__global__ __launch_bounds__(64) void walk_reg_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (RETRY) {
        retry_loop(p, [&](uint32_t qi) { walk_reg_one<METRIC, STEPS, OFF32, R, false, AUX>(p, qi, smem, p.ovf2_count, p.ovf2_list); });
    } else {
        walk_reg_one<METRIC, STEPS, OFF32, R, ONE_CHUNK, AUX>(p, walk_query_of(p, blockIdx.x), smem, p.ovf_count, p.ovf_list);
    }
}

// The reference's deep shape (96 -> 48: 192-byte rows) and 256-byte rows at ef <= 64, compact index, adjacency rows of one pass:
// the generic hop with the query in LDS, held to GBNNS_WIDE_VGPRS vector registers (6 wavefronts per SIMD instead of 5).
// LATE: its rows requested after the visited test (an instance of its own: with both orders in one kernel the register budget is gone).
template <int STEPS, bool LATE = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(GBNNS_WIDE_VGPRS))) void walk_reg_wide_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    walk_reg_one<0, STEPS, true, 1, true, false, false, true, LATE ? 1 : 0>(p, walk_query_of(p, blockIdx.x), smem, p.ovf_count, p.ovf_list);
}

// First pass with HBM visited bitmaps on register lists (128-byte rows, L2 or dot): persistent wavefronts.
template <int METRIC, int R>
__global__ __launch_bounds__(64) void walk_bitmap_reg_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* bitmap = p.fp_bitmap + (size_t)blockIdx.x * p.bitmap_words;
    while (true) {
        uint32_t w = 0;
        if (lane_id() == 0) w = atomicAdd(p.fp_cursor, 1u);
        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
        if (w >= p.nq) break;
        walk_reg_one<METRIC, 8, true, R, false, false, true>(p, walk_query_of(p, w), smem, p.ovf_count, p.ovf_list, bitmap);
        wave_sync();
    }
}

template <int METRIC, int STEPS = 8, bool ONE_PASS = false, bool LATE = false>
__global__ __launch_bounds__(64) void walk_bitmap_big_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* bitmap = p.fp_bitmap + (size_t)blockIdx.x * p.bitmap_words;
    while (true) {
        uint32_t w = 0;
        if (lane_id() == 0) w = atomicAdd(p.fp_cursor, 1u);
        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
        if (w >= p.nq) break;
        walk_reg_big_one<METRIC, STEPS, true, false, true, ONE_PASS, LATE>(p, walk_query_of(p, w), smem, p.ovf_count, p.ovf_list, bitmap);
        wave_sync();
    }
}

}  // namespace

}  // namespace gbnns
