// walk_general.hip -- the end of the hand-over chain: walk_general_kernel, exact for every input (any ef, any number of ties,
// any visited count, several entry points; everything in global memory), and the tests' merge probe.
#include <algorithm>

#include "launch_util.h"
#include "walk_lists.h"

namespace gbnns {

namespace {

// ---- general kernel: exact for every input (any ef, any number of ties, any visited count) ----
//
// Persistent wavefronts pull query indices from the hand-over list.  Visited set = one bit per
// node in a per-slot global bitmap (cleared per query); result list and tie list in global
// memory (tie capacity n: every node can be in it at most once).

template <int METRIC>
__global__ __launch_bounds__(64) void walk_general_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const uint32_t slot = blockIdx.x;
    float* qf = reinterpret_cast<float*>(smem);
    const float4* qs = reinterpret_cast<const float4*>(qf);
    uint32_t* bitmap = p.g_bitmap + (size_t)slot * 2u * p.bitmap_words;  // [visited bits][tie bits]
    const uint32_t n_ent = p.n_entries ? p.n_entries : 1u;
    uint64_t* keys = p.g_keys + (size_t)slot * ((size_t)p.ef + n_ent - 1u);  // one extra slot per extra entry point
    // tie set: one bit per node (second half of the slot's bitmap block); the host zeroes it when it allocates the
    // block, clear() leaves it all zero after every use
    TieBits tie{bitmap + p.bitmap_words, 0xFFFFFFFFu, 0u};
    const int ef = p.ef;
    const uint32_t total = p.all_general ? p.nq : *p.ovf2_count;
    if (slot == 0 && lane == 0 && total) atomicAdd(p.g_total, total);
    if (slot == 0 && lane < 7 && lane != 5 && p.next_ctrl) p.next_ctrl[lane] = 0u;  // control words of the next call (word 5 of block 0 is the persistent general-kernel total)

    while (true) {
        uint32_t w = 0;
        if (lane == 0) w = atomicAdd(p.g_cursor, 1u);
        w = (uint32_t)__shfl((int)w, 0);
        if (w >= total) break;
        const uint32_t qi = p.all_general ? w : p.ovf2_list[w];

        for (uint32_t i = lane; i < p.dstride; i += 64)
            qf[i] = (i < p.dim) ? p.q[(size_t)qi * p.qstride + i] : 0.f;
        wave_sync();

        {
            bool bad = false;  // an entry id outside the index: empty result, no row is touched
            for (uint32_t e = 0; e < n_ent; ++e) bad |= (p.entries ? p.entries[(size_t)qi * n_ent + e] : 0u) >= p.n;
            if (bad) {
                write_bad_entry(p, qi, lane);
                wave_sync();
                continue;
            }
        }
        WalkState st;
        st.size = 0; st.tsize = 0; st.first_un = 0; st.hops = 0; st.dist_calc = 1; st.edges = 0;
        // search_function.h:54-64: one walk per entry point -- fresh candidate set (every result so far counts as
        // expanded, the tie list is dropped) and fresh visited set; the result heap, hops and dist_calc carry
        // over; the entry's own distance is not counted and it is pushed without the size test (so the heap
        // stays one longer per extra entry point: makeStep pops once per push, :36-37)
        for (uint32_t e = 0; e < n_ent; ++e) {
        for (uint32_t i = lane; i < p.bitmap_words; i += 64) bitmap[i] = 0u;
        if (e > 0) {
            for (int i = lane; i < st.size; i += 64) keys[i] = keys[i] | 1ull;
            tie.clear(st.tsize, lane);
        }
        wave_sync();
        const uint32_t entry = p.entries ? p.entries[(size_t)qi * n_ent + e] : 0u;
        {
            const float d0 =
                metric_dist<METRIC>(qs, reinterpret_cast<const float4*>(p.db + (size_t)entry * p.dstride),
                                    p.dim);
            if (e == 0) {
                if (lane == 0) keys[0] = make_key(fkey(d0), entry);
                st.size = 1;
            } else {
                uint64_t ev;
                bool did;
                const int pos = list_insert(keys, st.size, 0x7FFFFFFF, make_key(fkey(d0), entry), ev, did, lane);
                st.first_un = pos;
            }
            if (lane == 0) bitmap[entry >> 5] = 1u << (entry & 31u);
            wave_sync();
        }
        uint32_t node;
        auto make_step = [&](const uint32_t* row, uint32_t stride, bool& found) {  // search_function.h:15-40
            for (uint32_t c = 0; c < stride; c += 64) {
                const uint32_t nb = (c + lane < stride) ? row[c + lane] : kInvalidId;
                const bool valid = nb != kInvalidId;
                const uint64_t mv = __ballot(valid);
                if (!mv) break;
                st.edges += __popcll(mv);
                bool fresh = false;
                if (valid) {
                    const uint32_t bit = 1u << (nb & 31u);
                    fresh = !(atomicOr(&bitmap[nb >> 5], bit) & bit);
                }
                uint32_t dk = 0xFFFFFFFFu;
                if (fresh)
                    dk = fkey(metric_dist<METRIC>(
                        qs, reinterpret_cast<const float4*>(p.db + (size_t)nb * p.dstride), p.dim));
                const uint64_t mf = __ballot(fresh);
                st.dist_calc += __popcll(mf);
                const uint32_t worst0 = key_hi(keys[st.size - 1]);
                uint64_t m = __ballot(fresh && (st.size < ef || dk < worst0));
                if (m) found = true;
                while (m) {
                    const int l = __ffsll((unsigned long long)m) - 1;
                    m &= m - 1;
                    const uint32_t dl = (uint32_t)__shfl((int)dk, l);
                    const uint32_t il = (uint32_t)__shfl((int)nb, l);
                    offer(keys, tie, st, ef, dl, il, lane);
                }
            }
        };
        while (select_candidate(keys, tie, st, node, lane)) {
            bool found = false;
            if (p.aux_ell && (uint32_t)st.hops < p.hops_bound)  // :73-80
                make_step(p.aux_ell + (size_t)node * p.aux_stride, p.aux_stride, found);
            if (!(found && p.llf))                                // :82-89
                make_step(p.ell + (size_t)node * p.ell_stride, p.ell_stride, found);
            st.hops += 1;
        }
        }  // entry points
        tie.clear(st.tsize, lane);  // leaves the tie bits all zero for the next query
        write_results(p, qi, keys, st, lane);
        if (p.rr_db) {
            const int kept = st.size < p.k ? st.size : p.k;
            fused_rerank(p, qi, kept, smem, lane, [&](int rank) { return key_id(keys[rank]); });
        }
        wave_sync();
    }
}

// Diagnostic kernel (tests only): runs one batch merge on a list / survivor set supplied by the host.
template <int R>
__global__ __launch_bounds__(64) void debug_merge_kernel(const uint64_t* entries, int size, const uint64_t* surv, int ef,
                                                         uint64_t* out, int* out_size) {
    __shared__ uint64_t stage[64 * R + 2];
    const int lane = lane_id();
    RegList<R> L;
    L.clear();
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (r * 64 + lane < size) {
            L.lo[r] = (uint32_t)entries[r * 64 + lane];
            L.hi[r] = (uint32_t)(entries[r * 64 + lane] >> 32);
        }
    const uint64_t sk = surv[lane];
    const bool is_surv = sk != ~0ull;
    const uint64_t m = __ballot(is_surv);
    uint32_t worst = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t t = readlane_u32(L.hi[r], (size - 1) & 63);
        if (((size - 1) >> 6) == r) worst = t;
    }
    int tsize = 0;
    bool ok;
    if constexpr (R == 1) ok = reg_merge(m, is_surv, (uint32_t)(sk >> 32), (uint32_t)sk >> 1, L, size, worst, tsize, stage, ef, lane);
    else ok = reg_merge_multi<R>(m, is_surv, (uint32_t)(sk >> 32), (uint32_t)sk >> 1, L, size, worst, tsize, stage, ef, lane);
#pragma unroll
    for (int r = 0; r < R; ++r) out[r * 64 + lane] = ((uint64_t)L.hi[r] << 32) | L.lo[r];
    if (lane == 0) {
        out_size[0] = size;
        out_size[1] = ok ? 1 : 0;
        out_size[2] = (int)worst;
    }
}

}  // namespace

hipError_t launch_walk_general(const WalkParams& p, int metric, hipStream_t s) {
    if (p.nq == 0) return hipSuccess;
    const size_t lds = std::max((size_t)p.dstride * 4, p.rr_db ? (size_t)p.rr_dstride * 4 : (size_t)0);
    if (metric == 1) {
        hipError_t e = set_lds(walk_general_kernel<1>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((walk_general_kernel<1>), dim3(kGeneralSlots), dim3(64), lds, s, p);
    } else {
        hipError_t e = set_lds(walk_general_kernel<0>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((walk_general_kernel<0>), dim3(kGeneralSlots), dim3(64), lds, s, p);
    }
    return hipGetLastError();
}

hipError_t launch_debug_merge(int regs, const uint64_t* entries, int size, const uint64_t* surv, int ef, uint64_t* out,
                              int* out_size, hipStream_t s) {
    if (regs == 1) hipLaunchKernelGGL((debug_merge_kernel<1>), dim3(1), dim3(64), 0, s, entries, size, surv, ef, out, out_size);
    else if (regs == 2) hipLaunchKernelGGL((debug_merge_kernel<2>), dim3(1), dim3(64), 0, s, entries, size, surv, ef, out, out_size);
    else hipLaunchKernelGGL((debug_merge_kernel<4>), dim3(1), dim3(64), 0, s, entries, size, surv, ef, out, out_size);
    return hipGetLastError();
}


}  // namespace gbnns
