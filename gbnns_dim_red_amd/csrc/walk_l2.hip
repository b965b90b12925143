// walk_l2.hip -- the L2 walks over generic and 128-byte rows, the first-pass dispatcher (launch_walk_fast / _retry) and
// the host-side sizing helpers of the walk kernels (which kernel serves a shape, LDS bytes per wavefront, visited-set forms).
#include "walk_launch.h"

namespace gbnns {

// The LDS-list kernel serves ef beyond the register lists, and auxiliary-graph walks over tables >= 4 GiB.
bool walk_uses_lds_list(const WalkParams& p) { return p.ef > kRegListMaxEf || (p.aux_ell && !walk_off32(p)); }

bool walk_uses_hot(const WalkParams& p, int metric) {
    const bool off32 = walk_off32(p);
    if (p.coop) return false;  // (the two-wavefront walk has its own launcher and LDS layout)
    return (metric == 0 || metric == 1) && p.dim == 32u && p.dstride == 32u && p.ef <= kBigMaxEf && p.ell_stride <= 64u && off32 && (!p.stamps_on || (p.ef > kHot2MaxEf && !getenv("GBNNS_STAMPS_GENERIC"))) &&
           !p.aux_ell;  // (off32 includes n < 2^24: its visited set stores 24-bit ids)
}

// LDS of one wavefront without the visited set.  Register kernels: tie list + merge buffer + query; the
// hot kernel stages the query inside the merge buffer (it lives in registers once the walk starts).
size_t walk_fast_lds_fixed_bytes(int ef, uint32_t dstride, bool hot, bool lds_list, int coop) {
    if (coop) return big_list_fixed_bytes(ef) + (size_t)dstride * 4 + kCoopExtraLds + (coop >= 2 ? kCoop3MoreLds : 0);  // (walk_coop.hip: the two-list layout + result buffers)
    if (hot)  // tie list + merge buffer of 1 / 2 list registers; ef > 128: + the base list and the flush's flag bytes (walk_hot_big)
        return ef <= 64 ? (size_t)kRegTieCap * 8 + (size_t)kRegStageSlots * 8 + (GBNNS_HOT1_QLDS ? 128 : 0)   // (+ the query, re-read every hop)
                        : (ef <= kHot2MaxEf ? (size_t)kRegTieCap * 8 + (size_t)(64 * 2 + 2) * 8 : big_list_fixed_bytes(ef));
    if (ef <= kRegListMaxEf && !lds_list) {  // tie list + merge buffer (ranks 0..ef of the 1 / 2 / 4-register list) + query
        if (ef > kHot2MaxEf) return big_list_fixed_bytes(ef) + (size_t)dstride * 4;  // walk_reg_big_one
        const int regs = ef <= 64 ? 1 : 2;
        return (size_t)kRegTieCap * 8 + (size_t)(64 * regs + 2) * 8 + (size_t)dstride * 4;
    }
    const size_t ef_pad = ((size_t)ef + 63) & ~(size_t)63;
    return ef_pad * 8 + (size_t)kTieCap * 8 + (size_t)dstride * 4;
}

// Visited set of `entries` ids: 4-byte slots in 4-slot buckets; the hot kernel packs five 24-bit ids and a
// counter byte into each 16-byte bucket (3.2 bytes per id).
// The quotient form (hot first pass, small enough n: GBNNS_VS_ASM) packs seven 16-bit entries per bucket (2.29 bytes per id).
size_t walk_hash_bytes(uint32_t entries, int form) {
    return form == 2 ? (size_t)(entries / 7u) * 16 : form == 1 ? (size_t)(entries / 5u) * 16 : (size_t)entries * 4;
}
uint32_t walk_hash_entries(size_t bytes, int form) {
    return form == 2 ? (uint32_t)(bytes / 16) * 7u : form == 1 ? (uint32_t)(bytes / 16) * 5u : ((uint32_t)(bytes / 4) & ~3u);
}
int walk_hash_form(const WalkParams& p, bool) { return p.vs_shr ? 2 : (walk_uses_packed(p) ? 1 : 0); }  // (vs_shr is set only where the first-pass kernel reads it)
// First-pass kernels that know the quotient form: the walk_hot* family and the generic register-list / two-list kernels of a compact index.
bool walk_knows_quotient(const WalkParams& p, int metric) {
    return walk_uses_hot(p, metric) || (walk_off32(p) && !walk_uses_lds_list(p) && !p.aux_ell);
}

// Every LDS kernel packs its visited set when ids fit 24 bits (the register-list kernels: in their compact,
// 32-bit-offset instantiations).
bool walk_uses_packed(const WalkParams& p) { return walk_uses_lds_list(p) ? (p.n <= 0xFFFFFFu && !p.force_wide) : walk_off32(p); }

size_t walk_fast_lds_bytes(const WalkParams& p, bool hot) {
    return walk_fast_lds_fixed_bytes(p.ef, p.dstride, hot, walk_uses_lds_list(p), p.coop) + walk_hash_bytes(p.hash_cap, walk_hash_form(p, hot));
}

thread_local const void* g_walk_first_fn = nullptr;
const char* walk_first_pass_name(hipStream_t s) {
    return g_walk_first_fn ? hipKernelNameRefByPtr(g_walk_first_fn, s) : nullptr;
}

constexpr int kPlain512PairMinEf = 200;  // 512-byte rows (PLAIN walks over sift vectors): beams beyond this take the pair-form two-list instance

static hipError_t launch_walk_any(const WalkParams& p, int metric, bool retry, hipStream_t s) {
    if (p.nq == 0) return hipSuccess;
    if (p.coop && !retry) return launch_walk_coop(p, s);  // (the retry pass re-runs hand-overs on the one-wavefront kernels)
    if (metric == 1) return launch_walk_dot(p, retry, s);
    // (A/B switch: GBNNS_WIDE2=0 sends the 384- / 512-byte rows to the run-time-length instances at every beam)
    static const bool wide2 = !getenv("GBNNS_WIDE2") || atoi(getenv("GBNNS_WIDE2")) != 0;
    if (p.dstride == p.dim) {
        switch (p.dim) {
            case 32: return launch_fast_t<0, 8>(p, retry, s);
            case 48: return launch_walk_wide(p, 12, retry, s);
            case 64: return launch_walk_wide(p, 16, retry, s);
            case 144: return launch_walk_wide(p, 36, retry, s);
            // 384- / 512-byte rows (PLAIN walks over deep / sift vectors): pair-form instances by beam, the rest on the run-time-length ones
            case 96:
                // (never a shape launch_fast_t gives to the LDS-list kernel -- an auxiliary-graph walk over a non-compact index: the two-list
                // instance's auxiliary branch is the 32-bit-offset one, and the host has sized the LDS for the LDS-list layout)
                if (wide2 && p.ef > kHot2MaxEf && p.ef <= kRegListMaxEf && !walk_uses_lds_list(p)) return launch_walk_wide2(p, 24, retry, s);
                // (shorter beams: the pair form in the one- / two-register list kernels for the first pass of a compact index over one-pass
                // adjacency rows (two-pass ones: <= 64 slots) -- a lane per row ran the reference's deep efs_hnsw 40 / 80 / 120 at 0.61 - 0.65 of the HBM peak)
                if (wide2 && p.ef <= kHot2MaxEf && !retry && walk_off32(p) && !p.aux_ell && (p.ell_stride <= 32u || (p.ell_stride <= 64u && p.ef <= 64)) && !p.stamps_on)
                    return launch_walk_wide2_list(p, s);
                break;
            // (512-byte rows: up to ef = 200 the run-time-length two-list instance with four lanes per row is the faster one -- 10 000-query
            // batches in flight at ef 130 / 200: 1.72 / 2.77 ms against 1.93 / 2.87 on the pair form; at ef 300 / 400 5.25 / 8.25 against
            // 4.20 / 5.51: tools/ref_sweep.py --config sift --only plain --efs ..., GBNNS_WIDE2=0 / 1)
            case 128: if (wide2 && p.ef > kPlain512PairMinEf && p.ef <= kRegListMaxEf && !walk_uses_lds_list(p)) return launch_walk_wide2(p, 32, retry, s); break;
            default: break;
        }
    }
    return launch_fast_t<0, 0>(p, retry, s);
}

hipError_t launch_walk_fast(const WalkParams& p, int metric, hipStream_t s) { return launch_walk_any(p, metric, false, s); }
hipError_t launch_walk_retry(const WalkParams& p, int metric, hipStream_t s) { return launch_walk_any(p, metric, true, s); }

}  // namespace gbnns
