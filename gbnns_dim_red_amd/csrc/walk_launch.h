// walk_launch.h -- which generic walk instance serves a shape: launch_fast_t<METRIC, STEPS> -> launch_reg_t<METRIC, STEPS, R>.
// Included by the translation units that instantiate them: walk_l2.hip (L2; generic and 128-byte rows), walk_dot.hip (dot
// metric), walk_wide.hip (L2, 192- and 256-byte rows) -- three units so that the instantiations build in parallel.
#pragma once

#include "launch_util.h"
#include "walk_generic.h"

namespace gbnns {

template <int METRIC, int STEPS, int R>
static hipError_t launch_reg_t(const WalkParams& p, bool retry, size_t lds, hipStream_t s) {
    // 32-bit byte offsets when both tables are < 4 GiB
    const bool off32 = walk_off32(p);
    // the register-list / two-list kernels have their auxiliary-graph hop in the 32-bit-offset instances only: an auxiliary-graph walk over a
    // non-compact index belongs to the LDS-list kernel (walk_uses_lds_list; launch_fast_t sends it there, and so must every other caller)
    if (p.aux_ell && !off32) return hipErrorInvalidValue;
    if constexpr (R >= 4) {
        // ef > 128: base list in LDS + front list in one register (walk_reg_big_one), whatever the shape
        if constexpr ((METRIC == 0 || METRIC == 1) && STEPS == 8) {
            if (!retry && walk_uses_hot(p, METRIC)) return launch_walk_hot(p, METRIC, s);  // (walk_hot.hip)
        }
        if (p.aux_ell)
            return retry ? launch_walk_k(walk_reg_big_kernel<METRIC, STEPS, true, true, true>, p, true, lds, s)
                         : launch_walk_k(walk_reg_big_kernel<METRIC, STEPS, true, false, true>, p, false, lds, s);
        if (off32) {
            // the common shape (compact index, adjacency rows of one pass) gets the hop without the pass loop
            if (!retry && p.ell_stride <= ((STEPS == 8 || ((STEPS == 12 || STEPS == 16 || STEPS >= 24) && METRIC == 0)) ? 32u : 64u)) {  // pair form: 32 slots per pass
                if constexpr (STEPS >= 24 && METRIC == 0) {  // (the shapes with an instance whose rows are requested after the visited test: WalkParams::late_rows)
                    if (p.late_rows) return launch_walk_k(walk_reg_big_kernel<METRIC, STEPS, true, false, false, true, true>, p, false, lds, s);
                }
                return launch_walk_k(walk_reg_big_kernel<METRIC, STEPS, true, false, false, true>, p, false, lds, s);
            }
            // (adjacency rows of two passes -- the reference's M18 / M20 hnsw graphs have rows of up to 36 / 40 slots: the rows-after-the-test
            // order in the pass loop too; glove 300 -> 144 on a GD(M = 20) graph at ef 300 / 400 / 600: 6.96 / 9.64 / 16.1 ms with the rows first)
            if constexpr (STEPS >= 24 && METRIC == 0) {
                if (!retry && p.late_rows) return launch_walk_k(walk_reg_big_kernel<METRIC, STEPS, true, false, false, false, true>, p, false, lds, s);
            }
            return retry ? launch_walk_k(walk_reg_big_kernel<METRIC, STEPS, true, true>, p, true, lds, s)
                         : launch_walk_k(walk_reg_big_kernel<METRIC, STEPS, true, false>, p, false, lds, s);
        }
        return retry ? launch_walk_k(walk_reg_big_kernel<METRIC, STEPS, false, true>, p, true, lds, s)
                     : launch_walk_k(walk_reg_big_kernel<METRIC, STEPS, false, false>, p, false, lds, s);
    } else {
    if (p.aux_ell)  // auxiliary-graph walk (32-bit offsets only; otherwise launch_fast_t took the LDS-list kernel)
        return retry ? launch_walk_k(walk_reg_kernel<METRIC, STEPS, true, true, R, false, true>, p, true, lds, s)
                     : launch_walk_k(walk_reg_kernel<METRIC, STEPS, true, false, R, false, true>, p, false, lds, s);
    if constexpr (R == 1) {
        // the common shape (ef <= 64, adjacency rows of one pass) gets a loop-free expansion;
        // 128-byte rows with L2 additionally the hand-laid-out hop of walk_hot_one
        if constexpr ((METRIC == 0 || METRIC == 1) && STEPS == 8) {
            if (!retry && walk_uses_hot(p, METRIC)) return launch_walk_hot(p, METRIC, s);  // (walk_hot.hip: the hand-laid-out hop)
        }
        if (off32 && !retry && p.ell_stride <= ((STEPS == 8 || ((STEPS == 12 || STEPS == 16) && METRIC == 0)) ? 32u : 64u)) {  // pair form: 32 slots per pass
            if constexpr (METRIC == 0 && (STEPS == 12 || STEPS == 16)) {
                if (!p.stamps_on)
                    return p.late_rows ? launch_walk_k(walk_reg_wide_kernel<STEPS, true>, p, false, lds, s)
                                       : launch_walk_k(walk_reg_wide_kernel<STEPS, false>, p, false, lds, s);
            }
            return launch_walk_k(walk_reg_kernel<METRIC, STEPS, true, false, 1, true>, p, false, lds, s);
        }
    }
    if constexpr (R == 2 && (METRIC == 0 || METRIC == 1) && STEPS == 8) {
        // the hot shape, 64 < ef <= 128: two list registers (measured: 0.85 ms against 0.93 ms with the two-list structure)
        if (!retry && walk_uses_hot(p, METRIC)) return launch_walk_hot(p, METRIC, s);
    }
    if (off32)
        return retry ? launch_walk_k(walk_reg_kernel<METRIC, STEPS, true, true, R>, p, true, lds, s)
                     : launch_walk_k(walk_reg_kernel<METRIC, STEPS, true, false, R>, p, false, lds, s);
    return retry ? launch_walk_k(walk_reg_kernel<METRIC, STEPS, false, true, R>, p, true, lds, s)
                 : launch_walk_k(walk_reg_kernel<METRIC, STEPS, false, false, R>, p, false, lds, s);
    }
}

// ef <= 64: one list register per lane (all row-length specialisations); ef <= 128 / 256: two / four
// registers (generic or 128-B-row distance); beyond that the list lives in LDS.
template <int METRIC, int STEPS>
static hipError_t launch_fast_t(const WalkParams& p, bool retry, hipStream_t s) {
    const size_t lds = walk_fast_lds_bytes(p, false);
    constexpr int kWideSteps = (STEPS == 8) ? 8 : 0;
    // 256-byte rows (d_low = 64, the GIST shape, whose efs start at 200): the 4- and 8-register lists keep the
    // unrolled distance with early row loads as well
    constexpr int kWideSteps48 = (STEPS == 8 || STEPS == 12 || STEPS == 16) ? STEPS : 0;
    if (walk_uses_lds_list(p)) {
        if (walk_uses_packed(p))
            return retry ? launch_walk_k(walk_fast_kernel<METRIC, STEPS, true, true>, p, true, lds, s)
                         : launch_walk_k(walk_fast_kernel<METRIC, STEPS, false, true>, p, false, lds, s);
        return retry ? launch_walk_k(walk_fast_kernel<METRIC, STEPS, true, false>, p, true, lds, s)
                     : launch_walk_k(walk_fast_kernel<METRIC, STEPS, false, false>, p, false, lds, s);
    }
    // (576-byte rows -- the reference's glove 300 -> 144 -- and the 384- / 512-byte rows of its PLAIN walks over deep / sift vectors
    // have the pair form in the two-list kernels: 2 x 48 .. 72 registers of row and query; shorter beams take the generic instances
    // here -- 384-byte rows have pair-form list instances of their own, dispatched before this function: walk_wide3.hip)
    constexpr bool kBigOnly = STEPS >= 24;
    constexpr int kListSteps = kBigOnly ? 0 : STEPS;
    constexpr int kBigSteps = kBigOnly ? STEPS : kWideSteps48;
    if (p.ef <= 64) return launch_reg_t<METRIC, kListSteps, 1>(p, retry, lds, s);
    if (p.ef <= kHot2MaxEf) return launch_reg_t<METRIC, kBigOnly ? 0 : kWideSteps48, 2>(p, retry, lds, s);  // (12- / 16-step rows keep the unrolled distance)
    return launch_reg_t<METRIC, kBigSteps, 4>(p, retry, lds, s);  // (R >= 4: the two-list kernels, one instance for every ef up to 512)
}

}  // namespace gbnns
