// walk_wide3.hip -- the L2 register-list walks over 384-byte rows in the pair form (the reference's PLAIN walks over deep (d = 96) vectors at
// efs_hnsw of up to 128, final_test.cpp:84): first pass of a compact index over adjacency rows of one or two passes; every other case stays on the
// run-time-length instances (same LDS layout).  A unit of its own so that the instantiations build in parallel.
#include "walk_launch.h"

namespace gbnns {

hipError_t launch_walk_wide2_list(const WalkParams& p, hipStream_t s) {
    const size_t lds = walk_fast_lds_bytes(p, false);
    // adjacency rows of two passes (the reference's M18 hnsw graph over the deep vectors: up to 36 slots): the one-register list only
    // (GD(M = 30) graph, rows of up to 45 slots, ef 40: 0.678 against 0.763 ms a lane per row; the two-register list 1.254 / 1.834 at ef 80 / 120
    // against 1.255 / 1.814 -- a lane per row takes such a row in one pass -- and stays on the run-time-length instance: the caller checks)
    if (p.ell_stride > 32u) return launch_walk_k(walk_reg_kernel<0, 24, true, false, 1, false>, p, false, lds, s);
    return p.ef <= 64 ? launch_walk_k(walk_reg_kernel<0, 24, true, false, 1, true>, p, false, lds, s)
                      : launch_walk_k(walk_reg_kernel<0, 24, true, false, 2, true>, p, false, lds, s);
}

}  // namespace gbnns
