// walk_wide3.hip -- the L2 register-list walks over 384-byte rows in the pair form (the reference's PLAIN walks over deep (d = 96) vectors at
// efs_hnsw of up to 128, final_test.cpp:84): first pass of a compact index over one-pass adjacency rows; every other case stays on the
// run-time-length instances (same LDS layout).  A unit of its own so that the instantiations build in parallel.
#include "walk_launch.h"

namespace gbnns {

hipError_t launch_walk_wide2_list(const WalkParams& p, hipStream_t s) {
    const size_t lds = walk_fast_lds_bytes(p, false);
    return p.ef <= 64 ? launch_walk_k(walk_reg_kernel<0, 24, true, false, 1, true>, p, false, lds, s)
                      : launch_walk_k(walk_reg_kernel<0, 24, true, false, 2, true>, p, false, lds, s);
}

}  // namespace gbnns
