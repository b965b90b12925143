// walk_wide2.hip -- the L2 two-list walks over 384- and 512-byte rows in the pair form (the reference's PLAIN walks over deep (d = 96) and
// sift (d = 128) vectors at efs_hnsw of more than 128, final_test.cpp:84); a unit of its own so that the instantiations build in parallel.
#include "walk_launch.h"

namespace gbnns {

hipError_t launch_walk_wide2(const WalkParams& p, int steps, bool retry, hipStream_t s) {
    // (the caller sends beams of 129 .. 1 024 only: the two-list instances, straight)
    const size_t lds = walk_fast_lds_bytes(p, false);
    return steps == 24 ? launch_reg_t<0, 24, 4>(p, retry, lds, s) : launch_reg_t<0, 32, 4>(p, retry, lds, s);
}

}  // namespace gbnns
