// walk_wide.hip -- the L2 walks over 192-, 256- and 576-byte rows (d_low = 48 / 64 / 144: the reference's deep row, the GIST shape, the
// reference's glove row).
#include "walk_launch.h"

namespace gbnns {

hipError_t launch_walk_wide(const WalkParams& p, int steps, bool retry, hipStream_t s) {
    if (steps == 36) return launch_fast_t<0, 36>(p, retry, s);
    return steps == 12 ? launch_fast_t<0, 12>(p, retry, s) : launch_fast_t<0, 16>(p, retry, s);
}

}  // namespace gbnns
