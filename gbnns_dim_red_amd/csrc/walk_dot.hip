// walk_dot.hip -- the generic walks with the negative-dot metric (Angular::Dist).
#include "walk_launch.h"

namespace gbnns {

hipError_t launch_walk_dot(const WalkParams& p, bool retry, hipStream_t s) {
    if (p.dstride == p.dim && p.dim == 32) return launch_fast_t<1, 8>(p, retry, s);  // 128-byte rows: pair form
    return launch_fast_t<1, 0>(p, retry, s);
}

}  // namespace gbnns
