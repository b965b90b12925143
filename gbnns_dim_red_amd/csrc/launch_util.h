// launch_util.h -- host-side helpers shared by the kernel families' launchers.
#pragma once

#include "kernels.h"

namespace gbnns {

// Shape served by the walk_hot* kernels (first pass only): L2, 128-byte rows, adjacency rows of one 32-slot pass
// (walk_hotw*: 33 .. 64 slots, two passes), 32-bit byte offsets.
static bool walk_off32(const WalkParams& p) {  // "compact" index: every table the walk indexes is < 4 GiB, ids fit 24 bits
    return !p.force_wide && (uint64_t)p.n * p.dstride * 4 < (1ull << 32) && (uint64_t)p.n * p.ell_stride * 4 < (1ull << 32) &&
           (!p.aux_ell || (uint64_t)p.n * p.aux_stride * 4 < (1ull << 32)) && p.n <= 0xFFFFFFu;
}

template <typename K>
static hipError_t set_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return hipSuccess;
}

// Host function of the last first-pass walk kernel this thread launched (profiling: gbnns_profile.walk_kernel); defined in walk_l2.hip.
extern thread_local const void* g_walk_first_fn;

template <typename K>
static hipError_t launch_walk_k(K kernel, const WalkParams& p, bool retry, size_t lds, hipStream_t s) {
    hipError_t e = set_lds(kernel, lds);
    if (e != hipSuccess) return e;
    if (!retry) g_walk_first_fn = reinterpret_cast<const void*>(kernel);
    const unsigned grid = retry ? (unsigned)kRetrySlots : p.nq;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), lds, s, p);
    return hipGetLastError();
}

// The hand-laid-out instances (walk_hot.hip): the kernel of p's beam class (ef <= 64, <= 128, above), adjacency width and metric.
hipError_t launch_walk_hot(const WalkParams& p, int metric, hipStream_t s);
// the walk launchers of the dot-metric and the wide-row (192 / 256-byte, L2) translation units
hipError_t launch_walk_dot(const WalkParams& p, bool retry, hipStream_t s);
hipError_t launch_walk_wide(const WalkParams& p, int steps, bool retry, hipStream_t s);
hipError_t launch_walk_wide2(const WalkParams& p, int steps, bool retry, hipStream_t s);
hipError_t launch_walk_wide2_list(const WalkParams& p, hipStream_t s);  // 384-byte rows, ef <= 128: pair-form register-list instances (walk_wide3.hip)

}  // namespace gbnns
