// walk_bitmap.hip -- the first pass with the visited sets as bitmaps in HBM (large ef on deep batches): persistent
// wavefronts, one bitmap slot each; launch and sizing.
#include "launch_util.h"
#include "walk_generic.h"

namespace gbnns {

// The bitmap first pass runs the register-list (ef <= 128, L2) / two-list (128 < ef <= 1 024, both metrics) walk for
// 128-byte rows of a compact index, the two-list walk for 256-byte rows with L2, else the LDS-list walk.
bool walk_bitmap_uses_reg(const WalkParams& p, int metric) {
    const bool rows128 = p.dim == 32u && p.dstride == 32u, rows256 = p.dim == 64u && p.dstride == 64u && metric == 0 && p.ef > kHot2MaxEf;
    const bool rows576 = p.dim == 144u && p.dstride == 144u && metric == 0 && p.ef > kHot2MaxEf;  // (the reference's glove 300 -> 144)
    return (metric == 0 || p.ef > kHot2MaxEf) && p.ef <= kRegListMaxEf && (rows128 || rows256 || rows576) && walk_off32(p) && !p.aux_ell;
}

// LDS of the bitmap first pass: result list (or merge buffer) + tie list + query (no visited table)
size_t walk_bitmap_lds_bytes(const WalkParams& p, int metric) {
    const bool reg = walk_bitmap_uses_reg(p, metric);
    // (two-list kernel: the re-rank query cannot overlay the base list it reads its candidates from)
    return walk_fast_lds_fixed_bytes(p.ef, p.dstride, false, !reg) + (reg && p.ef > kHot2MaxEf ? p.rr_reserve : 0u);
}

// Room the fused re-rank has for the original-space query (it is staged once the walk is over).
size_t walk_rr_room(const WalkParams& p, int metric, bool hot, bool bitmap_pass) {
    if (bitmap_pass) return walk_bitmap_uses_reg(p, metric) && p.ef > kHot2MaxEf ? (size_t)p.rr_reserve : walk_bitmap_lds_bytes(p, metric);
    if (p.ef > kHot2MaxEf && !walk_uses_lds_list(p)) return walk_hash_bytes(p.hash_cap, walk_hash_form(p, hot));  // two-list kernels: the visited-set area
    return walk_fast_lds_bytes(p, hot);
}

template <int R>
static hipError_t launch_bitmap_reg(const WalkParams& p, unsigned slots, size_t lds, hipStream_t s) {
    hipError_t e = set_lds(walk_bitmap_reg_kernel<0, R>, lds);
    if (e != hipSuccess) return e;
    g_walk_first_fn = reinterpret_cast<const void*>(walk_bitmap_reg_kernel<0, R>);
    hipLaunchKernelGGL((walk_bitmap_reg_kernel<0, R>), dim3(slots), dim3(64), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_walk_bitmap(const WalkParams& p, int metric, unsigned slots, hipStream_t s) {
    if (p.nq == 0) return hipSuccess;
    const size_t lds = walk_bitmap_lds_bytes(p, metric);
    if (walk_bitmap_uses_reg(p, metric)) {
        if (p.ef <= 64) return launch_bitmap_reg<1>(p, slots, lds, s);
        if (p.ef <= kHot2MaxEf) return launch_bitmap_reg<2>(p, slots, lds, s);
        // (adjacency rows of one 32-slot pass -- the common case -- take the instance without the pass loop)
        auto go = [&](auto kernel) -> hipError_t {
            hipError_t e = set_lds(kernel, lds);
            if (e != hipSuccess) return e;
            g_walk_first_fn = reinterpret_cast<const void*>(kernel);
            hipLaunchKernelGGL(kernel, dim3(slots), dim3(64), lds, s, p);
            return hipGetLastError();
        };
        const bool one = p.ell_stride <= 32u;
        if (p.dim == 64u) return one ? go(walk_bitmap_big_kernel<0, 16, true>) : go(walk_bitmap_big_kernel<0, 16, false>);  // 256-byte rows, L2 (pair form)
        if (p.dim == 144u) {  // 576-byte rows, L2 (pair form; WalkParams::late_rows: the rows after the bit test)
            if (p.late_rows) return one ? go(walk_bitmap_big_kernel<0, 36, true, true>) : go(walk_bitmap_big_kernel<0, 36, false, true>);
            return one ? go(walk_bitmap_big_kernel<0, 36, true>) : go(walk_bitmap_big_kernel<0, 36, false>);
        }
        if (metric == 1) return one ? go(walk_bitmap_big_kernel<1, 8, true>) : go(walk_bitmap_big_kernel<1, 8, false>);
        return one ? go(walk_bitmap_big_kernel<0, 8, true>) : go(walk_bitmap_big_kernel<0, 8, false>);
    }
    if (metric == 1) {
        hipError_t e = set_lds(walk_bitmap_kernel<1, 0>, lds);
        if (e != hipSuccess) return e;
        g_walk_first_fn = reinterpret_cast<const void*>(walk_bitmap_kernel<1, 0>);
        hipLaunchKernelGGL((walk_bitmap_kernel<1, 0>), dim3(slots), dim3(64), lds, s, p);
    } else if (p.dstride == p.dim && p.dim == 32) {
        hipError_t e = set_lds(walk_bitmap_kernel<0, 8>, lds);
        if (e != hipSuccess) return e;
        g_walk_first_fn = reinterpret_cast<const void*>(walk_bitmap_kernel<0, 8>);
        hipLaunchKernelGGL((walk_bitmap_kernel<0, 8>), dim3(slots), dim3(64), lds, s, p);
    } else {
        hipError_t e = set_lds(walk_bitmap_kernel<0, 0>, lds);
        if (e != hipSuccess) return e;
        g_walk_first_fn = reinterpret_cast<const void*>(walk_bitmap_kernel<0, 0>);
        hipLaunchKernelGGL((walk_bitmap_kernel<0, 0>), dim3(slots), dim3(64), lds, s, p);
    }
    return hipGetLastError();
}

}  // namespace gbnns
