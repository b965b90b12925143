// rerank.hip -- the stand-alone re-rank kernels (getRealNearest, search_function.h:105-125): gbnns_rerank, d % 8 != 0 and
// GBNNS_FLAG_NO_FUSED_RERANK; the walk kernels re-rank their own query through the same core (walk_common.h).
#include "launch_util.h"
#include "walk_common.h"

namespace gbnns {

namespace {

// ------------------------------------------------------------------------------------------
// re-rank (search_function.h:105-125 getRealNearest)
// ------------------------------------------------------------------------------------------
// One candidate per lane; each lane streams its own row with 16-B loads against the query staged
// in LDS.  Winner = strict minimum in pop order  <=>  min over (distance, pop index).

template <int METRIC>
__global__ __launch_bounds__(64) void rerank_kernel(RerankParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const uint32_t qi = blockIdx.x;
    float* qf = reinterpret_cast<float*>(smem);
    const float4* qs = reinterpret_cast<const float4*>(qf);
    for (uint32_t i = lane; i < p.dstride; i += 64)
        qf[i] = (i < p.dim) ? p.q[(size_t)qi * p.qstride + i] : 0.f;
    wave_sync();
    const int cnt = p.count[qi];
    const uint32_t* cand = p.cand + (size_t)qi * p.cand_stride;
    uint64_t bestk = ~0ull;
    for (int base = 0; base < cnt; base += 64) {
        const int r = base + lane;
        if (r < cnt) {
            uint32_t id = cand[r];
            id = id < p.n ? id : 0u;  // (never dereference an id outside the table)
            const float dv = metric_dist<METRIC>(
                reinterpret_cast<const float4*>(p.db + (size_t)id * p.dstride), qs, p.dim);
            const uint64_t kv = ((uint64_t)fkey(dv) << 32) | (uint32_t)r;
            bestk = kv < bestk ? kv : bestk;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t o = shfl_u64(bestk, lane ^ off);
        bestk = o < bestk ? o : bestk;
    }
    if (lane == 0) p.out[qi] = (cnt > 0) ? cand[(uint32_t)(bestk & 0xFFFFFFFFu)] : kInvalidId;
}

template <int METRIC>
__global__ __launch_bounds__(64) void rerank_pair_kernel(RerankParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const uint32_t qi = blockIdx.x;
    const int cnt = p.count[qi];
    const uint32_t* cand = p.cand + (size_t)qi * p.cand_stride;
    RerankSrc a{p.q, p.qstride, p.db, p.dstride, p.dim, p.n};
    const int win = (METRIC == 0 && p.dim >= 384u)
                        ? rerank_pairs_core<METRIC, 24>(a, qi, cnt, reinterpret_cast<float*>(smem), lane, [&](int r) { return cand[r]; })
                        : rerank_pairs_core<METRIC>(a, qi, cnt, reinterpret_cast<float*>(smem), lane, [&](int r) { return cand[r]; });
    if (lane == 0) p.out[qi] = (win >= 0) ? cand[win] : kInvalidId;
}

}  // namespace

hipError_t launch_rerank(const RerankParams& p, int metric, hipStream_t s) {
    if (p.nq == 0) return hipSuccess;
    const size_t lds = (size_t)p.dstride * 4;
    // pair form: both metrics at dim % 8 == 0; L2 at dim % 8 == 4 too (the last 16-byte step is the even lane's alone)
    const bool pairs = p.dim > 0 && (p.dim % 8 == 0 || (metric == 0 && p.dim % 4 == 0));
    hipError_t e;
    if (metric == 1) {
        if (pairs) {
            e = set_lds(rerank_pair_kernel<1>, lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((rerank_pair_kernel<1>), dim3(p.nq), dim3(64), lds, s, p);
        } else {
            e = set_lds(rerank_kernel<1>, lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((rerank_kernel<1>), dim3(p.nq), dim3(64), lds, s, p);
        }
    } else if (pairs) {
        e = set_lds(rerank_pair_kernel<0>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((rerank_pair_kernel<0>), dim3(p.nq), dim3(64), lds, s, p);
    } else {
        e = set_lds(rerank_kernel<0>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((rerank_kernel<0>), dim3(p.nq), dim3(64), lds, s, p);
    }
    return hipGetLastError();
}

}  // namespace gbnns
