// gd_order.hip -- GD pruning of a kNN graph on the device (hnswlikeGD, support_func.h:521-563) and the locality order of deep
// query batches (counting sort on sign bits of the walked-space query).
#include "launch_util.h"
#include "walk_common.h"

namespace gbnns {

namespace {

// ------------------------------------------------------------------------------------------
// GD pruning of a kNN graph (support_func.h:521-563 hnswlikeGD, need_const_degree = false), per node
// ------------------------------------------------------------------------------------------
// One wavefront per node i.  (1) every lane scores its share of the candidate list (Dist(i, c), candidates at
// distance <= 1e-10 dropped, :535); (2) the (distance, list position) keys are sorted in LDS (bitonic) -- the
// reference calls std::sort on the distance alone, which leaves the order of EQUAL distances to the library's
// algorithm, so a node whose list contains equal distances is not decided here: it is flagged for the host,
// which runs that very std::sort (gbnns_internal_gd_finish); (3) greedy pruning in sorted order: candidate c is
// kept iff for every neighbour g kept so far  !(Dist(c, i) + eps > Dist(c, g))  (:548-552) -- the kept
// neighbours sit one per lane, so one candidate costs one wave-wide distance and a ballot; stops at M kept
// (:555); (4) the M/2 nearest are always linked (:559-563).  Lists longer than kGdMaxList or M > 64 go to the host.
constexpr int kGdMaxList = 1024;

template <int METRIC>
__global__ __launch_bounds__(64) void gd_prune_kernel(GdParams p) {
    __shared__ uint64_t keys[kGdMaxList];
    __shared__ __attribute__((aligned(16))) float pi_s[132];
    const int lane = lane_id();
    const uint32_t i = blockIdx.x;
    const uint64_t o0 = p.knn_off[i], o1 = p.knn_off[i + 1];
    const uint32_t cnt = (uint32_t)(o1 - o0);
    uint32_t* g = p.adj + (size_t)i * (2u * p.M);
    if (cnt > (uint32_t)kGdMaxList) {
        if (lane == 0) p.deg[i] = 0xFFFFFFFFu;
        return;
    }
    for (uint32_t t = lane; t < p.dstride; t += 64) pi_s[t] = p.ds[(size_t)i * p.dstride + t];
    wave_sync();
    const float4* pi4 = reinterpret_cast<const float4*>(pi_s);
    const float eps = 1e-10f;
    // (1) scores
    uint32_t npow = 64;
    while (npow < cnt) npow <<= 1;
    bool bad = false;
    for (uint32_t j = lane; j < npow; j += 64) {
        uint64_t key = ~0ull;
        if (j < cnt) {
            const uint32_t c = p.knn_nbr[o0 + j];
            if (c >= p.n) bad = true;
            else {
                const float dc = metric_dist<METRIC>(pi4, reinterpret_cast<const float4*>(p.ds + (size_t)c * p.dstride), p.dim);
                if (dc > eps) key = ((uint64_t)fkey(dc) << 32) | j;
            }
        }
        keys[j] = key;
    }
    if (__ballot(bad)) {  // an id outside the set: the host reports it
        if (lane == 0) p.deg[i] = 0xFFFFFFFFu;
        return;
    }
    wave_sync();
    // (2) bitonic sort, ascending (all-ones = dropped candidates, at the end)
    for (uint32_t k = 2; k <= npow; k <<= 1) {
        for (uint32_t jj = k >> 1; jj > 0; jj >>= 1) {
            for (uint32_t t = lane; t < npow; t += 64) {
                const uint32_t x = t ^ jj;
                if (x > t) {
                    const uint64_t a = keys[t], b = keys[x];
                    const bool up = (t & k) == 0;
                    if ((a > b) == up) { keys[t] = b; keys[x] = a; }
                }
            }
            wave_sync();
        }
    }
    // equal distances among the kept candidates -> host
    bool tie = false;
    uint32_t valid = 0;
    for (uint32_t t = lane; t < npow; t += 64) {
        const uint64_t a = keys[t];
        if (a != ~0ull) {
            valid += 1;
            if (t + 1 < npow) {
                const uint64_t b = keys[t + 1];
                if (b != ~0ull && (uint32_t)(a >> 32) == (uint32_t)(b >> 32)) tie = true;
            }
        }
    }
    if (__ballot(tie)) {
        if (lane == 0) p.deg[i] = 0xFFFFFFFFu;
        return;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) valid += (uint32_t)__shfl_xor((int)valid, off);
    if (valid == 0) {
        if (lane == 0) p.deg[i] = 0u;
        return;
    }
    // (3) greedy pruning; lane l holds kept neighbour l (its id in `mine`)
    uint32_t m = 1;
    uint32_t mine = kInvalidId;
    {
        const uint32_t id0 = p.knn_nbr[o0 + (uint32_t)keys[0]];
        if (lane == 0) mine = id0;
    }
    for (uint32_t j = 1; j < valid && m < (uint32_t)p.M; ++j) {
        const uint64_t kv = keys[j];
        const uint32_t c = p.knn_nbr[o0 + (uint32_t)kv];
        const float dci = fkey_inv((uint32_t)(kv >> 32));  // Dist(c, i): the same bits as Dist(i, c) (the sums are symmetric)
        bool closer_to_kept = false;
        if ((uint32_t)lane < m) {
            const float dl = metric_dist<METRIC>(reinterpret_cast<const float4*>(p.ds + (size_t)c * p.dstride),
                                                 reinterpret_cast<const float4*>(p.ds + (size_t)mine * p.dstride), p.dim);
            closer_to_kept = dci + eps > dl;
        }
        if (!__ballot(closer_to_kept)) {
            if ((uint32_t)lane == m) mine = c;
            m += 1;
        }
    }
    // kept list in order, then (4) the M/2 nearest that are missing
    if ((uint32_t)lane < m) g[lane] = mine;
    wave_sync();
    uint32_t deg = m;
    for (uint32_t j = 0; j < (uint32_t)p.M / 2u && j < valid; ++j) {
        const uint32_t c = p.knn_nbr[o0 + (uint32_t)keys[j]];
        bool have = false;
        for (uint32_t t = lane; t < deg; t += 64) have |= g[t] == c;
        if (!__ballot(have)) {
            if (lane == 0) g[deg] = c;
            deg += 1;
            wave_sync();
        }
    }
    if (lane == 0) p.deg[i] = deg;
}

// ------------------------------------------------------------------------------------------
// locality order of a deep batch (WalkParams::order)
// ------------------------------------------------------------------------------------------
// key = the sign bits of the first `bits` (10 .. 16, default 12) coordinates of the query in the walked space (queries of
// one bucket lie in one orthant, their walks end in the same region); counting sort in three small launches -- histogram,
// scan of the counters by one workgroup, scatter (the order inside a bucket is whatever the atomics give: it
// does not matter).  Only the ORDER of the work changes; answers go to the queries' own output slots.
__device__ __forceinline__ uint32_t order_key(const float* q, uint32_t bits) {  // bits <= dim
    uint32_t k = 0;
    for (uint32_t j = 0; j < bits; ++j) k |= (q[j] > 0.f ? 1u : 0u) << (bits - 1u - j);  // coordinate 0 = most significant
    return k;
}

__global__ __launch_bounds__(256) void order_hist_kernel(const float* q, uint32_t qstride, uint32_t bits, uint32_t nq, uint32_t* hist) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < nq) atomicAdd(&hist[order_key(q + (size_t)i * qstride, bits)], 1u);
}

// exclusive scan of the 2^bits counters (a multiple of 1 024) in place: they become the buckets' cursors
__global__ __launch_bounds__(1024) void order_scan_kernel(uint32_t* hist, uint32_t per) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    uint32_t sum = 0;
    for (uint32_t j = 0; j < per; ++j) sum += hist[per * t + j];
    part[t] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        const uint32_t add = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    uint32_t base = part[t] - sum;
    for (uint32_t j = 0; j < per; ++j) {
        const uint32_t v = hist[per * t + j];
        hist[per * t + j] = base;
        base += v;
    }
}

__global__ __launch_bounds__(256) void order_scatter_kernel(const float* q, uint32_t qstride, uint32_t bits, uint32_t nq, uint32_t* cursor,
                                                            uint32_t* order) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < nq) order[atomicAdd(&cursor[order_key(q + (size_t)i * qstride, bits)], 1u)] = i;
}
}  // namespace

hipError_t launch_gd_prune(const GdParams& p, int metric, hipStream_t s) {
    if (p.n == 0) return hipSuccess;
    if (metric == 1) hipLaunchKernelGGL((gd_prune_kernel<1>), dim3((unsigned)p.n), dim3(64), 0, s, p);
    else hipLaunchKernelGGL((gd_prune_kernel<0>), dim3((unsigned)p.n), dim3(64), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_query_order(const float* q, uint32_t qstride, uint32_t dim, uint32_t nq, uint32_t bits, uint32_t* hist, uint32_t* order,
                              hipStream_t s) {
    if (nq == 0) return hipSuccess;
    bits = bits < 10u ? 10u : (bits > 16u ? 16u : bits);
    if (dim < bits) return hipErrorInvalidValue;  // (the caller orders only when the walked space has that many coordinates)
    hipError_t e = hipMemsetAsync(hist, 0, (size_t)4 << bits, s);
    if (e != hipSuccess) return e;
    const unsigned grid = (nq + 255u) / 256u;
    hipLaunchKernelGGL(order_hist_kernel, dim3(grid), dim3(256), 0, s, q, qstride, bits, nq, hist);
    hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(1024), 0, s, hist, (1u << bits) / 1024u);
    hipLaunchKernelGGL(order_scatter_kernel, dim3(grid), dim3(256), 0, s, q, qstride, bits, nq, hist, order);
    return hipGetLastError();
}

}  // namespace gbnns
