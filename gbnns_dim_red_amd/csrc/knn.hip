// knn.hip -- exact brute-force k-nearest-neighbour scan for gfx950 (MI355X).
//
// What it replaces in the reference: the brute-force ground truth getTruth (search/support_func.h:270-290,
// k = 1, strict `<` over ascending ids) and the exact kNN lists that feed the graph builder
// (dim_red/support_func.py:374-384 get_nearestneighbors_partly -> `<name>_knn_1k_<style>.ivecs`, read by
// search/prepare_graph.cpp:66).  Result of query i: the k smallest (Dist(base_j, q_i), j) pairs in ascending
// pair order -- distances in the reference's own arithmetic (L2Metric::Dist / Angular::Dist, support_func.h:
// 107-163: same running sums, same order, one rounding per operation), ties broken towards the lower id,
// which is what getTruth's strict `<` over ascending j does.
//
// Shape of the work: n_q x n distances of d dims, every one in a fixed summation order, so this is VALU work
// (sub / mul / add, packed two wide), not an MFMA GEMM: a matrix-core product would round differently.
// One thread owns QPT queries held in registers (d <= 128; wider rows: knn_scan_wide_kernel below); a workgroup of 128 threads streams the base set
// through LDS in tiles of 64 rows, every thread reading the same row (LDS broadcast reads, one row serves
// QPT x 128 queries).  Selection: a max-heap of k (distance key, id) pairs per query in global memory
// ([k][n_q] so that lanes touching the same heap level coalesce); a row is offered only when it beats the
// heap's root (kept in a register), which after the first few thousand rows is rare
// (about k * ln(n / k) times per query), then the heap is sorted in place.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace gbnns {

namespace {

__device__ __forceinline__ uint32_t knn_fkey(float x) {  // order-preserving map float -> u32 (-0 == +0)
    x = x + 0.0f;
    const uint32_t b = __float_as_uint(x);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float knn_fkey_inv(uint32_t k) {
    const uint32_t b = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
    return __uint_as_float(b);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kKnnThreads = 128;
constexpr int kKnnTile = 64;  // base rows per LDS tile

// Replaces the root of the max-heap h[0..k) (stride `st` elements between levels) by `key` and restores the
// heap; returns the new root.
__device__ __forceinline__ uint64_t heap_replace_root(uint64_t* h, size_t st, int k, uint64_t key) {
    int i = 0;
    while (true) {
        const int l = 2 * i + 1;
        if (l >= k) break;
        uint64_t cv = h[(size_t)l * st];
        int c = l;
        if (l + 1 < k) {
            const uint64_t cr = h[(size_t)(l + 1) * st];
            if (cr > cv) { cv = cr; c = l + 1; }
        }
        if (cv <= key) break;
        h[(size_t)i * st] = cv;
        i = c;
    }
    h[(size_t)i * st] = key;
    return h[0];
}

// S = 16-byte steps of a query held in registers (d <= 4 * S), QPT = queries per thread.
template <int METRIC, int S, int QPT>
__global__ __launch_bounds__(kKnnThreads) void knn_scan_kernel(KnnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4* tile = reinterpret_cast<float4*>(smem);  // [kKnnTile][S]
    const int t = threadIdx.x;
    // L2Metric::Dist uses the first 4 * floor(d / 4) dims only; the dot form is offered for d % 8 == 0
    const uint32_t steps = p.dim >> 2;
    const bool vec_ok = (p.bstride & 3u) == 0;   // rows 16-B aligned -> float4 loads
    const bool qvec_ok = (p.qstride & 3u) == 0;

    float4 q[QPT][S];
    uint32_t qi[QPT];
    bool live[QPT];
    uint64_t root[QPT];
#pragma unroll
    for (int a = 0; a < QPT; ++a) {
        qi[a] = (blockIdx.x * QPT + a) * kKnnThreads + t;
        live[a] = qi[a] < p.nq;
        root[a] = ~0ull;  // the heap starts filled with all-ones keys
        const float* qp = p.q + (size_t)(live[a] ? qi[a] : 0u) * p.qstride;
#pragma unroll
        for (int c = 0; c < S; ++c) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((uint32_t)c < steps) {
                if (qvec_ok) v = reinterpret_cast<const float4*>(qp)[c];
                else v = make_float4(qp[4 * c], qp[4 * c + 1], qp[4 * c + 2], qp[4 * c + 3]);
            }
            q[a][c] = v;
        }
    }

    for (uint64_t base0 = 0; base0 < p.n; base0 += kKnnTile) {
        __syncthreads();  // everyone is done with the previous tile
        for (int e = t; e < kKnnTile * S; e += kKnnThreads) {
            const int r = e / S, c = e % S;
            const uint64_t row = base0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < p.n && (uint32_t)c < steps) {
                const float* bp = p.base + (size_t)row * p.bstride;
                if (vec_ok) v = reinterpret_cast<const float4*>(bp)[c];
                else v = make_float4(bp[4 * c], bp[4 * c + 1], bp[4 * c + 2], bp[4 * c + 3]);
            }
            tile[e] = v;
        }
        __syncthreads();
        const int rows = (p.n - base0 < (uint64_t)kKnnTile) ? (int)(p.n - base0) : kKnnTile;
        for (int r = 0; r < rows; ++r) {
            float dist[QPT];
            if constexpr (METRIC == 0) {
                // two-wide vectors so that the compiler emits v_pk_add_f32 / v_pk_mul_f32 (each lane of a packed
                // instruction is an ordinary IEEE operation: same results as the scalar form)
                f32x2 s01[QPT], s23[QPT];
#pragma unroll
                for (int a = 0; a < QPT; ++a) s01[a] = s23[a] = f32x2{0.f, 0.f};
#pragma unroll
                for (int c = 0; c < S; ++c) {
                    const float4 b = tile[r * S + c];  // same address in every lane: LDS broadcast
                    const f32x2 b01{b.x, b.y}, b23{b.z, b.w};
#pragma unroll
                    for (int a = 0; a < QPT; ++a) {  // support_func.h:113-124: e = a - b; sum += e * e, four lanes
                        const f32x2 e01 = b01 - f32x2{q[a][c].x, q[a][c].y};
                        const f32x2 e23 = b23 - f32x2{q[a][c].z, q[a][c].w};
                        s01[a] = s01[a] + e01 * e01;
                        s23[a] = s23[a] + e23 * e23;
                    }
                }
#pragma unroll
                for (int a = 0; a < QPT; ++a) dist[a] = ((s01[a].x + s01[a].y) + s23[a].x) + s23[a].y;  // :125-126
            } else {
                f32x2 cs[QPT][4];  // eight running sums (k mod 8) as four pairs: {0,1} {2,3} {4,5} {6,7}
#pragma unroll
                for (int a = 0; a < QPT; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) cs[a][j] = f32x2{0.f, 0.f};
#pragma unroll
                for (int c = 0; c < S; ++c) {
                    const float4 b = tile[r * S + c];
                    const f32x2 b01{b.x, b.y}, b23{b.z, b.w};
                    const int o = (c & 1) * 2;  // support_func.h:140-147
#pragma unroll
                    for (int a = 0; a < QPT; ++a) {
                        cs[a][o + 0] = cs[a][o + 0] + b01 * f32x2{q[a][c].x, q[a][c].y};
                        cs[a][o + 1] = cs[a][o + 1] + b23 * f32x2{q[a][c].z, q[a][c].w};
                    }
                }
#pragma unroll
                for (int a = 0; a < QPT; ++a) {
                    const f32x2 m01 = cs[a][2] + cs[a][0], m23 = cs[a][3] + cs[a][1];  // :148 hi half onto lo half
                    dist[a] = -((m01.x + m01.y) + (m23.x + m23.y));                     // :160-162
                }
            }
            const uint64_t row = base0 + r;
#pragma unroll
            for (int a = 0; a < QPT; ++a) {
                const uint64_t key = ((uint64_t)knn_fkey(dist[a]) << 32) | (uint32_t)row;
                const bool self = p.self_offset >= 0 && row == (uint64_t)qi[a] + (uint64_t)p.self_offset;
                if (live[a] && !self && key < root[a])
                    root[a] = heap_replace_root(p.heap + qi[a], p.heap_stride, p.k, key);
            }
        }
    }

    // heap sort in place (ascending), then the outputs; every lane runs the same trip counts
#pragma unroll
    for (int a = 0; a < QPT; ++a) {
        if (!live[a]) continue;
        uint64_t* h = p.heap + qi[a];
        const size_t st = p.heap_stride;
        for (int m = p.k - 1; m > 0; --m) {
            const uint64_t last = h[(size_t)m * st];
            h[(size_t)m * st] = h[0];
            heap_replace_root(h, st, m, last);
        }
        for (int e = 0; e < p.k; ++e) {
            const uint64_t kv = h[(size_t)e * st];
            const uint32_t id = (uint32_t)kv;
            p.out_ids[(size_t)qi[a] * p.k + e] = id;  // all-ones (never offered) -> 0xFFFFFFFF: fewer than k rows
            if (p.out_dist)
                p.out_dist[(size_t)qi[a] * p.k + e] = (kv == ~0ull) ? __builtin_inff() : knn_fkey_inv((uint32_t)(kv >> 32));
        }
    }
}

// Wide rows (d > 128): a query no longer fits a thread's registers next to its running sums, so the loops are
// turned around.  A thread still owns one query, but walks the base set in tiles of R rows whose running sums
// (R x 4 for L2, R x 8 for the dot form) it keeps in registers while it streams its own query through in chunks
// of eight 16-byte steps (read from global memory: its row is re-read once per tile and stays in the caches);
// the tile [R][steps rounded up to 8, zero filled] sits in LDS and is read by broadcast.  Per distance the
// operations and their order are exactly those of the narrow kernel (a zero-filled step adds +0 to sums that
// are never -0).  Offers to the heap happen in ascending row order, as before.
template <int METRIC, int R>
__global__ __launch_bounds__(kKnnThreads) void knn_scan_wide_kernel(KnnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4* tile = reinterpret_cast<float4*>(smem);  // [R][steps8]
    const int t = threadIdx.x;
    const uint32_t steps = p.dim >> 2;
    const uint32_t steps8 = (steps + 7u) & ~7u;
    const bool vec_ok = (p.bstride & 3u) == 0;
    const bool qvec_ok = (p.qstride & 3u) == 0;
    const uint32_t qi = blockIdx.x * kKnnThreads + t;
    const bool live = qi < p.nq;
    const float* qp = p.q + (size_t)(live ? qi : 0u) * p.qstride;
    uint64_t root = ~0ull;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    for (uint64_t base0 = 0; base0 < p.n; base0 += R) {
        __syncthreads();  // everyone is done with the previous tile
        for (uint32_t e = t; e < (uint32_t)R * steps8; e += kKnnThreads) {
            const uint32_t r = e / steps8, c = e % steps8;
            const uint64_t row = base0 + r;
            float4 v = zero4;
            if (row < p.n && c < steps) {
                const float* bp = p.base + (size_t)row * p.bstride;
                if (vec_ok) v = reinterpret_cast<const float4*>(bp)[c];
                else v = make_float4(bp[4 * c], bp[4 * c + 1], bp[4 * c + 2], bp[4 * c + 3]);
            }
            tile[e] = v;
        }
        __syncthreads();
        float dist[R];
        if constexpr (METRIC == 0) {
            f32x2 s01[R], s23[R];
#pragma unroll
            for (int r = 0; r < R; ++r) s01[r] = s23[r] = f32x2{0.f, 0.f};
            for (uint32_t c0 = 0; c0 < steps8; c0 += 8) {
                float4 qv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t c = c0 + j;
                    qv[j] = zero4;
                    if (c < steps) {
                        if (qvec_ok) qv[j] = reinterpret_cast<const float4*>(qp)[c];
                        else qv[j] = make_float4(qp[4 * c], qp[4 * c + 1], qp[4 * c + 2], qp[4 * c + 3]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x2 q01{qv[j].x, qv[j].y}, q23{qv[j].z, qv[j].w};
#pragma unroll
                    for (int r = 0; r < R; ++r) {  // support_func.h:113-124: e = a - b; sum += e * e, four lanes
                        const float4 b = tile[r * steps8 + c0 + j];  // same address in every lane: LDS broadcast
                        const f32x2 e01 = f32x2{b.x, b.y} - q01;
                        const f32x2 e23 = f32x2{b.z, b.w} - q23;
                        s01[r] = s01[r] + e01 * e01;
                        s23[r] = s23[r] + e23 * e23;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) dist[r] = ((s01[r].x + s01[r].y) + s23[r].x) + s23[r].y;  // :125-126
        } else {
            f32x2 cs[R][4];  // eight running sums (k mod 8) as four pairs: {0,1} {2,3} {4,5} {6,7}
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) cs[r][j] = f32x2{0.f, 0.f};
            for (uint32_t c0 = 0; c0 < steps8; c0 += 8) {
                float4 qv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t c = c0 + j;
                    qv[j] = zero4;
                    if (c < steps) {
                        if (qvec_ok) qv[j] = reinterpret_cast<const float4*>(qp)[c];
                        else qv[j] = make_float4(qp[4 * c], qp[4 * c + 1], qp[4 * c + 2], qp[4 * c + 3]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x2 q01{qv[j].x, qv[j].y}, q23{qv[j].z, qv[j].w};
                    const int o = (j & 1) * 2;  // support_func.h:140-147 (c0 is a multiple of 8: c & 1 == j & 1)
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float4 b = tile[r * steps8 + c0 + j];
                        cs[r][o + 0] = cs[r][o + 0] + f32x2{b.x, b.y} * q01;
                        cs[r][o + 1] = cs[r][o + 1] + f32x2{b.z, b.w} * q23;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const f32x2 m01 = cs[r][2] + cs[r][0], m23 = cs[r][3] + cs[r][1];  // :148 hi half onto lo half
                dist[r] = -((m01.x + m01.y) + (m23.x + m23.y));                     // :160-162
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t row = base0 + r;
            const uint64_t key = ((uint64_t)knn_fkey(dist[r]) << 32) | (uint32_t)row;
            const bool self = p.self_offset >= 0 && row == (uint64_t)qi + (uint64_t)p.self_offset;
            if (live && row < p.n && !self && key < root)
                root = heap_replace_root(p.heap + qi, p.heap_stride, p.k, key);
        }
    }
    if (!live) return;
    uint64_t* h = p.heap + qi;
    const size_t st = p.heap_stride;
    for (int m = p.k - 1; m > 0; --m) {  // heap sort in place (ascending)
        const uint64_t last = h[(size_t)m * st];
        h[(size_t)m * st] = h[0];
        heap_replace_root(h, st, m, last);
    }
    for (int e = 0; e < p.k; ++e) {
        const uint64_t kv = h[(size_t)e * st];
        p.out_ids[(size_t)qi * p.k + e] = (uint32_t)kv;
        if (p.out_dist)
            p.out_dist[(size_t)qi * p.k + e] = (kv == ~0ull) ? __builtin_inff() : knn_fkey_inv((uint32_t)(kv >> 32));
    }
}

template <int METRIC, int R>
hipError_t launch_knn_wide(const KnnParams& p, hipStream_t s) {
    const uint32_t steps8 = ((p.dim >> 2) + 7u) & ~7u;
    const size_t lds = (size_t)R * steps8 * 16;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_scan_wide_kernel<METRIC, R>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const unsigned grid = (unsigned)((p.nq + (uint64_t)kKnnThreads - 1) / (uint64_t)kKnnThreads);
    hipLaunchKernelGGL((knn_scan_wide_kernel<METRIC, R>), dim3(grid), dim3(kKnnThreads), lds, s, p);
    return hipGetLastError();
}

template <int METRIC, int S, int QPT>
hipError_t launch_knn_t(const KnnParams& p, hipStream_t s) {
    const size_t lds = (size_t)kKnnTile * S * 16;
    const unsigned grid = (unsigned)((p.nq + (uint64_t)kKnnThreads * QPT - 1) / ((uint64_t)kKnnThreads * QPT));
    hipLaunchKernelGGL((knn_scan_kernel<METRIC, S, QPT>), dim3(grid), dim3(kKnnThreads), lds, s, p);
    return hipGetLastError();
}

template <int METRIC>
hipError_t launch_knn_m(const KnnParams& p, hipStream_t s) {
    if (p.dim <= 32) return launch_knn_t<METRIC, 8, 2>(p, s);
    if (p.dim <= 64) return launch_knn_t<METRIC, 16, 2>(p, s);
    if (p.dim <= 128) return launch_knn_t<METRIC, 32, 1>(p, s);
    // wide rows: tiles of 16 (L2) / 8 (dot) rows with their running sums in registers; the [R][d] tile must fit the
    // 160 KB of LDS (64 d bytes at R = 16), so very wide rows take fewer rows per tile: d <= 2560 / 5120 / 8192
    const size_t row_bytes = (size_t)(((p.dim >> 2) + 7u) & ~7u) * 16;
    if constexpr (METRIC == 0) {
        if (16 * row_bytes <= 160 * 1024) return launch_knn_wide<0, 16>(p, s);
    }
    if (8 * row_bytes <= 160 * 1024) return launch_knn_wide<METRIC, 8>(p, s);
    if (4 * row_bytes <= 160 * 1024) return launch_knn_wide<METRIC, 4>(p, s);
    return hipErrorInvalidValue;  // d > 10240: refused by gbnns_exact_knn (d <= 8192) before it gets here
}

}  // namespace

hipError_t launch_knn_scan(const KnnParams& p, int metric, hipStream_t s) {
    if (p.nq == 0) return hipSuccess;
    return metric == 1 ? launch_knn_m<1>(p, s) : launch_knn_m<0>(p, s);
}

}  // namespace gbnns
