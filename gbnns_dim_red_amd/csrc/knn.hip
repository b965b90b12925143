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
        root[a] = live[a] ? p.heap[qi[a]] : ~0ull;  // (all-ones in a fresh heap; an earlier slice's root otherwise)
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
            const uint64_t row = p.row0 + base0 + r;
#pragma unroll
            for (int a = 0; a < QPT; ++a) {
                const uint64_t key = ((uint64_t)knn_fkey(dist[a]) << 32) | (uint32_t)row;
                const bool self = p.self_offset >= 0 && row == (uint64_t)qi[a] + (uint64_t)p.self_offset;
                if (live[a] && !self && key < root[a])
                    root[a] = heap_replace_root(p.heap + qi[a], p.heap_stride, p.k, key);
            }
        }
    }

    if (p.keep_heap) return;
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
    uint64_t root = live ? p.heap[qi] : ~0ull;
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
            const uint64_t row = p.row0 + base0 + r;
            const uint64_t key = ((uint64_t)knn_fkey(dist[r]) << 32) | (uint32_t)row;
            const bool self = p.self_offset >= 0 && row == (uint64_t)qi + (uint64_t)p.self_offset;
            if (live && base0 + r < p.n && !self && key < root)
                root = heap_replace_root(p.heap + qi, p.heap_stride, p.k, key);
        }
    }
    if (!live || p.keep_heap) return;
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


// ------------------------------------------------------------------------------------------
// Matrix-core filter (round 4).  The kNN lists that feed the graph builder are the one stage of this path where
// the reference has no arithmetic of its own to match: they come from faiss / torch.mm (dim_red/support_func.py:20-74,
// 374-384).  What gbnns_exact_knn promises is the k smallest (L2Metric::Dist, id) pairs -- so an approximate distance
// with a KNOWN error bound may decide which rows are worth an exact distance, as long as no row that belongs to the
// answer can be dropped:
//   a(q, x) = |q|^2 + |x|^2 - 2 q.x  with q.x on the matrix cores (v_mfma_f32_32x32x16_bf16, every float split into
//   bf16 hi + lo: q.x ~ qh.xh + qh.xl + ql.xh), |a - Dist| <= eps(q, x) := c (|q|^2 + |x|^2), c = 2^-12 (bound below);
//   a row is kept when a <= T_q + eps, T_q = the query's current exact k-th distance (its heap's root).  A dropped row has
//   Dist >= a - eps > T_q >= the final k-th distance: it is not among the k smallest (distance, id) pairs, ties included.
//   Kept rows get their exact distance in the reference's order (knn_rescore_kernel) and are offered to the same heap
//   as before: the heap ends up holding the same k keys as the full scan's, whatever the order of the offers.
// Error bound: x = xh + xl + r with |xl| <= 2^-8 |x|, |r| <= 2^-16 |x| (two round-to-nearest bf16 steps); the three
// products kept are exact in f32, the three dropped ones sum to <= 3 * 2^-16 |q_k x_k| per dimension; f32 accumulation of
// 3 d products <= 3 d 2^-24 sum |q_k x_k|; sum |q_k x_k| <= |q||x| <= (|q|^2 + |x|^2) / 2.  For d <= 128 the dot
// product is off by <= 3.8e-5 (|q|^2 + |x|^2), a by twice that plus the norms' own rounding (d 2^-24 each) plus the
// reference distance's rounding against the real value (<= (d / 4 + 3) 2^-23 Dist): < 9e-5 (|q|^2 + |x|^2) so far.
// (Round-4 review: one term was missing here.)  The accumulator does not start at zero but at -0.5 (1 - c) |x|^2, so every
// one of the <= 3 d / 16 accumulate steps of the matrix pipe rounds against |x|^2 / 2 + sum |q_k x_k| <= |q|^2 + |x|^2, not
// against sum |q_k x_k| alone -- and its internal adds may truncate (2^-23) instead of rounding to nearest: with 24 steps at
// d = 128 that adds <= 24 * 2^-23 (|q|^2 + |x|^2) = 2.9e-6 on the dot product's scale, 5.7e-6 on a's ... taken with the
// worst case of every other term on top of each other the error on a stays < 1.6e-4 (|q|^2 + |x|^2): c = 2^-12 = 2.44e-4
// leaves a factor 1.5, not the 2.7 claimed before -- and none to spare for longer rows: the shape guard below (dp <= 128)
// and the host's (d <= 128, gbnns_exact_knn) are what the bound rests on.
// ------------------------------------------------------------------------------------------
// the slack c must cover 4 dp 2^-23 (accumulation on the scale |q|^2 + |x|^2, truncating adds) + 2^-14 (the dropped
// lo x lo products and the norms' and the reference distance's own roundings) at the longest row the filter takes
constexpr uint32_t kKnnFilterMaxDim = 128;
static_assert(kKnnFilterSlack >= 4.0f * kKnnFilterMaxDim / 8388608.0f + 1.0f / 16384.0f, "kKnnFilterSlack no longer covers the filter's error bound at its longest row");
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint16_t bf16_rne(float x) {  // round to nearest even (finite inputs)
    const uint32_t u = __float_as_uint(x);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_val(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// one thread per (row, group of 8 dims): writes the 8 hi and 8 lo parts (32 contiguous bytes); thread of group 0 also the norm
__global__ __launch_bounds__(256) void knn_pack_kernel(const float* x, uint32_t stride, uint32_t dim, uint64_t rows, uint32_t groups,
                                                       uint16_t* packed, float* norms) {
    const uint64_t e = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t row = e / groups;
    const uint32_t g = (uint32_t)(e % groups);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * stride;
    uint16_t hi[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t dcol = g * 8u + j;
        const float v = dcol < dim ? xr[dcol] : 0.f;
        hi[j] = bf16_rne(v);
        lo[j] = bf16_rne(v - bf16_val(hi[j]));
    }
    uint4 a, b;
    a.x = hi[0] | (uint32_t)hi[1] << 16; a.y = hi[2] | (uint32_t)hi[3] << 16; a.z = hi[4] | (uint32_t)hi[5] << 16; a.w = hi[6] | (uint32_t)hi[7] << 16;
    b.x = lo[0] | (uint32_t)lo[1] << 16; b.y = lo[2] | (uint32_t)lo[3] << 16; b.z = lo[4] | (uint32_t)lo[5] << 16; b.w = lo[6] | (uint32_t)lo[7] << 16;
    uint4* out = reinterpret_cast<uint4*>(packed + ((size_t)row * groups + g) * 16);
    out[0] = a;
    out[1] = b;
    if (g == 0) {
        float n2 = 0.f;
        for (uint32_t c = 0; c < dim; ++c) n2 += xr[c] * xr[c];
        norms[row] = n2;
    }
}

// A workgroup = 4 wavefronts x QB blocks of 32 queries (operand B of the product, in registers for the whole launch: the
// lane's column of every 32 x 32 tile is ITS query, so one threshold per lane and block); blockIdx.y picks a slab of the
// chunk's rows, walked in blocks of 32 (operand A straight from global memory, the next block's fragments requested
// before the current block's products: a 32 K-row chunk of packed rows is 4 MB at d = 32, L2 / Infinity-Cache resident
// while every workgroup sweeps it).  The accumulators start at -0.5 (1 - c) |x|^2 of their rows (16 rows of the block
// per lane: four float4 loads of the norms), so a kept pair is  max over the lane's 16 rows >= rhs[query]  -- eight
// v_max3 and one compare per tile beside its six matrix instructions.
template <int KSTEPS, int QB>
__global__ __launch_bounds__(256) void knn_filter_kernel(KnnFilterParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t r = lane & 31, h = lane >> 5;
    const uint32_t q0 = (blockIdx.x * 4u + wave) * (QB * 32u);
    if (q0 >= p.nq) return;
    const uint32_t groups = p.dp >> 3;  // 8-dim groups per row = 2 KSTEPS
    // B fragments: query q0 + 32 b + r, k = 16 s + 8 h .. + 7 -> group 2 s + h: hi at halfword 16 (2 s + h), lo 8 further
    bf16x8 qh[QB][KSTEPS], ql[QB][KSTEPS];
    float rhs[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const uint32_t qi = q0 + 32u * b + r;
        const bool ok = qi < p.nq;
        const uint16_t* qp = p.qpack + (size_t)(ok ? qi : 0u) * groups * 16;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const uint4* src = reinterpret_cast<const uint4*>(qp + (size_t)(2 * s + h) * 16);
            qh[b][s] = __builtin_bit_cast(bf16x8, src[0]);
            ql[b][s] = __builtin_bit_cast(bf16x8, src[1]);
        }
        rhs[b] = ok ? p.rhs[qi] : __builtin_inff();  // (a query beyond the batch keeps nothing)
    }
    // (every compiler-issued load is used here, once: its waits then sit in this prologue and not inside the sweep)
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        asm volatile("" ::"v"(rhs[b]));
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) asm volatile("" ::"v"(qh[b][s]), "v"(ql[b][s]));
    }
    const uint32_t blocks = (p.rows + 31u) >> 5;
    const uint32_t per = (blocks + gridDim.y - 1) / gridDim.y;
    const uint32_t b_lo = blockIdx.y * per, b_hi = min(blocks, b_lo + per);
    if (b_lo >= b_hi) return;
    const float scale = -0.5f * (1.0f - kKnnFilterSlack);
    // Fragments and norms of one block of 32 rows (rows beyond the chunk read row 0), requested by hand-written loads: the
    // compiler's own bookkeeping put a full wait (vmcnt(0)) BEHIND the next block's requests -- no overlap at all --
    // whatever the order in the source; these requests are invisible to it, and `landed` below is the one wait, placed in
    // front of the next requests.  (No branch here: the norms' array is readable 64 entries beyond the set and +inf
    // there, and the chunk starts on a multiple of 64 rows; a row of the NEXT chunk that slips through the test is
    // refused where the hits are stored.)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    auto request_block = [&](uint32_t rb, u32x4 (&xh)[KSTEPS], u32x4 (&xl)[KSTEPS], f32x4v (&nx)[4]) {
        const uint32_t j = rb * 32u + r;
        const uint16_t* bp = p.bpack + ((size_t)(j < p.rows ? j : 0u) * groups + h) * 16;
        const float* np = p.bnorm + rb * 32u + 4u * h;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {  // group 2 s + h: 64 bytes per K step further on
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xh[s]) : "v"(bp + 32 * s));
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(xl[s]) : "v"(bp + 32 * s));
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)  // accumulator registers 4 g .. 4 g + 3 = rows 8 g + 4 h + 0 .. 3 of the block
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(nx[g]) : "v"(np + 8 * g));
    };
    auto landed = [&](u32x4 (&xh)[KSTEPS], u32x4 (&xl)[KSTEPS], f32x4v (&nx)[4]) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(nx[0]));
#pragma unroll
        for (int g = 1; g < 4; ++g) asm volatile("" : "+v"(nx[g]));
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) asm volatile("" : "+v"(xh[s]), "+v"(xl[s]));
    };
    // the products and the test of one block of rows against the QB query blocks
    auto sweep_block = [&](uint32_t rb, const u32x4 (&xhr)[KSTEPS], const u32x4 (&xlr)[KSTEPS], const f32x4v (&nx)[4]) {
        bf16x8 xh[KSTEPS], xl[KSTEPS];
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) { xh[s] = __builtin_bit_cast(bf16x8, xhr[s]); xl[s] = __builtin_bit_cast(bf16x8, xlr[s]); }
        f32x16 cinit;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            cinit[4 * g + 0] = scale * nx[g].x; cinit[4 * g + 1] = scale * nx[g].y;
            cinit[4 * g + 2] = scale * nx[g].z; cinit[4 * g + 3] = scale * nx[g].w;
        }
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[0], qh[b][0], cinit, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[0], qh[b][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[0], ql[b][0], acc, 0, 0, 0);
#pragma unroll
            for (int s = 1; s < KSTEPS; ++s) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[s], qh[b][s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[s], qh[b][s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[s], ql[b][s], acc, 0, 0, 0);
            }
            float m = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
            for (int v = 3; v < 15; v += 2) m = fmaxf(fmaxf(m, acc[v]), acc[v + 1]);
            m = fmaxf(m, acc[15]);
            if (__builtin_expect(__ballot(m >= rhs[b]) != 0ull, 0)) {
                // one returning atomic per lane with hits (its 16 rows of the tile): the places of all of them at once -- a
                // returning atomic per hit serialised up to sixteen memory round trips in every tile that had any, and in
                // the early chunks (thresholds still loose) that is most tiles
                const uint32_t qv = q0 + 32u * b + r;   // (rhs = +inf beyond the batch: never here with qv >= nq)
                uint32_t hits = 0;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const uint32_t j = rb * 32u + (uint32_t)((v & 3) + 8 * (v >> 2)) + 4u * h;
                    hits |= (acc[v] >= rhs[b] && j < p.rows) ? 1u << v : 0u;  // (a threshold of -inf -- list not full yet -- lets -inf through too)
                }
                if (hits) {
                    uint32_t pos = atomicAdd(&p.count[qv], (uint32_t)__popc(hits));
                    do {
                        const uint32_t v = (uint32_t)__ffs((int)hits) - 1u;
                        hits &= hits - 1u;
                        if (pos < p.cap) p.cand[(size_t)qv * p.cap + pos] = p.row0 + rb * 32u + ((v & 3u) + 8u * (v >> 2)) + 4u * h;
                        else *p.overflow = 1u;
                        pos += 1u;
                    } while (hits);
                }
            }
        }
    };
    // two register sets in turn: wait for the set at hand, request the next block into the other one, sweep
    u32x4 xh0[KSTEPS], xl0[KSTEPS], xh1[KSTEPS], xl1[KSTEPS];
    f32x4v nx0[4], nx1[4];
    request_block(b_lo, xh0, xl0, nx0);
    for (uint32_t rb = b_lo;;) {
        landed(xh0, xl0, nx0);
        if (rb + 1u < b_hi) request_block(rb + 1u, xh1, xl1, nx1);
        sweep_block(rb, xh0, xl0, nx0);
        if (++rb >= b_hi) break;
        landed(xh1, xl1, nx1);
        if (rb + 1u < b_hi) request_block(rb + 1u, xh0, xl0, nx0);
        sweep_block(rb, xh1, xl1, nx1);
        if (++rb >= b_hi) break;
    }
}

__global__ __launch_bounds__(256) void knn_thresholds_kernel(const uint64_t* heap, int k, const float* qnorm, uint32_t nq, float* rhs) {
    const uint32_t qi = blockIdx.x * 256u + threadIdx.x;
    if (qi >= nq) return;
    const uint64_t root = heap[qi];
    // heap not full yet (an all-ones key at the root): every row passes
    rhs[qi] = root == ~0ull ? -__builtin_inff()
                            : 0.5f * ((1.0f - kKnnFilterSlack) * qnorm[qi] - knn_fkey_inv((uint32_t)(root >> 32)));
    (void)k;
}

// One thread per query: the exact distance (L2Metric::Dist, support_func.h:107-128, the scan kernels' operations and order)
// of every candidate the filter kept, offered to the query's heap; then the query's new threshold.  d <= 128, query in
// registers, candidate rows straight from global memory.
template <int S>
__global__ __launch_bounds__(128) void knn_rescore_kernel(KnnRescoreParams p) {
    const KnnParams& kp = p.k;
    const uint32_t qi = blockIdx.x * 128u + threadIdx.x;
    if (qi >= kp.nq) return;
    const uint32_t steps = kp.dim >> 2;
    const float* qp = kp.q + (size_t)qi * kp.qstride;
    float4 q[S];
#pragma unroll
    for (int c = 0; c < S; ++c)
        q[c] = (uint32_t)c < steps ? make_float4(qp[4 * c], qp[4 * c + 1], qp[4 * c + 2], qp[4 * c + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    uint64_t root = kp.heap[qi];
    const uint32_t cnt = min(p.count[qi], p.cap);
    for (uint32_t e = 0; e < cnt; ++e) {
        const uint32_t row = p.cand[(size_t)qi * p.cap + e];
        const float* bp = kp.base + (size_t)row * kp.bstride;
        f32x2 s01{0.f, 0.f}, s23{0.f, 0.f};
#pragma unroll
        for (int c = 0; c < S; ++c) {
            if ((uint32_t)c < steps) {
                const f32x2 b01{bp[4 * c], bp[4 * c + 1]}, b23{bp[4 * c + 2], bp[4 * c + 3]};
                const f32x2 e01 = b01 - f32x2{q[c].x, q[c].y};
                const f32x2 e23 = b23 - f32x2{q[c].z, q[c].w};
                s01 = s01 + e01 * e01;
                s23 = s23 + e23 * e23;
            }
        }
        const float dist = ((s01.x + s01.y) + s23.x) + s23.y;
        const uint64_t key = ((uint64_t)knn_fkey(dist) << 32) | row;
        const bool self = kp.self_offset >= 0 && (uint64_t)row == (uint64_t)qi + (uint64_t)kp.self_offset;
        if (!self && key < root) root = heap_replace_root(kp.heap + qi, kp.heap_stride, kp.k, key);
    }
    p.rhs[qi] = root == ~0ull ? -__builtin_inff()
                              : 0.5f * ((1.0f - kKnnFilterSlack) * p.qnorm[qi] - knn_fkey_inv((uint32_t)(root >> 32)));
}

__global__ __launch_bounds__(128) void knn_finalize_kernel(KnnParams p) {
    const uint32_t qi = blockIdx.x * 128u + threadIdx.x;
    if (qi >= p.nq) return;
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

// ------------------------------------------------------------------------------------------
// Long lists (k >= 64; the reference's graph builder reads 1 000-NN lists, dim_red/support_func.py:374-384 ->
// prepare_graph.cpp:66).  A binary heap per query in global memory costs ~3 log2 k scattered accesses per entrant and a
// query sees about k ln(n / k) entrants (7 000 at k = 1 000, n = 10^6): 20 TB of 64-byte lines for 10^6 queries, seven
// seconds.  Here a query's current k smallest keys are an UNORDERED pool [k] in global memory and one wavefront serves
// one query per chunk: it reads the pool into LDS (coalesced), computes the exact distances (the reference's operations
// and order, one candidate per lane) of the rows the filter kept, appends the keys below the pool's largest, and when
// the appended keys fill their buffer -- or the chunk ends -- keeps the k smallest of the union with a radix select on the
// 64-bit (distance, id) keys: a 256-bin histogram per digit from the top, until the bin that holds the k-th key holds
// nothing else.  Same set of keys as the heaps would hold, whatever the order; knn_pool_finalize_kernel sorts it.
// ------------------------------------------------------------------------------------------
constexpr int kPoolNew = 1024;  // appended keys per selection round

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)v, d);
        v += lane >= d ? t : 0u;
    }
    return v;
}

// k-th smallest (1-based) of keys[0 .. M), M > k; afterwards keys[0 .. k) hold the k smallest (any order; all-ones keys
// are "no entry" and may repeat, every other key is distinct).  Returns the largest of them.
__device__ __forceinline__ uint64_t pool_select(uint64_t* keys, int k, int M, uint32_t* hist, int lane) {
    uint64_t prefix = 0, mask = 0;
    uint32_t need = (uint32_t)k;
    for (int shift = 56; shift >= 0; shift -= 8) {
        for (int j = lane; j < 256; j += 64) hist[j] = 0u;
        __syncthreads();
        for (int i = lane; i < M; i += 64) {
            const uint64_t key = keys[i];
            if ((key & mask) == prefix) atomicAdd(&hist[(uint32_t)(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        const uint32_t c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
        const uint32_t sum = c0 + c1 + c2 + c3;
        const uint32_t incl = wave_incl_scan_u32(sum, lane), excl = incl - sum;
        const bool has = excl < need && need <= incl;  // exactly one lane
        const uint32_t r = need - excl;
        uint32_t b = 3, below = c0 + c1 + c2, cb = c3;
        if (r <= c0) { b = 0; below = 0; cb = c0; }
        else if (r <= c0 + c1) { b = 1; below = c0; cb = c1; }
        else if (r <= c0 + c1 + c2) { b = 2; below = c0 + c1; cb = c2; }
        const int src = __ffsll((unsigned long long)__ballot(has)) - 1;
        const uint32_t bin = (uint32_t)__shfl((int)(4u * (uint32_t)lane + b), src);
        const uint32_t skipped = (uint32_t)__shfl((int)(excl + below), src);
        const uint32_t in_bin = (uint32_t)__shfl((int)cb, src);
        need -= skipped;
        prefix |= (uint64_t)bin << shift;
        mask |= 255ull << shift;
        if (in_bin == 1u) break;
    }
    uint64_t tau = prefix;  // every digit fixed (only all-ones keys repeat), or: the one key under the prefix
    if (mask != ~0ull) {
        uint64_t found = 0;
        for (int i = lane; i < M; i += 64) {
            const uint64_t key = keys[i];
            if ((key & mask) == prefix) found = key;
        }
        const int src = __ffsll((unsigned long long)__ballot(found != 0)) - 1;  // (a key is never 0: distance keys have a bit set)
        tau = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(found >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)found, src);
    }
    int c = 0;  // keys below tau to the front, in place (a key moves to a position at or below its own)
    for (int i0 = 0; i0 < M; i0 += 64) {
        const int i = i0 + lane;
        const uint64_t key = i < M ? keys[i] : ~0ull;
        const bool keep = i < M && key < tau;
        const uint64_t m = __ballot(keep);
        const int pos = c + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        __syncthreads();
        if (keep) keys[pos] = key;
        c += __popcll(m);
        __syncthreads();
    }
    for (int i = c + lane; i < k; i += 64) keys[i] = tau;  // the k-th itself (all-ones: every free place)
    __syncthreads();
    return tau;
}

template <int S>
__global__ __launch_bounds__(64) void knn_pool_update_kernel(KnnPoolParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char knn_pool_smem[];
    const KnnParams& kp = p.k;
    const uint32_t qi = blockIdx.x;
    const int lane = threadIdx.x;
    const int k = kp.k;
    const uint32_t cnt = min(p.count[qi], p.cap);
    if (cnt == 0) return;  // (threshold unchanged)
    uint64_t* keys = reinterpret_cast<uint64_t*>(knn_pool_smem);
    uint32_t* hist = reinterpret_cast<uint32_t*>(keys + k + kPoolNew);
    uint64_t* pool = p.pool + (size_t)qi * (size_t)k;
    for (int i = lane; i < k; i += 64) keys[i] = pool[i];
    uint64_t root = p.root[qi];
    const uint32_t steps = kp.dim >> 2;
    const float* qp = kp.q + (size_t)qi * kp.qstride;
    float4 q[S];
#pragma unroll
    for (int c = 0; c < S; ++c)
        q[c] = (uint32_t)c < steps ? make_float4(qp[4 * c], qp[4 * c + 1], qp[4 * c + 2], qp[4 * c + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    int nnew = 0;
    for (uint32_t e0 = 0; e0 < cnt; e0 += 64) {
        const uint32_t e = e0 + (uint32_t)lane;
        const bool valid = e < cnt;
        const uint32_t row = valid ? p.cand[(size_t)qi * p.cap + e] : 0u;
        const float* bp = kp.base + (size_t)row * kp.bstride;
        f32x2 s01{0.f, 0.f}, s23{0.f, 0.f};  // L2Metric::Dist, the scan kernels' operations and order
#pragma unroll
        for (int c = 0; c < S; ++c) {
            if ((uint32_t)c < steps) {
                const float4 b4 = *reinterpret_cast<const float4*>(bp + 4 * c);
                const f32x2 e01 = f32x2{b4.x, b4.y} - f32x2{q[c].x, q[c].y};
                const f32x2 e23 = f32x2{b4.z, b4.w} - f32x2{q[c].z, q[c].w};
                s01 = s01 + e01 * e01;
                s23 = s23 + e23 * e23;
            }
        }
        const float dist = ((s01.x + s01.y) + s23.x) + s23.y;
        const uint64_t key = ((uint64_t)knn_fkey(dist) << 32) | row;
        const bool self = kp.self_offset >= 0 && (uint64_t)row == (uint64_t)qi + (uint64_t)kp.self_offset;
        const bool keep = valid && !self && key < root;
        const uint64_t m = __ballot(keep);
        if (keep) keys[k + nnew + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = key;
        nnew += __popcll(m);
        if (nnew > kPoolNew - 64) {
            __syncthreads();
            root = pool_select(keys, k, k + nnew, hist, lane);
            nnew = 0;
        }
    }
    __syncthreads();
    if (nnew) root = pool_select(keys, k, k + nnew, hist, lane);
    for (int i = lane; i < k; i += 64) pool[i] = keys[i];
    if (lane == 0) {
        p.root[qi] = root;
        p.rhs[qi] = root == ~0ull ? -__builtin_inff()
                                  : 0.5f * ((1.0f - kKnnFilterSlack) * p.qnorm[qi] - knn_fkey_inv((uint32_t)(root >> 32)));
    }
}

// ascending (distance, id) order of a query's pool and the outputs: one wavefront per query, bitonic network in LDS
__global__ __launch_bounds__(64) void knn_pool_finalize_kernel(KnnPoolParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char knn_pool_smem[];
    const KnnParams& kp = p.k;
    const uint32_t qi = blockIdx.x;
    const int lane = threadIdx.x;
    const int k = kp.k;
    int N = 64;
    while (N < k) N <<= 1;
    uint64_t* keys = reinterpret_cast<uint64_t*>(knn_pool_smem);
    const uint64_t* pool = p.pool + (size_t)qi * (size_t)k;
    for (int i = lane; i < N; i += 64) keys[i] = i < k ? pool[i] : ~0ull;
    __syncthreads();
    for (int size = 2; size <= N; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = lane; t < (N >> 1); t += 64) {
                const int i = 2 * t - (t & (stride - 1)), j = i + stride;
                const bool up = (i & size) == 0;
                const uint64_t a = keys[i], b = keys[j];
                if ((a > b) == up) { keys[i] = b; keys[j] = a; }
            }
            __syncthreads();
        }
    }
    for (int e = lane; e < k; e += 64) {
        const uint64_t kv = keys[e];
        kp.out_ids[(size_t)qi * k + e] = (uint32_t)kv;
        if (kp.out_dist) kp.out_dist[(size_t)qi * k + e] = (kv == ~0ull) ? __builtin_inff() : knn_fkey_inv((uint32_t)(kv >> 32));
    }
}

template <int KSTEPS, int QB>
hipError_t launch_knn_filter_t(const KnnFilterParams& p, hipStream_t s) {
    const unsigned gx = (p.nq + QB * 128u - 1) / (QB * 128u);
    const unsigned blocks = (p.rows + 31u) >> 5;
    static const unsigned simds = [] {  // 4 SIMDs per CU of the current device (the MI355X: 1 024)
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return 4u * (unsigned)cus;
    }();
    unsigned gy = gx >= simds ? 1u : (simds + gx - 1) / gx;   // enough workgroups for every SIMD
    if (gy > blocks) gy = blocks;
    if (gy < 1u) gy = 1u;
    hipLaunchKernelGGL((knn_filter_kernel<KSTEPS, QB>), dim3(gx, gy), dim3(256), 0, s, p);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_knn_scan(const KnnParams& p, int metric, hipStream_t s) {
    if (p.nq == 0) return hipSuccess;
    return metric == 1 ? launch_knn_m<1>(p, s) : launch_knn_m<0>(p, s);
}

hipError_t launch_knn_pack(const float* x, uint32_t stride, uint32_t dim, uint64_t rows, uint16_t* packed, float* norms, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    const uint32_t groups = ((dim + 15u) & ~15u) >> 3;
    const uint64_t threads = rows * groups;
    hipLaunchKernelGGL(knn_pack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, x, stride, dim, rows, groups, packed, norms);
    return hipGetLastError();
}

hipError_t launch_knn_filter(const KnnFilterParams& p, hipStream_t s) {
    if (p.nq == 0 || p.rows == 0) return hipSuccess;
    switch (p.dp >> 4) {   // K steps of 16; query blocks per wavefront by what the operands leave of the register file
        case 1: return launch_knn_filter_t<1, 4>(p, s);
        case 2: return launch_knn_filter_t<2, 4>(p, s);
        case 3: return launch_knn_filter_t<3, 2>(p, s);
        case 4: return launch_knn_filter_t<4, 2>(p, s);
        case 5: return launch_knn_filter_t<5, 1>(p, s);
        case 6: return launch_knn_filter_t<6, 1>(p, s);
        case 7: return launch_knn_filter_t<7, 1>(p, s);
        case 8: return launch_knn_filter_t<8, 1>(p, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_knn_thresholds(const uint64_t* heap, size_t heap_stride, int k, const float* qnorm, uint32_t nq, float* rhs, hipStream_t s) {
    (void)heap_stride;
    if (nq == 0) return hipSuccess;
    hipLaunchKernelGGL(knn_thresholds_kernel, dim3((nq + 255u) / 256u), dim3(256), 0, s, heap, k, qnorm, nq, rhs);
    return hipGetLastError();
}

hipError_t launch_knn_rescore(const KnnRescoreParams& p, hipStream_t s) {
    if (p.k.nq == 0) return hipSuccess;
    const dim3 grid((p.k.nq + 127u) / 128u), block(128);
    if (p.k.dim <= 32) hipLaunchKernelGGL(knn_rescore_kernel<8>, grid, block, 0, s, p);
    else if (p.k.dim <= 64) hipLaunchKernelGGL(knn_rescore_kernel<16>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(knn_rescore_kernel<32>, grid, block, 0, s, p);
    return hipGetLastError();
}

hipError_t launch_knn_pool_update(const KnnPoolParams& p, hipStream_t s) {
    if (p.k.nq == 0) return hipSuccess;
    const size_t lds = ((size_t)p.k.k + kPoolNew) * 8 + 256 * 4;
    const dim3 grid(p.k.nq), block(64);
    auto go = [&](auto kernel) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kernel, grid, block, lds, s, p);
        return hipGetLastError();
    };
    if (p.k.dim <= 32) return go(knn_pool_update_kernel<8>);
    if (p.k.dim <= 64) return go(knn_pool_update_kernel<16>);
    return go(knn_pool_update_kernel<32>);
}

hipError_t launch_knn_pool_finalize(const KnnPoolParams& p, hipStream_t s) {
    if (p.k.nq == 0) return hipSuccess;
    size_t N = 64;
    while (N < (size_t)p.k.k) N <<= 1;
    hipLaunchKernelGGL(knn_pool_finalize_kernel, dim3(p.k.nq), dim3(64), N * 8, s, p);
    return hipGetLastError();
}

hipError_t launch_knn_finalize(const KnnParams& p, hipStream_t s) {
    if (p.nq == 0) return hipSuccess;
    hipLaunchKernelGGL(knn_finalize_kernel, dim3((p.nq + 127u) / 128u), dim3(128), 0, s, p);
    return hipGetLastError();
}

}  // namespace gbnns
