// mlp.hip -- the projection (GetLowQueryFromNet / computeNetLayer / normalizeVector, support_func.h:624-658): tiled layer
// kernels in the reference's 8-sum order, the narrow last layer with the normalise step fused, and two small fill kernels.
#include <algorithm>

#include "launch_util.h"
#include "walk_common.h"

namespace gbnns {

namespace {

// ------------------------------------------------------------------------------------------
// MLP projection (support_func.h:624-633 computeNetLayer over a batch)
// ------------------------------------------------------------------------------------------
// out[q][o] = act( dot8(W[o,:], x[q,:]) + b[o] ) with dot8 = the 8-running-sum order of
// Angular::Dist.  Block = 256 threads = 32 queries x 64 neurons; thread = 2 queries x 4 neurons,
// 8 running sums each; x / W tiles of 32 k-values staged through LDS (rows padded to 36 floats so
// the 16-B fragment reads of 16 consecutive rows hit distinct bank groups).

constexpr int kTQ = 32, kTO = 64, kKC = 32, kLd = kKC + 4;

// NORM (last layer, dout <= 64, one block column): the block also applies normalizeVector
// (support_func.h:636-642) to its 32 output rows -- same arithmetic as normalize_kernel, one launch less.
constexpr int kNormLd = kTO + 1;

__device__ __forceinline__ void mlp_normalize_rows(const LayerParams& p, const float* ys, uint32_t qbase, int t) {
    // 8 threads per query: threads 0..3 of a query run the four running sums of L2Metric::Dist(y, 0)
    // (support_func.h:107-128, d % 4 tail ignored), then every thread divides its share of the outputs.
    __shared__ float nsum[kTQ][4];
    const int q = t >> 3, part = t & 7;
    const uint32_t qg = qbase + q;
    const float* y = ys + q * kNormLd;
    if (part < 4) {
        const uint32_t steps = p.dout >> 2;
        float sc = 0.f;
        for (uint32_t k = 0; k < steps; ++k) {
            const float e = y[4 * k + part] - 0.f;
            sc = sc + e * e;
        }
        nsum[q][part] = sc;
    }
    __syncthreads();
    if (qg >= p.nq) return;
    float norm = ((nsum[q][0] + nsum[q][1]) + nsum[q][2]) + nsum[q][3];
    norm = __builtin_sqrtf(norm);  // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt); __fsqrt_rn maps to the native sqrt here
    float* r = p.out + (size_t)qg * p.ostride;
    for (uint32_t i = part; i < p.dout; i += 8) r[i] = __fdiv_rn(y[i], norm);
    for (uint32_t i = p.dout + part; i < p.ostride; i += 8) r[i] = 0.f;
}

template <bool RELU, bool NORM = false>
__global__ __launch_bounds__(256) void mlp_layer_kernel(LayerParams p) {
    __shared__ __attribute__((aligned(16))) float xs[kTQ * kLd];
    __shared__ __attribute__((aligned(16))) float ws[kTO * kLd];
    const int t = threadIdx.x;
    const int tq = t >> 4;   // 0..15 -> queries 2*tq, 2*tq+1
    const int to = t & 15;   // neurons to + 16*j
    const uint32_t qbase = blockIdx.x * kTQ;
    const uint32_t obase = blockIdx.y * kTO;
    const uint32_t kmain = (p.din >> 3) << 3;

    float acc[2][4][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int l = 0; l < 8; ++l) acc[a][b][l] = 0.f;

    for (uint32_t k0 = 0; k0 < kmain; k0 += kKC) {
        const uint32_t kc = (kmain - k0 < (uint32_t)kKC) ? (kmain - k0) : (uint32_t)kKC;
        for (int e = t; e < kTQ * kKC; e += 256) {
            const int r = e / kKC, c = e % kKC;
            const uint32_t qg = qbase + r;
            xs[r * kLd + c] = (qg < p.nq && (uint32_t)c < kc) ? p.x[(size_t)qg * p.xstride + k0 + c] : 0.f;
        }
        for (int e = t; e < kTO * kKC; e += 256) {
            const int r = e / kKC, c = e % kKC;
            const uint32_t og = obase + r;
            ws[r * kLd + c] = (og < p.dout && (uint32_t)c < kc) ? p.w[(size_t)og * p.wstride + k0 + c] : 0.f;
        }
        __syncthreads();
        for (uint32_t s = 0; s < kc; s += 8) {
            float4 xv[2][2], wv[4][2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float4* xp = reinterpret_cast<const float4*>(&xs[(2 * tq + a) * kLd + s]);
                xv[a][0] = xp[0];
                xv[a][1] = xp[1];
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float4* wp = reinterpret_cast<const float4*>(&ws[(to + 16 * b) * kLd + s]);
                wv[b][0] = wp[0];
                wv[b][1] = wp[1];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    acc[a][b][0] = acc[a][b][0] + wv[b][0].x * xv[a][0].x;
                    acc[a][b][1] = acc[a][b][1] + wv[b][0].y * xv[a][0].y;
                    acc[a][b][2] = acc[a][b][2] + wv[b][0].z * xv[a][0].z;
                    acc[a][b][3] = acc[a][b][3] + wv[b][0].w * xv[a][0].w;
                    acc[a][b][4] = acc[a][b][4] + wv[b][1].x * xv[a][1].x;
                    acc[a][b][5] = acc[a][b][5] + wv[b][1].y * xv[a][1].y;
                    acc[a][b][6] = acc[a][b][6] + wv[b][1].z * xv[a][1].z;
                    acc[a][b][7] = acc[a][b][7] + wv[b][1].w * xv[a][1].w;
                }
        }
        __syncthreads();
    }

    const uint32_t rem8 = p.din & 7;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const uint32_t qg = qbase + 2 * tq + a;
        if (qg >= p.nq) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t og = obase + to + 16 * b;
            if (og >= p.dout) continue;
            float m[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = acc[a][b][j + 4] + acc[a][b][j];
            uint32_t kk = kmain, rem = rem8;
            const float* xr = p.x + (size_t)qg * p.xstride;
            const float* wr = p.w + (size_t)og * p.wstride;
            if (rem >= 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) m[j] = m[j] + wr[kk + j] * xr[kk + j];
                kk += 4;
                rem -= 4;
            }
            if (rem > 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xv = ((uint32_t)j < rem) ? xr[kk + j] : 0.f;
                    const float wv = ((uint32_t)j < rem) ? wr[kk + j] : 0.f;
                    m[j] = m[j] + wv * xv;
                }
            }
            const float dist = -((m[0] + m[1]) + (m[2] + m[3]));  // Angular::Dist
            float v = 0.f;
            v = v - dist;               // support_func.h:627
            v = v + p.bias[og];         // :628
            if (RELU && v < 0.f) v = 0.f;  // :629-631
            if constexpr (NORM) ws[(2 * tq + a) * kNormLd + to + 16 * b] = v;  // ws is free after the k loop
            else p.out[(size_t)qg * p.ostride + og] = v;
        }
    }
    if constexpr (NORM) {
        __syncthreads();
        mlp_normalize_rows(p, ws, qbase, t);
    }
}

// Same tile and arithmetic as mlp_layer_kernel, for 16-B aligned operands (xstride % 4 == 0):
// 16-B global loads, and the next k-chunk is fetched into registers while the current one is
// being consumed from LDS (the generic kernel exposes one global round trip per chunk).
// (second launch bound: 1 = as many registers as the tile wants -- 162; 4 = at most 128, so that a block fits beside six 64-register
// walk wavefronts per SIMD: the A/B switch of the round-4 pipeline experiment, profiles/r04_ab.txt)
#ifndef GBNNS_MLP_WAVES
#define GBNNS_MLP_WAVES 1
#endif
template <bool RELU, bool NORM = false>
__global__ __launch_bounds__(256, GBNNS_MLP_WAVES) void mlp_layer_vec_kernel(LayerParams p) {
    __shared__ __attribute__((aligned(16))) float xs[kTQ * kLd];
    __shared__ __attribute__((aligned(16))) float ws[kTO * kLd];
    const int t = threadIdx.x;
    const int tq = t >> 4;   // 0..15 -> queries 2*tq, 2*tq+1
    const int to = t & 15;   // neurons to + 16*j
    const uint32_t qbase = blockIdx.x * kTQ;
    const uint32_t obase = blockIdx.y * kTO;
    const uint32_t kmain = (p.din >> 3) << 3;
    // staging role of this thread: one float4 of the x tile, two of the w tile
    const int srow = t >> 3, sc4 = (t & 7) * 4;
    const uint32_t xq = qbase + srow;
    const uint32_t wo0 = obase + srow, wo1 = obase + srow + 32;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    auto fetch = [&](uint32_t k0, float4& fx, float4& fw0, float4& fw1) {
        const bool kin = k0 + sc4 < kmain;  // kmain is a multiple of 8 and sc4 of 4: whole float4 in or out
        fx = (kin && xq < p.nq) ? *reinterpret_cast<const float4*>(p.x + (size_t)xq * p.xstride + k0 + sc4) : zero4;
        fw0 = (kin && wo0 < p.dout) ? *reinterpret_cast<const float4*>(p.w + (size_t)wo0 * p.wstride + k0 + sc4) : zero4;
        fw1 = (kin && wo1 < p.dout) ? *reinterpret_cast<const float4*>(p.w + (size_t)wo1 * p.wstride + k0 + sc4) : zero4;
    };

    float acc[2][4][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int l = 0; l < 8; ++l) acc[a][b][l] = 0.f;

    float4 fx, fw0, fw1;
    fetch(0, fx, fw0, fw1);
    for (uint32_t k0 = 0; k0 < kmain; k0 += kKC) {
        const uint32_t kc = (kmain - k0 < (uint32_t)kKC) ? (kmain - k0) : (uint32_t)kKC;
        *reinterpret_cast<float4*>(&xs[srow * kLd + sc4]) = fx;
        *reinterpret_cast<float4*>(&ws[srow * kLd + sc4]) = fw0;
        *reinterpret_cast<float4*>(&ws[(srow + 32) * kLd + sc4]) = fw1;
        __syncthreads();
        if (k0 + kKC < kmain) fetch(k0 + kKC, fx, fw0, fw1);  // in flight during the compute below
        float4 xv[2][2], wv[4][2];
        auto lds_step = [&](uint32_t s, float4 (&xo)[2][2], float4 (&wo)[4][2]) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float4* xp = reinterpret_cast<const float4*>(&xs[(2 * tq + a) * kLd + s]);
                xo[a][0] = xp[0];
                xo[a][1] = xp[1];
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float4* wp = reinterpret_cast<const float4*>(&ws[(to + 16 * b) * kLd + s]);
                wo[b][0] = wp[0];
                wo[b][1] = wp[1];
            }
        };
        lds_step(0, xv, wv);
        for (uint32_t s = 0; s < kc; s += 8) {
            // (the next step's operands are requested before this step's products: one LDS round trip less in the chain)
            float4 xn[2][2], wn[4][2];
            if (s + 8 < kc) lds_step(s + 8, xn, wn);
            else lds_step(s, xn, wn);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    acc[a][b][0] = acc[a][b][0] + wv[b][0].x * xv[a][0].x;
                    acc[a][b][1] = acc[a][b][1] + wv[b][0].y * xv[a][0].y;
                    acc[a][b][2] = acc[a][b][2] + wv[b][0].z * xv[a][0].z;
                    acc[a][b][3] = acc[a][b][3] + wv[b][0].w * xv[a][0].w;
                    acc[a][b][4] = acc[a][b][4] + wv[b][1].x * xv[a][1].x;
                    acc[a][b][5] = acc[a][b][5] + wv[b][1].y * xv[a][1].y;
                    acc[a][b][6] = acc[a][b][6] + wv[b][1].z * xv[a][1].z;
                    acc[a][b][7] = acc[a][b][7] + wv[b][1].w * xv[a][1].w;
                }
#pragma unroll
            for (int a = 0; a < 2; ++a) { xv[a][0] = xn[a][0]; xv[a][1] = xn[a][1]; }
#pragma unroll
            for (int b = 0; b < 4; ++b) { wv[b][0] = wn[b][0]; wv[b][1] = wn[b][1]; }
        }
        __syncthreads();
    }

    const uint32_t rem8 = p.din & 7;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const uint32_t qg = qbase + 2 * tq + a;
        if (qg >= p.nq) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t og = obase + to + 16 * b;
            if (og >= p.dout) continue;
            float m[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = acc[a][b][j + 4] + acc[a][b][j];
            uint32_t kk = kmain, rem = rem8;
            const float* xr = p.x + (size_t)qg * p.xstride;
            const float* wr = p.w + (size_t)og * p.wstride;
            if (rem >= 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) m[j] = m[j] + wr[kk + j] * xr[kk + j];
                kk += 4;
                rem -= 4;
            }
            if (rem > 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xv = ((uint32_t)j < rem) ? xr[kk + j] : 0.f;
                    const float wv = ((uint32_t)j < rem) ? wr[kk + j] : 0.f;
                    m[j] = m[j] + wv * xv;
                }
            }
            const float dist = -((m[0] + m[1]) + (m[2] + m[3]));  // Angular::Dist
            float v = 0.f;
            v = v - dist;               // support_func.h:627
            v = v + p.bias[og];         // :628
            if (RELU && v < 0.f) v = 0.f;  // :629-631
            if constexpr (NORM) ws[(2 * tq + a) * kNormLd + to + 16 * b] = v;  // ws is free after the k loop
            else p.out[(size_t)qg * p.ostride + og] = v;
        }
    }
    if constexpr (NORM) {
        __syncthreads();
        mlp_normalize_rows(p, ws, qbase, t);
    }
}

// Hidden layers of batches IN FLIGHT (round 4; LayerParams::small_footprint): lane = query, the wavefront's four neurons are
// wave-uniform, so their weights come through SCALAR loads (s_load_dwordx8, read through the constant address space) and
// are the scalar source of the packed multiplies; LDS only carries the x tile (64 queries x 32 k, one 32-byte read per
// lane and k step).  60 vector registers and 9 KB of LDS per block instead of 162 and 14 KB: its blocks find room beside
// the walk wavefronts of the batch before, which mlp_layer_vec_kernel's do not.  Alone it is the SLOWER
// kernel (0.080 against 0.070 ms per 10 000 x 128 -> 256 -> 256 -> 32 projection: scalar loads that miss the 16 KB K$ take
// long and cannot be pipelined inside a wavefront -- they return out of order, every wait is lgkmcnt(0); two queries per
// lane made it 0.128 ms), so a batch that runs alone keeps the kernel above.  Same arithmetic: eight running sums per
// (query, neuron) over k mod 8, the folds and the tail rules of the kernel above, bit for bit.
// Block = 4 wavefronts = 64 queries x 16 neurons (wavefront w: neurons 4 w .. 4 w + 3 of the block's 16).
constexpr int kSwQ = 64, kSwO = 16;
template <bool RELU>
__global__ __launch_bounds__(256) void mlp_layer_sw_kernel(LayerParams p) {
    __shared__ __attribute__((aligned(16))) float xs[kSwQ * kLd];
    const int t = threadIdx.x, lane = t & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t qbase = blockIdx.x * kSwQ;
    const uint32_t obase = blockIdx.y * kSwO + 4u * wave;
    const uint32_t kmain = (p.din >> 3) << 3;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // staging role: float4 number t and t + 256 of the 64 x 32 tile (row = number / 8)
    const int r0 = t >> 3, r1 = r0 + 32, sc4 = (t & 7) * 4;
    auto fetch = [&](uint32_t k0, float4& f0, float4& f1) {
        const bool kin = k0 + sc4 < kmain;
        f0 = (kin && qbase + r0 < p.nq) ? *reinterpret_cast<const float4*>(p.x + (size_t)(qbase + r0) * p.xstride + k0 + sc4) : zero4;
        f1 = (kin && qbase + r1 < p.nq) ? *reinterpret_cast<const float4*>(p.x + (size_t)(qbase + r1) * p.xstride + k0 + sc4) : zero4;
    };
    // the wavefront's four weight rows (neurons beyond dout read the last row: their results are dropped)
    typedef const __attribute__((address_space(4))) float* cfloat_p;
    cfloat_p wr[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const uint32_t og = obase + o < p.dout ? obase + o : p.dout - 1u;
        wr[o] = (cfloat_p)(uintptr_t)(p.w + (size_t)og * p.wstride);
    }
    float acc[4][8];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int l = 0; l < 8; ++l) acc[o][l] = 0.f;

    float4 f0, f1;
    fetch(0, f0, f1);
    for (uint32_t k0 = 0; k0 < kmain; k0 += kKC) {
        const uint32_t kc = (kmain - k0 < (uint32_t)kKC) ? (kmain - k0) : (uint32_t)kKC;
        *reinterpret_cast<float4*>(&xs[r0 * kLd + sc4]) = f0;
        *reinterpret_cast<float4*>(&xs[r1 * kLd + sc4]) = f1;
        __syncthreads();
        if (k0 + kKC < kmain) fetch(k0 + kKC, f0, f1);  // in flight during the compute below
        for (uint32_t s = 0; s < kc; s += 8) {
            const float4* xp = reinterpret_cast<const float4*>(&xs[lane * kLd + s]);
            const float4 xa = xp[0], xb = xp[1];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                cfloat_p w = wr[o] + k0 + s;  // wave-uniform address: scalar loads
                acc[o][0] = acc[o][0] + w[0] * xa.x;
                acc[o][1] = acc[o][1] + w[1] * xa.y;
                acc[o][2] = acc[o][2] + w[2] * xa.z;
                acc[o][3] = acc[o][3] + w[3] * xa.w;
                acc[o][4] = acc[o][4] + w[4] * xb.x;
                acc[o][5] = acc[o][5] + w[5] * xb.y;
                acc[o][6] = acc[o][6] + w[6] * xb.z;
                acc[o][7] = acc[o][7] + w[7] * xb.w;
            }
        }
        __syncthreads();
    }

    const uint32_t rem8 = p.din & 7;
    const uint32_t qg = qbase + lane;
    if (qg >= p.nq) return;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const uint32_t og = obase + o;
        if (og >= p.dout) continue;
        float m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = acc[o][j + 4] + acc[o][j];
        uint32_t kk = kmain, rem = rem8;
        const float* xr = p.x + (size_t)qg * p.xstride;
        const float* wt = p.w + (size_t)og * p.wstride;
        if (rem >= 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = m[j] + wt[kk + j] * xr[kk + j];
            kk += 4;
            rem -= 4;
        }
        if (rem > 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xv = ((uint32_t)j < rem) ? xr[kk + j] : 0.f;
                const float wv = ((uint32_t)j < rem) ? wt[kk + j] : 0.f;
                m[j] = m[j] + wv * xv;
            }
        }
        const float dist = -((m[0] + m[1]) + (m[2] + m[3]));  // Angular::Dist
        float v = 0.f;
        v = v - dist;               // support_func.h:627
        v = v + p.bias[og];         // :628
        if (RELU && v < 0.f) v = 0.f;  // :629-631
        p.out[(size_t)qg * p.ostride + og] = v;
    }
}

// Narrow layers (dout <= 32, din <= 256: the last projection layer): the chunked kernels above spend
// their time waiting -- eight dependent chunk round trips for two microseconds of arithmetic.  Here a
// block stages its whole x tile [32 queries x din] and W tile [32 neurons x din] in one go (all loads in
// flight together), then computes; thread = 2 queries x 2 neurons x 8 running sums, same order, same
// tail rules, optional fused normalizeVector.  Dynamic LDS: 64 rows x (din8 + 4) floats (at least 32 x 65 for the normalise step).
template <bool RELU, bool NORM>
__global__ __launch_bounds__(256) void mlp_narrow_kernel(LayerParams p) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    const uint32_t kpad = (p.din + 7u) & ~7u;
    const uint32_t ld = kpad + 4;  // row stride: 16-B reads of 16 consecutive rows hit distinct bank groups
    float* xs = smf;               // [32][ld]
    float* ws = smf + 32 * ld;     // [32][ld]
    const int t = threadIdx.x;
    const int tq = t >> 4;         // queries 2*tq, 2*tq+1
    const int to = t & 15;         // neurons to, to+16
    const uint32_t qbase = blockIdx.x * 32;
    const uint32_t c4n = kpad >> 2;  // float4 per row
    // staging: 8 loads in flight per thread, then 8 LDS stores (a load -> store loop would pay one
    // round trip per 16 bytes)
    for (uint32_t e0 = t; e0 < 64 * c4n; e0 += 256 * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t e = e0 + 256u * u;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < 64 * c4n) {
                const uint32_t r = e / c4n, c4 = (e % c4n) * 4;
                const bool isx = r < 32;
                const uint32_t row = isx ? qbase + r : r - 32;
                const bool in = isx ? row < p.nq : row < p.dout;
                const float* src = isx ? p.x + (size_t)row * p.xstride + c4 : p.w + (size_t)row * p.wstride + c4;
                if (in) {
                    if (c4 + 4 <= p.din) v[u] = *reinterpret_cast<const float4*>(src);
                    else {
                        if (c4 + 0 < p.din) v[u].x = src[0];
                        if (c4 + 1 < p.din) v[u].y = src[1];
                        if (c4 + 2 < p.din) v[u].z = src[2];
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t e = e0 + 256u * u;
            if (e < 64 * c4n) {
                const uint32_t r = e / c4n, c4 = (e % c4n) * 4;
                *reinterpret_cast<float4*>(&smf[r * ld + c4]) = v[u];
            }
        }
    }
    __syncthreads();
    float acc[2][2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int l = 0; l < 8; ++l) acc[a][b][l] = 0.f;
    const uint32_t kmain = (p.din >> 3) << 3;
    for (uint32_t k = 0; k < kmain; k += 8) {
        float4 xv[2][2], wv[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float4* xp = reinterpret_cast<const float4*>(&xs[(2 * tq + a) * ld + k]);
            xv[a][0] = xp[0]; xv[a][1] = xp[1];
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float4* wp = reinterpret_cast<const float4*>(&ws[(to + 16 * b) * ld + k]);
            wv[b][0] = wp[0]; wv[b][1] = wp[1];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acc[a][b][0] = acc[a][b][0] + wv[b][0].x * xv[a][0].x;
                acc[a][b][1] = acc[a][b][1] + wv[b][0].y * xv[a][0].y;
                acc[a][b][2] = acc[a][b][2] + wv[b][0].z * xv[a][0].z;
                acc[a][b][3] = acc[a][b][3] + wv[b][0].w * xv[a][0].w;
                acc[a][b][4] = acc[a][b][4] + wv[b][1].x * xv[a][1].x;
                acc[a][b][5] = acc[a][b][5] + wv[b][1].y * xv[a][1].y;
                acc[a][b][6] = acc[a][b][6] + wv[b][1].z * xv[a][1].z;
                acc[a][b][7] = acc[a][b][7] + wv[b][1].w * xv[a][1].w;
            }
    }
    // fold, tail steps (x and W tiles are zero beyond din, so the masked step is a full one), bias, ReLU
    const uint32_t rem = p.din & 7u;
    float outv[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float m[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = acc[a][b][j + 4] + acc[a][b][j];
            const float* xr = &xs[(2 * tq + a) * ld + kmain];
            const float* wr = &ws[(to + 16 * b) * ld + kmain];
            uint32_t kk = 0;
            if (rem >= 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) m[j] = m[j] + wr[j] * xr[j];
                kk = 4;
            }
            if (rem > kk) {
#pragma unroll
                for (int j = 0; j < 4; ++j) m[j] = m[j] + wr[kk + j] * xr[kk + j];
            }
            const uint32_t og = to + 16 * b;
            const float dist = -((m[0] + m[1]) + (m[2] + m[3]));  // Angular::Dist
            float v = 0.f;
            v = v - dist;                                  // support_func.h:627
            v = v + (og < p.dout ? p.bias[og] : 0.f);      // :628
            if (RELU && v < 0.f) v = 0.f;                  // :629-631
            outv[a][b] = v;
        }
    if constexpr (NORM) {
        __syncthreads();  // tiles are dead: reuse xs as [32][kNormLd] output rows
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                if (to + 16 * b < (int)p.dout) smf[(2 * tq + a) * kNormLd + to + 16 * b] = outv[a][b];
        __syncthreads();
        mlp_normalize_rows(p, smf, qbase, t);
    } else {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const uint32_t qg = qbase + 2 * tq + a;
            if (qg >= p.nq) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b)
                if (to + 16 * b < (int)p.dout) p.out[(size_t)qg * p.ostride + to + 16 * b] = outv[a][b];
        }
    }
}

// ------------------------------------------------------------------------------------------
// The THROUGHPUT OPTION (GBNNS_FLAG_MFMA_PROJECTION; never the default, never in the bench's `value`): a layer as a GEMM on
// the matrix cores, v_mfma_f32_32x32x2_f32.  Its sums are ONE k-ordered fma chain per output (one rounding per product-and-
// sum), not the reference's eight separately rounded running sums (support_func.h:131-163): outputs differ from the exact
// kernels' in the last bits (<= a few 1e-7 on unit-norm q_low), and through the walk's compare-driven control flow a few
// answers of a batch can differ -- the bench reports how many (throughput_option.id_mismatches_vs_reference).
// Block = 4 wavefronts = 64 queries x 64 neurons (a 32 x 32 accumulator tile each), x / W tiles of 32 k-values through LDS
// (rows padded to 33 floats: the 32 rows an operand read touches hit 32 banks).
constexpr int kMQ = 64, kMO = 64, kMK = 32, kMLd = kMK + 1;
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool RELU>
__global__ __launch_bounds__(256) void mlp_layer_mfma_kernel(LayerParams p) {
    __shared__ float xs[kMQ * kMLd];
    __shared__ float ws[kMO * kMLd];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t qbase = blockIdx.x * kMQ, obase = blockIdx.y * kMO;
    const int qoff = (wave & 1) * 32, ooff = (wave >> 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // staging role: rows t / 8 and t / 8 + 32 of each tile, floats 4 (t % 8) .. + 3 of the chunk
    const int srow = t >> 3, sc4 = (t & 7) * 4;
    for (uint32_t k0 = 0; k0 < p.din; k0 += kMK) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t r = srow + 32 * h, qg = qbase + r, og = obase + r;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t k = k0 + sc4 + i;
                xs[r * kMLd + sc4 + i] = (qg < p.nq && k < p.din) ? p.x[(size_t)qg * p.xstride + k] : 0.f;
                ws[r * kMLd + sc4 + i] = (og < p.dout && k < p.din) ? p.w[(size_t)og * p.wstride + k] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int s2 = 0; s2 < kMK / 2; ++s2) {
            const float a = xs[(qoff + (lane & 31)) * kMLd + 2 * s2 + (lane >> 5)];
            const float b = ws[(ooff + (lane & 31)) * kMLd + 2 * s2 + (lane >> 5)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // accumulator register r of lane l: query row 8 (r / 4) + 4 (l / 32) + r % 4, neuron column l % 32
    const uint32_t og = obase + ooff + (lane & 31);
    const float bs = og < p.dout ? p.bias[og] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const uint32_t qg = qbase + qoff + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        float v = acc[r] + bs;
        if (RELU && v < 0.f) v = 0.f;
        if (qg < p.nq && og < p.dout) p.out[(size_t)qg * p.ostride + og] = v;
    }
}

// support_func.h:636-642 normalizeVector: norm = sqrt(L2Metric.Dist(y, zeros)); y[i] /= norm.
__global__ __launch_bounds__(256) void normalize_kernel(float* y, uint32_t stride, uint32_t dim,
                                                        uint32_t nq) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    float* r = y + (size_t)q * stride;
    const uint32_t steps = dim >> 2;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (uint32_t t = 0; t < steps; ++t) {
        const float e0 = r[4 * t + 0] - 0.f, e1 = r[4 * t + 1] - 0.f;
        const float e2 = r[4 * t + 2] - 0.f, e3 = r[4 * t + 3] - 0.f;
        s0 = s0 + e0 * e0; s1 = s1 + e1 * e1; s2 = s2 + e2 * e2; s3 = s3 + e3 * e3;
    }
    float norm = ((s0 + s1) + s2) + s3;
    norm = __builtin_sqrtf(norm);  // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt); __fsqrt_rn maps to the native sqrt here
    for (uint32_t i = 0; i < dim; ++i) r[i] = __fdiv_rn(r[i], norm);
    for (uint32_t i = dim; i < stride; ++i) r[i] = 0.f;
}

__global__ void fill_u32_kernel(uint32_t* p, uint32_t v, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += step) p[i] = v;
}

}  // namespace

hipError_t launch_mlp_layer(const LayerParams& p, hipStream_t s) {
    if (p.nq == 0 || p.dout == 0) return hipSuccess;
    const dim3 grid((p.nq + kTQ - 1) / kTQ, (p.dout + kTO - 1) / kTO);
    const bool aligned = p.xstride % 4 == 0 && p.wstride % 4 == 0 && (reinterpret_cast<uintptr_t>(p.x) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(p.w) & 15) == 0;
    if (aligned && p.dout <= 32u && p.din <= 256u && (!p.normalize || !p.relu)) {
        // narrow layer: whole-K staging, one load phase (the last projection layer)
        const uint32_t kpad = (p.din + 7u) & ~7u;
        const size_t lds = std::max<size_t>((size_t)64 * (kpad + 4), (size_t)32 * kNormLd) * sizeof(float);  // tiles, or the rows to normalise
        const dim3 gn((p.nq + 31) / 32);
        hipError_t e = hipSuccess;
        if (p.normalize) {
            e = set_lds(mlp_narrow_kernel<false, true>, lds);
            if (e == hipSuccess) hipLaunchKernelGGL((mlp_narrow_kernel<false, true>), gn, dim3(256), lds, s, p);
        } else if (p.relu) {
            e = set_lds(mlp_narrow_kernel<true, false>, lds);
            if (e == hipSuccess) hipLaunchKernelGGL((mlp_narrow_kernel<true, false>), gn, dim3(256), lds, s, p);
        } else {
            e = set_lds(mlp_narrow_kernel<false, false>, lds);
            if (e == hipSuccess) hipLaunchKernelGGL((mlp_narrow_kernel<false, false>), gn, dim3(256), lds, s, p);
        }
        return e != hipSuccess ? e : hipGetLastError();
    }
    if (p.normalize && !p.relu && p.dout <= (uint32_t)kTO) {  // fused normalizeVector (one block column)
        if (aligned) hipLaunchKernelGGL((mlp_layer_vec_kernel<false, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((mlp_layer_kernel<false, true>), grid, dim3(256), 0, s, p);
        return hipGetLastError();
    }
    if (aligned && p.small_footprint && !p.normalize && p.dout >= 64u) {
        const dim3 gsw((p.nq + kSwQ - 1) / kSwQ, (p.dout + kSwO - 1) / kSwO);
        if (p.relu) hipLaunchKernelGGL((mlp_layer_sw_kernel<true>), gsw, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((mlp_layer_sw_kernel<false>), gsw, dim3(256), 0, s, p);
    } else if (aligned) {
        if (p.relu) hipLaunchKernelGGL((mlp_layer_vec_kernel<true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((mlp_layer_vec_kernel<false>), grid, dim3(256), 0, s, p);
    } else {
        if (p.relu) hipLaunchKernelGGL((mlp_layer_kernel<true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((mlp_layer_kernel<false>), grid, dim3(256), 0, s, p);
    }
    if (p.normalize) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        return launch_normalize(p.out, p.ostride, p.dout, p.nq, s);
    }
    return hipGetLastError();
}

hipError_t launch_mlp_layer_mfma(const LayerParams& p, hipStream_t s) {
    if (p.nq == 0 || p.dout == 0) return hipSuccess;
    const dim3 grid((p.nq + kMQ - 1) / kMQ, (p.dout + kMO - 1) / kMO);
    if (p.relu) hipLaunchKernelGGL((mlp_layer_mfma_kernel<true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((mlp_layer_mfma_kernel<false>), grid, dim3(256), 0, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !p.normalize) return e;
    return launch_normalize(p.out, p.ostride, p.dout, p.nq, s);  // (the exact normalise step on the approximate outputs)
}

hipError_t launch_normalize(float* y, uint32_t stride, uint32_t dim, uint32_t nq, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    hipLaunchKernelGGL(normalize_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, y, stride, dim, nq);
    return hipGetLastError();
}

hipError_t launch_fill_u32(uint32_t* p, uint32_t v, size_t count, hipStream_t s) {
    if (count == 0) return hipSuccess;
    size_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, v, count);
    return hipGetLastError();
}

}  // namespace gbnns
