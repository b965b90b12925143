// project.hip -- the three layers of GetLowQueryFromNet (support_func.h:645-658) in ONE launch, for nets whose
// activations fit a CU's LDS (SIFT 128-256-256-32, GloVe 200-256-256-32, DEEP 96-128-128-32).
//
// The per-layer kernels (kernels.hip: mlp_layer_vec_kernel x2 + mlp_narrow_kernel) take 0.07 ms per 10 000-query
// batch where the arithmetic alone is 0.027 ms: the hidden activations travel through HBM twice, every launch ends
// in a partly filled last "round" of workgroups, and the narrow last layer is latency bound.  Here a workgroup takes
// M queries through all three layers: x, h1 and h2 live in LDS ([M][ld] f32, two buffers used alternately), only
// the weights stream in (tiles of 128 neurons x 32 inputs, next tile prefetched into registers), normalizeVector
// (:636-642) runs on the M finished rows, and M is chosen on the host so that the grid is one balanced set of
// workgroups (mlp_fused_plan).
//
// Arithmetic = the per-layer kernels' = the reference's: per output neuron Angular::Dist's 8 running sums over the
// inputs in order (:134-147, separate multiply and add), fold 8 -> 4, optional 4-wide and masked steps (:148-159),
// (m0 + m1) + (m2 + m3), `0 - dist`, `+ bias`, ReLU (:624-633); the norm is L2Metric::Dist(y, 0) (4 running sums,
// d_low % 4 tail ignored), correctly rounded sqrt and divide.  Bit-identical outputs (tests compare q_low bits).
//
// 512 threads = 8 wavefronts: wavefront (rg, nh) owns rows [rg Mw, (rg + 1) Mw) (Mw = M / 4 <= 12) and neurons
// nh * 64 .. + 63 of the current 128-neuron pass; lane = (qsub = lane / 16, to = lane % 16): rows qsub, qsub + 4,
// qsub + 8 of the wavefront's rows x neurons to, to + 16, to + 32, to + 48 -- 3 x 4 outputs x 8 running sums.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace gbnns {

namespace {

constexpr int kFThreads = 512, kFNeurons = 128, kFK = 32, kFLd = kFK + 4;

__device__ __forceinline__ uint32_t round8(uint32_t v) { return (v + 7u) & ~7u; }

// One layer over the workgroup's rows: in [M][ldin] (LDS, zero beyond din up to a multiple of 8) -> outb [M][ldout].
template <bool RELU>
__device__ __forceinline__ void fused_layer(const float* in, uint32_t ldin, uint32_t din, const float* __restrict__ w,
                                            uint32_t wstride, const float* __restrict__ bias, uint32_t dout, float* outb,
                                            uint32_t ldout, float* wt, int t, int mw) {
    const int wave = t >> 6, lane = t & 63;
    const int rg = wave >> 1, nh = wave & 1;
    const int qsub = lane >> 4, to = lane & 15;
    const uint32_t kmain = (din >> 3) << 3;
    const uint32_t rem8 = din & 7u;
    // staging role: rows srow and srow + 64 of the 128-neuron tile, columns sc4 .. sc4 + 3 of the 32-wide k chunk
    const int srow = t >> 3, sc4 = (t & 7) * 4;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int row[3];
    bool rv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        rv[i] = qsub + 4 * i < mw;
        row[i] = rg * mw + (rv[i] ? qsub + 4 * i : 0);
    }
    for (uint32_t obase = 0; obase < dout; obase += kFNeurons) {
        float acc[3][4][8];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int l = 0; l < 8; ++l) acc[i][b][l] = 0.f;
        const uint32_t o0 = obase + srow, o1 = obase + srow + 64;
        auto fetch = [&](uint32_t k0, float4& f0, float4& f1) {
            const bool kin = k0 + sc4 < kmain;  // kmain is a multiple of 8, sc4 of 4: a float4 is wholly in or out
            f0 = (kin && o0 < dout) ? *reinterpret_cast<const float4*>(w + (size_t)o0 * wstride + k0 + sc4) : zero4;
            f1 = (kin && o1 < dout) ? *reinterpret_cast<const float4*>(w + (size_t)o1 * wstride + k0 + sc4) : zero4;
        };
        float4 f0, f1;
        fetch(0, f0, f1);
        for (uint32_t k0 = 0; k0 < kmain; k0 += kFK) {
            const uint32_t kc = (kmain - k0 < (uint32_t)kFK) ? (kmain - k0) : (uint32_t)kFK;
            *reinterpret_cast<float4*>(&wt[srow * kFLd + sc4]) = f0;
            *reinterpret_cast<float4*>(&wt[(srow + 64) * kFLd + sc4]) = f1;
            __syncthreads();
            if (k0 + kFK < kmain) fetch(k0 + kFK, f0, f1);  // in flight during the arithmetic below
            for (uint32_t s = 0; s < kc; s += 8) {
                float4 xv[3][2], wv[4][2];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float4* xp = reinterpret_cast<const float4*>(&in[(size_t)row[i] * ldin + k0 + s]);
                    xv[i][0] = xp[0];
                    xv[i][1] = xp[1];
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float4* wp = reinterpret_cast<const float4*>(&wt[(nh * 64 + to + 16 * b) * kFLd + s]);
                    wv[b][0] = wp[0];
                    wv[b][1] = wp[1];
                }
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        acc[i][b][0] = acc[i][b][0] + wv[b][0].x * xv[i][0].x;
                        acc[i][b][1] = acc[i][b][1] + wv[b][0].y * xv[i][0].y;
                        acc[i][b][2] = acc[i][b][2] + wv[b][0].z * xv[i][0].z;
                        acc[i][b][3] = acc[i][b][3] + wv[b][0].w * xv[i][0].w;
                        acc[i][b][4] = acc[i][b][4] + wv[b][1].x * xv[i][1].x;
                        acc[i][b][5] = acc[i][b][5] + wv[b][1].y * xv[i][1].y;
                        acc[i][b][6] = acc[i][b][6] + wv[b][1].z * xv[i][1].z;
                        acc[i][b][7] = acc[i][b][7] + wv[b][1].w * xv[i][1].w;
                    }
            }
            __syncthreads();
        }
        // fold, tail steps (the input rows are zero beyond din, the weight rows are zero padded to a multiple of 8:
        // the masked step is a full one), bias, ReLU
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (!rv[i]) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t og = obase + nh * 64 + to + 16 * b;
                if (og >= dout) continue;
                float m[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) m[j] = acc[i][b][j + 4] + acc[i][b][j];
                const float* xr = &in[(size_t)row[i] * ldin + kmain];
                const float* wr = w + (size_t)og * wstride + kmain;
                uint32_t kk = 0;
                if (rem8 >= 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) m[j] = m[j] + wr[j] * xr[j];
                    kk = 4;
                }
                if (rem8 > kk) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) m[j] = m[j] + wr[kk + j] * xr[kk + j];
                }
                const float dist = -((m[0] + m[1]) + (m[2] + m[3]));  // Angular::Dist
                float v = 0.f;
                v = v - dist;          // support_func.h:627
                v = v + bias[og];      // :628
                if (RELU && v < 0.f) v = 0.f;  // :629-631
                outb[(size_t)row[i] * ldout + og] = v;
            }
        }
    }
    __syncthreads();  // the layer's outputs are complete (and the weight tile is free)
}

__global__ __launch_bounds__(kFThreads) void mlp_fused_kernel(FusedMlpParams p) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    const int t = threadIdx.x;
    const int mw = (int)p.rows_per_wave, M = 4 * mw;
    const uint32_t lda = p.lda, ldb = p.ldb;
    float* bufa = smf;                          // x, then h2
    float* bufb = bufa + (size_t)M * lda;       // h1, then y
    float* wt = bufb + (size_t)M * ldb;         // [128][36] weight tile
    float* nsum = wt + kFNeurons * kFLd;        // [M][4] partial sums of the norm
    const uint32_t qbase = blockIdx.x * (uint32_t)M;
    // zero both activation buffers (pad columns must read as zero), then stage the workgroup's queries
    for (uint32_t e = t; e < (uint32_t)M * (lda + ldb) / 4u; e += kFThreads)
        reinterpret_cast<float4*>(smf)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (p.xvec) {
        const uint32_t c4n = p.d >> 2;
        for (uint32_t e = t; e < (uint32_t)M * c4n; e += kFThreads) {
            const uint32_t r = e / c4n, c = e % c4n;
            if (qbase + r < p.nq)
                *reinterpret_cast<float4*>(&bufa[(size_t)r * lda + 4 * c]) =
                    *reinterpret_cast<const float4*>(p.x + (size_t)(qbase + r) * p.xstride + 4 * c);
        }
    } else {
        for (uint32_t e = t; e < (uint32_t)M * p.d; e += kFThreads) {
            const uint32_t r = e / p.d, c = e % p.d;
            if (qbase + r < p.nq) bufa[(size_t)r * lda + c] = p.x[(size_t)(qbase + r) * p.xstride + c];
        }
    }
    __syncthreads();
    fused_layer<true>(bufa, lda, p.d, p.w1, p.ws1, p.b1, p.dh, bufb, ldb, wt, t, mw);
    // h1's pad columns are zero already; bufa still holds x beyond h2's columns only where d > dh: clear that part
    if (p.d > p.dh) {
        const uint32_t from = p.dh, to_ = round8(p.d);
        for (uint32_t e = t; e < (uint32_t)M * (to_ - from); e += kFThreads)
            bufa[(size_t)(e / (to_ - from)) * lda + from + e % (to_ - from)] = 0.f;
        __syncthreads();
    }
    fused_layer<true>(bufb, ldb, p.dh, p.w2, p.ws2, p.b2, p.dh, bufa, lda, wt, t, mw);
    fused_layer<false>(bufa, lda, p.dh, p.w3, p.ws3, p.b3, p.dl, bufb, ldb, wt, t, mw);
    // normalizeVector (support_func.h:636-642) on the finished rows: 8 threads per row
    const int q = t >> 3, part = t & 7;
    if (q < M) {
        const float* y = bufb + (size_t)q * ldb;
        if (part < 4) {
            const uint32_t steps = p.dl >> 2;
            float sc = 0.f;
            for (uint32_t k = 0; k < steps; ++k) {
                const float e = y[4 * k + part] - 0.f;
                sc = sc + e * e;
            }
            nsum[q * 4 + part] = sc;
        }
    }
    __syncthreads();
    if (q < M && qbase + q < p.nq) {
        const float* y = bufb + (size_t)q * ldb;
        float norm = ((nsum[q * 4 + 0] + nsum[q * 4 + 1]) + nsum[q * 4 + 2]) + nsum[q * 4 + 3];
        norm = __builtin_sqrtf(norm);  // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
        float* r = p.out + (size_t)(qbase + q) * p.ostride;
        for (uint32_t i = part; i < p.dl; i += 8) r[i] = __fdiv_rn(y[i], norm);
        for (uint32_t i = p.dl + part; i < p.ostride; i += 8) r[i] = 0.f;
    }
}

}  // namespace

// Rows per wavefront (1..12) that minimise the launch's makespan on 256 CUs, one workgroup per CU at a time:
// ceil(workgroups / 256) rounds x ceil(rows / 4) row slots per lane; 0 = the net does not fit the LDS this way.
uint32_t mlp_fused_plan(uint32_t d, uint32_t dh, uint32_t dl, uint32_t nq, uint32_t* lda, uint32_t* ldb, size_t* lds_bytes) {
    if (nq == 0 || d == 0 || dh == 0 || dl == 0) return 0;
    const uint32_t r8d = (d + 7u) & ~7u, r8h = (dh + 7u) & ~7u, r8l = (dl + 7u) & ~7u;
    const uint32_t a = (r8d > r8h ? r8d : r8h) + 4u, b = (r8h > r8l ? r8h : r8l) + 4u;
    uint32_t best = 0;
    uint64_t best_cost = ~0ull;
    for (uint32_t mw = 1; mw <= 12; ++mw) {
        const size_t bytes = ((size_t)4 * mw * (a + b) + (size_t)kFNeurons * kFLd + (size_t)16 * mw) * 4;
        if (bytes > 160 * 1024) break;
        const uint64_t groups = ((uint64_t)nq + 4u * mw - 1) / (4u * mw);
        const uint64_t cost = ((groups + 255) / 256) * ((mw + 3) / 4);
        if (cost <= best_cost) {  // ties -> more rows per workgroup (the weights are read once per workgroup)
            best_cost = cost;
            best = mw;
        }
    }
    if (best) {
        *lda = a;
        *ldb = b;
        *lds_bytes = ((size_t)4 * best * (a + b) + (size_t)kFNeurons * kFLd + (size_t)16 * best) * 4;
    }
    return best;
}

hipError_t launch_mlp_fused(const FusedMlpParams& p, size_t lds_bytes, hipStream_t s) {
    if (p.nq == 0) return hipSuccess;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fused_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    const uint32_t M = 4u * p.rows_per_wave;
    hipLaunchKernelGGL(mlp_fused_kernel, dim3((p.nq + M - 1) / M), dim3(kFThreads), lds_bytes, s, p);
    return hipGetLastError();
}

}  // namespace gbnns
