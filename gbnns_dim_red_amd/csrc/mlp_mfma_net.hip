// mlp_mfma_net.hip -- the throughput option of the projection (GBNNS_FLAG_MFMA_PROJECTION; NOT bit-exact, never the default): the whole
// three-layer net (GetLowQueryFromNet, support_func.h:645-658: relu, relu, normalise) in ONE launch on the matrix cores,
// v_mfma_f32_16x16x4_f32, activations in LDS from the first layer's input to the normalised output.
//
// Round 5's option was three per-layer launches, each staging 64 x 64 tiles of x AND W through LDS behind two barriers per 16 inputs:
// 0.096 ms on the SIFT net, 13 % of the f32 matrix peak, slower than the exact packed-f32 kernel (0.053 ms).  What this kernel does:
//   * ONE workgroup per CU takes that CU's share of the batch -- ceil(n_q / CUs) queries, up to three row blocks of 16 -- so every CU
//     has the same work (tiles of 32 queries left 57 CUs with two tiles and 199 with one on a 10 000-query batch) and the net's weights
//     are streamed out of L2 ONCE per CU, each B operand multiplied into all of the workgroup's row blocks (a 16-query tile per weight
//     pass asks the L2 for more than a CU can take: measured);
//   * weights never pass through LDS: they are repacked once per handle into the order the instruction reads its B operand -- for every
//     block of 16 neurons and every four instructions (16 inputs) 64 lanes x 4 floats, lane = [input mod 4][neuron] -- so a wavefront's B
//     operands of four instructions are ONE coalesced 1 KB load, 16 bytes per lane (the packed net is 0.43 MB: it stays in L2);
//   * activations live in LDS transposed, [input][rows + 1]: the A operands are conflict-free reads of consecutive floats, and so are
//     the epilogue's stores (lane = neuron, odd stride);
//   * eight wavefronts (two per SIMD), each owning two of a layer's sixteen neuron blocks x all row blocks: six accumulation chains per
//     wavefront, no barrier inside a layer; three register sets of B operands rotate, so the loads of the next two groups of eight
//     instructions are in flight while a group is multiplied and the wait in front of a group is a counted one;
//   * the last layer (two or four neuron blocks) splits its INPUTS over the wavefronts, partial sums meet in LDS, bias, norm
//     (4 running sums, support_func.h:636-642 shape) and the divide follow in the same launch.
// 10 000 x 128 -> 256 -> 256 -> 32: 250 workgroups of 40 queries (48 rows of matrix work); 2.13 GFLOP against a dense f32 matrix peak of
// 157 TFLOP/s = 13.6 us.
// Sums: one k-ordered fma chain per output instead of the reference's eight separately rounded running sums -- tests bound the error
// (2e-6 on unit-norm outputs) and the share of answers that differ from the exact path's (<= 0.5 %).
#include "kernels.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace gbnns {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kGroup = 8;      // instructions (steps of four inputs) whose B operands come in together: two float4 per lane
constexpr int kWaves = 8;      // wavefronts per workgroup: two per SIMD (sixteen: the same 31 us -- shorter layers, longer barriers)
constexpr int kMaxRB = 3;      // row blocks of 16 queries per workgroup (LDS: two activation images of up to 304 inputs x 49 floats)

struct MfmaNet {
    const float* x;            // [nq x xstride]
    uint32_t xstride, nq, d;
    uint32_t rows;             // queries per workgroup (<= 16 RB)
    const float* wp[3];        // packed weights (mlp_mfma_pack_kernel)
    const float* bp[3];        // biases padded with zeros to whole neuron blocks
    uint32_t kp[3];            // inputs of a layer, padded to whole groups of instructions (per wavefront share where the inputs are split)
    uint32_t nt[3];            // neuron blocks of 16 per layer
    float* out;                // [nq x ostride]
    uint32_t ostride, d_low;
    uint32_t rows_a, rows_b;   // input rows of the two activation images
    unsigned long long* stamps;   // diagnostic (GBNNS_MFMA_STAMPS=1): [workgroups x 8] s_memtime at the phase ends, wavefront 0
};

// W [dout x wstride] -> packed [nt][kp / 16][64 lanes][4]: lane l = 16 h + n of group g holds, in element e, the B-operand word of
// instruction j = 4 g + e: W[16 t + n][4 j + h] (zero outside) -- one 16-byte load per lane feeds four instructions; bias -> [16 nt], zero padded
__global__ void mlp_mfma_pack_kernel(const float* w, uint32_t wstride, uint32_t din, uint32_t dout, const float* bias, uint32_t kp, uint32_t nt,
                                     float* wp, float* bp) {
    const size_t total = (size_t)nt * (kp / 4) * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t e = (uint32_t)(i & 3), l = (uint32_t)((i >> 2) & 63), n = l & 15u, h = l >> 4;
        const size_t tg = i >> 8;
        const uint32_t groups4 = kp / 16u;
        const uint32_t g = (uint32_t)(tg % groups4), t = (uint32_t)(tg / groups4);
        const uint32_t o = 16u * t + n, k = 4u * (4u * g + e) + h;
        wp[i] = (o < dout && k < din) ? w[(size_t)o * wstride + k] : 0.f;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)nt * 16; i += (size_t)gridDim.x * blockDim.x)
        bp[i] = i < dout ? bias[i] : 0.f;
}

// Neuron blocks t0 (and t1 when TWO) of one layer, all RB row blocks: in [kp][PAD] -> out [16 nt][PAD].
// accumulator register r of lane l = (h, m) of an instruction: query row 4 h + r of its row block, neuron column m
template <int RB, bool RELU, bool TWO>
__device__ __forceinline__ void mfma_blocks(const float* in, float* outp, const float* wp, const float* bp, uint32_t kp, uint32_t t0, uint32_t t1, int lane) {
    constexpr int PAD = 16 * RB + 1;
    const uint32_t steps = kp / 4u;
    const uint32_t m = (uint32_t)lane & 15u, h = (uint32_t)lane >> 4;
    const float* a_base = in + h * PAD + m;   // A operand of instruction j, row block rb: in[4 j + h][16 rb + m]
    const float4* b0 = reinterpret_cast<const float4*>(wp + (size_t)t0 * steps * 64) + lane;
    const float4* b1 = reinterpret_cast<const float4*>(wp + (size_t)t1 * steps * 64) + lane;
    f32x4 acc0[RB], acc1[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc0[rb][r] = 0.f; acc1[rb][r] = 0.f; }
    const uint32_t groups = steps / kGroup;   // (steps is a multiple of kGroup: the host pads)
    const uint32_t last = groups - 1u;
    // three register sets in rotation, no copies between them: a set is requested again right after its group has been multiplied, two
    // groups before it is needed, and every group requests exactly as many loads as every other (the last ones re-request the last
    // group): the wait in front of a group is a counted one.  The scheduling barriers keep the compiler from sinking the loads down to
    // their uses (it did: one load, a full wait, four instructions, the next load ...).
    struct Set { float4 x0, y0, x1, y1; };
    Set sa, sb, sc;
    auto request = [&](uint32_t g, Set& q) {
        const uint32_t gg = g < last ? g : last;
        q.x0 = b0[(size_t)(2u * gg) * 64]; q.y0 = b0[(size_t)(2u * gg + 1u) * 64];
        if constexpr (TWO) { q.x1 = b1[(size_t)(2u * gg) * 64]; q.y1 = b1[(size_t)(2u * gg + 1u) * 64]; }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto multiply = [&](uint32_t g, const Set& q) {
        const float c0[kGroup] = {q.x0.x, q.x0.y, q.x0.z, q.x0.w, q.y0.x, q.y0.y, q.y0.z, q.y0.w};
        const float c1[kGroup] = {q.x1.x, q.x1.y, q.x1.z, q.x1.w, q.y1.x, q.y1.y, q.y1.z, q.y1.w};
#pragma unroll
        for (int u = 0; u < kGroup; ++u) {
            const float* ar = a_base + (size_t)(g * kGroup + u) * 4 * PAD;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float a = ar[16 * rb];
                acc0[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, c0[u], acc0[rb], 0, 0, 0);
                if constexpr (TWO) acc1[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, c1[u], acc1[rb], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    request(0, sa);
    request(1, sb);
    request(2, sc);
    for (uint32_t g = 0; g < groups; g += 3) {
        multiply(g, sa);
        request(g + 3u, sa);
        if (g + 1u < groups) { multiply(g + 1u, sb); request(g + 4u, sb); }
        if (g + 2u < groups) { multiply(g + 2u, sc); request(g + 5u, sc); }
    }
    auto store = [&](const f32x4* acc, uint32_t t) {
        const float bs = bp[16u * t + m];
        float* col = outp + (size_t)(16u * t + m) * PAD + 4u * h;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[rb][r] + bs;
                if (RELU && v < 0.f) v = 0.f;
                col[16 * rb + r] = v;
            }
    };
    store(acc0, t0);
    if constexpr (TWO) store(acc1, t1);
}
template <int RB, bool RELU>
__device__ __forceinline__ void mfma_layer(const float* in, float* outp, const float* wp, const float* bp, uint32_t kp, uint32_t nt, int wave, int lane) {
    for (uint32_t t0 = (uint32_t)wave; t0 < nt; t0 += 2 * kWaves) {   // this wavefront's neuron blocks w, w + kWaves, ...: two at a time
        const uint32_t t1 = t0 + kWaves;
        if (t1 < nt) mfma_blocks<RB, RELU, true>(in, outp, wp, bp, kp, t0, t1, lane);
        else mfma_blocks<RB, RELU, false>(in, outp, wp, bp, kp, t0, t0, lane);
    }
}

template <int RB>
__global__ __launch_bounds__(64 * kWaves) void mlp_mfma_net_kernel(MfmaNet p) {
    constexpr int PAD = 16 * RB + 1, ROWS = 16 * RB;
    extern __shared__ __attribute__((aligned(16))) float act[];
    float* const A = act;                              // x, then the second hidden layer, then the summed outputs
    float* const B = act + (size_t)p.rows_a * PAD;     // the first hidden layer, then the last layer's partial sums
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    const uint32_t q0 = blockIdx.x * p.rows;
    const uint32_t nrow = min(p.rows, p.nq - q0);       // queries of this workgroup (the grid has no empty ones)
    auto stamp = [&](int i) {
        if (p.stamps && threadIdx.x == 0) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            p.stamps[(size_t)blockIdx.x * 8 + i] = t;
        }
    };
    stamp(0);
    // ---- the workgroup's inputs, transposed: A[k][m] = x[q0 + m][k]; rows beyond its queries and inputs d .. kp[0] zero
    {
        const uint32_t d4 = p.d >> 2;   // (d % 4 == 0: the host checks)
        for (uint32_t i = threadIdx.x; i < (uint32_t)ROWS * d4; i += 64 * kWaves) {
            const uint32_t m = i / d4, k4 = i % d4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < nrow) v = *reinterpret_cast<const float4*>(p.x + (size_t)(q0 + m) * p.xstride + 4u * k4);
            float* dst = A + (size_t)(4u * k4) * PAD + m;
            dst[0] = v.x; dst[PAD] = v.y; dst[2 * PAD] = v.z; dst[3 * PAD] = v.w;
        }
        for (uint32_t i = threadIdx.x; i < (p.kp[0] - p.d) * (uint32_t)ROWS; i += 64 * kWaves) A[(size_t)(p.d + i / ROWS) * PAD + (i % ROWS)] = 0.f;
    }
    __syncthreads();
    stamp(1);
    mfma_layer<RB, true>(A, B, p.wp[0], p.bp[0], p.kp[0], p.nt[0], wave, lane);
    stamp(2);
    // (rows 16 nt[0] .. kp[1] of B: the padding of the next layer's inputs)
    for (uint32_t i = threadIdx.x; i < (p.kp[1] - 16u * p.nt[0]) * (uint32_t)ROWS; i += 64 * kWaves) B[(size_t)(16u * p.nt[0] + i / ROWS) * PAD + (i % ROWS)] = 0.f;
    __syncthreads();
    stamp(3);
    mfma_layer<RB, true>(B, A, p.wp[1], p.bp[1], p.kp[1], p.nt[1], wave, lane);
    stamp(4);
    for (uint32_t i = threadIdx.x; i < (p.kp[2] - 16u * p.nt[1]) * (uint32_t)ROWS; i += 64 * kWaves) A[(size_t)(16u * p.nt[1] + i / ROWS) * PAD + (i % ROWS)] = 0.f;
    // (the last layer's first B operands do not depend on the activations: requested in front of the barrier, they arrive during it)
    const uint32_t ksplit = (uint32_t)kWaves / p.nt[2];
    const uint32_t steps3 = p.kp[2] / 4u, share3 = steps3 / ksplit;   // (a multiple of kGroup: the host pads)
    const uint32_t t3 = (uint32_t)wave % p.nt[2], ks3 = min((uint32_t)wave / p.nt[2], ksplit - 1u);
    const float4* b3 = reinterpret_cast<const float4*>(p.wp[2] + (size_t)t3 * steps3 * 64) + lane;
    float4 pa = b3[(size_t)(ks3 * share3 / 4u) * 64], pb = b3[(size_t)(ks3 * share3 / 4u + 1u) * 64];
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    stamp(5);
    // ---- last layer: its nt[2] (2 .. 4: d_low 32 .. 64) neuron blocks x ksplit = kWaves / nt[2] shares of the inputs, one (block, share)
    // per wavefront (three blocks leave two wavefronts idle), all row blocks; partial sums -> B [share][block][neuron][PAD]
    (void)pa; (void)pb;
    if ((uint32_t)wave < ksplit * p.nt[2]) {
        const uint32_t steps = steps3, share = share3;
        const uint32_t m = (uint32_t)lane & 15u, h = (uint32_t)lane >> 4;
        const uint32_t t = t3, ks = ks3;
        const float* a_base = A + h * PAD + m;
        const float4* b0 = b3;
        (void)steps;
        f32x4 acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[rb][r] = 0.f;
        for (uint32_t j = ks * share; j < (ks + 1u) * share; j += kGroup) {
            const float4 ca = pa, cb = pb;
            const uint32_t jn = min(j + kGroup, (ks + 1u) * share - kGroup);   // (the next group, requested before this one is multiplied)
            pa = b0[(size_t)(jn / 4u) * 64]; pb = b0[(size_t)(jn / 4u + 1u) * 64];
            __builtin_amdgcn_sched_barrier(0);
            const float c[kGroup] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
#pragma unroll
            for (int u = 0; u < kGroup; ++u)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_base[(size_t)(j + u) * 4 * PAD + 16 * rb], c[u], acc[rb], 0, 0, 0);
        }
        float* col = B + ((size_t)(ks * p.nt[2] + t) * 16u + m) * PAD + 4u * h;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) col[16 * rb + r] = acc[rb][r];
    }
    __syncthreads();
    // ---- sum of the partial images + bias -> A [neuron][PAD] (the second hidden layer is no longer needed)
    {
        const uint32_t n3 = 16u * p.nt[2];
        for (uint32_t i = threadIdx.x; i < n3 * (uint32_t)ROWS; i += 64 * kWaves) {
            const uint32_t n = i / ROWS, m = i % ROWS;
            float v = 0.f;
            for (uint32_t w = 0; w < ksplit; ++w) v += B[((size_t)w * n3 + n) * PAD + m];
            A[(size_t)n * PAD + m] = v + p.bp[2][n];
        }
    }
    __syncthreads();
    stamp(6);
    // ---- normalizeVector + the output rows (eight threads per query: the norm by four running sums, as the reference shapes it)
    if (threadIdx.x < 8u * nrow) {
        const uint32_t m = threadIdx.x >> 3, part = threadIdx.x & 7u;
        const uint32_t dl = p.d_low;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (uint32_t t = 0; 4u * t + 3u < dl; ++t) {
            const float e0 = A[(size_t)(4u * t) * PAD + m], e1 = A[(size_t)(4u * t + 1u) * PAD + m];
            const float e2 = A[(size_t)(4u * t + 2u) * PAD + m], e3 = A[(size_t)(4u * t + 3u) * PAD + m];
            s0 += e0 * e0; s1 += e1 * e1; s2 += e2 * e2; s3 += e3 * e3;
        }
        const float norm = __builtin_sqrtf(((s0 + s1) + s2) + s3);
        float* row = p.out + (size_t)(q0 + m) * p.ostride;
        for (uint32_t n = part; n < p.ostride; n += 8) row[n] = n < dl ? A[(size_t)n * PAD + m] / norm : 0.f;
    }
}

// the padded shape of a net: k1 / k2 / k3 inputs, n1 / n3 neurons, LDS rows of the two activation images
struct Dims { uint32_t k1, k2, k3, n1, n3, rows_a, rows_b; };
uint32_t pad_to(uint32_t v, uint32_t m) { return (v + m - 1u) / m * m; }
Dims dims_of(uint32_t d, uint32_t dh, uint32_t dl) {
    const uint32_t g = 4u * kGroup;   // inputs per group of instructions
    Dims q;
    q.n1 = pad_to(dh, 16u);
    q.n3 = pad_to(dl, 16u);
    q.k1 = pad_to(d, g);
    q.k2 = pad_to(q.n1, g);
    q.k3 = pad_to(q.n1, g * ((uint32_t)kWaves / (q.n3 / 16u)));
    q.rows_a = std::max(std::max(q.k1, q.k3), q.n3);
    q.rows_b = std::max(std::max(q.k2, q.n1), (uint32_t)kWaves * 16u);
    return q;
}
size_t lds_bytes(const Dims& q, int rb) { return (size_t)(q.rows_a + q.rows_b) * (16 * rb + 1) * 4; }
// row blocks a workgroup can hold (0: the net does not fit even one)
int max_rb(uint32_t d, uint32_t dh, uint32_t dl) {
    if (d % 4u != 0 || d == 0 || dh == 0 || dl == 0) return 0;
    const uint32_t n3 = pad_to(dl, 16u);
    if (n3 / 16u > (uint32_t)kWaves) return 0;   // (one wavefront per neuron block of the last layer at least)
    const Dims q = dims_of(d, dh, dl);
    // every workgroup streams the whole packed net out of L2: worth it while that is a fraction of a megabyte (the SIFT / DEEP / GloVe
    // nets: 0.2 - 0.5 MB), not for the GIST net's 8 MB -- that one keeps the per-layer kernels
    if (((size_t)q.n1 * q.k1 + (size_t)q.n1 * q.k2 + (size_t)q.n3 * q.k3) * 4 > (size_t)2 << 20) return 0;
    int rb = kMaxRB;
    while (rb > 0 && lds_bytes(q, rb) > (size_t)150 * 1024) --rb;
    return rb;
}

template <int RB>
hipError_t launch_t(const NetLaunch& n, const float* packed, uint32_t rows, hipStream_t s) {
    const uint32_t d = n.din[0], dh = n.dout[0], dl = n.dout[2];
    const Dims q = dims_of(d, dh, dl);
    MfmaNet p{};
    p.x = n.x; p.xstride = n.xstride; p.nq = n.nq; p.d = d; p.rows = rows;
    p.wp[0] = packed; p.wp[1] = p.wp[0] + (size_t)q.n1 * q.k1; p.wp[2] = p.wp[1] + (size_t)q.n1 * q.k2;
    p.bp[0] = p.wp[2] + (size_t)q.n3 * q.k3; p.bp[1] = p.bp[0] + q.n1; p.bp[2] = p.bp[1] + q.n1;
    p.kp[0] = q.k1; p.kp[1] = q.k2; p.kp[2] = q.k3;
    p.nt[0] = q.n1 / 16u; p.nt[1] = q.n1 / 16u; p.nt[2] = q.n3 / 16u;
    p.out = n.out; p.ostride = n.ostride; p.d_low = dl;
    p.rows_a = q.rows_a; p.rows_b = q.rows_b;
    const size_t lds = lds_bytes(q, RB);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_mfma_net_kernel<RB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    static const bool want_stamps = getenv("GBNNS_MFMA_STAMPS") != nullptr;
    const unsigned wgs = (n.nq + rows - 1u) / rows;
    if (want_stamps) {
        if (hipMalloc(reinterpret_cast<void**>(&p.stamps), (size_t)wgs * 64) != hipSuccess) p.stamps = nullptr;
        else (void)hipMemsetAsync(p.stamps, 0, (size_t)wgs * 64, s);
    }
    hipLaunchKernelGGL(mlp_mfma_net_kernel<RB>, dim3(wgs), dim3(64 * kWaves), lds, s, p);
    if (want_stamps && p.stamps) {   // diagnostic: mean phase lengths of the workgroups' first wavefronts (s_memtime ticks), once per call
        std::vector<unsigned long long> h((size_t)wgs * 8);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), p.stamps, h.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(p.stamps);
        double seg[6] = {0, 0, 0, 0, 0, 0};
        for (unsigned t = 0; t < wgs; ++t)
            for (int i = 0; i < 6; ++i) seg[i] += (double)(h[(size_t)t * 8 + i + 1] - h[(size_t)t * 8 + i]);
        std::fprintf(stderr, "[gbnns mfma stamps] %u workgroups of %u queries (%d row blocks): staging %.0f, layer 1 %.0f, (zero + barrier %.0f), layer 2 %.0f, (barrier %.0f), layer 3 + sums %.0f ticks\n",
                     wgs, rows, RB, seg[0] / wgs, seg[1] / wgs, seg[2] / wgs, seg[3] / wgs, seg[4] / wgs, seg[5] / wgs);
    }
    return hipGetLastError();
}

}  // namespace

bool mlp_mfma_net_serves(uint32_t d, uint32_t d_hidden, uint32_t d_low) { return max_rb(d, d_hidden, d_low) > 0; }
size_t mlp_mfma_net_packed_floats(uint32_t d, uint32_t d_hidden, uint32_t d_low) {
    const Dims q = dims_of(d, d_hidden, d_low);
    return (size_t)q.n1 * q.k1 + (size_t)q.n1 * q.k2 + (size_t)q.n3 * q.k3 + 2u * (size_t)q.n1 + q.n3;
}
hipError_t launch_mlp_mfma_pack(const NetLaunch& n, float* packed, hipStream_t s) {
    const uint32_t d = n.din[0], dh = n.dout[0], dl = n.dout[2];
    const Dims q = dims_of(d, dh, dl);
    float* w1 = packed; float* w2 = w1 + (size_t)q.n1 * q.k1; float* w3 = w2 + (size_t)q.n1 * q.k2;
    float* b1 = w3 + (size_t)q.n3 * q.k3; float* b2 = b1 + q.n1; float* b3 = b2 + q.n1;
    hipLaunchKernelGGL(mlp_mfma_pack_kernel, dim3(256), dim3(256), 0, s, n.w[0], n.wstride[0], d, dh, n.bias[0], q.k1, q.n1 / 16u, w1, b1);
    hipLaunchKernelGGL(mlp_mfma_pack_kernel, dim3(256), dim3(256), 0, s, n.w[1], n.wstride[1], dh, dh, n.bias[1], q.k2, q.n1 / 16u, w2, b2);
    hipLaunchKernelGGL(mlp_mfma_pack_kernel, dim3(64), dim3(256), 0, s, n.w[2], n.wstride[2], dh, dl, n.bias[2], q.k3, q.n3 / 16u, w3, b3);
    return hipGetLastError();
}
// One workgroup per CU where the batch allows: ceil(nq / CUs) queries each, in as many row blocks of 16 as that takes -- at most what the
// LDS holds (then more workgroups than CUs).  GBNNS_MFMA_ROWS=<n> forces the queries per workgroup (A/B runs).
hipError_t launch_mlp_mfma_net(const NetLaunch& n, const float* packed, hipStream_t s) {
    if (n.nq == 0) return hipSuccess;
    const int cap = max_rb(n.din[0], n.dout[0], n.dout[2]);
    if (cap <= 0) return hipErrorInvalidValue;
    const uint32_t cus = (uint32_t)(n.cus > 0 ? n.cus : 256);
    static const uint32_t forced = getenv("GBNNS_MFMA_ROWS") ? (uint32_t)atoi(getenv("GBNNS_MFMA_ROWS")) : 0u;
    uint32_t rows = forced ? forced : (n.nq + cus - 1u) / cus;
    rows = std::max(1u, std::min(rows, 16u * (uint32_t)cap));
    const int rb = (int)((rows + 15u) / 16u);
    return rb == 1 ? launch_t<1>(n, packed, rows, s) : (rb == 2 ? launch_t<2>(n, packed, rows, s) : launch_t<3>(n, packed, rows, s));
}

}  // namespace gbnns
