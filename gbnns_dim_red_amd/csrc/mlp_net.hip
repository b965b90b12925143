// mlp_net.hip -- the projection (GetLowQueryFromNet, support_func.h:645-658) as ONE launch: a workgroup takes a strip of
// queries through the three layers, activations never leave LDS, weights stream from L2 through wavefront-private LDS.
//
// Arithmetic (bit for bit the per-layer kernels' of mlp.hip): per neuron  out = 0 - Angular::Dist(row, in) + bias
// (support_func.h:624-633), Angular::Dist = eight running sums over k mod 8, fold m_j = c_{j+4} + c_j, (m0 + m1) + (m2 + m3)
// (support_func.h:131-163); products and sums rounded separately (v_pk_mul_f32 / v_pk_add_f32, never a fused multiply-add);
// normalizeVector (:636-642) on the last layer's outputs.
//
// Layout of the work.  Block = 16 wavefronts (4 per SIMD), Q = 8 A queries.  A wavefront owns 2 B neurons of a layer pass
// and ALL Q queries: its weights are its own (no workgroup barrier inside a layer), the activations are shared.  The eight
// running sums of an output are split over FOUR lanes: lane j keeps the pair (c_j, c_{j+4}) in one 64-bit register pair, so
// one v_pk_mul_f32 + one v_pk_add_f32 advance an output by one k-step of 8.  Lane = 16 j + 8 go + gq: gq = 0..7 picks the
// lane's A queries (gq, gq + 8, ...), go = 0..1 its B neurons, j = the lane's ROW of 16 -- so that the fold across the four
// lanes of an output is a reduce-scatter on whole rows: v_permlane16_swap + add gives m0 + m1 | m2 + m3 for two outputs at
// once, v_permlane32_swap + add the final (m0 + m1) + (m2 + m3) for four, and row j ends up OWNING output 4 g + j of every
// group of four: bias, ReLU and the store run once per output, not once per lane.  Operand floats per multiply-add:
// (A + B) / (A B) = 0.45 at A = 5, B = 4 (the per-layer kernels read 0.75 and are LDS-bound); 2 A B accumulator registers.
//
// Activations in LDS: [gq][16-float block kb][a][16 floats], the 16 floats permuted so that one ds_read_b128 at float
// offset 4 j returns (x[16kb+j], x[16kb+4+j], x[16kb+8+j], x[16kb+12+j]) = the operands of sums (j, j + 4) for k-steps 2 kb
// and 2 kb + 1.  A lane's A queries are 64 bytes apart: every read of the k loop is one base register + an immediate.  The
// gq groups are 32 bytes (mod 256) apart: the 16 lanes one LDS cycle serves hit 16 different 16-byte bank groups.
#include <algorithm>

#include "launch_util.h"

namespace gbnns {

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));


__host__ __device__ constexpr uint32_t net_pad16(uint32_t k) { return (k + 15u) & ~15u; }
// floats between the gq groups of an activation image with k inputs and A queries per lane
__host__ __device__ constexpr uint32_t net_gstride(uint32_t k, uint32_t A) { return net_pad16(k) * A + 8u; }
// position of input k inside its 16-float block
__device__ __forceinline__ uint32_t net_swz16(uint32_t k) { return ((k & 3u) << 2) | (((k >> 3) & 1u) << 1) | ((k >> 2) & 1u); }
// float offset of input k of query row q (= gq + 8 a) in an image of group stride sg
template <int A>
__device__ __forceinline__ uint32_t net_xpos(uint32_t q, uint32_t k, uint32_t sg) {
    return (q & 7u) * sg + ((k >> 4) * A + (q >> 3)) * 16u + net_swz16(k);
}

template <int B>
struct NetGeom {
    static constexpr int CK = B >= 4 ? 32 : 64;       // k-values per staged chunk (2 B rows x CK floats: 1 KB, 2 KB at B = 8)
    static constexpr int LDW = CK + 4;        // staged row stride
    static constexpr int P = CK / 4;          // 16-byte pieces per row
    static constexpr int BUF = 2 * B * LDW;   // floats per staging buffer
    static constexpr int NU = CK / 16;        // double k-steps per chunk
    static constexpr int NF = 2 * B * P / 64; // 16-byte pieces per lane and chunk
};
template <int B>
struct NetChunk {
    float4 v[NetGeom<B>::NF];
};
template <>
struct NetChunk<0> {};
__host__ __device__ constexpr uint32_t net_padk(uint32_t k, uint32_t ck) { return (k + ck - 1u) / ck * ck; }

// the 16-byte pieces of a weight chunk this lane fetches: rows obase .. obase + 2B - 1 (clamped to the last row: the results
// of rows beyond dout are dropped), inputs k0 .. k0 + CK - 1, zero from the row's padded end (k16 <= wstride) on
template <int B>
__device__ __forceinline__ NetChunk<B> net_fetch(const float* __restrict__ W, uint32_t wstride, uint32_t k16, uint32_t dout,
                                                 uint32_t obase, uint32_t k0, int lane) {
    using G = NetGeom<B>;
    NetChunk<B> g;
#pragma unroll
    for (int f = 0; f < G::NF; ++f) {
        const uint32_t e = (uint32_t)lane + 64u * f;
        uint32_t o = obase + e / G::P;
        o = o < dout ? o : dout - 1u;
        const uint32_t k = k0 + 4u * (e % G::P);
        g.v[f] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < k16) g.v[f] = *reinterpret_cast<const float4*>(W + (size_t)o * wstride + k);
    }
    return g;
}

template <int B>
__device__ __forceinline__ void net_stage(float* wb, const NetChunk<B>& g, int lane) {
    using G = NetGeom<B>;
#pragma unroll
    for (int f = 0; f < G::NF; ++f) {
        const uint32_t e = (uint32_t)lane + 64u * f;
        const uint32_t r = e / G::P, p = e % G::P;
        float* d = wb + r * G::LDW + (p >> 2) * 16u + ((p >> 1) & 1u) * 2u + (p & 1u);
        d[0] = g.v[f].x; d[4] = g.v[f].y; d[8] = g.v[f].z; d[12] = g.v[f].w;
    }
}

// p + (p of the row ^ 1) in the even rows, q + (q of the row ^ 1) in the odd rows
// (inline asm: with __builtin_amdgcn_permlane16_swap hipcc 7.2 adds the FIRST result to itself here -- `v_add_f32 v17, v7, v7`
// after `v_permlane16_swap_b32 v7, v10`; the s_nop covers the two wait states a VALU write of either operand needs before the swap)
__device__ __forceinline__ float net_fold16(float p, float q) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(p), "+v"(q));
    return p + q;
}
// p + (p of the other half) in lanes 0..31, q + (q of the other half) in lanes 32..63
__device__ __forceinline__ float net_fold32(float p, float q) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
    return p + q;
}

// Two k-steps of one query against four of the lane's neurons: (c_j, c_{j+4}) += w * x, products first (four in flight),
// then the sums; x = this query's 16 bytes of the current 16-input block, w0..w3 the neurons'.
__device__ __forceinline__ void net_row4(f2& c0, f2& c1, f2& c2, f2& c3, const float4& x, const float4& w0, const float4& w1,
                                         const float4& w2, const float4& w3) {
    f2 t0, t1, t2, t3;
    asm volatile(
        "v_pk_mul_f32 %4, %10, %8\n\t"
        "v_pk_mul_f32 %5, %12, %8\n\t"
        "v_pk_mul_f32 %6, %14, %8\n\t"
        "v_pk_mul_f32 %7, %16, %8\n\t"
        "v_pk_add_f32 %0, %0, %4\n\t"
        "v_pk_add_f32 %1, %1, %5\n\t"
        "v_pk_add_f32 %2, %2, %6\n\t"
        "v_pk_add_f32 %3, %3, %7\n\t"
        "v_pk_mul_f32 %4, %11, %9\n\t"
        "v_pk_mul_f32 %5, %13, %9\n\t"
        "v_pk_mul_f32 %6, %15, %9\n\t"
        "v_pk_mul_f32 %7, %17, %9\n\t"
        "v_pk_add_f32 %0, %0, %4\n\t"
        "v_pk_add_f32 %1, %1, %5\n\t"
        "v_pk_add_f32 %2, %2, %6\n\t"
        "v_pk_add_f32 %3, %3, %7"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(f2{x.x, x.y}), "v"(f2{x.z, x.w}), "v"(f2{w0.x, w0.y}), "v"(f2{w0.z, w0.w}), "v"(f2{w1.x, w1.y}), "v"(f2{w1.z, w1.w}),
          "v"(f2{w2.x, w2.y}), "v"(f2{w2.z, w2.w}), "v"(f2{w3.x, w3.y}), "v"(f2{w3.z, w3.w}));
}
__device__ __forceinline__ void net_row2(f2& c0, f2& c1, const float4& x, const float4& w0, const float4& w1) {
    f2 t0, t1, t2, t3;
    asm volatile(
        "v_pk_mul_f32 %2, %8, %6\n\t"
        "v_pk_mul_f32 %3, %10, %6\n\t"
        "v_pk_mul_f32 %4, %9, %7\n\t"
        "v_pk_mul_f32 %5, %11, %7\n\t"
        "v_pk_add_f32 %0, %0, %2\n\t"
        "v_pk_add_f32 %1, %1, %3\n\t"
        "s_nop 0\n\t"
        "v_pk_add_f32 %0, %0, %4\n\t"
        "v_pk_add_f32 %1, %1, %5"
        : "+v"(c0), "+v"(c1), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(f2{x.x, x.y}), "v"(f2{x.z, x.w}), "v"(f2{w0.x, w0.y}), "v"(f2{w0.z, w0.w}), "v"(f2{w1.x, w1.y}), "v"(f2{w1.z, w1.w}));
}

// One layer for the block's Q = 8 A rows.  xs: input image (group stride sx, inputs padded with zeros to a multiple of the
// layer's chunk); outs: the next layer's image (group stride so) when OIMG, else plain rows y[q][so].
// The k loop is a software pipeline over double k-steps u (16 inputs): the NEXT step's weights are requested at the top of a
// step, query a's 16 bytes of the next step right after query a's products of this step (into the same registers); the weight
// chunks (CK inputs of the wavefront's 2 B rows) run two ahead: chunk i + 2 is on its way from L2 while chunk i + 1 sits
// staged in the wavefront's other LDS buffer.  Chunks and steps are numbered THROUGH the layer's passes.
// Which 2 B-neuron slices of a layer a wavefront takes, in order: round robin.  (The vector pipe is arbitrated by age -- with two
// slices each the four older wavefronts of a block are through a 256-neuron layer in 16 us, the younger in 25 -- but handing
// the older ones three slices and the younger one made the older ones the slow ones: 25.2 / 17.4 us, 52.5 us per projection
// against 48.5.  Measured and dropped, round 5.)
template <int NW>
__device__ __forceinline__ uint32_t net_slice(uint32_t wave, uint32_t k) { return k * NW + wave; }  // the k-th slice of `wave`
template <int NW>
__device__ __forceinline__ uint32_t net_slices(uint32_t wave, uint32_t total) {  // how many of `total` slices `wave` takes
    uint32_t n = 0;
    while (net_slice<NW>(wave, n) < total) ++n;
    return n;
}

template <int NW, int B>
struct NetW;
template <int NW>
struct NetW<NW, 0> {};
template <int NW, int B>
struct NetW {            // one layer's weights as a wavefront's stream of chunks
    const float* __restrict__ W;
    uint32_t wstride, k16, dout, nch;
    __device__ __forceinline__ NetW(const float* w, uint32_t ws, uint32_t din, uint32_t dout_)
        : W(w), wstride(ws), k16(net_pad16(din)), dout(dout_), nch(net_padk(din, NetGeom<B>::CK) / NetGeom<B>::CK) {}
    __device__ __forceinline__ NetChunk<B> fetch(uint32_t i, int lane, int wave) const {  // chunk i of the stream
        const uint32_t ps = i / nch, c = i - ps * nch;
        return net_fetch<B>(W, wstride, k16, dout, net_slice<NW>((uint32_t)wave, ps) * 2u * B, c * NetGeom<B>::CK, lane);
    }
};

template <int NW, int A, int B, int NB, bool RELU, bool OIMG>
__device__ __forceinline__ void net_layer(const float* xs, uint32_t sx, const NetW<NW, B>& w, const float* __restrict__ bias, float* wb,
                                          float* outs, uint32_t so, const NetChunk<B>& g0, const NetChunk<B>& g1,
                                          const NetW<NW, NB>& nw, NetChunk<NB>& n0, NetChunk<NB>& n1, int lane, int wave,
                                          unsigned long long* st = nullptr) {
    using G = NetGeom<B>;
    static_assert(B == 2 || B == 4 || B == 8, "neurons per lane group");
    const uint32_t j = (uint32_t)lane >> 4, go = ((uint32_t)lane >> 3) & 1u, gq = (uint32_t)lane & 7u;
    const uint32_t dout = w.dout, nch = w.nch;
    const uint32_t npass = net_slices<NW>((uint32_t)wave, (dout + 2 * B - 1) / (2 * B));  // this wavefront's slices of 2 B neurons
    const uint32_t total = nch * npass;
    if (npass == 0) {  // (a layer of fewer slices than wavefronts) nothing here but the next layer's first chunks
        if constexpr (NB > 0) {
            n0 = nw.fetch(0, lane, wave);
            n1 = nw.fetch(1, lane, wave);
        }
        return;
    }
    const float* xl = xs + gq * sx + 4 * j;
    const float* wl = wb + go * B * G::LDW + 4 * j;
    net_stage<B>(wb, g0, lane);
    NetChunk<B> g = g1;
    constexpr int NBV = B >= 4 ? B / 4 : 1;  // distinct neurons among a lane's outputs: b = (4 gi + j) mod B
    float bsv[NBV];
    auto load_bias = [&](uint32_t ps) {
        const uint32_t ob = net_slice<NW>((uint32_t)wave, ps) * 2u * B + go * B;
#pragma unroll
        for (int tb = 0; tb < NBV; ++tb) {
            const uint32_t o = ob + (B >= 4 ? j + 4u * tb : j % B);
            bsv[tb] = o < dout ? bias[o] : 0.f;
        }
    };
    load_bias(0);
    f2 acc[A][B];
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
        for (int b = 0; b < B; ++b) acc[a][b] = f2{0.f, 0.f};
    // operands of step 0
    float4 xv[A], wv[2][B];
#pragma unroll
    for (int b = 0; b < B; ++b) wv[0][b] = *reinterpret_cast<const float4*>(wl + b * G::LDW);
#pragma unroll
    for (int a = 0; a < A; ++a) xv[a] = *reinterpret_cast<const float4*>(xl + a * 16);
    static_assert(G::NU % 2 == 0, "the weight registers alternate by step: a chunk must end on the set it began with");
    uint32_t i = 0;  // chunk of the layer's stream
    for (uint32_t ps = 0; ps < npass; ++ps) {
#ifdef GBNNS_NET_STAMPS
        unsigned long long ts0 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma clang loop unroll(disable)
        for (uint32_t c = 0; c < nch; ++c, ++i) {
            // (a wavefront's LDS operations execute in order: its own staged rows are visible to its later reads, and the rows
            //  of chunk i - 1 were last read before these stores; past the layer's last chunk the store and the fetch repeat it)
            net_stage<B>(wb + ((i + 1) & 1u) * G::BUF, g, lane);
            g = w.fetch(i + 2 < total ? i + 2 : total - 1, lane, wave);
            const float* wcur = wl + (i & 1u) * G::BUF;
            const float* wnxt = wl + ((i + 1) & 1u) * G::BUF;
            const float* xc = xl + c * (G::NU * A * 16);
            const float* xn = c + 1 == nch ? xl : xc + G::NU * A * 16;  // the next chunk's queries (the next pass starts over)
#pragma unroll
            for (int u = 0; u < G::NU; ++u) {
                const int cb = u & 1, nb = cb ^ 1;
                // the next step's weights: this chunk's next 16 inputs, or the next chunk's first
#pragma unroll
                for (int b = 0; b < B; ++b)
                    wv[nb][b] = u + 1 < G::NU ? *reinterpret_cast<const float4*>(wcur + b * G::LDW + 16 * (u + 1))
                                              : *reinterpret_cast<const float4*>(wnxt + b * G::LDW);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < A; ++a) {
                    if constexpr (B == 2) net_row2(acc[a][0], acc[a][1], xv[a], wv[cb][0], wv[cb][1]);
                    else {
#pragma unroll
                        for (int b = 0; b < B; b += 4)
                            net_row4(acc[a][b], acc[a][b + 1], acc[a][b + 2], acc[a][b + 3], xv[a], wv[cb][b], wv[cb][b + 1],
                                     wv[cb][b + 2], wv[cb][b + 3]);
                    }
                    xv[a] = u + 1 < G::NU ? *reinterpret_cast<const float4*>(xc + ((u + 1) * A + a) * 16)
                                          : *reinterpret_cast<const float4*>(xn + a * 16);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#ifdef GBNNS_NET_STAMPS
        unsigned long long ts1 = __builtin_amdgcn_s_memrealtime();
        if (st && lane == 0) st[0] += ts1 - ts0;
#endif
        // end of a pass.  The next layer's first two chunks set out now, under the fold and the barrier.
        if constexpr (NB > 0) {
            if (ps + 1 == npass) {
                n0 = nw.fetch(0, lane, wave);
                n1 = nw.fetch(1, lane, wave);
            }
        }
        // fold (support_func.h:159-161): m = c_{j+4} + c_j in the lane, then the rows' reduce-scatter; row j owns output 4 gi + j
        const uint32_t obase = net_slice<NW>((uint32_t)wave, ps) * 2u * B;
        constexpr int N = A * B, NG = (N + 3) / 4;
        float res[NG];
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            float m[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int idx = 4 * gi + r < N ? 4 * gi + r : 4 * gi;
                m[r] = acc[idx / B][idx % B].y + acc[idx / B][idx % B].x;
            }
            res[gi] = net_fold32(net_fold16(m[0], m[1]), net_fold16(m[2], m[3]));  // (m0 + m1) + (m2 + m3)
        }
#pragma unroll
        for (int a = 0; a < A; ++a)
#pragma unroll
            for (int b = 0; b < B; ++b) acc[a][b] = f2{0.f, 0.f};
        // bias, ReLU (support_func.h:627-631) and the store, once per output
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const uint32_t idx = 4u * gi + j, a = idx / B, b = idx % B;
            const uint32_t o = obase + go * B + b;
            const bool live = idx < (uint32_t)N && o < dout;
            const float bs = bsv[B >= 4 ? gi % NBV : 0];
            const float dist = -res[gi];                 // Angular::Dist
            float v = 0.f;
            v = v - dist;                                // support_func.h:627
            v = v + bs;                                  // :628
            if (RELU && v < 0.f) v = 0.f;                // :629-631
            if (live) {
                if (OIMG) outs[gq * so + ((o >> 4) * A + a) * 16u + net_swz16(o)] = v;
                else outs[(size_t)(gq + 8u * a) * so + o] = v;
            }
        }
        if (ps + 1 < npass) load_bias(ps + 1);
#ifdef GBNNS_NET_STAMPS
        if (st && lane == 0) st[1] += __builtin_amdgcn_s_memrealtime() - ts1;
#endif
    }
}

struct NetParams {
    const float* x;          // [nq x xstride]
    uint32_t xstride, nq;
    const float* w[3];       // [dout x wstride] each, rows zero padded to a multiple of 16 floats
    uint32_t wstride[3];
    const float* bias[3];
    uint32_t din[3], dout[3];
    float* out;              // [nq x ostride]; columns [dout[2], ostride) are written as zero
    uint32_t ostride;
    uint32_t bufa, bufb;     // floats of the two activation buffers
    unsigned long long* stamps;  // diagnostic builds only (GBNNS_NET_STAMPS): [blocks x 8] s_memrealtime at the phase ends
};

#ifdef GBNNS_NET_STAMPS
#define NET_STAMP(i)                                                                                      \
    do {                                                                                                  \
        if (p.stamps && t == 0) p.stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define NET_STAMP(i) do { } while (0)
#endif

template <int NW, int A, int BH, int B3>
__global__ __launch_bounds__(NW * 64) void mlp_net_kernel(NetParams p) {
    extern __shared__ __attribute__((aligned(16))) float nsm[];
    constexpr int Q = 8 * A, NT = NW * 64;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    float* bufa = nsm;                     // x, then h2
    float* bufb = bufa + p.bufa;           // h1, then y
    float* wb = bufb + p.bufb + (size_t)wave * (2 * NetGeom<BH>::BUF);  // (the hidden layers' staging pair is the largest)
    const uint32_t qbase = blockIdx.x * Q;
    const uint32_t s0 = net_gstride(net_padk(p.din[0], NetGeom<BH>::CK), A), s1 = net_gstride(net_padk(p.dout[0], NetGeom<BH>::CK), A),
                   s2 = net_gstride(net_padk(p.dout[1], NetGeom<B3>::CK), A);
    NET_STAMP(0);

    const NetW<NW, BH> w1(p.w[0], p.wstride[0], p.din[0], p.dout[0]), w2(p.w[1], p.wstride[1], p.din[1], p.dout[1]);
    const NetW<NW, B3> w3(p.w[2], p.wstride[2], p.din[2], p.dout[2]);
    const NetW<NW, 0> w_none;
    NetChunk<BH> ga = w1.fetch(0, lane, wave), gb = w1.fetch(1, lane, wave);  // (a net's first layer has two chunks: din > 32)
    // the block's queries: 16 floats per thread and turn, permuted on the way into LDS (zeros beyond din)
    {
        const uint32_t nb = net_padk(p.din[0], NetGeom<BH>::CK) / 16u;
        for (uint32_t e = t; e < (uint32_t)Q * nb; e += NT) {
            const uint32_t row = e / nb, blk = e % nb, qg = qbase + row;
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t k = 16u * blk + 4u * i;
                v[i] = (qg < p.nq && k < p.din[0]) ? *reinterpret_cast<const float4*>(p.x + (size_t)qg * p.xstride + k)
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float4* d = reinterpret_cast<float4*>(bufa + net_xpos<A>(row, 16u * blk, s0));
            d[0] = make_float4(v[0].x, v[1].x, v[2].x, v[3].x);
            d[1] = make_float4(v[0].y, v[1].y, v[2].y, v[3].y);
            d[2] = make_float4(v[0].z, v[1].z, v[2].z, v[3].z);
            d[3] = make_float4(v[0].w, v[1].w, v[2].w, v[3].w);
        }
    }
    // h1 / h2 columns beyond dout (up to the reading layer's chunk multiple) are read by the next layer: zero them once
    {
        const uint32_t h = p.dout[0], hp = net_padk(h, NetGeom<BH>::CK) - h;
        for (uint32_t e = t; e < (uint32_t)Q * hp; e += NT) bufb[net_xpos<A>(e / hp, h + e % hp, s1)] = 0.f;
    }
    __syncthreads();
    NET_STAMP(1);
    NetChunk<BH> gc, gd;
    net_layer<NW, A, BH, BH, true, true>(bufa, s0, w1, p.bias[0], wb, bufb, s1, ga, gb, w2, gc, gd, lane, wave);
    NET_STAMP(2);
    __syncthreads();
    NET_STAMP(3);
    {
        const uint32_t h = p.dout[1], hp = net_padk(h, NetGeom<B3>::CK) - h;
        for (uint32_t e = t; e < (uint32_t)Q * hp; e += NT) bufa[net_xpos<A>(e / hp, h + e % hp, s2)] = 0.f;
    }
    NetChunk<B3> ge, gf;
#ifdef GBNNS_NET_STAMPS
    net_layer<NW, A, BH, B3, true, true>(bufb, s1, w2, p.bias[1], wb, bufa, s2, gc, gd, w3, ge, gf, lane, wave,
                                         p.stamps ? p.stamps + 8 * 1024 + 2 * (blockIdx.x * NW + wave) : nullptr);
#else
    net_layer<NW, A, BH, B3, true, true>(bufb, s1, w2, p.bias[1], wb, bufa, s2, gc, gd, w3, ge, gf, lane, wave);
#endif
    __syncthreads();
    NET_STAMP(4);
    const uint32_t ldy = p.dout[2] + 1u;
    NetChunk<0> gz;
    net_layer<NW, A, B3, 0, false, false>(bufa, s2, w3, p.bias[2], wb, bufb, ldy, ge, gf, w_none, gz, gz, lane, wave);
    __syncthreads();
    NET_STAMP(5);
    // normalizeVector (support_func.h:636-642): 8 threads per query; threads 0..3 of a query run the four running sums of
    // L2Metric::Dist(y, 0) (support_func.h:107-128, d % 4 tail ignored), then every thread divides its share of the outputs
    float* nsum = bufa;  // [Q][4]
    const int part = t & 7;
    for (int q = t >> 3; q < Q; q += NT / 8) {
        if (part < 4) {
            const float* y = bufb + (size_t)q * ldy;
            const uint32_t steps = p.dout[2] >> 2;
            float sc = 0.f;
            for (uint32_t k = 0; k < steps; ++k) {
                const float e = y[4 * k + part] - 0.f;
                sc = sc + e * e;
            }
            nsum[q * 4 + part] = sc;
        }
    }
    __syncthreads();
    for (int q = t >> 3; q < Q; q += NT / 8) {
        if (qbase + q >= p.nq) break;
        const float* y = bufb + (size_t)q * ldy;
        float norm = ((nsum[q * 4 + 0] + nsum[q * 4 + 1]) + nsum[q * 4 + 2]) + nsum[q * 4 + 3];
        norm = __builtin_sqrtf(norm);  // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
        float* r = p.out + (size_t)(qbase + q) * p.ostride;
        for (uint32_t i = part; i < p.dout[2]; i += 8) r[i] = __fdiv_rn(y[i], norm);
        for (uint32_t i = p.dout[2] + part; i < p.ostride; i += 8) r[i] = 0.f;
    }
    NET_STAMP(6);
}

// ------------------------------------------------------------------------------------------
// One LAYER with the same inner loop, for wide layers of small batches (the GIST shape: 1 000 queries through 960 -> 1 024 -> 1 024):
// the one-launch kernel cannot hold such activations in LDS, and the per-layer kernel of mlp.hip runs them at two wavefronts per
// SIMD with a compiler-scheduled loop.  A workgroup takes 8 A queries x 128 neurons (8 wavefronts x 16) through the whole k range;
// the queries' inputs pass through LDS a SLAB of 256 at a time, two images used alternately: the next slab's floats are requested
// into registers before the current slab's arithmetic and written behind it (one barrier per slab); accumulators stay in
// registers across slabs.  Same arithmetic as net_layer, bit for bit.
constexpr int kSlabK = 256;

struct SlabParams {
    LayerParams l;
    uint32_t sg;  // floats between the gq groups of a slab image
};

template <int NW, int A, bool RELU, bool NORM>
__global__ __launch_bounds__(NW * 64) void mlp_slab_kernel(SlabParams sp) {
    extern __shared__ __attribute__((aligned(16))) float nsm[];
    constexpr int B = 8, NT = NW * 64, Q = 8 * A;
    constexpr int NBLK = kSlabK / 16;                 // 16-float blocks of a query's slab
    constexpr int E = (Q * NBLK + NT - 1) / NT;       // ... of them per thread
    using G = NetGeom<B>;
    const LayerParams& p = sp.l;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t isz = 8u * sp.sg;                  // floats of one image
    float* wb = nsm + 2u * isz + (size_t)wave * (2 * G::BUF);
    const uint32_t qbase = blockIdx.x * Q;
    const uint32_t obase = (blockIdx.y * NW + (uint32_t)wave) * 2u * B;
    const uint32_t j = (uint32_t)lane >> 4, go = ((uint32_t)lane >> 3) & 1u, gq = (uint32_t)lane & 7u;
    const uint32_t k16 = net_pad16(p.din), kpad = net_padk(p.din, G::CK), total = kpad / G::CK;
    auto fetch = [&](uint32_t i) { return net_fetch<B>(p.w, p.wstride, k16, p.dout, obase, i * G::CK, lane); };
    // a thread's share of a slab: blocks e = t, t + NT, ... of the Q x NBLK grid, requested as four float4 each
    float4 sv[E][4];
    auto slab_request = [&](uint32_t k0) {
#pragma unroll
        for (int ei = 0; ei < E; ++ei) {
            const uint32_t e = (uint32_t)t + (uint32_t)ei * NT, row = e / NBLK, blk = e % NBLK, qg = qbase + row;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const uint32_t k = k0 + 16u * blk + 4u * c4;
                sv[ei][c4] = (e < (uint32_t)Q * NBLK && qg < p.nq && k < p.din)
                                 ? *reinterpret_cast<const float4*>(p.x + (size_t)qg * p.xstride + k)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto slab_write = [&](float* img) {
#pragma unroll
        for (int ei = 0; ei < E; ++ei) {
            const uint32_t e = (uint32_t)t + (uint32_t)ei * NT, row = e / NBLK, blk = e % NBLK;
            if (e < (uint32_t)Q * NBLK) {
                float4* d = reinterpret_cast<float4*>(img + net_xpos<A>(row, 16u * blk, sp.sg));
                d[0] = make_float4(sv[ei][0].x, sv[ei][1].x, sv[ei][2].x, sv[ei][3].x);
                d[1] = make_float4(sv[ei][0].y, sv[ei][1].y, sv[ei][2].y, sv[ei][3].y);
                d[2] = make_float4(sv[ei][0].z, sv[ei][1].z, sv[ei][2].z, sv[ei][3].z);
                d[3] = make_float4(sv[ei][0].w, sv[ei][1].w, sv[ei][2].w, sv[ei][3].w);
            }
        }
    };
    slab_request(0);
    NetChunk<B> g = fetch(0);
    net_stage<B>(wb, g, lane);
    g = fetch(total > 1 ? 1 : 0);
    float bsv[2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const uint32_t o = obase + go * B + j + 4u * tb;
        bsv[tb] = o < p.dout ? p.bias[o] : 0.f;
    }
    f2 acc[A][B];
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
        for (int b = 0; b < B; ++b) acc[a][b] = f2{0.f, 0.f};
    const float* wl = wb + go * B * G::LDW + 4 * j;
    float4 xv[A], wv[2][B];
#pragma unroll
    for (int b = 0; b < B; ++b) wv[0][b] = *reinterpret_cast<const float4*>(wl + b * G::LDW);
    slab_write(nsm);
    __syncthreads();
    uint32_t i = 0, par = 0;  // chunk of the layer; image in use
    for (uint32_t k0 = 0; k0 < kpad; k0 += kSlabK, par ^= 1u) {
        const uint32_t ks = kpad - k0 < (uint32_t)kSlabK ? kpad - k0 : (uint32_t)kSlabK;  // inputs of this slab (a multiple of CK)
        const bool next_slab = k0 + kSlabK < kpad;
        if (next_slab) slab_request(k0 + kSlabK);
        const float* xl = nsm + par * isz + gq * sp.sg + 4 * j;
#pragma unroll
        for (int a = 0; a < A; ++a) xv[a] = *reinterpret_cast<const float4*>(xl + a * 16);
        const uint32_t cps = ks / G::CK;
#pragma clang loop unroll(disable)
        for (uint32_t cs = 0; cs < cps; ++cs, ++i) {
            net_stage<B>(wb + ((i + 1) & 1u) * G::BUF, g, lane);
            g = fetch(i + 2 < total ? i + 2 : total - 1);
            const float* wcur = wl + (i & 1u) * G::BUF;
            const float* wnxt = wl + ((i + 1) & 1u) * G::BUF;
            const float* xc = xl + cs * (G::NU * A * 16);
            const bool more = cs + 1 < cps;  // (the slab's last step has no next step in this image)
#pragma unroll
            for (int u = 0; u < G::NU; ++u) {
                const int cb = u & 1, nb = cb ^ 1;
#pragma unroll
                for (int b = 0; b < B; ++b)
                    wv[nb][b] = u + 1 < G::NU ? *reinterpret_cast<const float4*>(wcur + b * G::LDW + 16 * (u + 1))
                                              : *reinterpret_cast<const float4*>(wnxt + b * G::LDW);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < A; ++a) {
#pragma unroll
                    for (int b = 0; b < B; b += 4)
                        net_row4(acc[a][b], acc[a][b + 1], acc[a][b + 2], acc[a][b + 3], xv[a], wv[cb][b], wv[cb][b + 1], wv[cb][b + 2],
                                 wv[cb][b + 3]);
                    if (u + 1 < G::NU) xv[a] = *reinterpret_cast<const float4*>(xc + ((u + 1) * A + a) * 16);
                    else if (more) xv[a] = *reinterpret_cast<const float4*>(xc + (G::NU * A + a) * 16);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (next_slab) {
            slab_write(nsm + (par ^ 1u) * isz);  // (last read one slab ago: everyone has passed the barrier since)
            __syncthreads();
        }
    }
    // fold (support_func.h:159-161) by rows, then bias, ReLU (:627-631) and the store, once per output (as net_layer)
    constexpr int N = A * B, NG = (N + 3) / 4;
    float res[NG];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        float m[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int idx = 4 * gi + r < N ? 4 * gi + r : 4 * gi;
            m[r] = acc[idx / B][idx % B].y + acc[idx / B][idx % B].x;
        }
        res[gi] = net_fold32(net_fold16(m[0], m[1]), net_fold16(m[2], m[3]));
    }
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        const uint32_t idx = 4u * gi + j, a = idx / B, b = idx % B;
        const uint32_t o = obase + go * B + b, qg = qbase + gq + 8u * a;
        const float dist = -res[gi];                 // Angular::Dist
        float v = 0.f;
        v = v - dist;                                // support_func.h:627
        v = v + bsv[gi % 2];                         // :628
        if (RELU && v < 0.f) v = 0.f;                // :629-631
        if constexpr (NORM) res[gi] = v;
        else if (idx < (uint32_t)N && o < p.dout && qg < p.nq) p.out[(size_t)qg * p.ostride + o] = v;
    }
    if constexpr (NORM) {
        // the workgroup owns every output of its queries (one block column): normalizeVector (support_func.h:636-642) here, as in
        // mlp_net_kernel -- 8 threads per query; threads 0..3 run the four running sums of L2Metric::Dist(y, 0) (:107-128, d % 4 tail
        // ignored), then every thread divides its share of the outputs
        const uint32_t ldy = p.dout + 1u;
        float* ybuf = nsm;                       // [Q][ldy] (the slab images are dead once everyone has left the k loop)
        float* nsum = nsm + (size_t)Q * ldy;     // [Q][4]
        __syncthreads();
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const uint32_t idx = 4u * gi + j, a = idx / B, b = idx % B;
            const uint32_t o = obase + go * B + b, ql = gq + 8u * a;
            if (idx < (uint32_t)N && o < p.dout) ybuf[ql * ldy + o] = res[gi];
        }
        __syncthreads();
        const int part = t & 7;
        for (int q = t >> 3; q < Q; q += NT / 8) {
            if (part < 4) {
                const float* y = ybuf + (size_t)q * ldy;
                const uint32_t steps = p.dout >> 2;
                float sc = 0.f;
                for (uint32_t k = 0; k < steps; ++k) {
                    const float e = y[4 * k + part] - 0.f;
                    sc = sc + e * e;
                }
                nsum[q * 4 + part] = sc;
            }
        }
        __syncthreads();
        for (int q = t >> 3; q < Q; q += NT / 8) {
            if (qbase + q >= p.nq) break;
            const float* y = ybuf + (size_t)q * ldy;
            float norm = ((nsum[q * 4 + 0] + nsum[q * 4 + 1]) + nsum[q * 4 + 2]) + nsum[q * 4 + 3];
            norm = __builtin_sqrtf(norm);  // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
            float* r = p.out + (size_t)(qbase + q) * p.ostride;
            for (uint32_t i = part; i < p.dout; i += 8) r[i] = __fdiv_rn(y[i], norm);
            for (uint32_t i = p.dout + part; i < p.ostride; i += 8) r[i] = 0.f;
        }
    }
}

}  // namespace

// LDS bytes of the one-launch projection for Q = 8 A queries per block, NW wavefronts, BH neurons per lane group in the hidden layers
static size_t net_lds_bytes(const NetLaunch& n, int A, int nw, int bh, uint32_t* bufa, uint32_t* bufb) {
    const uint32_t Q = 8u * A;
    // (images padded to 64 inputs: the largest chunk of any layer form)
    *bufa = 8u * std::max(net_gstride(net_padk(n.din[0], 64), A), net_gstride(net_padk(n.dout[1], 64), A));
    *bufb = (std::max(8u * net_gstride(net_padk(n.dout[0], 64), A), Q * (n.dout[2] + 1u)) + 3u) & ~3u;
    const size_t stage = (size_t)2 * 2 * bh * (32 + 4);  // (>= the last layer's 2 x 2 B3 x 68 for B3 <= bh / 2 ... checked below)
    return ((size_t)*bufa + *bufb + (size_t)nw * stage) * sizeof(float);
}

bool mlp_net_serves(const NetLaunch& n) {
    // Batch sizes it wins at (tools/ubench/mlp_lab, one launch against the three per-layer launches): 2 048 queries 26.6 / 32.3 us,
    // 10 000 49.6 / 76.0 (96 -> 128 -> 32: 20.4 / 37.0), 100 000 479 / 526 (96 -> 128: 187 / 193) -- but a 1 M-query batch of the
    // small 96 -> 128 -> 32 net 2 205 / 1 843: one block per CU at a time leaves a block's staging, barriers and stores uncovered,
    // which a hundred rounds of it pay a hundred times while the per-layer kernels' co-resident blocks cover each other's.
    if (n.nq < 2048u || n.nq > 200000u) return false;
    for (int l = 0; l < 3; ++l) {
        if (n.din[l] % 8u || n.wstride[l] % 4u || n.wstride[l] < net_pad16(n.din[l]) || (reinterpret_cast<uintptr_t>(n.w[l]) & 15u))
            return false;
        if (l && n.din[l] != n.dout[l - 1]) return false;
    }
    if (n.xstride % 4u || (reinterpret_cast<uintptr_t>(n.x) & 15u)) return false;
    if (n.dout[2] > 128u) return false;
    // (strips of at least 32 queries must fit: hidden layers of 512 and more leave room for 24 -- 256 -> 512 -> 512 -> 64 then
    // takes 223 us against 215)
    uint32_t ba, bb;
    return net_lds_bytes(n, 4, 8, 8, &ba, &bb) <= 160u * 1024u;
}

template <int NW, int A, int BH, int B3>
static hipError_t net_launch(const NetParams& p, size_t lds, hipStream_t s) {
    hipError_t e = set_lds(mlp_net_kernel<NW, A, BH, B3>, lds);
    if (e != hipSuccess) return e;
    const unsigned grid = (p.nq + 8u * A - 1u) / (8u * A);
    hipLaunchKernelGGL((mlp_net_kernel<NW, A, BH, B3>), dim3(grid), dim3(NW * 64), lds, s, p);
    return hipGetLastError();
}

template <int NW, int BH>
static hipError_t net_launch_a(const NetParams& p, int A, int b3, size_t lds, hipStream_t s) {
#define GBNNS_NET_CASE(AA)                                                        \
    case AA:                                                                      \
        if (b3 == 2) return net_launch<NW, AA, BH, 2>(p, lds, s);                 \
        return net_launch<NW, AA, BH, 4>(p, lds, s);
    switch (A) {
        GBNNS_NET_CASE(2)
        GBNNS_NET_CASE(3)
        GBNNS_NET_CASE(4)
        GBNNS_NET_CASE(5)
    }
#undef GBNNS_NET_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_mlp_net(const NetLaunch& n, hipStream_t s) {
    if (n.nq == 0) return hipSuccess;
    if (!mlp_net_serves(n)) return hipErrorInvalidValue;
    constexpr int nw = 8, bh = 8;  // (blocks of 16 wavefronts x 8 neurons and of 4 x 16 were measured too: DESIGN.md 5.3)
    // queries per lane group: the A that needs the fewest rounds of the machine x A (a block per CU: one round of 8 A queries)
    int cus = n.cus > 0 ? n.cus : 256;
    int bestA = 0;
    uint64_t best = ~0ull;
    NetParams p{};
    for (int A = 5; A >= 2; --A) {
        uint32_t ba, bb;
        if (net_lds_bytes(n, A, nw, bh, &ba, &bb) > 160u * 1024u) continue;
        if (n.force_a && n.force_a != A) continue;
        const uint64_t blocks = (n.nq + 8u * A - 1u) / (8u * A);
        const uint64_t cost = ((blocks + cus - 1) / cus) * A;
        if (cost < best) { best = cost; bestA = A; }
    }
    if (!bestA) return hipErrorInvalidValue;
    const size_t lds = net_lds_bytes(n, bestA, nw, bh, &p.bufa, &p.bufb);
    p.x = n.x; p.xstride = n.xstride; p.nq = n.nq; p.out = n.out; p.ostride = n.ostride; p.stamps = n.stamps;
    for (int l = 0; l < 3; ++l) {
        p.w[l] = n.w[l]; p.wstride[l] = n.wstride[l]; p.bias[l] = n.bias[l]; p.din[l] = n.din[l]; p.dout[l] = n.dout[l];
    }
    // neurons per lane group in the last layer: a pass covers NW x 2 B3 neurons
    const uint32_t per = (n.dout[2] + 2u * nw - 1u) / (2u * nw);
    const int b3 = per <= 2 ? 2 : 4;
    return net_launch_a<8, 8>(p, bestA, b3, lds, s);
}

bool mlp_slab_serves(const LayerParams& p) {
    return p.nq > 0 && p.dout > 0 && p.din % 8u == 0 && p.xstride % 4u == 0 && p.wstride % 4u == 0 && p.wstride >= net_pad16(p.din) &&
           (reinterpret_cast<uintptr_t>(p.x) & 15u) == 0 && (reinterpret_cast<uintptr_t>(p.w) & 15u) == 0;
}

template <int NW, int A>
static hipError_t slab_launch(const SlabParams& sp, unsigned gx, unsigned gy, size_t lds, bool norm, hipStream_t s) {
    hipError_t e;
    if constexpr (NW <= 4) {
        if (norm) {  // (the last layer: no ReLU)
            if ((e = set_lds(mlp_slab_kernel<NW, A, false, true>, lds)) != hipSuccess) return e;
            hipLaunchKernelGGL((mlp_slab_kernel<NW, A, false, true>), dim3(gx, gy), dim3(NW * 64), lds, s, sp);
            return hipGetLastError();
        }
    }
    if (sp.l.relu) {
        if ((e = set_lds(mlp_slab_kernel<NW, A, true, false>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((mlp_slab_kernel<NW, A, true, false>), dim3(gx, gy), dim3(NW * 64), lds, s, sp);
    } else {
        if ((e = set_lds(mlp_slab_kernel<NW, A, false, false>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((mlp_slab_kernel<NW, A, false, false>), dim3(gx, gy), dim3(NW * 64), lds, s, sp);
    }
    return hipGetLastError();
}

// Workgroup shape of the slab kernel for a layer: wide layers 8 wavefronts (128 neurons) x 16 .. 40 queries, layers of up to 64
// neurons one wavefront per 16 neurons x 8 or 16 queries (a 1 000-query batch through 1 024 -> 64 is then 125 workgroups instead of
// the 32 of the per-layer kernel's 32-query tiles: 42 us of a 154-us projection were that layer's).
static void slab_shape(const LayerParams& p, int cus, int force_a, int& nw, int& bestA, uint64_t& blocks) {
    const bool narrow = p.dout <= 64u;
    nw = narrow ? (p.dout <= 32u ? 2 : 4) : 8;
    const unsigned gy = (p.dout + 16u * nw - 1u) / (16u * nw);
    bestA = narrow ? 1 : 2;
    uint64_t best = ~0ull;
    for (int A = narrow ? 2 : 5; A >= (narrow ? 1 : 2); --A) {  // the strip length that needs the fewest rounds of the machine x A
        if (force_a && A != force_a) continue;
        const uint64_t b = (uint64_t)((p.nq + 8u * A - 1u) / (8u * A)) * gy;
        const uint64_t cost = ((b + cus - 1) / cus) * A;
        if (cost < best) { best = cost; bestA = A; blocks = b; }
    }
}

// (measured, tools/ubench/mlp_lab MLP_LAB_SLAB=1: it wins where the layer is one round of the machine -- 1 000 x 960 -> 1 024 50.7
// against 59.2 us, 1 000 x 200 -> 256 9.0 / 11.0 -- and loses beyond: 4 000 x 1 024 -> 1 024 201 / 183, 10 000 x 512 -> 512 123 / 119)
bool mlp_slab_wins(const LayerParams& p, int cus) {
    if (!mlp_slab_serves(p)) return false;
    if (cus <= 0) cus = 256;
    int nw, a;
    uint64_t blocks = 0;
    slab_shape(p, cus, 0, nw, a, blocks);
    return blocks <= (uint64_t)cus && p.din >= 64u;
}

hipError_t launch_mlp_slab(const LayerParams& p, int cus, hipStream_t s, int force_a) {
    if (!mlp_slab_serves(p)) return hipErrorInvalidValue;
    if (cus <= 0) cus = 256;
    int nw, bestA;
    uint64_t blocks = 0;
    slab_shape(p, cus, force_a, nw, bestA, blocks);
    SlabParams sp{};
    sp.l = p;
    sp.sg = net_gstride(kSlabK, bestA);
    const size_t lds = ((size_t)2 * 8 * sp.sg + (size_t)nw * 2 * NetGeom<8>::BUF) * sizeof(float);
    const unsigned gx = (p.nq + 8u * bestA - 1u) / (8u * bestA), gy = (p.dout + 16u * nw - 1u) / (16u * nw);
    // normalizeVector inside the launch when a workgroup holds all the outputs of its queries (and the layer has no ReLU)
    const bool norm = p.normalize && !p.relu && gy == 1 && nw <= 4;
    hipError_t e = hipErrorInvalidValue;
    switch (nw * 8 + bestA) {
        case 8 * 8 + 2: e = slab_launch<8, 2>(sp, gx, gy, lds, norm, s); break;
        case 8 * 8 + 3: e = slab_launch<8, 3>(sp, gx, gy, lds, norm, s); break;
        case 8 * 8 + 4: e = slab_launch<8, 4>(sp, gx, gy, lds, norm, s); break;
        case 8 * 8 + 5: e = slab_launch<8, 5>(sp, gx, gy, lds, norm, s); break;
        case 4 * 8 + 1: e = slab_launch<4, 1>(sp, gx, gy, lds, norm, s); break;
        case 4 * 8 + 2: e = slab_launch<4, 2>(sp, gx, gy, lds, norm, s); break;
        case 2 * 8 + 1: e = slab_launch<2, 1>(sp, gx, gy, lds, norm, s); break;
        case 2 * 8 + 2: e = slab_launch<2, 2>(sp, gx, gy, lds, norm, s); break;
    }
    if (e != hipSuccess || !p.normalize || norm) return e;
    return launch_normalize(p.out, p.ostride, p.dout, p.nq, s);
}

}  // namespace gbnns
