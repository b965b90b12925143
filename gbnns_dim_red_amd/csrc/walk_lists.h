// walk_lists.h -- the result list in registers (RegList: one entry per lane and register, batch merge, sequential offer)
// and the two-list structure for 128 < ef <= 1 024 (BigList: sorted base list in LDS + front list in one register);
// row-in-registers distances of the register kernels.  Shared by the generic walks (walk_generic.h) and the
// hand-laid-out instances (walk_hot.hip).
#pragma once

#include "walk_common.h"

namespace gbnns {

namespace {

// ---- register kernel (ef <= 64): the result list lives in registers, one entry per lane -------
//
// Lane i holds the i-th smallest (dist, id) entry as two dwords: hi = fkey(dist), lo = id<<1 |
// expanded.  Empty lanes hold all-ones (which reads as "expanded", so they are never selected).
// Insertion is a u64 compare + popcount for the position and one wave-wide DPP shift
// (v_mov_b32 wave_shr:1) for the move -- no LDS traffic, a dozen instructions per insert instead
// of five dependent LDS round trips.  LDS keeps only the visited hash set, the query and the tie
// list: [tie: kTieCap x u64][q: dstride x f32][hash: cap x u32].

// vdst[lane `l`] = val (wave-uniform val and l).  The lane select goes through M0: two different
// SGPR operands would exceed the constant-bus limit of gfx9-class VALU instructions.  Nothing else in
// these kernels uses M0.
__device__ __forceinline__ uint32_t writelane_u32(uint32_t vdst, uint32_t val, int l) {
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(vdst) : "s"(val), "s"(l) : "m0");
    return vdst;
}
__device__ __forceinline__ uint64_t clear_bit64(uint64_t m, int bit) {  // wave-uniform operands
    asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(bit));
    return m;
}

// Layout of the one-register hot instances (ef <= 64), A/B switches: GBNNS_HOT1_QLDS = the query is re-read from LDS every
// hop (64 vector registers: 8 wavefronts per SIMD) instead of living in 16 registers (72: 7 per SIMD); GBNNS_HOT1_SPEC = the
// rows are requested before the visited test (speculatively, for every valid slot) instead of after it (new ids only).
#ifndef GBNNS_HOT1_QLDS
#define GBNNS_HOT1_QLDS 1
#endif
#ifndef GBNNS_HOT1_SPEC
#define GBNNS_HOT1_SPEC 0
#endif
// GBNNS_HOT1_PF2_IN_MERGE = 1: the second prefetch's closest survivor comes out of the merge's rank loop (one scalar minimum per
// survivor) instead of a DPP butterfly in front of the merge -- 20 instructions per hop less, and the prefetch ~60
// instructions later: measured 1 - 2 % SLOWER on the SIFT / GloVe shapes at ef = 36 / 64 (profiles/r04_ab.txt), so off.
#ifndef GBNNS_WIDE_VGPRS
#define GBNNS_WIDE_VGPRS 80  // walk_reg_wide_kernel's vector-register budget
#endif
#ifndef GBNNS_HOT1_PF2_IN_MERGE
#define GBNNS_HOT1_PF2_IN_MERGE 0
#endif
constexpr int kRegTieCap = 16;       // tie list of the register kernel (LDS, 128 B)
constexpr int kRegListMaxEf = 1024;  // largest ef served by the register-list / two-list kernels (beyond: result list as one sorted LDS array)

__device__ __forceinline__ uint32_t dpp_wave_shr1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}

#ifdef GBNNS_STAMPS
// Diagnostic build only (make STAMPS=1): per-segment cycle sums of the hop loop, accumulated in
// scalar registers and added to p.stamps[] once per wave.  Never enabled in the shipped library.
#define STAMP(var)                                                                        \
    unsigned long long var;                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);
#define STAMP_ADD(i, a, b) seg[i] += (b) - (a);
#else
#define STAMP(var)
#define STAMP_ADD(i, a, b)
#endif

// Row distance split in two halves so the 16-B row loads can be issued early (before the visited
// test) and consumed late: load_row<STEPS>() then l2_from_regs<STEPS>().
template <int STEPS>
struct RowRegs {
    float4 v[STEPS > 0 ? STEPS : 1];
};

template <int STEPS>
__device__ __forceinline__ void load_row(RowRegs<STEPS>& r, const float* row) {
    const float4* r4 = reinterpret_cast<const float4*>(row);
#pragma unroll
    for (int t = 0; t < STEPS; ++t) r.v[t] = r4[t];
}

template <int STEPS, typename QP>
__device__ __forceinline__ float l2_from_regs(const RowRegs<STEPS>& r, QP qs) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int t = 0; t < STEPS; ++t) {
        const float4 bv = qs[t];
        float e;
        e = r.v[t].x - bv.x; s0 = s0 + e * e;  e = r.v[t].y - bv.y; s1 = s1 + e * e;
        e = r.v[t].z - bv.z; s2 = s2 + e * e;  e = r.v[t].w - bv.w; s3 = s3 + e * e;
    }
    return ((s0 + s1) + s2) + s3;
}

// 128-byte rows (8 steps): the same arithmetic, hand-scheduled.  The compiler emits the two packed
// chains (x,y) and (z,w) one after the other, every dependent pair separated by an s_nop (a packed
// f32 result needs one wait state before it is read); interleaving the chains fills those slots
// with useful instructions: ~49 instead of ~77 issue slots per distance.  Operation order and
// rounding are those of l2_from_regs (v_pk_add/v_pk_mul are exact IEEE f32 per half, no fma).
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define GBNNS_L2_STEP(T)                                                        \
    "v_pk_add_f32 %[ta], %[a" #T "], %[qa" #T "] neg_lo:[0,1] neg_hi:[0,1]\n\t" \
    "v_pk_add_f32 %[tb], %[b" #T "], %[qb" #T "] neg_lo:[0,1] neg_hi:[0,1]\n\t" \
    "v_pk_mul_f32 %[ta], %[ta], %[ta]\n\t"                                     \
    "v_pk_mul_f32 %[tb], %[tb], %[tb]\n\t"                                     \
    "v_pk_add_f32 %[sa], %[sa], %[ta]\n\t"                                     \
    "v_pk_add_f32 %[sb], %[sb], %[tb]\n\t"

template <typename QP>
__device__ __forceinline__ float l2_from_regs8(const RowRegs<8>& r, QP qs) {
    f32x2 sa, sb, ta, tb;  // sa = (s0, s1), sb = (s2, s3)
#define GBNNS_PAIRS(T)                                                                          \
    [a##T] "v"(f32x2{r.v[T].x, r.v[T].y}), [b##T] "v"(f32x2{r.v[T].z, r.v[T].w}),                \
    [qa##T] "v"(f32x2{qs[T].x, qs[T].y}), [qb##T] "v"(f32x2{qs[T].z, qs[T].w})
    asm("v_pk_add_f32 %[ta], %[a0], %[qa0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[tb], %[b0], %[qb0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[sa], %[ta], %[ta]\n\t"   // 0 + e*e == e*e
        "v_pk_mul_f32 %[sb], %[tb], %[tb]\n\t"
        GBNNS_L2_STEP(1) GBNNS_L2_STEP(2) GBNNS_L2_STEP(3)
        : [sa] "=&v"(sa), [sb] "=&v"(sb), [ta] "=&v"(ta), [tb] "=&v"(tb)
        : GBNNS_PAIRS(0), GBNNS_PAIRS(1), GBNNS_PAIRS(2), GBNNS_PAIRS(3));
    asm(GBNNS_L2_STEP(4) GBNNS_L2_STEP(5) GBNNS_L2_STEP(6) GBNNS_L2_STEP(7)
        : [sa] "+v"(sa), [sb] "+v"(sb), [ta] "=&v"(ta), [tb] "=&v"(tb)
        : GBNNS_PAIRS(4), GBNNS_PAIRS(5), GBNNS_PAIRS(6), GBNNS_PAIRS(7));
#undef GBNNS_PAIRS
    return ((sa.x + sa.y) + sb.x) + sb.y;
}
#undef GBNNS_L2_STEP

// Pair form of the same distance: the two lanes 2i / 2i+1 hold the lower / upper 64 bytes of row i
// (four 16-B steps each) and the matching half of the query.  Every lane squares its four steps; the
// even lane runs the reference's chain over steps 0..3, hands its four running sums to the odd lane
// (DPP quad_perm 0,0,2,2), which continues the chain over steps 4..7 and folds ((s0+s1)+s2)+s3:
// the odd lane ends up with exactly the value l2_from_regs8 computes (same operations, same order).
// Why: a lane that streams a whole 128-B row alone costs the CU's vector-memory path one cache-line
// access per 16-B load; two lanes per row halve that (tools/ubench/gather_cost.hip: 36 -> 51 G rows/s
// at ~16 rows per instruction), and the walk was bound by exactly that path.
// Uses v[64:71] as scratch (contiguous pairs are needed for the packed sums).
#define GBNNS_P_SUB(T)                                                                  \
    "v_pk_add_f32 %[pa" #T "], %[a" #T "], %[qa" #T "] neg_lo:[0,1] neg_hi:[0,1]\n\t" \
    "v_pk_add_f32 %[pb" #T "], %[b" #T "], %[qb" #T "] neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define GBNNS_P_MUL(T)                                           \
    "v_pk_mul_f32 %[pa" #T "], %[pa" #T "], %[pa" #T "]\n\t" \
    "v_pk_mul_f32 %[pb" #T "], %[pb" #T "], %[pb" #T "]\n\t"
#define GBNNS_P_ACC(T)                                         \
    "v_pk_add_f32 v[64:65], v[64:65], %[pa" #T "]\n\t"      \
    "v_pk_add_f32 v[66:67], v[66:67], %[pb" #T "]\n\t"
template <typename QP>
__device__ __forceinline__ float l2_pair_from_regs(const RowRegs<4>& r, QP qh) {
    f32x2 pa0, pb0, pa1, pb1, pa2, pb2, pa3, pb3;  // squared differences of this lane's four steps
    float d;
#define GBNNS_Q(T)                                                                               \
    [a##T] "v"(f32x2{r.v[T].x, r.v[T].y}), [b##T] "v"(f32x2{r.v[T].z, r.v[T].w}),                \
    [qa##T] "v"(f32x2{qh[T].x, qh[T].y}), [qb##T] "v"(f32x2{qh[T].z, qh[T].w})
    asm(GBNNS_P_SUB(0) GBNNS_P_SUB(1) GBNNS_P_MUL(0) GBNNS_P_MUL(1) GBNNS_P_SUB(2) GBNNS_P_SUB(3) GBNNS_P_MUL(2) GBNNS_P_MUL(3)
        "v_pk_add_f32 v[68:69], %[pa0], %[pa1]\n\t"      // even lane: steps 0..3 (0 + e*e == e*e)
        "v_pk_add_f32 v[70:71], %[pb0], %[pb1]\n\t"
        "v_pk_add_f32 v[68:69], v[68:69], %[pa2]\n\t"
        "v_pk_add_f32 v[70:71], v[70:71], %[pb2]\n\t"
        "v_pk_add_f32 v[68:69], v[68:69], %[pa3]\n\t"
        "v_pk_add_f32 v[70:71], v[70:71], %[pb3]\n\t"
        "s_nop 1\n\t"                                    // VALU write -> DPP read of the same register
        "v_mov_b32_dpp v64, v68 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp v65, v69 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp v66, v70 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp v67, v71 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
        GBNNS_P_ACC(0) GBNNS_P_ACC(1) GBNNS_P_ACC(2) GBNNS_P_ACC(3)   // odd lane: steps 4..7 on top
        "v_add_f32 %[d], v64, v65\n\t"
        "v_add_f32 %[d], %[d], v66\n\t"
        "v_add_f32 %[d], %[d], v67"
        : [d] "=&v"(d), [pa0] "=&v"(pa0), [pb0] "=&v"(pb0), [pa1] "=&v"(pa1), [pb1] "=&v"(pb1), [pa2] "=&v"(pa2),
          [pb2] "=&v"(pb2), [pa3] "=&v"(pa3), [pb3] "=&v"(pb3)
        : GBNNS_Q(0), GBNNS_Q(1), GBNNS_Q(2), GBNNS_Q(3)
        : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71");
#undef GBNNS_Q
    return d;
}
#undef GBNNS_P_SUB
#undef GBNNS_P_MUL
#undef GBNNS_P_ACC

// Pair form of Angular::Dist for 128-byte rows (support_func.h:131-163, dim = 32: four steps of eight): the
// lanes 2i / 2i+1 hold the EVEN / ODD 16-byte pieces of row i, i.e. the even lane owns the running sums 0..3
// and the odd lane the sums 4..7 of every step -- independent chains.  The fold m_j = c_{j+4} + c_j happens in
// the odd lane (its own sums + the even lane's through DPP quad_perm 0,0,2,2), then -((m0 + m1) + (m2 + m3)):
// the reference's operations in the reference's order; the odd lane holds the distance.
__device__ __forceinline__ float dpp_from_even(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xA0, 0xf, 0xf, false));
}
// Pair form of L2Metric::Dist for rows of 2 H steps (256-byte rows: H = 8), plain C++: the even lane of a pair holds steps
// 0 .. H-1, the odd lane steps H .. 2H-1.  Every lane squares its H steps; chain A adds them in order onto the four
// running sums (meaningful in the even lane: the reference's sums after step H-1; 0 + e*e == e*e), a quad-permute DPP
// hands the even lane's sums to the odd lane, chain B continues there over steps H .. 2H-1; ((s0 + s1) + s2) + s3 as
// the reference.  Same operations in the same order as one lane walking the whole row (support_func.h:107-128).
template <int H, typename QP>
__device__ __forceinline__ float l2_pair_from_regs_wide(const RowRegs<H>& r, QP qh) {
    // (two-wide vector types: the compiler issues v_pk_add_f32 / v_pk_mul_f32, half the instructions of the scalar form --
    // on a GIST-shaped batch the hop is one wavefront's instruction chain; same IEEE operations per component, no contraction)
    f32x2 ea[H], eb[H];
#pragma unroll
    for (int t = 0; t < H; ++t) {
        const float4 a = r.v[t], b = qh[t];
        const f32x2 da = f32x2{a.x, a.y} - f32x2{b.x, b.y}, db = f32x2{a.z, a.w} - f32x2{b.z, b.w};
        ea[t] = da * da;
        eb[t] = db * db;
    }
    f32x2 sa = ea[0], sb = eb[0];
#pragma unroll
    for (int t = 1; t < H; ++t) { sa = sa + ea[t]; sb = sb + eb[t]; }
    auto from_even = [](float v) {  // lanes 2i and 2i + 1 <- lane 2i   (quad_perm [0, 0, 2, 2])
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xA0, 0xf, 0xf, false));
    };
    f32x2 ta = f32x2{from_even(sa.x), from_even(sa.y)}, tb = f32x2{from_even(sb.x), from_even(sb.y)};
#pragma unroll
    for (int t = 0; t < H; ++t) { ta = ta + ea[t]; tb = tb + eb[t]; }
    return ((ta.x + ta.y) + tb.x) + tb.y;
}

template <typename QP>
__device__ __forceinline__ float dot_pair_from_regs(const RowRegs<4>& r, QP qh) {
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        c0 = c0 + r.v[t].x * qh[t].x; c1 = c1 + r.v[t].y * qh[t].y;
        c2 = c2 + r.v[t].z * qh[t].z; c3 = c3 + r.v[t].w * qh[t].w;
    }
    const float m0 = c0 + dpp_from_even(c0), m1 = c1 + dpp_from_even(c1);
    const float m2 = c2 + dpp_from_even(c2), m3 = c3 + dpp_from_even(c3);
    return -((m0 + m1) + (m2 + m3));
}
// four 16-B loads at a stride of 32 bytes (the even or the odd pieces of a 128-byte row)
__device__ __forceinline__ void load_row_alt(RowRegs<4>& r, const float* row) {
    const float4* r4 = reinterpret_cast<const float4*>(row);
#pragma unroll
    for (int t = 0; t < 4; ++t) r.v[t] = r4[2 * t];
}

// Row address.  OFF32: every byte offset into the table fits 32 bits, so the load can use the
// "scalar base + 32-bit lane offset" form (one address VGPR instead of two, no 64-bit multiply).
template <bool OFF32>
__device__ __forceinline__ const float* row_ptr(const float* base, uint32_t id, uint32_t stride_floats) {
    if constexpr (OFF32) {
        const uint32_t off = id * (stride_floats * 4u);
        return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + off);
    } else {
        return base + (size_t)id * stride_floats;
    }
}

// L2Metric::Dist (support_func.h:107-128) for LONG rows whose length is a run-time value -- the PLAIN walks over gist / glove / sift /
// deep vectors at the beams the pair-form instances do not take -- four lanes per row: lane j of a quad owns the reference's running
// sum j (the four sums are independent chains: no hand-over between lanes), reads element j of every 16-byte step -- the quad reads 16
// contiguous bytes per load instruction, 16 rows per instruction instead of the 64 cache lines of one lane per row -- and the quad
// folds ((s0 + s1) + s2) + s3 at the end.  `mfresh` = the lanes that need the distance of their row `nb`; 16 of them per round.
// Returns the distance in those lanes.  Same operations in the same order as l2_ordered ((q - r)^2, sum = sum + that, dim % 4 ignored).
template <bool OFF32>
__device__ __forceinline__ float l2_quad_rows(uint64_t mfresh, uint32_t nb, const float* db, uint32_t dstride, uint32_t dim, const float* qf, int lane) {
    float out = 0.f;
    const uint32_t steps = dim >> 2;
    const int g = lane >> 2, j = lane & 3;
    const float* q = qf + j;
    uint64_t m = mfresh;
    while (m) {
        // this round's rows: the lowest (up to) sixteen set bits; quad i takes the i-th of them
        uint64_t round = 0ull, mm = m;
        int src = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int pos = mm ? __ffsll((unsigned long long)mm) - 1 : 0;
            round |= mm ? (1ull << pos) : 0ull;
            mm &= mm - 1ull;
            src = (g == i) ? pos : src;
        }
        const int nrows = __popcll(round);
        uint32_t id = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)nb);
        id = g < nrows ? id : 0u;  // (idle quads read row 0)
        const float* row = row_ptr<OFF32>(db, id, dstride) + j;
        float sum = 0.f;
        // sixteen loads in flight per round trip; what is left -- fewer than sixteen steps -- as ONE masked batch (steps beyond the row read
        // nothing and add +0 to a sum of squares: no change): step by step it costs a round trip per step (d = 300: 75 steps = 4 x 16 + 11)
        uint32_t t = 0;
        for (; t + 16 <= steps; t += 16) {
            float r[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) r[k] = row[4u * (t + k)];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float e = q[4u * (t + k)] - r[k];
                sum = sum + e * e;
            }
        }
        if (t < steps) {
            float r[16], qv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const bool in = t + k < steps;
                r[k] = in ? row[4u * (t + k)] : 0.f;
                qv[k] = in ? q[4u * (t + k)] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float e = qv[k] - r[k];
                sum = sum + e * e;
            }
        }
        const int si = __float_as_int(sum);
        const float s0 = __int_as_float(__builtin_amdgcn_update_dpp(0, si, 0x00, 0xf, 0xf, false));  // quad_perm [0,0,0,0]
        const float s1 = __int_as_float(__builtin_amdgcn_update_dpp(0, si, 0x55, 0xf, 0xf, false));  // [1,1,1,1]
        const float s2 = __int_as_float(__builtin_amdgcn_update_dpp(0, si, 0xAA, 0xf, 0xf, false));  // [2,2,2,2]
        const float s3 = __int_as_float(__builtin_amdgcn_update_dpp(0, si, 0xFF, 0xf, 0xf, false));  // [3,3,3,3]
        const float d = ((s0 + s1) + s2) + s3;
        // an owner lane is the rank-th row of the round: its distance sits in quad `rank`
        const int rank = __popcll(round & ((1ull << lane) - 1ull));
        const float dv = __int_as_float(__builtin_amdgcn_ds_bpermute(rank << 4, __float_as_int(d)));
        out = ((round >> lane) & 1ull) ? dv : out;
        m &= ~round;
    }
    return out;
}

// Register-resident result list: entry of rank i lives in register i / 64 of lane i % 64
// (R registers per lane, ef <= 64 * R).  Empty slots hold all-ones, which reads as "expanded".
template <int R>
struct RegList {
    uint32_t lo[R], hi[R];

    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int r = 0; r < R; ++r) lo[r] = hi[r] = 0xFFFFFFFFu;
    }
    // wave-uniform rank -> value (all R readlanes are issued, a scalar select keeps the right one)
    __device__ __forceinline__ uint32_t lo_at(int rank) const {
        uint32_t v = readlane_u32(lo[0], rank & 63);
#pragma unroll
        for (int r = 1; r < R; ++r) {
            const uint32_t t = readlane_u32(lo[r], rank & 63);
            if ((rank >> 6) == r) v = t;
        }
        return v;
    }
    __device__ __forceinline__ uint32_t hi_at(int rank) const {
        uint32_t v = readlane_u32(hi[0], rank & 63);
#pragma unroll
        for (int r = 1; r < R; ++r) {
            const uint32_t t = readlane_u32(hi[r], rank & 63);
            if ((rank >> 6) == r) v = t;
        }
        return v;
    }
    __device__ __forceinline__ void mark_expanded(int rank, int lane) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if ((rank >> 6) == r && lane == (rank & 63)) lo[r] |= 1u;
    }
    // lanes of register r that hold list entries (rank < ef)
    __device__ __forceinline__ static uint64_t lane_mask(int r, int ef) {
        const int left = ef - r * 64;
        return left >= 64 ? ~0ull : (left <= 0 ? 0ull : ((1ull << left) - 1ull));
    }
};

// One offer to the register-resident result list, reference rule (search_function.h:31-37):
// insert when worst.dist > dist || size < ef, evict the largest pair when full.  Returns false
// when the tie list overflowed (query is handed to the general kernel).
// Lanes of rank >= ef are scratch (they receive what falls off the end); readers mask them out.
// The placement is decided per lane without a scalar round trip: a lane whose key is >= the new
// key takes its left neighbour's entry, unless that neighbour's key is < the new key -- then it
// is the insertion point and takes the new key.  Register r+1's lane 0 has register r's lane 63
// as its left neighbour (carried through scalar registers).
template <int R>
__device__ __forceinline__ bool reg_offer(uint32_t dl, uint32_t nlo, RegList<R>& L, int& size, uint32_t& worst,
                                          int& tsize, uint64_t* tie, int ef, int lane) {
    const bool full = size >= ef;
    if (full && !(dl < worst)) return true;  // re-test against the CURRENT worst
    const uint64_t nk = ((uint64_t)dl << 32) | nlo;
    const uint32_t ev_lo = L.lo_at(ef - 1);  // evicted entry when full (its hi == worst)
    uint32_t c_lt = 1u, c_lo = 0u, c_hi = 0u;  // left neighbour of lane 0 (rank 0: "smaller" -> insertion point)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint64_t key = ((uint64_t)L.hi[r] << 32) | L.lo[r];
        const bool lt = key < nk;
        const uint32_t lt_i = lt ? 1u : 0u;
        const uint32_t lt_left = (uint32_t)__builtin_amdgcn_update_dpp((int)c_lt, (int)lt_i, 0x138, 0xf, 0xf, false);
        const uint32_t slo = (uint32_t)__builtin_amdgcn_update_dpp((int)c_lo, (int)L.lo[r], 0x138, 0xf, 0xf, false);
        const uint32_t shi = (uint32_t)__builtin_amdgcn_update_dpp((int)c_hi, (int)L.hi[r], 0x138, 0xf, 0xf, false);
        if (r + 1 < R) {
            c_lt = readlane_u32(lt_i, 63);
            c_lo = readlane_u32(L.lo[r], 63);
            c_hi = readlane_u32(L.hi[r], 63);
        }
        if (!lt) {
            L.lo[r] = lt_left ? nlo : slo;
            L.hi[r] = lt_left ? dl : shi;
        }
    }
    if (!full) {
        size += 1;
        worst = L.hi_at(size - 1);
        return true;
    }
    const uint32_t nw = L.hi_at(ef - 1);
    if (nw != worst) {
        tsize = 0;                  // worst distance decreased: old ties are dead
    } else if (!(ev_lo & 1u)) {     // evicted unexpanded at a distance that is still the worst
        if (tsize >= kRegTieCap) return false;
        if (lane == 0) tie[tsize] = ((uint64_t)worst << 32) | ev_lo;
        tsize += 1;
        wave_sync();
    }
    worst = nw;
    return true;
}

// Batch merge of one hop's survivors into a one-register list (ef <= 64).  Offering the survivors
// one by one (reg_offer) is a long serial chain; here every survivor's final position and every
// list entry's shift are counted in one pass over the survivors (independent compares), the new
// list is scattered through a small LDS buffer, and the result equals the sequential rule
// (search_function.h:31-37) whenever no dropped element (evicted entry or rejected survivor) ties
// the new worst distance: an element the sequential rule rejects has dist >= the worst distance of
// its moment >= the final worst distance, so it lies outside the top-ef by (dist, id) unless it
// TIES the final worst distance -- and an accepted element is only ever displaced by smaller keys.
// On such a tie (returns false, list untouched) the caller falls back to the sequential offers.
#ifdef GBNNS_MERGE_S98
#define GBNNS_MERGE_SLO "s98"
#define GBNNS_MERGE_SHI "s99"
#define GBNNS_MERGE_SPAIR "s[98:99]"
#else
#define GBNNS_MERGE_SLO "s46"
#define GBNNS_MERGE_SHI "s47"
#define GBNNS_MERGE_SPAIR "s[46:47]"
#endif
constexpr int kRegStageSlots = 66;   // merge scatter buffer: ranks 0..ef (ef <= 64), padded to 16 B

// WANT_MIN: the loop also keeps the smallest survivor distance key (one scalar instruction per survivor) and hands it to
// `after_ranks(dmin)` right behind the loop -- the hot instance requests its second prefetch there (round 4; before, a
// 25-instruction DPP butterfly in front of the merge found that minimum, every hop, in a kernel that is bound by the
// CU's instruction issue).
template <bool WANT_MIN, typename AfterRanks>
__device__ __forceinline__ bool reg_merge_cb(uint64_t m, bool is_surv, uint32_t dk, uint32_t nb, RegList<1>& L, int& size,
                                             uint32_t& worst, int& tsize, uint64_t* stage, int ef, int lane, AfterRanks&& after_ranks) {
    const int ns = __popcll(m);
    const uint64_t key = ((uint64_t)L.hi[0] << 32) | L.lo[0];
    const uint32_t slo = nb << 1;
    const uint64_t skey = ((uint64_t)dk << 32) | slo;
    const bool is_entry = lane < size;
    // A lane plays two roles: it holds list entry `lane` and (maybe) a survivor.  One pass over the
    // survivors (hand-scheduled: the compiler's version of this loop is 19 instructions, 13 of them
    // scalar, and the walk is bound by scalar issue): survivor `sl`'s key is broadcast through
    // s[46:47]; every entry counts the survivors below it (shift), every survivor the survivors below
    // it (rank), and the number of entries below survivor `sl` -- the zero bits of the compare mask,
    // because lanes that hold no entry hold all-ones or evicted keys, both greater than any survivor --
    // is dropped into lane `sl` (below).  Keys are distinct (a survivor was never visited).
    uint32_t shift, rank, below, sl_, t_, smin = 0xFFFFFFFFu;
    uint64_t ma, mb, mm = m;
#define GBNNS_RANK_LOOP(MIN_STEP)                                                                                       \
    asm volatile(                                                                                                      \
        "v_mov_b32 %[shift], 0\n\t"                                                                                    \
        "v_mov_b32 %[rank], 0\n\t"                                                                                     \
        "v_mov_b32 %[below], 0\n"                                                                                      \
        "1:\n\t"                                                                                                       \
        "s_ff1_i32_b64 %[sl], %[mm]\n\t"                                                                               \
        "v_readlane_b32 " GBNNS_MERGE_SHI ", %[dk], %[sl]\n\t"                                                                         \
        "v_readlane_b32 " GBNNS_MERGE_SLO ", %[slo], %[sl]\n\t"                                                                        \
        "s_bitset0_b64 %[mm], %[sl]\n\t"                                                                               \
        "s_mov_b32 m0, %[sl]\n\t"                                                                                      \
        "v_cmp_gt_u64_e64 %[ma], %[key], " GBNNS_MERGE_SPAIR "\n\t"                                                                 \
        "v_cmp_gt_u64_e64 %[mb], %[skey], " GBNNS_MERGE_SPAIR "\n\t" MIN_STEP                                                       \
        "s_bcnt0_i32_b64 %[t], %[ma]\n\t"                                                                              \
        "v_addc_co_u32_e64 %[shift], vcc, 0, %[shift], %[ma]\n\t"                                                      \
        "v_addc_co_u32_e64 %[rank], vcc, 0, %[rank], %[mb]\n\t"                                                        \
        "v_writelane_b32 %[below], %[t], m0\n\t"                                                                       \
        "s_cmp_lg_u64 %[mm], 0\n\t"                                                                                    \
        "s_cbranch_scc1 1b"                                                                                            \
        : [shift] "=&v"(shift), [rank] "=&v"(rank), [below] "=&v"(below), [sl] "=&s"(sl_), [t] "=&s"(t_), [ma] "=&s"(ma), \
          [mb] "=&s"(mb), [mm] "+s"(mm), [smin] "+s"(smin)                                                              \
        : [dk] "v"(dk), [slo] "v"(slo), [key] "v"(key), [skey] "v"(skey)                                                \
        : "vcc", "scc", "m0", GBNNS_MERGE_SLO, GBNNS_MERGE_SHI)
    if constexpr (WANT_MIN) {
        GBNNS_RANK_LOOP("s_min_u32 %[smin], %[smin], " GBNNS_MERGE_SHI "\n\t");
        after_ranks(smin);
    } else {
        GBNNS_RANK_LOOP("");
    }
#undef GBNNS_RANK_LOOP
    const int dst_e = lane + (int)shift, dst_s = (int)(rank + below);
    const int total = size + ns;
    const int new_size = total < ef ? total : ef;
    // rank ef (the first element that falls off) is staged too: it decides the boundary-tie test
    if (is_entry && dst_e <= ef) stage[dst_e] = key;
    if (is_surv && dst_s <= ef) stage[dst_s] = skey;
    wave_sync();
    const uint64_t nkey = lane < new_size ? stage[lane] : ~0ull;
    const uint32_t nw = readlane_u32((uint32_t)(nkey >> 32), new_size - 1);
    if (total > ef) {
        // dropped elements are the merged ranks >= ef, ascending: one of them ties the new worst
        // distance iff the first one does -> order matters, go sequential (list untouched)
        const uint32_t first_dropped = (uint32_t)__builtin_amdgcn_readfirstlane((int)(stage[ef] >> 32));
        if (first_dropped == nw) return false;
        tsize = 0;  // something was evicted and (no tie) the worst distance decreased
    }
    L.lo[0] = (uint32_t)nkey;
    L.hi[0] = (uint32_t)(nkey >> 32);
    size = new_size;
    worst = nw;
    return true;
}

__device__ __forceinline__ bool reg_merge(uint64_t m, bool is_surv, uint32_t dk, uint32_t nb, RegList<1>& L, int& size,
                                          uint32_t& worst, int& tsize, uint64_t* stage, int ef, int lane) {
    return reg_merge_cb<false>(m, is_surv, dk, nb, L, size, worst, tsize, stage, ef, lane, [](uint32_t) {});
}

// The same batch merge for lists of R = 2 / 4 registers per lane (64 < ef <= 256): entry of rank i lives in
// register i / 64 of lane i % 64, the scatter buffer holds ranks 0..ef.  Plain C++ (these instances are not
// the hot one); same rule, same fallback on a boundary tie.
__device__ __forceinline__ constexpr int reg_stage_slots(int R) { return 64 * R + 2; }

template <int R>
__device__ __forceinline__ bool reg_merge_multi(uint64_t m, bool is_surv, uint32_t dk, uint32_t nb, RegList<R>& L, int& size,
                                                uint32_t& worst, int& tsize, uint64_t* stage, int ef, int lane) {
    const int ns = __popcll(m);
    uint64_t key[R];
#pragma unroll
    for (int r = 0; r < R; ++r) key[r] = ((uint64_t)L.hi[r] << 32) | L.lo[r];
    const uint32_t slo = nb << 1;
    const uint64_t skey = ((uint64_t)dk << 32) | slo;
    uint32_t shift[R];
#pragma unroll
    for (int r = 0; r < R; ++r) shift[r] = 0;
    uint32_t rank = 0, below = 0;
    uint64_t mm = m;
    do {
        const int sl = __ffsll((unsigned long long)mm) - 1;
        mm = clear_bit64(mm, sl);
        const uint64_t ks = ((uint64_t)readlane_u32(dk, sl) << 32) | readlane_u32(slo, sl);
        uint32_t cnt = 0;  // entries below this survivor: lanes without an entry hold all-ones / evicted keys (greater)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool gt = key[r] > ks;
            shift[r] += gt ? 1u : 0u;
            cnt += (uint32_t)__popcll(~__ballot(gt));
        }
        rank += skey > ks ? 1u : 0u;
        below = writelane_u32(below, cnt, sl);
    } while (mm);
    const int total = size + ns;
    const int new_size = total < ef ? total : ef;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int re = r * 64 + lane, dst = re + (int)shift[r];
        if (re < size && dst <= ef) stage[dst] = key[r];
    }
    {
        const int dst_s = (int)(rank + below);
        if (is_surv && dst_s <= ef) stage[dst_s] = skey;
    }
    wave_sync();
    // the element of merged rank ef (the first one that falls off) decides the boundary-tie test
    uint32_t first_dropped = 0;
    if (total > ef) first_dropped = (uint32_t)__builtin_amdgcn_readfirstlane((int)(stage[ef] >> 32));
    uint32_t nlo[R], nhi[R], nw = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint64_t v = (r * 64 + lane < new_size) ? stage[r * 64 + lane] : ~0ull;
        nlo[r] = (uint32_t)v;
        nhi[r] = (uint32_t)(v >> 32);
        const uint32_t t = readlane_u32(nhi[r], (new_size - 1) & 63);
        if (((new_size - 1) >> 6) == r) nw = t;
    }
    wave_sync();  // (the buffer is reused by the next merge)
    if (total > ef && first_dropped == nw) return false;  // order matters: the caller goes sequential, list untouched
    if (total > ef) tsize = 0;  // something was evicted and (no tie) the worst distance decreased
#pragma unroll
    for (int r = 0; r < R; ++r) {
        L.lo[r] = nlo[r];
        L.hi[r] = nhi[r];
    }
    size = new_size;
    worst = nw;
    return true;
}

// Results of a register-list walk in POP order (worst -> best): rank i goes to position kept-1-i.
template <int R>
__device__ __forceinline__ void reg_write_results(const WalkParams& p, uint32_t qi, const RegList<R>& L, int size, int hops,
                                                  int dist_calc, int edges, int lane) {
    const int kept = size < p.k ? size : p.k;
    // PLAIN answer = topk.top() after trimming the heap to k (search_function.h:174-181): the k-th best
    const uint32_t kth = L.lo_at(kept > 0 ? kept - 1 : 0) >> 1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int rank = r * 64 + lane;
        if (rank < (int)p.cand_stride) {
            if (rank < kept) {
                p.cand[(size_t)qi * p.cand_stride + (kept - 1 - rank)] = L.lo[r] >> 1;
                if (p.cand_dist) p.cand_dist[(size_t)qi * p.cand_stride + (kept - 1 - rank)] = fkey_inv_out(L.hi[r], p.zero_dist_bits);
            } else {
                p.cand[(size_t)qi * p.cand_stride + rank] = kInvalidId;
                if (p.cand_dist) p.cand_dist[(size_t)qi * p.cand_stride + rank] = __builtin_inff();
            }
        }
    }
    if (lane == 0) {
        p.count[qi] = kept;
        p.hops[qi] = hops;
        p.dist_calc[qi] = dist_calc;
        atomicMax(p.max_dc, (uint32_t)dist_calc);
        if (p.edges) p.edges[qi] = edges;
        if (p.best) p.best[qi] = kth;
    }
}

// the id of list rank `rank` (per-lane rank; all lanes call)
template <int R>
__device__ __forceinline__ uint32_t reg_id_at_rank(const RegList<R>& L, int rank) {
    uint32_t v = (uint32_t)__shfl((int)(L.lo[0] >> 1), rank & 63);
#pragma unroll
    for (int r = 1; r < R; ++r) {
        const uint32_t t = (uint32_t)__shfl((int)(L.lo[r] >> 1), rank & 63);
        if ((rank >> 6) == r) v = t;
    }
    return v;
}

// ---- hot instance for 128 < ef <= 1024: the result list as a sorted BASE list in LDS + a sorted FRONT list in a register
//
// With the whole list in R = ceil(ef / 64) registers per lane (walk_hot_one<R>, reg_merge_multi), finding the next node
// costs ~16 R instructions EVERY hop and a merge rewrites all R registers; the walk is instruction-issue bound, so ef =
// 300 ran at a third of the ef = 64 rate per distance -- although 2 hops in 3 insert at most one entry (measured
// histogram: 44 % of the hops at ef = 180 have no survivor, 20 % one).  Here
//   base  L: sorted keys in LDS, `l` live entries; between two flushes it only loses entries from its end and gets
//            "expanded" bits set.  Its two closest unexpanded entries are cached in scalar registers (found through a
//            per-chunk unexpanded mask kept in two vector registers), so a hop that does not pick from it pays nothing.
//   front F: sorted, ONE register per lane, `f` <= 64 live entries: every insertion goes here, with the one-register
//            machinery of the ef <= 64 instance (reg_merge / reg_offer), whatever ef is.
// The reference's result heap (search_function.h:50) is the union: l + f <= ef entries; its worst element is the
// larger of the two tails, the next node is the closer of the two first unexpanded entries.  When a hop's E = l + f
// - ef entries have to go, they are the E largest of the two tails: one vector step finds how many come from which
// list (lane j tests the split "j from the base list, E - j from the front list").  When the front list would
// overflow (and at the end of the walk) it is merged into the base list in place: every front entry finds its rank by
// bisection, every destination rank then gathers its entry (chunks of 64 ranks, top down).  Exactly the same results:
// the union holds the same keys as the single list did, selection and eviction see the same total order; the
// sequential fallback on a boundary tie and the tie list work as before.  Nothing depends on R any more: one kernel.

constexpr int kBigMaxEf = kRegListMaxEf;  // (the structure itself reaches 64 chunks = 4 096 entries: one mask lane per chunk)
#ifndef GBNNS_HOT2_MAX
#define GBNNS_HOT2_MAX 128  // (64: experiments with the two-list kernels from ef = 65 on)
#endif
constexpr int kHot2MaxEf = GBNNS_HOT2_MAX;  // up to here the two-register lists (walk_hot_one<2>, walk_reg_one<2>) are the faster ones

__device__ __forceinline__ uint64_t dpp_wave_shl1_u64(uint64_t v) {  // lane j <- lane j + 1
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x130, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x130, 0xf, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}

// LDS of the instance besides the visited set: [tie list][front-merge buffer: 66 keys][base list: ef_pad keys]
// [flush flags: ef_pad + 64 bytes], ef_pad = ef rounded up to 64.  The flush flags live inside the front-merge buffer
// when they fit (ef <= 448: the two are never in use at the same time) -- at ef = 140 .. 180 those 256 bytes are what
// separates 14 / 13 / 12 resident wavefronts per CU from 15 / 14 / 13.
__host__ __device__ __forceinline__ constexpr bool big_list_flags_in_stage(int ef) {
    return (size_t)((ef + 63) / 64 * 64) + 64 <= (size_t)kRegStageSlots * 8;
}
__host__ __device__ __forceinline__ constexpr size_t big_list_fixed_bytes(int ef) {
    return (size_t)kRegTieCap * 8 + (size_t)kRegStageSlots * 8 + (size_t)((ef + 63) / 64 * 64) * 8 +
           (big_list_flags_in_stage(ef) ? 0 : (size_t)((ef + 63) / 64 * 64) + 64);
}

struct BigList {
    uint64_t* tie;          // [kRegTieCap]
    uint64_t* stage;        // [kRegStageSlots] scatter buffer of the front-list merge
    uint64_t* base;         // [ef_pad] base list, ascending; ranks >= l are dead
    unsigned char* flags;   // [ef_pad + 64] flush scratch
    RegList<1> F;           // front list; lanes >= f hold all-ones
    int ef, l, f, tsize;
    uint32_t worst;         // hi of the union's worst entry (valid once l + f == ef)
    uint32_t fworst;        // hi of the front list's last entry
    uint32_t mu_lo, mu_hi;  // lane c: mask of the unexpanded live entries of base ranks 64 c .. 64 c + 63
    // the two closest unexpanded base entries (c_valid: the cache reflects the masks)
    bool c_valid;
    int p1, p2;             // ranks, -1 = none
    uint32_t h1, n1, h2, n2;
#ifdef GBNNS_COOP_HINT
    uint32_t hint3;         // select(): the id after the runner-up, probably (unvalidated: may be garbage when the lists run dry -- a prefetch hint only)
#endif
#ifdef GBNNS_STAMPS
    unsigned long long st_flush = 0, st_refresh = 0, st_evict = 0;  // cycles inside flush / refresh_cache / the eviction step
    unsigned st_nflush = 0, st_nrefresh = 0, st_nbase = 0, st_nseq = 0, st_ninsert = 0, st_slow = 0;
#endif

    // NOTE on lane-dependent updates: they are written as selects / unconditional same-value stores, never as
    // `if (lane == x) ...`.  A lane-dependent branch inside these functions lets the optimiser thread scalar list
    // state through its two arms; the divergence analysis then takes l, f, worst ... for divergent, keeps them in
    // vector registers and turns the list's scalar control flow into exec-masked regions (measured: 3 x the vector
    // instructions per hop).
    __device__ __forceinline__ void kill_front_from(int first_dead, int lane) {  // lanes >= first_dead <- all-ones
        const bool dead = lane >= first_dead;
        F.lo[0] = dead ? 0xFFFFFFFFu : F.lo[0];
        F.hi[0] = dead ? 0xFFFFFFFFu : F.hi[0];
    }

    __device__ __forceinline__ uint64_t base_at(int rank) const {  // wave-uniform rank
        const uint64_t v = base[rank];
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    }
    __device__ __forceinline__ uint32_t union_worst() const {
        const uint32_t lw = l > 0 ? (uint32_t)(base_at(l - 1) >> 32) : 0u;
        const uint32_t fw = f > 0 ? readlane_u32(F.hi[0], f - 1) : 0u;
        return lw > fw ? lw : fw;
    }
    // drops the mask bits of base ranks >= l (after the base list lost entries from its end)
    __device__ __forceinline__ void trim_masks(int lane) {
        const int left = l - 64 * lane;
        const uint64_t keep = left >= 64 ? ~0ull : (left <= 0 ? 0ull : ((1ull << left) - 1ull));
        mu_lo &= (uint32_t)keep;
        mu_hi &= (uint32_t)(keep >> 32);
        if (p1 >= l || p2 >= l) c_valid = false;
    }
    // the two lowest set bits of the masks -> (p1, h1, n1), (p2, h2, n2)
    __device__ __forceinline__ void refresh_cache(int lane) {
        STAMP(tr0)
#ifdef GBNNS_STAMPS
        st_nrefresh += 1;
#endif
        p1 = p2 = -1;
        h1 = h2 = 0xFFFFFFFFu;
        n1 = n2 = 0u;
        uint64_t nz = __ballot((mu_lo | mu_hi) != 0u);
        if (nz) {
            const int c1 = __ffsll((unsigned long long)nz) - 1;
            uint64_t m1 = ((uint64_t)readlane_u32(mu_hi, c1) << 32) | readlane_u32(mu_lo, c1);
            p1 = 64 * c1 + __ffsll((unsigned long long)m1) - 1;
            m1 &= m1 - 1;
            if (m1) {
                p2 = 64 * c1 + __ffsll((unsigned long long)m1) - 1;
            } else {
                nz &= nz - 1;
                if (nz) {
                    const int c2 = __ffsll((unsigned long long)nz) - 1;
                    const uint64_t m2 = ((uint64_t)readlane_u32(mu_hi, c2) << 32) | readlane_u32(mu_lo, c2);
                    p2 = 64 * c2 + __ffsll((unsigned long long)m2) - 1;
                }
            }
            const uint64_t kv = base[lane == 0 ? p1 : (p2 >= 0 ? p2 : p1)];  // lane 0: first, lane 1: second
            h1 = readlane_u32((uint32_t)(kv >> 32), 0);
            n1 = readlane_u32((uint32_t)kv, 0) >> 1;
            if (p2 >= 0) {
                h2 = readlane_u32((uint32_t)(kv >> 32), 1);
                n2 = readlane_u32((uint32_t)kv, 1) >> 1;
            }
        }
        c_valid = true;
        STAMP(tr1)
#ifdef GBNNS_STAMPS
        st_refresh += tr1 - tr0;
#endif
    }
    // marks base rank `p` expanded: the key's flag bit in LDS and the mask bit
    __device__ __forceinline__ void expand_base(int p, int lane) {
        p = __builtin_amdgcn_readfirstlane(p);  // wave-uniform by construction; tell the compiler
#ifdef GBNNS_STAMPS
        st_nbase += 1;
#endif
        reinterpret_cast<uint32_t*>(base)[2 * p] |= 1u;  // (every lane: same address, same value)
        const int c = p >> 6;
        uint64_t m = ((uint64_t)readlane_u32(mu_hi, c) << 32) | readlane_u32(mu_lo, c);
        m = clear_bit64(m, p & 63);
        mu_lo = writelane_u32(mu_lo, (uint32_t)m, c);
        mu_hi = writelane_u32(mu_hi, (uint32_t)(m >> 32), c);
        c_valid = false;
    }

    // Merges the front list into the base list in place (both sorted; keys are distinct).  Afterwards l += f, f = 0,
    // the front register holds all-ones, the masks and the cache are rebuilt lazily.
    __device__ __forceinline__ void flush(int lane) {
        if (f == 0) return;
        STAMP(tf0)
        l = __builtin_amdgcn_readfirstlane(l);  // wave-uniform by construction; tell the compiler (loop counters
        f = __builtin_amdgcn_readfirstlane(f);  // below index lanes through scalar registers)
        const int total = l + f;
        const int chunks = (total + 63) >> 6;
        {   // zero the flag bytes of ranks 0 .. 64 chunks + 63: unconditional (clamped) stores, 64 words per round
            const int words = chunks * 8 + 8;
            for (int w0 = 0; w0 < words; w0 += 64)
                reinterpret_cast<uint64_t*>(flags)[w0 + lane < words ? w0 + lane : 0] = 0ull;
        }
        wave_sync();
        // every front entry: number of base entries below it (lower bound by bisection)
        const uint64_t fk = ((uint64_t)F.hi[0] << 32) | F.lo[0];
        int lo = 0, hi = l;
        const int iters = 32 - __clz(l);  // covers 0 .. l
        for (int it = 0; it < iters; ++it) {
            const int mid = (lo + hi) >> 1;
            const uint64_t v = base[mid < l ? mid : 0];
            const bool go = lo < hi;
            const bool less = v < fk;
            lo = (go && less) ? mid + 1 : lo;
            hi = (go && !less) ? mid : hi;
        }
        // final rank = base entries below + front entries below (= its lane); lanes without an entry hit a byte past
        // the last chunk (zeroed again by the next flush)
        flags[lane < f ? lo + lane : 64 * chunks + 1] = 1;
        wave_sync();
        // front entries below each chunk (lane c of `below`)
        uint32_t below = 0;
        {
            int carry = 0;
            for (int c = 0; c < chunks; ++c) {
                below = writelane_u32(below, (uint32_t)__builtin_amdgcn_readfirstlane(carry), c);
                carry += __popcll(__ballot(flags[64 * c + lane] != 0));
            }
        }
        // every destination rank takes its entry, chunks top down (a chunk reads base ranks of itself and of the chunk
        // below only, so writing in place is safe in this order)
        mu_lo = mu_hi = 0u;
        for (int c = chunks - 1; c >= 0; --c) {
            const int rank = 64 * c + lane;
            const bool flagged = flags[rank] != 0;
            const uint64_t mk = __ballot(flagged);
            const int cc = (int)readlane_u32(below, c) +
                           (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
            const uint32_t flo = (uint32_t)__builtin_amdgcn_ds_bpermute((cc & 63) << 2, (int)F.lo[0]);
            const uint32_t fhi = (uint32_t)__builtin_amdgcn_ds_bpermute((cc & 63) << 2, (int)F.hi[0]);
            const int src = rank - cc;
            const uint64_t bv = (rank < total && src < l) ? base[src] : ~0ull;
            const uint64_t nk = flagged ? (((uint64_t)fhi << 32) | flo) : bv;
            base[rank] = nk;
            const uint64_t un = __ballot(rank < total && !((uint32_t)nk & 1u));
            mu_lo = writelane_u32(mu_lo, (uint32_t)un, c);
            mu_hi = writelane_u32(mu_hi, (uint32_t)(un >> 32), c);
        }
        wave_sync();
        l = total;
        f = 0;
        F.clear();
        c_valid = false;
        STAMP(tf1)
#ifdef GBNNS_STAMPS
        st_flush += tf1 - tf0;
        st_nflush += 1;
#endif
    }

    // One offer with the reference's rule (search_function.h:31-37): the sequential path (single survivors into a full
    // union, boundary ties).  False: the tie list overflowed.
    __device__ __forceinline__ bool offer_one(uint32_t dl, uint32_t nlo, int lane) {
        const bool full = l + f >= ef;
        if (full && !(dl < worst)) return true;
        if (f == 64) flush(lane);
        int ts_unused = 0;
        reg_offer<1>(dl, nlo, F, f, fworst, ts_unused, tie, 64, lane);  // f < 64: a plain sorted insert
        if (!full) {
            if (l + f == ef) worst = union_worst();
            return true;
        }
        // evict the union's largest entry: the larger of the two tails
        const uint64_t lt = l > 0 ? base_at(l - 1) : 0ull;
        const uint64_t ft = ((uint64_t)readlane_u32(F.hi[0], f - 1) << 32) | readlane_u32(F.lo[0], f - 1);
        uint64_t ev;
        if (lt > ft) {
            ev = lt;
            l -= 1;
            trim_masks(lane);
        } else {
            ev = ft;
            f -= 1;
            kill_front_from(f, lane);
        }
        const uint32_t nw = union_worst();
        if (nw != worst) {
            tsize = 0;  // the worst distance decreased: old ties are dead
        } else if (!(ev & 1ull)) {  // evicted unexpanded at a distance that is still the worst
            if (tsize >= kRegTieCap) return false;
            tie[tsize] = ev;  // (every lane stores the same value: no lane-dependent branch, see the note above)
            tsize += 1;
            wave_sync();
        }
        worst = nw;
        return true;
    }

    // A hop's survivors (mask m, keys dk / ids nb in their lanes; at most 32) into the union.  False: hand over.
    __device__ __forceinline__ bool insert(uint64_t m, uint32_t dk, uint32_t nb, int lane) {
        const int ns = __popcll(m);
        bool sequential = false;
        if (l + f + ns <= ef || (m & (m - 1)) != 0) {
            // ---- batch: survivors into the front list, then the E largest of the union go
            if (f + ns > 64) flush(lane);
            const uint32_t keep_lo = F.lo[0], keep_hi = F.hi[0], keep_fw = fworst;
            const int keep_f = f;
            int ts_unused = 0;
            if ((m & (m - 1)) != 0) {
                reg_merge(m, __builtin_amdgcn_inverse_ballot_w64(m), dk, nb, F, f, fworst, ts_unused, stage, 64, lane);
            } else {
                const int sl = __ffsll((unsigned long long)m) - 1;
                reg_offer<1>(readlane_u32(dk, sl), readlane_u32(nb, sl) << 1, F, f, fworst, ts_unused, tie, 64, lane);
            }
            const int E = l + f - ef;
            STAMP(te0)
            if (E > 0) {
                // lane j: "the base list drops its top j entries, the front list its top E - j" (0 <= j <= E <= 32)
                const int j = lane;
                const int bi = l - j;  // smallest base entry dropped (j = 0: none -> all-ones; below rank 0: zero)
                uint64_t H = base[(bi >= 0 && bi < l) ? bi : 0];
                H = bi < 0 ? 0ull : H;
                H = bi >= l ? ~0ull : H;
                const int fi = f - E - 1 + j;  // largest front entry kept (< 0: none)
                const uint32_t glo = (uint32_t)__builtin_amdgcn_ds_bpermute((fi & 63) << 2, (int)F.lo[0]);
                const uint32_t ghi = (uint32_t)__builtin_amdgcn_ds_bpermute((fi & 63) << 2, (int)F.hi[0]);
                uint64_t G = ((uint64_t)ghi << 32) | glo;
                G = fi < 0 ? 0ull : G;
                G = fi >= f ? ~0ull : G;
                const uint64_t La = H, Lb = dpp_wave_shl1_u64(H);  // Lb = base entry of rank l - j - 1 (largest kept)
                const uint64_t Fb = G, Fa = dpp_wave_shl1_u64(G);  // Fa = front entry of rank f - E + j (smallest dropped)
                const bool cand = j <= E && j <= l && E - j <= f;
                const uint64_t good = __ballot(cand && La > Fb && Fa > Lb);
                const int x = __ffsll((unsigned long long)good) - 1;  // exactly one lane (keys are distinct)
                const uint32_t la_hi = readlane_u32((uint32_t)(La >> 32), x), fa_hi = readlane_u32((uint32_t)(Fa >> 32), x);
                const uint32_t lb_hi = readlane_u32((uint32_t)(Lb >> 32), x), fb_hi = readlane_u32((uint32_t)(Fb >> 32), x);
                const uint32_t first_dropped = la_hi < fa_hi ? la_hi : fa_hi;  // distance of the smallest dropped entry
                const uint32_t nw = lb_hi > fb_hi ? lb_hi : fb_hi;            // distance of the largest kept entry
                if (first_dropped == nw) {
                    // a dropped entry ties the new worst distance: order matters -> undo, go sequential
                    F.lo[0] = keep_lo; F.hi[0] = keep_hi; f = keep_f; fworst = keep_fw;
                    sequential = true;
                } else {
                    if (x > 0) {
                        l -= x;
                        trim_masks(lane);
                    }
                    f -= E - x;
                    kill_front_from(f, lane);
                    tsize = 0;  // something was evicted and (no tie) the worst distance decreased
                    worst = nw;
                }
            } else if (E == 0) {
                worst = union_worst();  // the union just became full
            }
            STAMP(te1)
#ifdef GBNNS_STAMPS
            st_evict += te1 - te0;
            st_ninsert += 1;
#endif
        } else {
            sequential = true;  // a single survivor into a full union: one offer
        }
        if (sequential) {
#ifdef GBNNS_STAMPS
            st_nseq += 1;
#endif
            do {
                const int sl = __ffsll((unsigned long long)m) - 1;
                m &= m - 1;
                if (!offer_one(readlane_u32(dk, sl), readlane_u32(nb, sl) << 1, lane)) return false;
            } while (m);
        }
        return true;
    }

    // The next node to expand: the closest unexpanded entry of the union, ties -> largest id (the candidate heap is
    // keyed (-dist, id)); `pred` / `h2k` = the runner-up's id / distance key when it is well defined.  False: nothing
    // is left (the reference's loop exit).
    __device__ __forceinline__ bool select(uint32_t& node, uint32_t& pred, uint32_t& h2k, int lane) {
        if (!c_valid) refresh_cache(lane);
        // Common case as one branch-free block (the walk is instruction-issue bound; the compiler's version of this
        // logic is twice as long): the front list's two closest unexpanded entries (dead lanes hold all-ones = read
        // as expanded), the winner against the cached base entries, the runner-up as the prediction.
        //   ok   = the closest distance is unique and the tie list is empty (else: slow path below)
        //   pred = runner-up id, -1 when there is none or the two runner-up candidates have equal distances
        uint32_t ok, fb, q1, q2, hf1, hf2, nf1, nf2, hw, ha, na, hb, nb2, hr, t0;
        uint64_t fm;
#ifdef GBNNS_COOP_HINT   // one more scalar select inside the block below: the candidate the runner-up's comparison did NOT pick
        uint32_t hint_;
#define GBNNS_SEL_HINT_LINE "s_cselect_b32 %[hint], %[nb2], %[na]\n\t"
#define GBNNS_SEL_HINT_OUT , [hint] "=&s"(hint_)
#else
#define GBNNS_SEL_HINT_LINE
#define GBNNS_SEL_HINT_OUT
#endif
        // (rfl: a no-op where the compiler already keeps the cache in scalar registers; where it chose vector registers for
        // it -- it may, the values come out of LDS -- the asm below still gets scalars)
        auto rfl = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
        asm volatile(
            "v_and_b32 %[t0], 1, %[flo]\n\t"
            "v_cmp_eq_u32 vcc, 0, %[t0]\n\t"
            "s_mov_b64 %[fm], vcc\n\t"
            "s_ff1_i32_b64 %[q1], %[fm]\n\t"               // -1 when the front list has no unexpanded entry
            "s_bitset0_b64 %[fm], %[q1]\n\t"
            "s_ff1_i32_b64 %[q2], %[fm]\n\t"
            "v_readlane_b32 %[hf1], %[fhi], %[q1]\n\t"      // (lane index taken mod 64; fixed up below)
            "v_readlane_b32 %[nf1], %[flo], %[q1]\n\t"
            "v_readlane_b32 %[hf2], %[fhi], %[q2]\n\t"
            "v_readlane_b32 %[nf2], %[flo], %[q2]\n\t"
            "s_cmp_lt_i32 %[q1], 0\n\t"
            "s_cselect_b32 %[hf1], -1, %[hf1]\n\t"
            "s_cmp_lt_i32 %[q2], 0\n\t"
            "s_cselect_b32 %[hf2], -1, %[hf2]\n\t"
            "s_lshr_b32 %[nf1], %[nf1], 1\n\t"
            "s_lshr_b32 %[nf2], %[nf2], 1\n\t"
            "s_cmp_lt_u32 %[h1], %[hf1]\n\t"                // the base list's closest entry wins
            "s_cselect_b32 %[fb], 1, 0\n\t"
            "s_cselect_b32 %[hw], %[h1], %[hf1]\n\t"
            "s_cselect_b32 %[node], %[n1], %[nf1]\n\t"
            "s_cselect_b32 %[ha], %[h2], %[hf2]\n\t"        // second entry of the winning list
            "s_cselect_b32 %[na], %[n2], %[nf2]\n\t"
            "s_cselect_b32 %[hb], %[hf1], %[h1]\n\t"        // first entry of the other list
            "s_cselect_b32 %[nb2], %[nf1], %[n1]\n\t"
            "s_min_u32 %[hr], %[ha], %[hb]\n\t"
            "s_cmp_lt_u32 %[ha], %[hb]\n\t"
            "s_cselect_b32 %[pred], %[na], %[nb2]\n\t"
            GBNNS_SEL_HINT_LINE
            "s_cmp_lg_u32 %[h1], %[hf1]\n\t"                // equal: a tie across the lists, or both lists empty
            "s_cselect_b32 %[ok], 1, 0\n\t"
            "s_cmp_lg_u32 %[hr], %[hw]\n\t"                 // the runner-up ties the winner
            "s_cselect_b32 %[ok], %[ok], 0\n\t"
            "s_cmp_eq_u32 %[ts], 0\n\t"
            "s_cselect_b32 %[ok], %[ok], 0\n\t"
            "s_cmp_lg_u32 %[ha], %[hb]\n\t"                 // ambiguous runner-up: no prediction
            "s_cselect_b32 %[pred], %[pred], -1\n\t"
            "s_cselect_b32 %[hr], %[hr], -1\n\t"
            "s_cmp_lg_u32 %[hr], -1\n\t"
            "s_cselect_b32 %[pred], %[pred], -1"
            : [ok] "=&s"(ok), [fb] "=&s"(fb), [q1] "=&s"(q1), [q2] "=&s"(q2), [hf1] "=&s"(hf1), [hf2] "=&s"(hf2),
              [nf1] "=&s"(nf1), [nf2] "=&s"(nf2), [hw] "=&s"(hw), [ha] "=&s"(ha), [na] "=&s"(na), [hb] "=&s"(hb),
              [nb2] "=&s"(nb2), [hr] "=&s"(hr), [t0] "=&v"(t0), [fm] "=&s"(fm), [node] "=&s"(node), [pred] "=&s"(pred) GBNNS_SEL_HINT_OUT
            : [flo] "v"(F.lo[0]), [fhi] "v"(F.hi[0]), [h1] "s"(rfl(h1)), [n1] "s"(rfl(n1)), [h2] "s"(rfl(h2)), [n2] "s"(rfl(n2)), [ts] "s"(rfl((uint32_t)tsize))
            : "vcc", "scc");
        h2k = hr;
#ifdef GBNNS_COOP_HINT
        hint3 = hint_;   // (walk_coop.hip) of {the winning list's second entry, the other list's first} the one that is NOT the runner-up
#endif
        if (__builtin_expect(ok != 0, 1)) {
            if (fb) expand_base(p1, lane);
            else F.lo[0] |= (lane == (int)q1) ? 1u : 0u;
            return true;
        }
        pred = kInvalidId;
        h2k = 0xFFFFFFFFu;
        const int pF = (int)q1;
        const uint32_t hF1 = hf1;
#ifdef GBNNS_STAMPS
        st_slow += 1;
#endif
        // rare: equal distances among the closest unexpanded entries, a non-empty tie list, or the end
        const bool any = p1 >= 0 || pF >= 0;
        const uint32_t hi_p = h1 < hF1 ? h1 : hF1;
        int bestL = -1, bestF = -1;  // largest id with that distance: the last unexpanded one of its run in either list
        uint32_t idL = 0, idF = 0;
        if (any) {
            if (p1 >= 0 && h1 == hi_p) {
                for (int b0 = p1; b0 < l; b0 += 64) {  // the run of equal distances starts at p1
                    const int r = b0 + lane;
                    uint64_t kv = base[r < l ? r : 0];
                    kv = r < l ? kv : ~0ull;
                    const bool same = (uint32_t)(kv >> 32) == hi_p;
                    const uint64_t ms = __ballot(same && !((uint32_t)kv & 1u));
                    if (ms) {
                        const int q = 63 - __clzll((long long)ms);
                        bestL = b0 + q;
                        idL = readlane_u32((uint32_t)kv, q) >> 1;
                    }
                    if (!((__ballot(same) >> 63) & 1ull)) break;  // the run ends inside this chunk
                }
            }
            const uint64_t msf = __ballot(!(F.lo[0] & 1u) && F.hi[0] == hi_p);
            if (msf) {
                bestF = 63 - __clzll((long long)msf);
                idF = readlane_u32(F.lo[0], bestF) >> 1;
            }
        }
        const bool pickL = bestL >= 0 && (bestF < 0 || idL > idF);
        const bool have = bestL >= 0 || bestF >= 0;
        const uint32_t lid = pickL ? idL : idF;
        if (tsize > 0 && (!have || hi_p == worst)) {
            // tie entries all sit at the worst distance: the largest id among them competes
            uint32_t v = (lane < tsize) ? key_id(tie[lane]) + 1u : 0u;
            int w = lane;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t ov = (uint32_t)__shfl_xor((int)v, off);
                const int ow = __shfl_xor(w, off);
                if (ov > v) { v = ov; w = ow; }
            }
            v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
            w = __builtin_amdgcn_readfirstlane(w);
            if (!have || v - 1u > lid) {
                node = v - 1u;
                tie[w] = tie[tsize - 1];  // (every lane: same value)
                tsize -= 1;
                wave_sync();
                return true;
            }
        }
        if (!have) return false;
        node = lid;
        if (pickL) expand_base(bestL, lane);
        else F.lo[0] |= (lane == bestF) ? 1u : 0u;
        return true;
    }
    // End of a walk: one sorted list (flush), the outputs in POP order (rank i goes to position kept - 1 - i), and --
    // when the walk kernels re-rank -- getRealNearest on this query with the original-space query staged in
    // `rr_scratch` (LDS that the walk no longer needs; the base list must stay readable).
    template <int DEEP = 8>
    __device__ __forceinline__ void finish(const WalkParams& p, uint32_t qi, int hops, int dist_calc, int edges,
                                           unsigned char* rr_scratch, int lane) {
        flush(lane);
        const int kept = l < p.k ? l : p.k;
        for (int rank = lane; rank < (int)p.cand_stride; rank += 64) {
            const uint64_t kv = base[rank < kept ? rank : 0];
            const size_t at = (size_t)qi * p.cand_stride + (rank < kept ? kept - 1 - rank : rank);
            p.cand[at] = rank < kept ? key_id(kv) : kInvalidId;
            if (p.cand_dist) p.cand_dist[at] = rank < kept ? fkey_inv_out(key_hi(kv), p.zero_dist_bits) : __builtin_inff();
        }
        if (lane == 0) {
            p.count[qi] = kept;
            p.hops[qi] = hops;
            p.dist_calc[qi] = dist_calc;
            atomicMax(p.max_dc, (uint32_t)dist_calc);
            if (p.edges) p.edges[qi] = edges;
            // PLAIN answer = topk.top() after trimming the heap to k (search_function.h:174-181): the k-th best
            if (p.best) p.best[qi] = kept > 0 ? key_id(base[kept - 1]) : kInvalidId;
        }
        if (p.rr_db) {
            const uint64_t* b = base;
            fused_rerank<DEEP>(p, qi, kept, rr_scratch, lane, [&](int rank) { return key_id(b[rank]); });
        }
    }

    // the state of an empty union around `entry` (the caller puts the entry's key into lane 0 of the front list)
    __device__ __forceinline__ void init(unsigned char* smem, int ef_) {
        const int ef_pad = (ef_ + 63) & ~63;
        tie = reinterpret_cast<uint64_t*>(smem);
        stage = tie + kRegTieCap;
        base = stage + kRegStageSlots;
        flags = big_list_flags_in_stage(ef_) ? reinterpret_cast<unsigned char*>(stage) : reinterpret_cast<unsigned char*>(base + ef_pad);
        F.clear();
        ef = ef_; l = 0; f = 1; tsize = 0;
        mu_lo = mu_hi = 0u;
        c_valid = false; p1 = p2 = -1; h1 = h2 = 0xFFFFFFFFu; n1 = n2 = 0u;
        worst = fworst = 0u;
    }
};
}  // namespace

}  // namespace gbnns
