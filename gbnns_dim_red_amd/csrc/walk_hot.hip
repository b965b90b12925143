// walk_hot.hip -- the hand-laid-out first-pass instances for 128-byte rows of a compact index: hot_expand
// (one asm block per hop: visited-set protocol, row loads, pair distance), walk_hot_one (ef <= 128, result list in one / two
// registers per lane), walk_hot_big (128 < ef <= 1 024, two-list), for the L2 and the negative-dot metric and for adjacency
// rows of <= 32 and 33 .. 64 slots.
#include "launch_util.h"
#include "walk_lists.h"

// (A/B switch: wavefronts per SIMD the ef <= 64 instances are compiled for.  8 = at most 80 scalar registers, 13 of them spilled
// to lanes of a vector register; 7 = 94, no spill: 0.313 against 0.306 ms alone, a tie with batches in flight)
#ifndef GBNNS_HOT1_LB
#define GBNNS_HOT1_LB (GBNNS_HOT1_QLDS ? 8 : 7)
#endif
namespace gbnns {

namespace {

// ---- hot instance: L2, 128-byte rows, ef <= 64, adjacency rows of <= 32 slots, 32-bit offsets ---------
//
// Same algorithm and data structures as walk_reg_one<0, 8, true, 1> in its pair form; the hop is laid
// out as one straight common path (hand-written selection, probe and distance blocks, every rare case
// out of line), because this instance is bound by instruction issue and by the CU's vector-memory path.

// Rare part of the selection (register list, one entry per lane): an equal-distance run among the
// unexpanded entries, a non-empty tie list, or nothing left.  Returns false at the end of the walk.
__device__ __forceinline__ bool reg1_select_slow(RegList<1>& L, uint64_t mu, int& tsize, uint64_t* tie, uint32_t worst,
                                                 uint64_t lmask, int lane, uint32_t& node) {
    int best = -1;
    uint32_t hi_p = 0;
    if (mu) {
        hi_p = readlane_u32(L.hi[0], __ffsll((unsigned long long)mu) - 1);
        const uint64_t ms = __ballot(!(L.lo[0] & 1u) && L.hi[0] == hi_p) & lmask;
        if (ms) best = 63 - __clzll((long long)ms);
    }
    if (tsize > 0 && (best < 0 || hi_p == worst)) {
        // tie entries all sit at the worst distance: the largest id among them competes
        uint32_t v = (lane < tsize) ? key_id(tie[lane]) + 1u : 0u;
        int w = lane;
        // (this rare path must not cost the hop registers: with __shfl_xor the six partner indices lane ^ 32 .. lane ^ 1 and
        // a second copy of the lane id are loop invariants that the compiler keeps in registers across the whole walk; the
        // lane id is laundered here so that they are computed on the spot)
        int lane_here = lane;
        asm volatile("" : "+v"(lane_here));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int partner = (lane_here ^ off) << 2;
            const uint32_t ov = (uint32_t)__builtin_amdgcn_ds_bpermute(partner, (int)v);
            const int ow = __builtin_amdgcn_ds_bpermute(partner, w);
            if (ov > v) { v = ov; w = ow; }
        }
        v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
        w = __builtin_amdgcn_readfirstlane(w);
        const uint32_t lid = (best >= 0) ? (readlane_u32(L.lo[0], best) >> 1) : 0u;
        if (best < 0 || v - 1u > lid) {
            node = v - 1u;
            if (lane == 0) tie[w] = tie[tsize - 1];
            tsize -= 1;
            wave_sync();
            return true;
        }
    }
    if (best < 0) return false;
    node = readlane_u32(L.lo[0], best) >> 1;
    if (lane == best) L.lo[0] |= 1u;
    return true;
}

// The expansion of one node in the hot instance, as ONE block (so that no compiler-chosen register can
// sit between the row loads and their use): issue this lane's four 16-B row loads (lanes of `valid`),
// run the visited-set protocol of visited_claim_mask on the even lanes while they are in flight, then
// the pair distance of l2_pair_from_regs.  Returns the sort key of the distance (meaningful in the odd
// lane of a pair whose id was new); `claimed` = even lanes whose id was new.
// The same for lists of R registers per lane (`p1` = rank of the closest unexpanded entry, -1 if none).
template <int R>
__device__ __forceinline__ bool regN_select_slow(RegList<R>& L, int p1, int& tsize, uint64_t* tie, uint32_t worst, int ef,
                                                 int lane, uint32_t& node) {
    int best = -1;
    uint32_t hi_p = 0;
    if (p1 >= 0) {
        hi_p = L.hi_at(p1);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t ms = __ballot(!(L.lo[r] & 1u) && L.hi[r] == hi_p) & RegList<R>::lane_mask(r, ef);
            if (ms) best = r * 64 + 63 - __clzll((long long)ms);
        }
    }
    if (tsize > 0 && (best < 0 || hi_p == worst)) {
        // tie entries all sit at the worst distance: the largest id among them competes
        uint32_t v = (lane < tsize) ? key_id(tie[lane]) + 1u : 0u;
        int w = lane;
        // (this rare path must not cost the hop registers: with __shfl_xor the six partner indices lane ^ 32 .. lane ^ 1 and
        // a second copy of the lane id are loop invariants that the compiler keeps in registers across the whole walk; the
        // lane id is laundered here so that they are computed on the spot)
        int lane_here = lane;
        asm volatile("" : "+v"(lane_here));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int partner = (lane_here ^ off) << 2;
            const uint32_t ov = (uint32_t)__builtin_amdgcn_ds_bpermute(partner, (int)v);
            const int ow = __builtin_amdgcn_ds_bpermute(partner, w);
            if (ov > v) { v = ov; w = ow; }
        }
        v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
        w = __builtin_amdgcn_readfirstlane(w);
        const uint32_t lid = (best >= 0) ? (L.lo_at(best) >> 1) : 0u;
        if (best < 0 || v - 1u > lid) {
            node = v - 1u;
            if (lane == 0) tie[w] = tie[tsize - 1];
            tsize -= 1;
            wave_sync();
            return true;
        }
    }
    if (best < 0) return false;
    node = L.lo_at(best) >> 1;
    L.mark_expanded(best, lane);
    return true;
}

// The visited-set section of hot_expand's asm block (shared by its two metric forms): even lanes of `valid` test and
// claim their id; %[fresh] = lanes whose id was new (even bits) and, shifted to their odd neighbours' bits, lanes that
// ran out of probe range (quotient form only).
//
// Packed form (%[shr] == 0): a 16-byte bucket holds five 24-bit ids (bits
// 24k .. 24k+23, all-ones = empty) and, in its top byte, the number of slots handed out.  An id is in the set iff it is
// found in a bucket of its probe sequence before a bucket with a free slot; a new id takes the slot index an atomic
// add on that counter returns (unique per lane, so no compare-and-swap and no retry inside a bucket) and writes its
// three bytes.  3.2 bytes per id.
//
// Quotient form (%[shr] != 0: shift count in its low five bits, 32 - W in bits 8 .. 12, bit 16 = thirteen remainder
// bits instead of twelve, displacement limit -- 15 -- in its top four; n <= 2^W): H = id * (0x9E3779B1 << (32 - W)) is a
// bijection of the ids onto the multiples of 2^(32-W); the home bucket is mulhi(H, buckets) and the low word of that
// product, shifted right by 32 - W + floor(log2 buckets), tells the ids of one home bucket apart in
// W - floor(log2 buckets) <= 12 bits (the host checks).  A bucket holds seven 16-bit entries -- displacement from the
// home bucket (0 .. 14) << 12 | those bits; 0xFFFF = empty -- and in its top halfword 0xF000 | slots handed out (no
// key has displacement 15, so neither that halfword nor an empty slot ever compares equal).  Same protocol as above;
// 2.29 bytes per id and nine instead of fifteen instructions per bucket test.  The "displacement" is the probe number:
// probe j + 1 looks 1 .. 8 buckets (by the key's low three bits) beyond probe j, so ids of neighbouring home buckets
// do not queue up behind one run of full buckets (with steps of one bucket a 10 000-query batch at ef = 140 handed a
// few queries over every time).  A probe sequence longer than fifteen buckets gives up: the lane is reported and its id
// goes to the stash (stash_claim).  Tables of 2^(W-13) .. 2^(W-12) buckets keep thirteen remainder bits and a 3-bit
// probe number (seven probes; bit 16 of %[shr]).
#define GBNNS_VS_ASM(T0, T1, T2, ADDR)                                                                                                   \
        "s_bfe_u32 %[mulc], %[shr], 0x50008\n\t"               /* 32 - W (0 in the packed form) */                     \
        "s_lshl_b32 %[mulc], 0x9E3779B1, %[mulc]\n\t"                                                                  \
        "v_mul_lo_u32 " T0 ", %[id], %[mulc]\n\t"                                                                       \
        "s_and_b32 exec_lo, exec_lo, 0x55555555\n\t"                                                                   \
        "s_and_b32 exec_hi, exec_hi, 0x55555555\n\t"                                                                   \
        "s_mov_b64 %[fresh], 0\n\t"                                                                                    \
        "s_cmp_lg_u32 %[shr], 0\n\t"                                                                                   \
        "v_mul_hi_u32 " T1 ", " T0 ", %[nb]\n\t"                                                                         \
        "v_lshl_add_u32 " ADDR ", " T1 ", 4, %[basev]\n\t"                                                               \
        "s_cbranch_scc1 4f\n"                                                                                          \
        "1:\n\t"                                                                                                       \
        "ds_read_b128 v[60:63], " ADDR "\n\t"                                                                           \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                     \
        "v_bfe_u32 v56, v60, 0, 24\n\t"                        /* slot 0 */                                            \
        "v_alignbit_b32 v57, v61, v60, 24\n\t"                 /* slot 1 (bits 24..47) in the low 24 bits */           \
        "v_alignbit_b32 v58, v62, v61, 16\n\t"                 /* slot 2 (bits 48..71) */                              \
        "v_lshrrev_b32 v59, 8, v62\n\t"                        /* slot 3 (bits 72..95) */                              \
        "v_bfe_u32 " T1 ", v63, 0, 24\n\t"                      /* slot 4 (bits 96..119) */                             \
        "v_bfe_u32 v57, v57, 0, 24\n\t"                                                                                \
        "v_bfe_u32 v58, v58, 0, 24\n\t"                                                                                \
        "v_xor_b32 v56, v56, %[id]\n\t"                                                                                \
        "v_xor_b32 v57, v57, %[id]\n\t"                                                                                \
        "v_xor_b32 v58, v58, %[id]\n\t"                                                                                \
        "v_xor_b32 v59, v59, %[id]\n\t"                                                                                \
        "v_xor_b32 " T1 ", " T1 ", %[id]\n\t"                                                                            \
        "v_min3_u32 v56, v56, v57, v58\n\t"                                                                            \
        "v_min3_u32 v56, v56, v59, " T1 "\n\t"                  /* 0 <=> id is in the bucket */                         \
        "v_lshrrev_b32 " T1 ", 24, v63\n\t"                     /* slots handed out */                                  \
        "v_cmp_ne_u32 vcc, 0, v56\n\t"                                                                                 \
        "s_and_b64 exec, exec, vcc\n\t"                        /* lanes that found their id are done */                \
        "s_cbranch_execz 9f\n\t"                                                                                       \
        "s_mov_b64 %[act], exec\n\t"                                                                                   \
        "v_cmp_gt_u32 vcc, 5, " T1 "\n\t"                                                                               \
        "s_and_b64 exec, exec, vcc\n\t"                        /* the bucket had room when it was read */              \
        "s_cbranch_execz 3f\n\t"                                                                                       \
        "v_mov_b32 " T1 ", 0x1000000\n\t"                                                                               \
        "ds_add_rtn_u32 " T0 ", " ADDR ", " T1 " offset:12\n\t"   /* take a slot number */                                \
        "v_lshrrev_b32 " T2 ", 8, %[id]\n\t"                                                                            \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                     \
        "v_lshrrev_b32 " T0 ", 24, " T0 "\n\t"                                                                           \
        "v_cmp_gt_u32 vcc, 5, " T0 "\n\t"                                                                               \
        "s_and_b64 exec, exec, vcc\n\t"                        /* lanes whose number is a real slot */                 \
        "s_cbranch_execz 3f\n\t"                                                                                       \
        "v_mad_u32_u24 " T0 ", " T0 ", 3, " ADDR "\n\t"           /* byte address of the slot */                          \
        "ds_write_b8 " T0 ", %[id]\n\t"                                                                                 \
        "ds_write_b8 " T0 ", " T2 " offset:1\n\t"                                                                        \
        "ds_write_b8_d16_hi " T0 ", %[id] offset:2\n\t"                                                                 \
        "s_or_b64 %[fresh], %[fresh], exec\n\t"                                                                        \
        "s_andn2_b64 %[act], %[act], exec\n"                                                                           \
        "3:\n\t"                                                                                                       \
        "s_mov_b64 exec, %[act]\n\t"                           /* absent and unplaced: their bucket is full */         \
        "s_cbranch_execz 9f\n\t"                                                                                       \
        "v_add_u32 " ADDR ", 16, " ADDR "\n\t"                                                                           \
        "v_cmp_eq_u32 vcc, %[end], " ADDR "\n\t"                                                                        \
        "v_cndmask_b32 " ADDR ", " ADDR ", %[basev], vcc\n\t"                                                            \
        "s_branch 1b\n"                                                                                                \
        "4:\n\t"                                               /* ---- quotient form ---- */                           \
        "s_lshl_b32 %[mulc], %[nb], 4\n\t"                                                                             \
        "v_mul_lo_u32 " T0 ", " T0 ", %[nb]\n\t"                 /* place inside the home bucket's range */              \
        "v_lshrrev_b32 " T0 ", %[shr], " T0 "\n\t"               /* < 2^12 */                                            \
        "v_lshl_or_b32 " T2 ", " T0 ", 16, " T0 "\n"              /* the key in both halves, displacement 0 */            \
        "5:\n\t"                                                                                                       \
        "ds_read_b128 v[60:63], " ADDR "\n\t"                                                                           \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                     \
        "v_xor_b32 v56, v60, " T2 "\n\t"                                                                                \
        "v_xor_b32 v57, v61, " T2 "\n\t"                                                                                \
        "v_xor_b32 v58, v62, " T2 "\n\t"                                                                                \
        "v_xor_b32 v59, v63, " T2 "\n\t"                                                                                \
        "v_pk_min_u16 v56, v56, v57\n\t"                                                                               \
        "v_pk_min_u16 v58, v58, v59\n\t"                                                                               \
        "v_bfe_u32 " T1 ", v63, 16, 12\n\t"                     /* slots handed out */                                  \
        "v_pk_min_u16 v56, v56, v58\n\t"                                                                               \
        "v_mad_u32_u16 v56, v56, v56, 0 op_sel:[0,1,0,0]\n\t"  /* low half x high half: 0 <=> the key is in the bucket */ \
        "v_cmp_ne_u32 vcc, 0, v56\n\t"                                                                                 \
        "s_and_b64 exec, exec, vcc\n\t"                                                                                \
        "s_cbranch_execz 9f\n\t"                                                                                       \
        "s_mov_b64 %[act], exec\n\t"                                                                                   \
        "v_cmp_gt_u32 vcc, 7, " T1 "\n\t"                                                                               \
        "s_and_b64 exec, exec, vcc\n\t"                                                                                \
        "s_cbranch_execz 6f\n\t"                                                                                       \
        "v_mov_b32 " T1 ", 0x10000\n\t"                                                                                 \
        "ds_add_rtn_u32 " T0 ", " ADDR ", " T1 " offset:12\n\t"                                                           \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                     \
        "v_bfe_u32 " T0 ", " T0 ", 16, 12\n\t"                                                                           \
        "v_cmp_gt_u32 vcc, 7, " T0 "\n\t"                                                                               \
        "s_and_b64 exec, exec, vcc\n\t"                                                                                \
        "s_cbranch_execz 6f\n\t"                                                                                       \
        "v_lshl_add_u32 " T0 ", " T0 ", 1, " ADDR "\n\t"                                                                  \
        "ds_write_b16 " T0 ", " T2 "\n\t"                                                                                \
        "s_or_b64 %[fresh], %[fresh], exec\n\t"                                                                        \
        "s_andn2_b64 %[act], %[act], exec\n"                                                                           \
        "6:\n\t"                                                                                                       \
        "s_mov_b64 exec, %[act]\n\t"                                                                                   \
        "s_cbranch_execz 9f\n\t"                                                                                       \
        "v_and_b32 " T0 ", 7, " T2 "\n\t"                        /* next probe: 1 .. 8 buckets on, by the key's low bits */ \
        "v_lshl_add_u32 " T0 ", " T0 ", 4, 16\n\t"               /* (no runs of full buckets shared by neighbouring homes) */ \
        "v_add_u32 " ADDR ", " ADDR ", " T0 "\n\t"                                                                        \
        "s_bfe_u32 vcc_lo, %[shr], 0x10010\n\t"                /* 13 remainder bits: the probe number sits one bit higher */ \
        "s_lshl_b32 vcc_lo, 0x10001000, vcc_lo\n\t"                                                                    \
        "v_add_u32 " T2 ", vcc_lo, " T2 "\n\t"                   /* one probe further from home */                       \
        "v_cmp_le_u32 vcc, %[end], " ADDR "\n\t"                                                                        \
        "v_subrev_u32 " T0 ", %[mulc], " ADDR "\n\t"             /* (%[mulc] holds the table's bytes by now) */          \
        "v_cndmask_b32 " ADDR ", " ADDR ", " T0 ", vcc\n\t"                                                               \
        "s_and_b32 vcc_lo, %[shr], 0xF0000000\n\t"           /* the probe-number field alone (ctl's low bits hold shifts and flags) */ \
        "v_cmp_gt_u32 vcc, vcc_lo, " T2 "\n\t"                  /* probe number still in range (below %[shr] >> 28) */  \
        "s_andn2_b64 %[act], exec, vcc\n\t"                    /* lanes out of range: reported in the odd bits of %[fresh] */ \
        "s_lshl_b64 %[act], %[act], 1\n\t"                                                                             \
        "s_or_b64 %[fresh], %[fresh], %[act]\n\t"                                                                      \
        "s_and_b64 exec, exec, vcc\n\t"                                                                                \
        "s_cbranch_execnz 5b\n"                                                                                        \
        "9:\n\t"

// METRIC 0: L2Metric::Dist, the lane holds 64 contiguous bytes of the row (roff = row + half * 64, loads at 0 / 16 / 32 / 48).
// METRIC 1: Angular::Dist, the lane holds the even (odd) 16-byte pieces (roff = row + half * 16, loads at 0 / 32 / 64 / 96):
// its eight running sums are independent chains, the even lane runs sums 0..3, the odd lane sums 4..7, and the fold
// m_j = c_{j+4} + c_j happens once, in the odd lane (dot_pair_from_regs).
//
// The lane's 16 query floats (the pieces that face its four row loads).  QLDS = false: all in registers (`qh`).
// QLDS = true (round 4: the ef <= 128 instances): none stays in registers; the pieces are re-read every hop from the
// wavefront's LDS copy of the query (`qaddr` = its byte address + this lane's piece offset; the pieces sit at the row
// loads' offsets) into the block's own temporaries -- v[36:39], free once the visited-set protocol is done, and v[56:63]
// -- right behind the row loads, i.e. in their shadow (>= 500 cycles); the fourth piece follows into v[36:39] as soon as
// step 0 has consumed the first.  The walk then holds 16 registers less across the hop, which is what lets these
// instances fit 64 registers (8 wavefronts per SIMD) without a spill.
#define GBNNS_L2_DIST_ASM(Q0A, Q0B, Q1A, Q1B, Q2A, Q2B, Q3A, Q3B, W1, W2, W3)                                                     \
        "s_waitcnt vmcnt(3)\n\t"                               /* loads return in order: square each step as it lands */ \
        "v_pk_add_f32 v[40:41], v[40:41], " Q0A " neg_lo:[0,1] neg_hi:[0,1]\n\t"                                        \
        "v_pk_add_f32 v[42:43], v[42:43], " Q0B " neg_lo:[0,1] neg_hi:[0,1]\n\t"                                        \
        "v_pk_mul_f32 v[40:41], v[40:41], v[40:41]\n\t"                                                                \
        "v_pk_mul_f32 v[42:43], v[42:43], v[42:43]\n\t"                                                                \
        W1                                                                                                             \
        "v_pk_add_f32 v[44:45], v[44:45], " Q1A " neg_lo:[0,1] neg_hi:[0,1]\n\t"                                        \
        "v_pk_add_f32 v[46:47], v[46:47], " Q1B " neg_lo:[0,1] neg_hi:[0,1]\n\t"                                        \
        "v_pk_mul_f32 v[44:45], v[44:45], v[44:45]\n\t"                                                                \
        "v_pk_mul_f32 v[46:47], v[46:47], v[46:47]\n\t"                                                                \
        W2                                                                                                             \
        "v_pk_add_f32 v[48:49], v[48:49], " Q2A " neg_lo:[0,1] neg_hi:[0,1]\n\t"                                        \
        "v_pk_add_f32 v[50:51], v[50:51], " Q2B " neg_lo:[0,1] neg_hi:[0,1]\n\t"                                        \
        "v_pk_mul_f32 v[48:49], v[48:49], v[48:49]\n\t"                                                                \
        "v_pk_mul_f32 v[50:51], v[50:51], v[50:51]\n\t"                                                                \
        W3                                                                                                             \
        "v_pk_add_f32 v[52:53], v[52:53], " Q3A " neg_lo:[0,1] neg_hi:[0,1]\n\t"                                        \
        "v_pk_add_f32 v[54:55], v[54:55], " Q3B " neg_lo:[0,1] neg_hi:[0,1]\n\t"                                        \
        "v_pk_mul_f32 v[52:53], v[52:53], v[52:53]\n\t"                                                                \
        "v_pk_mul_f32 v[54:55], v[54:55], v[54:55]\n\t"                                                                \
        "v_pk_add_f32 v[60:61], v[40:41], v[44:45]\n\t"      /* even lane: steps 0..3 */                               \
        "v_pk_add_f32 v[62:63], v[42:43], v[46:47]\n\t"                                                                \
        "v_pk_add_f32 v[60:61], v[60:61], v[48:49]\n\t"                                                                \
        "v_pk_add_f32 v[62:63], v[62:63], v[50:51]\n\t"                                                                \
        "v_pk_add_f32 v[60:61], v[60:61], v[52:53]\n\t"                                                                \
        "v_pk_add_f32 v[62:63], v[62:63], v[54:55]\n\t"                                                                \
        "s_nop 1\n\t"                                                                                                  \
        "v_mov_b32_dpp v56, v60 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"                                    \
        "v_mov_b32_dpp v57, v61 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"                                    \
        "v_mov_b32_dpp v58, v62 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"                                    \
        "v_mov_b32_dpp v59, v63 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"                                    \
        "v_pk_add_f32 v[56:57], v[56:57], v[40:41]\n\t"      /* odd lane: steps 4..7 on top */                         \
        "v_pk_add_f32 v[58:59], v[58:59], v[42:43]\n\t"                                                                \
        "v_pk_add_f32 v[56:57], v[56:57], v[44:45]\n\t"                                                                \
        "v_pk_add_f32 v[58:59], v[58:59], v[46:47]\n\t"                                                                \
        "v_pk_add_f32 v[56:57], v[56:57], v[48:49]\n\t"                                                                \
        "v_pk_add_f32 v[58:59], v[58:59], v[50:51]\n\t"                                                                \
        "v_pk_add_f32 v[56:57], v[56:57], v[52:53]\n\t"                                                                \
        "v_pk_add_f32 v[58:59], v[58:59], v[54:55]\n\t"                                                                \
        "v_add_f32 %[key], v56, v57\n\t"                                                                               \
        "v_add_f32 %[key], %[key], v58\n\t"                                                                            \
        "v_add_f32 %[key], %[key], v59\n\t"                                                                            \
        "v_or_b32 %[key], 0x80000000, %[key]"                  /* fkey of a non-negative float */

// (dot_pair_from_regs) products, then four running sums from +0 in load order
#define GBNNS_DOT_DIST_ASM(Q0A, Q0B, Q1A, Q1B, Q2A, Q2B, Q3A, Q3B, W1, W2, W3)                                                    \
        "s_waitcnt vmcnt(3)\n\t"                                                                                       \
        "v_pk_mul_f32 v[40:41], v[40:41], " Q0A "\n\t"                                                                 \
        "v_pk_mul_f32 v[42:43], v[42:43], " Q0B "\n\t"                                                                 \
        W1                                                                                                             \
        "v_pk_mul_f32 v[44:45], v[44:45], " Q1A "\n\t"                                                                 \
        "v_pk_mul_f32 v[46:47], v[46:47], " Q1B "\n\t"                                                                 \
        W2                                                                                                             \
        "v_pk_mul_f32 v[48:49], v[48:49], " Q2A "\n\t"                                                                 \
        "v_pk_mul_f32 v[50:51], v[50:51], " Q2B "\n\t"                                                                 \
        W3                                                                                                             \
        "v_pk_mul_f32 v[52:53], v[52:53], " Q3A "\n\t"                                                                 \
        "v_pk_mul_f32 v[54:55], v[54:55], " Q3B "\n\t"                                                                 \
        "v_mov_b32 v56, 0\n\t"                                 /* (the query pieces in v[56:63] are dead by now) */    \
        "v_mov_b32 v57, 0\n\t"                                                                                         \
        "v_pk_add_f32 v[60:61], v[56:57], v[40:41]\n\t"      /* 0 + p: a product of -0 must not make the sum -0 */     \
        "v_pk_add_f32 v[62:63], v[56:57], v[42:43]\n\t"                                                                \
        "v_pk_add_f32 v[60:61], v[60:61], v[44:45]\n\t"                                                                \
        "v_pk_add_f32 v[62:63], v[62:63], v[46:47]\n\t"                                                                \
        "v_pk_add_f32 v[60:61], v[60:61], v[48:49]\n\t"                                                                \
        "v_pk_add_f32 v[62:63], v[62:63], v[50:51]\n\t"                                                                \
        "v_pk_add_f32 v[60:61], v[60:61], v[52:53]\n\t"                                                                \
        "v_pk_add_f32 v[62:63], v[62:63], v[54:55]\n\t"                                                                \
        "s_nop 1\n\t"                                                                                                  \
        "v_mov_b32_dpp v56, v60 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"   /* the even lane's sums 0..3 */   \
        "v_mov_b32_dpp v57, v61 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"                                    \
        "v_mov_b32_dpp v58, v62 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"                                    \
        "v_mov_b32_dpp v59, v63 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"                                    \
        "v_pk_add_f32 v[56:57], v[60:61], v[56:57]\n\t"      /* odd lane: m_j = c_{j+4} + c_j */                       \
        "v_pk_add_f32 v[58:59], v[62:63], v[58:59]\n\t"                                                                \
        "v_add_f32 %[key], v56, v57\n\t"                       /* (m0 + m1) + (m2 + m3) */                             \
        "v_add_f32 v60, v58, v59\n\t"                        /* (v[60:63] are free again) */                       \
        "v_add_f32 %[key], %[key], v60\n\t"                                                                            \
        "v_xor_b32 %[key], 0x80000000, %[key]\n\t"             /* Angular::Dist = -(x . y) */                          \
        "v_add_f32 %[key], 0, %[key]\n\t"                      /* fkey: -0 -> +0, */                                   \
        "v_ashrrev_i32 v60, 31, %[key]\n\t"                    /* then flip all bits of a negative value, the sign bit of a positive one */ \
        "v_or_b32 v60, 0x80000000, v60\n\t"                                                                            \
        "v_xor_b32 %[key], %[key], v60"

// The four row loads of a lane, two layouts:
//  * SPEC (rounds 1-3; since round 4 only the instances for ef > 64): requested for every valid slot BEFORE the visited
//    test, which then runs in their shadow -- shortest hop, but the rows of already-visited ids are fetched for nothing
//    (15 % of the tested ids at ef = 64, two thirds at ef >= 180).
//  * tested first (round 4, the ef <= 64 instances): requested AFTER the test, for the new ids only (lanes 2i / 2i+1 of
//    a pair whose even lane claimed its id, or ran out of probe range and goes to the stash).  With 32 wavefronts per
//    CU resident the longer hop is hidden and the saved row traffic shows: SIFT-like ef = 64 0.335 -> 0.322 ms, 29.3 ->
//    31.0 M queries/s in flight; at ef = 128 / 180 / 300 (fewer wavefronts per CU: latency chains) it loses 7 - 12 %,
//    hence the split.  No new id at all: the distance section is skipped.
#define GBNNS_LOADS_SPEC(O1, O2, O3)                                    \
        "global_load_dwordx4 v[40:43], %[roff], %[db]\n\t"              \
        "global_load_dwordx4 v[44:47], %[roff], %[db] offset:" O1 "\n\t" \
        "global_load_dwordx4 v[48:51], %[roff], %[db] offset:" O2 "\n\t" \
        "global_load_dwordx4 v[52:55], %[roff], %[db] offset:" O3 "\n\t"
#define GBNNS_LOADS_TESTED(O1, O2, O3)                                                                                 \
        "s_lshr_b64 %[act], %[fresh], 1\n\t"                  /* out-of-range reports sit in the odd bits */           \
        "s_or_b64 exec, %[act], %[fresh]\n\t"                                                                          \
        "s_and_b32 exec_lo, exec_lo, 0x55555555\n\t"          /* even lanes with a new id */                           \
        "s_and_b32 exec_hi, exec_hi, 0x55555555\n\t"                                                                   \
        "s_lshl_b64 %[act], exec, 1\n\t"                                                                               \
        "s_or_b64 exec, exec, %[act]\n\t"                     /* ... and their odd partners */                         \
        "s_cbranch_execz 8f\n\t"                                                                                       \
        GBNNS_LOADS_SPEC(O1, O2, O3)                                                                                   \
        "s_mov_b64 exec, %[sv]\n\t"
// ... and both in one block, chosen by the wave-uniform %[spec] (WalkParams::spec_from: the wavefronts of a launch's last,
// partial round request before the test -- they walk a draining machine, where the shorter hop wins and the rows of
// already-visited ids cost nothing)
#define GBNNS_LOADS_DYN_BEFORE(O1, O2, O3)      \
        "s_cmp_eq_u32 %[spec], 0\n\t"           \
        "s_cbranch_scc1 12f\n\t"                \
        GBNNS_LOADS_SPEC(O1, O2, O3)            \
        "12:\n\t"
#define GBNNS_LOADS_DYN_AFTER(O1, O2, O3)       \
        "s_cmp_lg_u32 %[spec], 0\n\t"           \
        "s_cbranch_scc0 10f\n\t"                \
        "s_mov_b64 exec, %[sv]\n\t"             \
        "s_branch 11f\n"                        \
        "10:\n\t"                               \
        GBNNS_LOADS_TESTED(O1, O2, O3)          \
        "11:\n\t"
#define GBNNS_HOT_END "\n8:\n\ts_mov_b64 exec, %[sv]"

#define GBNNS_HOT_CLOBBERS_40                                                                                           \
    "vcc", "scc", "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54",  \
        "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"

template <int METRIC = 0, bool QLDS = false, bool SPEC = !QLDS, bool DYN = false, typename QP>
__device__ __forceinline__ uint32_t hot_expand(const char* db_base, uint32_t roff, uint32_t nb, uint64_t valid, uint32_t lds_base,
                                               uint32_t nbuckets, QP qh, uint32_t qaddr, uint64_t& claimed, uint32_t shr, uint64_t& overflowed,
                                               uint32_t spec = 0u) {
    const uint32_t end = lds_base + (nbuckets << 4);
    uint32_t basev = lds_base, key, mulc;
    uint64_t fresh, act, sv;
#define GBNNS_Q(T) [qa##T] "v"(f32x2{qh[T].x, qh[T].y}), [qb##T] "v"(f32x2{qh[T].z, qh[T].w})
#define GBNNS_HOT_OUT [fresh] "=&s"(fresh), [act] "=&s"(act), [sv] "=&s"(sv), [mulc] "=&s"(mulc), [key] "=&v"(key)
#define GBNNS_HOT_IN [id] "v"(nb), [valid] "s"(valid), [end] "s"(end), [basev] "v"(basev), [shr] "s"(shr), [nb] "s"(nbuckets), \
                     [roff] "v"(roff), [db] "s"(db_base)
    // one statement per (metric, query source, load placement); the pieces are the macros above
// (query in registers: the visited-set block's four temporaries are the compiler's to place, as in rounds 1-3; query in
// LDS: they are v[36:39], which the query pieces take over afterwards)
#define GBNNS_HOT_IN_QREG GBNNS_HOT_IN, GBNNS_Q(0), GBNNS_Q(1), GBNNS_Q(2), GBNNS_Q(3)
#define GBNNS_HOT_IN_QLDS GBNNS_HOT_IN, [qaddr] "v"(qaddr)
#define GBNNS_HOT_IN_QLDS_DYN GBNNS_HOT_IN_QLDS, [spec] "s"(spec)
#define GBNNS_HOT_OUT_QREG GBNNS_HOT_OUT, [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [addr] "=&v"(addr)
#define GBNNS_HOT_OUT_QLDS GBNNS_HOT_OUT
#define GBNNS_VS_QREG GBNNS_VS_ASM("%[t0]", "%[t1]", "%[t2]", "%[addr]")
#define GBNNS_VS_QLDS GBNNS_VS_ASM("v36", "v37", "v38", "v39")
#define GBNNS_CLOB_QREG GBNNS_HOT_CLOBBERS_40
#define GBNNS_CLOB_QLDS "v36", "v37", "v38", "v39", GBNNS_HOT_CLOBBERS_40
#define GBNNS_HOT_STMT(LOADS_BEFORE, VS, AFTER_VS, DIST, OUTS, OPS, CLOB)                      \
    asm volatile("s_mov_b64 %[sv], exec\n\t"                                                   \
                 "s_mov_b64 exec, %[valid]\n\t" LOADS_BEFORE VS AFTER_VS DIST GBNNS_HOT_END      \
                 : OUTS                                                                        \
                 : OPS                                                                         \
                 : CLOB)
#define GBNNS_RESTORE_EXEC "s_mov_b64 exec, %[sv]\n\t"
#define GBNNS_L2_QREG                                                                                                          \
    GBNNS_L2_DIST_ASM("%[qa0]", "%[qb0]", "%[qa1]", "%[qb1]", "%[qa2]", "%[qb2]", "%[qa3]", "%[qb3]", "s_waitcnt vmcnt(2)\n\t", \
                      "s_waitcnt vmcnt(1)\n\t", "s_waitcnt vmcnt(0)\n\t")
#define GBNNS_L2_QLDS                                                                                                          \
    "ds_read_b128 v[36:39], %[qaddr]\n\t"             /* the query pieces that face row loads 0 .. 2 (all lanes); */          \
    "ds_read_b128 v[56:59], %[qaddr] offset:16\n\t"   /* piece 3 follows into v[36:39] once step 0 has used piece 0 */        \
    "ds_read_b128 v[60:63], %[qaddr] offset:32\n\t"                                                                           \
    "s_waitcnt lgkmcnt(2)\n\t"                                                                                                \
    GBNNS_L2_DIST_ASM("v[36:37]", "v[38:39]", "v[56:57]", "v[58:59]", "v[60:61]", "v[62:63]", "v[36:37]", "v[38:39]",          \
                      "ds_read_b128 v[36:39], %[qaddr] offset:48\n\ts_waitcnt vmcnt(2) lgkmcnt(2)\n\t",                        \
                      "s_waitcnt vmcnt(1) lgkmcnt(1)\n\t", "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t")
#define GBNNS_DOT_QREG                                                                                                          \
    GBNNS_DOT_DIST_ASM("%[qa0]", "%[qb0]", "%[qa1]", "%[qb1]", "%[qa2]", "%[qb2]", "%[qa3]", "%[qb3]", "s_waitcnt vmcnt(2)\n\t", \
                       "s_waitcnt vmcnt(1)\n\t", "s_waitcnt vmcnt(0)\n\t")
#define GBNNS_DOT_QLDS                                                                                                         \
    "ds_read_b128 v[36:39], %[qaddr]\n\t"                                                                                     \
    "ds_read_b128 v[56:59], %[qaddr] offset:32\n\t"                                                                           \
    "ds_read_b128 v[60:63], %[qaddr] offset:64\n\t"                                                                           \
    "s_waitcnt lgkmcnt(2)\n\t"                                                                                                \
    GBNNS_DOT_DIST_ASM("v[36:37]", "v[38:39]", "v[56:57]", "v[58:59]", "v[60:61]", "v[62:63]", "v[36:37]", "v[38:39]",         \
                       "ds_read_b128 v[36:39], %[qaddr] offset:96\n\ts_waitcnt vmcnt(2) lgkmcnt(2)\n\t",                       \
                       "s_waitcnt vmcnt(1) lgkmcnt(1)\n\t", "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t")
    uint32_t t0, t1, t2, addr;  // (query-in-registers forms)
    (void)t0; (void)t1; (void)t2; (void)addr;
    if constexpr (DYN) {  // (QLDS instances only)
        static_assert(!DYN || QLDS, "the dynamic order exists for the query-in-LDS instances");
        if constexpr (METRIC == 0) GBNNS_HOT_STMT(GBNNS_LOADS_DYN_BEFORE("16", "32", "48"), GBNNS_VS_QLDS, GBNNS_LOADS_DYN_AFTER("16", "32", "48"), GBNNS_L2_QLDS, GBNNS_HOT_OUT_QLDS, GBNNS_HOT_IN_QLDS_DYN, GBNNS_CLOB_QLDS);
        else GBNNS_HOT_STMT(GBNNS_LOADS_DYN_BEFORE("32", "64", "96"), GBNNS_VS_QLDS, GBNNS_LOADS_DYN_AFTER("32", "64", "96"), GBNNS_DOT_QLDS, GBNNS_HOT_OUT_QLDS, GBNNS_HOT_IN_QLDS_DYN, GBNNS_CLOB_QLDS);
    } else if constexpr (METRIC == 0) {
        if constexpr (!QLDS && SPEC) GBNNS_HOT_STMT(GBNNS_LOADS_SPEC("16", "32", "48"), GBNNS_VS_QREG, GBNNS_RESTORE_EXEC, GBNNS_L2_QREG, GBNNS_HOT_OUT_QREG, GBNNS_HOT_IN_QREG, GBNNS_CLOB_QREG);
        else if constexpr (!QLDS) GBNNS_HOT_STMT("", GBNNS_VS_QREG, GBNNS_LOADS_TESTED("16", "32", "48"), GBNNS_L2_QREG, GBNNS_HOT_OUT_QREG, GBNNS_HOT_IN_QREG, GBNNS_CLOB_QREG);
        else if constexpr (SPEC) GBNNS_HOT_STMT(GBNNS_LOADS_SPEC("16", "32", "48"), GBNNS_VS_QLDS, GBNNS_RESTORE_EXEC, GBNNS_L2_QLDS, GBNNS_HOT_OUT_QLDS, GBNNS_HOT_IN_QLDS, GBNNS_CLOB_QLDS);
        else GBNNS_HOT_STMT("", GBNNS_VS_QLDS, GBNNS_LOADS_TESTED("16", "32", "48"), GBNNS_L2_QLDS, GBNNS_HOT_OUT_QLDS, GBNNS_HOT_IN_QLDS, GBNNS_CLOB_QLDS);
    } else {
        if constexpr (!QLDS && SPEC) GBNNS_HOT_STMT(GBNNS_LOADS_SPEC("32", "64", "96"), GBNNS_VS_QREG, GBNNS_RESTORE_EXEC, GBNNS_DOT_QREG, GBNNS_HOT_OUT_QREG, GBNNS_HOT_IN_QREG, GBNNS_CLOB_QREG);
        else if constexpr (!QLDS) GBNNS_HOT_STMT("", GBNNS_VS_QREG, GBNNS_LOADS_TESTED("32", "64", "96"), GBNNS_DOT_QREG, GBNNS_HOT_OUT_QREG, GBNNS_HOT_IN_QREG, GBNNS_CLOB_QREG);
        else if constexpr (SPEC) GBNNS_HOT_STMT(GBNNS_LOADS_SPEC("32", "64", "96"), GBNNS_VS_QLDS, GBNNS_RESTORE_EXEC, GBNNS_DOT_QLDS, GBNNS_HOT_OUT_QLDS, GBNNS_HOT_IN_QLDS, GBNNS_CLOB_QLDS);
        else GBNNS_HOT_STMT("", GBNNS_VS_QLDS, GBNNS_LOADS_TESTED("32", "64", "96"), GBNNS_DOT_QLDS, GBNNS_HOT_OUT_QLDS, GBNNS_HOT_IN_QLDS, GBNNS_CLOB_QLDS);
    }
#undef GBNNS_HOT_STMT
#undef GBNNS_HOT_IN_QREG
#undef GBNNS_HOT_IN_QLDS
#undef GBNNS_HOT_IN_QLDS_DYN
#undef GBNNS_CLOB_QLDS
#undef GBNNS_CLOB_QREG
#undef GBNNS_VS_QLDS
#undef GBNNS_VS_QREG
#undef GBNNS_HOT_OUT_QLDS
#undef GBNNS_HOT_OUT_QREG
#undef GBNNS_Q
#undef GBNNS_HOT_OUT
#undef GBNNS_HOT_IN
    claimed = fresh & 0x5555555555555555ull;
    overflowed = fresh & 0xAAAAAAAAAAAAAAAAull;
    return key;
}

// R = list registers per lane = ceil(ef / 64): 1 (every block hand-laid-out) .. 8 (ef <= 512): the same hop
// -- one-block expansion, packed visited set, both prefetches -- around the generic selection and merge of
// multi-register lists.
// WIDE: adjacency rows of 33 .. 64 slots (the level-0 lists of hnswlib M = 18 / 20 graphs, prepare_graph.cpp's M = 30):
// the same hop with a second expansion pass over slots 32 .. 63 when the node has that many neighbours.
template <int R, bool WIDE = false, int METRIC = 0, bool SPEC1 = GBNNS_HOT1_SPEC != 0>
__device__ __forceinline__ void walk_hot_one(const WalkParams& p, uint32_t qi, unsigned char* smem) {
    const int lane = lane_id();
    const uint32_t slot = (uint32_t)lane >> 1, half = (uint32_t)lane & 1u;  // lane = 2 * adjacency slot + row half
    const int ef = p.ef;
#ifdef GBNNS_LIFE_STAMPS   // diagnostic (EXTRA_DEFS=-DGBNNS_LIFE_STAMPS, tools/life_stamps.py): a wavefront's life in three parts, 100 MHz ticks
    const unsigned long long life0 = __builtin_amdgcn_s_memrealtime();
#endif
    // R = 1 (QLDS): the query stays in LDS and hot_expand re-reads a lane's four pieces every hop -- 64 registers, 8 wavefronts
    // per SIMD; rows requested after the visited test.  R = 2: the query in registers, speculative row loads (rounds 1-3 layout).
    constexpr bool QLDS = R == 1 && GBNNS_HOT1_QLDS, SPEC = R == 1 ? SPEC1 : true;
    // LDS: [tie list 128 B][merge buffer 528 B (R = 2: 1 040 B; its head stages the query until it is in registers)]
    //      [QLDS: the query, 128 B][visited set]
    uint64_t* tie = reinterpret_cast<uint64_t*>(smem);
    uint64_t* stage = tie + kRegTieCap;
    float* qf = reinterpret_cast<float*>(QLDS ? stage + reg_stage_slots(R) : stage);
    uint32_t* hash = reinterpret_cast<uint32_t*>(stage + reg_stage_slots(R)) + (QLDS ? 32 : 0);
    const float4* qs = reinterpret_cast<const float4*>(qf);
    const uint32_t cap = p.hash_cap;
    const uint32_t hash_lds = (uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(hash));

    // visited set (hot_expand, GBNNS_VS_ASM): cap / 5 buckets of 16 bytes with five 24-bit ids + a counter byte each,
    // or (p.vs_shr != 0) cap / 7 buckets with seven 16-bit quotient entries + a counter halfword
    const uint32_t vs_shr = p.vs_shr;
    const uint32_t nbuckets = vs_shr ? cap / 7u - kStashBuckets : cap / 5u;
    if (vs_shr) quotient_table_init(hash, nbuckets, lane);
    else packed_table_init(hash, nbuckets, 0u, lane);
    if (lane < 32) qf[lane] = p.q[(size_t)qi * p.qstride + lane];
    wave_sync();
    // this lane's half of the query: 64 contiguous bytes (L2) / the even or odd 16-byte pieces (dot): the LDS byte address
    // of its four pieces for hot_expand
    RowRegs<4> qreg;  // (R = 2: the pieces in registers)
    if constexpr (!QLDS) {
#pragma unroll
        for (int t = 0; t < 4; ++t) qreg.v[t] = METRIC == 0 ? qs[4 * half + t] : qs[2 * t + half];
    }
    const uint32_t qaddr = (uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(qf)) +
                           half * (METRIC == 0 ? 64u : 16u);

    RegList<R> L;
    L.clear();
    int size = 1, tsize = 0, hops = 0, dist_calc = 1, edges = 0;
    uint32_t worst;
    const uint32_t entry = p.entries ? p.entries[qi] : 0u;
    if (entry >= p.n) { write_bad_entry(p, qi, lane); return; }
    {
        const float d0 = walk_dist<METRIC, 8>(qs, row_ptr<true>(p.db, entry, 32u), 32u);
        worst = fkey(d0);
        if (lane == 0) {
            L.hi[0] = worst;
            L.lo[0] = entry << 1;
            if (vs_shr) quotient_table_put_first(hash, nbuckets, entry, vs_shr);
            else packed_table_put_first(hash, nbuckets, entry);
        }
        wave_sync();
    }

    const uint64_t lmask = RegList<R>::lane_mask(0, ef);
    const uint32_t ell_row_bytes = p.ell_stride * 4u;
    const bool slot_ok = slot < p.ell_stride;             // ell_stride is 16 or 32 here (WIDE: 48 or 64)
    const uint32_t slot_off = slot_ok ? slot * 4u : 0u;    // lanes beyond the row read slot 0 and are masked
    const bool slotw_ok = WIDE && slot + 32u < p.ell_stride;  // second pass: slots 32 .. 63
    const uint32_t slotw_off = slotw_ok ? (slot + 32u) * 4u : 0u;
    const char* ell_base = reinterpret_cast<const char*>(p.ell);
    const char* db_base = reinterpret_cast<const char*>(p.db);
    const uint32_t dc_limit = p.hash_limit >= 32u ? p.hash_limit - 32u : 0u;  // at most 32 new ids per pass
    bool handed_over = false;
    const uint32_t spec_tail = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x >= p.spec_from));  // wave-uniform
    uint32_t pf_node = kInvalidId, pf_val = kInvalidId;    // prefetch 1: the runner-up of the selection
    uint32_t pf2_node = kInvalidId, pf2_val = kInvalidId;  // prefetch 2: the closest new survivor (see below)
    uint32_t pf_valw = kInvalidId, pf2_valw = kInvalidId;  // WIDE: the rows' second halves
#ifdef GBNNS_LIFE_STAMPS
    const unsigned long long life1 = __builtin_amdgcn_s_memrealtime();
#endif

    while (true) {
        // ---- next node: closest unexpanded entry (ties -> largest id), and the runner-up as prediction
        uint32_t node, pred, ok, h2;
        if constexpr (R == 1) {
            uint64_t fm;
            uint32_t t0, q1, q2, h1;
            asm volatile(
                "v_and_b32 %[t0], 1, %[lo]\n\t"
                "v_cmp_eq_u32 vcc, 0, %[t0]\n\t"
                "s_and_b64 %[fm], vcc, %[lmask]\n\t"          // unexpanded list entries
                "s_cmp_eq_u32 %[tsize], 0\n\t"
                "s_cselect_b64 %[fm], %[fm], 0\n\t"            // a non-empty tie list -> slow path
                "s_ff1_i32_b64 %[q1], %[fm]\n\t"               // -1 when nothing is left
                "s_bitset0_b64 %[fm], %[q1]\n\t"
                "s_ff1_i32_b64 %[q2], %[fm]\n\t"               // runner-up, -1 when there is none
                "v_readlane_b32 %[h1], %[hi], %[q1]\n\t"
                "v_readlane_b32 %[node], %[lo], %[q1]\n\t"
                "v_readlane_b32 %[h2], %[hi], %[q2]\n\t"
                "v_readlane_b32 %[pred], %[lo], %[q2]\n\t"
                "s_lshr_b32 %[node], %[node], 1\n\t"
                "s_lshr_b32 %[pred], %[pred], 1\n\t"
                "s_cmp_lg_u32 %[h1], %[h2]\n\t"
                "s_cselect_b32 %[ok], 1, 0\n\t"                 // distinct distances: plain pick
                "s_cmp_lt_i32 %[q2], 0\n\t"
                "s_cselect_b32 %[ok], 1, %[ok]\n\t"             // no runner-up: plain pick, no prediction
                "s_cselect_b32 %[pred], -1, %[pred]\n\t"
                "s_cselect_b32 %[h2], -1, %[h2]\n\t"            // runner-up's distance key (all-ones: none)
                "s_cmp_lt_i32 %[q1], 0\n\t"
                "s_cselect_b32 %[ok], 0, %[ok]\n\t"             // nothing left (or tie list in play)
                "s_cmp_lg_u32 %[ok], 0\n\t"
                "s_cselect_b32 %[q1], %[q1], -1\n\t"
                "v_cmp_eq_u32 vcc, %[q1], %[lane]\n\t"          // mark the picked entry expanded
                "v_cndmask_b32 %[t0], 0, 1, vcc\n\t"
                "v_or_b32 %[lo], %[lo], %[t0]"
                : [lo] "+v"(L.lo[0]), [fm] "=&s"(fm), [t0] "=&v"(t0), [q1] "=&s"(q1), [q2] "=&s"(q2), [h1] "=&s"(h1),
                  [h2] "=&s"(h2), [node] "=&s"(node), [pred] "=&s"(pred), [ok] "=&s"(ok)
                : [hi] "v"(L.hi[0]), [lmask] "s"(lmask), [tsize] "s"(tsize), [lane] "v"(lane)
                : "vcc", "scc");
            if (__builtin_expect(ok == 0, 0)) {
                const uint64_t mu = __ballot(!(L.lo[0] & 1u)) & lmask;
                if (!reg1_select_slow(L, mu, tsize, tie, worst, lmask, lane, node)) break;
                pred = kInvalidId;
                h2 = 0xFFFFFFFFu;
            }
        } else {
            // the two closest unexpanded entries across the R registers (ranks p1 < p2)
            int p1 = -1, p2 = -1;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                uint64_t mm = __ballot(!(L.lo[r] & 1u)) & RegList<R>::lane_mask(r, ef);
                if (p1 < 0 && mm) {
                    p1 = r * 64 + __ffsll((unsigned long long)mm) - 1;
                    mm &= mm - 1;
                }
                if (p1 >= 0 && p2 < 0 && mm) p2 = r * 64 + __ffsll((unsigned long long)mm) - 1;
            }
            ok = 0; node = 0; pred = kInvalidId; h2 = 0xFFFFFFFFu;
            if (p1 >= 0 && tsize == 0) {
                if (p2 >= 0) {
                    const uint32_t hp2 = L.hi_at(p2);
                    if (L.hi_at(p1) != hp2) {  // distinct distances: plain pick, the runner-up is the prediction
                        ok = 1;
                        node = L.lo_at(p1) >> 1;
                        pred = L.lo_at(p2) >> 1;
                        h2 = hp2;
                        L.mark_expanded(p1, lane);
                    }
                } else {
                    ok = 1;
                    node = L.lo_at(p1) >> 1;
                    L.mark_expanded(p1, lane);
                }
            }
            if (__builtin_expect(ok == 0, 0)) {
                if (!regN_select_slow<R>(L, p1, tsize, tie, worst, ef, lane, node)) break;
            }
        }

        // ---- adjacency row of `node` (prefetched, or loaded now), then the prefetch for the next hop
        uint32_t nb, nbw = kInvalidId;
        if (node == pf_node) {
            nb = pf_val;
            if constexpr (WIDE) nbw = pf_valw;
        } else if (node == pf2_node) {
            nb = pf2_val;
            if constexpr (WIDE) nbw = pf2_valw;
        } else {
            nb = *reinterpret_cast<const uint32_t*>(ell_base + node * ell_row_bytes + slot_off);
            if constexpr (WIDE) nbw = *reinterpret_cast<const uint32_t*>(ell_base + node * ell_row_bytes + slotw_off);
        }
        nb = slot_ok ? nb : kInvalidId;
        if constexpr (WIDE) nbw = slotw_ok ? nbw : kInvalidId;
        pf2_node = kInvalidId;
        // (the ballots come before the prefetch loads below: their wait must cover this row only)
        const uint64_t mv0 = __ballot(nb != kInvalidId);
        const uint64_t mv1 = WIDE ? __ballot(nbw != kInvalidId) : 0ull;
        pf_node = pred;
        if (pred != kInvalidId) {
            pf_val = *reinterpret_cast<const uint32_t*>(ell_base + pred * ell_row_bytes + slot_off);
            if constexpr (WIDE) pf_valw = *reinterpret_cast<const uint32_t*>(ell_base + pred * ell_row_bytes + slotw_off);
        }
        // one expansion pass over <= 32 adjacency slots (lane = 2 * slot + row half); false = the query is handed over
        auto expand_pass = [&](const uint32_t nb, const uint64_t mv) -> bool {
            if (__builtin_expect(mv == 0, 0)) return true;
            if (__builtin_expect((uint32_t)dist_calc > dc_limit, 0)) return false;
            edges += __popcll(mv & 0x5555555555555555ull);
            // ---- gather (speculative: before the visited test), visited test, distances -----------
            uint64_t mclaimed, movf;
            // (R = 1, tested-first: the order is a run-time choice per wavefront, WalkParams::spec_from)
            const uint32_t kd = hot_expand<METRIC, QLDS, SPEC, (QLDS && !SPEC)>(db_base, (nb << 7) + half * (METRIC == 0 ? 64u : 16u), nb, mv, hash_lds, nbuckets, qreg.v, qaddr, mclaimed, vs_shr, movf, spec_tail);
            if (__builtin_expect(movf != 0, 0)) {  // a probe sequence ran out (quotient form): the stash takes the id
                // (the two-pass instances too since the second half of round 5: see walk_hot_big)
                if (!stash_claim(hash_lds, nbuckets, movf >> 1, nb, mclaimed, lane)) return false;  // (reported in the odd bits)
            }
            const uint64_t mfresh = mclaimed << 1;  // odd lanes hold the distances
            const uint32_t dk = __builtin_amdgcn_inverse_ballot_w64(mfresh) ? kd : 0xFFFFFFFFu;
            dist_calc += __popcll(mfresh);
            uint64_t m = size < ef ? mfresh : __ballot(dk < worst);
            // ---- survivors into the result list: batch merge, or one by one (reference order) --------
            if (m != 0) {
                // Prefetch 2: a survivor closer than the runner-up will be the next node (it becomes the
                // closest unexpanded entry); request its adjacency row now, before the merge's scatter and the next
                // selection, instead of after them.  Only when it is unique (ties go the slow way).
                auto prefetch2 = [&](const uint32_t dmin, const uint64_t me) {  // me = survivors at distance dmin
                    if (dmin < h2 && me != 0 && (me & (me - 1)) == 0) {
                        pf2_node = readlane_u32(nb, __ffsll((unsigned long long)me) - 1);
                        pf2_val = *reinterpret_cast<const uint32_t*>(ell_base + pf2_node * ell_row_bytes + slot_off);
                        if constexpr (WIDE) {
                            pf2_valw = *reinterpret_cast<const uint32_t*>(ell_base + pf2_node * ell_row_bytes + slotw_off);
                            h2 = dmin;  // the second pass overrides the prediction only with something closer still
                        }
                    }
                };
                bool merged = false;
                if constexpr (R == 1 && GBNNS_HOT1_PF2_IN_MERGE) {
                    // (the closest survivor comes out of the merge's rank loop -- one scalar minimum per survivor -- or, for
                    // a lone survivor, out of its lane)
                    if ((m & (m - 1)) != 0) {
                        merged = reg_merge_cb<true>(m, __builtin_amdgcn_inverse_ballot_w64(m), dk, nb, L, size, worst, tsize, stage, ef, lane,
                                                    [&](const uint32_t dmin) { if (dmin < h2) prefetch2(dmin, __ballot(dk == dmin) & m); });
                    } else {
                        prefetch2(readlane_u32(dk, __ffsll((unsigned long long)m) - 1), m);
                    }
                } else {
                    uint32_t x = dk;  // all-ones outside the new ids; survivors are below `worst`
                    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xf, 0xf, false));   // quad_perm 1,0,3,2
                    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xf, 0xf, false));   // quad_perm 2,3,0,1
                    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xf, 0xf, false));  // row_half_mirror
                    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xf, 0xf, false));  // row_mirror
                    const uint32_t dmin = min(min(readlane_u32(x, 0), readlane_u32(x, 16)), min(readlane_u32(x, 32), readlane_u32(x, 48)));
                    if (dmin < h2) prefetch2(dmin, __ballot(dk == dmin) & m);
                    if ((m & (m - 1)) != 0) {
                        if constexpr (R == 1) merged = reg_merge(m, __builtin_amdgcn_inverse_ballot_w64(m), dk, nb, L, size, worst, tsize, stage, ef, lane);
                        else merged = reg_merge_multi<R>(m, __builtin_amdgcn_inverse_ballot_w64(m), dk, nb, L, size, worst, tsize, stage, ef, lane);
                    }
                }
                if (!merged) {
                    do {
                        const int l = __ffsll((unsigned long long)m) - 1;
                        m &= m - 1;
                        if (!reg_offer<R>(readlane_u32(dk, l), readlane_u32(nb, l) << 1, L, size, worst, tsize, tie, ef, lane)) return false;
                    } while (m);
                }
            }
            return true;
        };
        if (!expand_pass(nb, mv0)) { handed_over = true; break; }
        if constexpr (WIDE) {
            if (!expand_pass(nbw, mv1)) { handed_over = true; break; }
        }
        hops += 1;
    }

    if (handed_over) {
        if (lane == 0) {
            const uint32_t s = atomicAdd(p.ovf_count, 1u);
            p.ovf_list[s] = qi;
        }
        return;
    }
#ifdef GBNNS_LIFE_STAMPS
    const unsigned long long life2 = __builtin_amdgcn_s_memrealtime();
#endif
    reg_write_results<R>(p, qi, L, size, hops, dist_calc, edges, lane);
    if (p.rr_db) {
        const int kept = size < p.k ? size : p.k;
        fused_rerank(p, qi, kept, smem, lane, [&](int rank) { return reg_id_at_rank<R>(L, rank); });
    }
#ifdef GBNNS_LIFE_STAMPS
    {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long life3 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && p.stamps) {
            atomicAdd(p.stamps + 0, life1 - life0); atomicAdd(p.stamps + 1, life2 - life1); atomicAdd(p.stamps + 2, life3 - life2);
            atomicAdd(p.stamps + 3, 1ull);
            atomicMax(p.stamps + 4, life3 - life0);
            atomicMax(p.stamps + 5, (1ull << 62) - life0); atomicMax(p.stamps + 6, life0); atomicMax(p.stamps + 7, life3);
        }
    }
#endif
}

template <bool WIDE = false, int METRIC = 0>
__device__ __forceinline__ void walk_hot_big(const WalkParams& p, uint32_t qi, unsigned char* smem) {
    const int lane = lane_id();
#ifdef GBNNS_STAMPS  // diagnostic build: cycles per segment of the hop (tools/stamps.py)
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned st_pf1 = 0, st_pf2 = 0;  // hops whose node was the runner-up prediction / the closest new survivor
    STAMP(t_begin)
    unsigned long long t_prev = t_begin;
#endif
    const uint32_t slot = (uint32_t)lane >> 1, half = (uint32_t)lane & 1u;  // lane = 2 * adjacency slot + row half
    const int ef = p.ef;
    BigList B;
    B.init(smem, ef);
    float* qf = reinterpret_cast<float*>(B.stage);  // the query is staged here until it sits in registers
    unsigned char* hash_bytes = smem + big_list_fixed_bytes(ef);
    uint32_t* hash = reinterpret_cast<uint32_t*>(hash_bytes);
    const float4* qs = reinterpret_cast<const float4*>(qf);
    const uint32_t cap = p.hash_cap;
    const uint32_t hash_lds = (uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(hash));
    const uint32_t vs_shr = p.vs_shr;  // (walk_hot_one: the two forms of the table)
    const uint32_t nbuckets = vs_shr ? cap / 7u - kStashBuckets : cap / 5u;
    if (vs_shr) quotient_table_init(hash, nbuckets, lane);
    else packed_table_init(hash, nbuckets, 0u, lane);
    if (lane < 32) qf[lane] = p.q[(size_t)qi * p.qstride + lane];
    wave_sync();
    RowRegs<4> qreg;
#pragma unroll
    for (int t = 0; t < 4; ++t) qreg.v[t] = METRIC == 0 ? qs[4 * half + t] : qs[2 * t + half];

    int hops = 0, dist_calc = 1, edges = 0;
    // (every lane computes the same entry id and distance; readfirstlane tells the compiler they are wave-uniform --
    // otherwise every piece of list state that is ever merged with them is kept in vector registers and the scalar
    // control flow of the list turns into exec-masked regions)
    const uint32_t entry = (uint32_t)__builtin_amdgcn_readfirstlane((int)(p.entries ? p.entries[qi] : 0u));
    if (entry >= p.n) { write_bad_entry(p, qi, lane); return; }
    {
        const float d0 = walk_dist<METRIC, 8>(qs, row_ptr<true>(p.db, entry, 32u), 32u);
        B.worst = B.fworst = (uint32_t)__builtin_amdgcn_readfirstlane((int)fkey(d0));
        if (lane == 0) {
            B.F.hi[0] = B.worst;
            B.F.lo[0] = entry << 1;
            if (vs_shr) quotient_table_put_first(hash, nbuckets, entry, vs_shr);
            else packed_table_put_first(hash, nbuckets, entry);
        }
        wave_sync();
    }

    const uint32_t ell_row_bytes = p.ell_stride * 4u;
    const bool slot_ok = slot < p.ell_stride;
    const uint32_t slot_off = slot_ok ? slot * 4u : 0u;
    const bool slotw_ok = WIDE && slot + 32u < p.ell_stride;  // second pass: slots 32 .. 63 (see walk_hot_one)
    const uint32_t slotw_off = slotw_ok ? (slot + 32u) * 4u : 0u;
    const char* ell_base = reinterpret_cast<const char*>(p.ell);
    const char* db_base = reinterpret_cast<const char*>(p.db);
    const uint32_t dc_limit = p.hash_limit >= 32u ? p.hash_limit - 32u : 0u;
    bool handed_over = false;
    uint32_t pf_node = kInvalidId, pf_val = kInvalidId, pf2_node = kInvalidId, pf2_val = kInvalidId;
    uint32_t pf_valw = kInvalidId, pf2_valw = kInvalidId;

    while (true) {
        uint32_t node, pred, h2;
        STAMP(t0)
        STAMP_ADD(7, t_prev, t0)
        if (!B.select(node, pred, h2, lane)) break;
        STAMP(t1)
        STAMP_ADD(0, t0, t1)

        // ---- adjacency row of `node` (prefetched, or loaded now), then the prefetch for the next hop
        uint32_t nb, nbw = kInvalidId;
#ifdef GBNNS_STAMPS
        if (node == pf_node) st_pf1 += 1;
        else if (node == pf2_node) st_pf2 += 1;
#endif
        if (node == pf_node) {
            nb = pf_val;
            if constexpr (WIDE) nbw = pf_valw;
        } else if (node == pf2_node) {
            nb = pf2_val;
            if constexpr (WIDE) nbw = pf2_valw;
        } else {
            nb = *reinterpret_cast<const uint32_t*>(ell_base + node * ell_row_bytes + slot_off);
            if constexpr (WIDE) nbw = *reinterpret_cast<const uint32_t*>(ell_base + node * ell_row_bytes + slotw_off);
        }
        nb = slot_ok ? nb : kInvalidId;
        if constexpr (WIDE) nbw = slotw_ok ? nbw : kInvalidId;
        pf2_node = kInvalidId;
        const uint64_t mv0 = __ballot(nb != kInvalidId);
        const uint64_t mv1 = WIDE ? __ballot(nbw != kInvalidId) : 0ull;
        STAMP(t2)
        STAMP_ADD(1, t1, t2)
        pf_node = pred;
        if (pred != kInvalidId) {
            pf_val = *reinterpret_cast<const uint32_t*>(ell_base + pred * ell_row_bytes + slot_off);
            if constexpr (WIDE) pf_valw = *reinterpret_cast<const uint32_t*>(ell_base + pred * ell_row_bytes + slotw_off);
        }
        // one expansion pass over <= 32 adjacency slots; false = the query is handed over
        auto expand_pass = [&](const uint32_t nb, const uint64_t mv) -> bool {
            if (__builtin_expect(mv == 0, 0)) return true;
            if (__builtin_expect((uint32_t)dist_calc > dc_limit, 0)) return false;
            edges += __popcll(mv & 0x5555555555555555ull);
            STAMP(t3)
            STAMP_ADD(2, t2, t3)
            uint64_t mclaimed, movf;
            const uint32_t kd = hot_expand<METRIC, false, true>(db_base, (nb << 7) + half * (METRIC == 0 ? 64u : 16u), nb, mv, hash_lds, nbuckets, qreg.v, 0u, mclaimed, vs_shr, movf);
            if (__builtin_expect(movf != 0, 0)) {  // a probe sequence ran out (quotient form): the stash takes the id
                // (second half of round 5: the two-pass instance too -- it used to hand the whole query over here, which made it
                // fall off a cliff as the table filled: GD(M = 30) graph, ef 180, first-pass kernel 2.33 ms at a fill of 0.86, 11.6 at
                // 0.95; this kernel's wavefronts per CU are bounded by LDS, not by its scalar registers)
                if (!stash_claim(hash_lds, nbuckets, movf >> 1, nb, mclaimed, lane)) return false;  // (reported in the odd bits)
            }
            const uint64_t mfresh = mclaimed << 1;  // odd lanes hold the distances
            const uint32_t dk = __builtin_amdgcn_inverse_ballot_w64(mfresh) ? kd : 0xFFFFFFFFu;
            dist_calc += __popcll(mfresh);
            const uint64_t m = (B.l + B.f < ef) ? mfresh : __ballot(dk < B.worst);
            STAMP(t5)
            STAMP_ADD(4, t3, t5)
            if (m != 0) {
                // prefetch 2: a unique survivor closer than the runner-up is the next node
                {
                    uint32_t x = dk;
                    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xf, 0xf, false));
                    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xf, 0xf, false));
                    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xf, 0xf, false));
                    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xf, 0xf, false));
                    const uint32_t dmin = min(min(readlane_u32(x, 0), readlane_u32(x, 16)), min(readlane_u32(x, 32), readlane_u32(x, 48)));
                    if (dmin < h2) {
                        const uint64_t me = __ballot(dk == dmin) & m;
                        if (me != 0 && (me & (me - 1)) == 0) {
                            pf2_node = readlane_u32(nb, __ffsll((unsigned long long)me) - 1);
                            pf2_val = *reinterpret_cast<const uint32_t*>(ell_base + pf2_node * ell_row_bytes + slot_off);
                            if constexpr (WIDE) {
                                pf2_valw = *reinterpret_cast<const uint32_t*>(ell_base + pf2_node * ell_row_bytes + slotw_off);
                                h2 = dmin;
                            }
                        }
                    }
                }
                if (!B.insert(m, dk, nb, lane)) return false;
            }
            STAMP(t6)
            STAMP_ADD(5, t5, t6)
#ifdef GBNNS_STAMPS
            t_prev = t6;
#endif
            return true;
        };
        if (!expand_pass(nb, mv0)) { handed_over = true; break; }
        if constexpr (WIDE) {
            if (!expand_pass(nbw, mv1)) { handed_over = true; break; }
        }
        hops += 1;
    }

#ifdef GBNNS_STAMPS
    {
        STAMP(t_end)
        seg[6] = t_end - t_begin;
        if (lane == 0 && p.stamps) {
            for (int i = 0; i < 7; ++i) atomicAdd(p.stamps + i, seg[i]);
            atomicAdd(p.stamps + 30, seg[7]);
            atomicAdd(p.stamps + 21, B.st_flush); atomicAdd(p.stamps + 22, B.st_refresh); atomicAdd(p.stamps + 23, B.st_evict);
            atomicAdd(p.stamps + 24, (unsigned long long)B.st_nflush); atomicAdd(p.stamps + 25, (unsigned long long)B.st_nrefresh);
            atomicAdd(p.stamps + 26, (unsigned long long)B.st_nbase); atomicAdd(p.stamps + 27, (unsigned long long)B.st_nseq);
            atomicAdd(p.stamps + 28, (unsigned long long)B.st_ninsert); atomicAdd(p.stamps + 29, (unsigned long long)B.st_slow);
            atomicAdd(p.stamps + 7, (unsigned long long)st_pf1); atomicAdd(p.stamps + 31, (unsigned long long)st_pf2);
        }
    }
#endif
    if (handed_over) {
        if (lane == 0) {
            const uint32_t s = atomicAdd(p.ovf_count, 1u);
            p.ovf_list[s] = qi;
        }
        return;
    }
    B.finish(p, qi, hops, dist_calc, edges, hash_bytes, lane);  // (re-rank query staged in the dead visited-set area)
}

__global__ __launch_bounds__(64) void walk_hot2_kernel(WalkParams p) {  // 64 < ef <= 128: two list registers per lane
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    walk_hot_one<2>(p, walk_query_of(p, blockIdx.x), smem);
}

__global__ __launch_bounds__(64) void walk_hot_big_kernel(WalkParams p) {  // 128 < ef <= 1024
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    walk_hot_big(p, walk_query_of(p, blockIdx.x), smem);
}

// (second launch bound = wavefronts per SIMD the register allocation must leave room for: 8 = 64 registers, which the hop needs
// anyway since round 4 -- hot_expand, QLDS; the bound only keeps the prologue's entry distance from taking more)
__global__ __launch_bounds__(64, GBNNS_HOT1_LB) void walk_hot_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef GBNNS_HOT1_CAP6  // experiment: 97+ scalar registers = 6 wavefronts per SIMD, leaving 128 vector registers per SIMD to other kernels
    asm volatile("" ::: "s96");
#endif
    walk_hot_one<1>(p, walk_query_of(p, blockIdx.x), smem);
}

// ... with the rows requested BEFORE the visited test (the rounds 1-3 order): launches of many rounds of wavefronts -- the
// DEEP10M-shaped 1 M-query batch in locality order -- are 6 % faster this way (21.7 against 23.1 ms: the hop is one LDS round
// trip shorter and the extra rows mostly hit the L2), the 10 000-query launch 1 - 2 % slower (profiles/r04_ab.txt)
__global__ __launch_bounds__(64, GBNNS_HOT1_LB) void walk_hot_spec_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    walk_hot_one<1, false, 0, true>(p, walk_query_of(p, blockIdx.x), smem);
}

// the same three for adjacency rows of 33 .. 64 slots (two expansion passes per hop)
__global__ __launch_bounds__(64, GBNNS_HOT1_LB) void walk_hotw_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    walk_hot_one<1, true>(p, walk_query_of(p, blockIdx.x), smem);
}

__global__ __launch_bounds__(64) void walk_hotw2_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    walk_hot_one<2, true>(p, walk_query_of(p, blockIdx.x), smem);
}

__global__ __launch_bounds__(64) void walk_hotw_big_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    walk_hot_big<true>(p, walk_query_of(p, blockIdx.x), smem);
}

// ... and the negative-dot metric (Angular::Dist) on the same shapes (round 3): R = 1 / 2 list registers, or the two-list form
template <int R, bool WIDE>
__global__ __launch_bounds__(64, R == 1 && GBNNS_HOT1_QLDS ? 8 : 1) void walk_hot_dot_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    walk_hot_one<R, WIDE, 1>(p, walk_query_of(p, blockIdx.x), smem);
}

template <bool WIDE>
__global__ __launch_bounds__(64) void walk_hot_dot_big_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    walk_hot_big<WIDE, 1>(p, walk_query_of(p, blockIdx.x), smem);
}

}  // namespace

hipError_t launch_walk_hot(const WalkParams& p, int metric, hipStream_t s) {
    const size_t lds = walk_fast_lds_bytes(p, true);
    const bool wide = p.ell_stride > 32u;  // adjacency rows of 33 .. 64 slots: the two-pass instances
    if (p.ef <= 64) {
        if (metric == 1) return wide ? launch_walk_k(walk_hot_dot_kernel<1, true>, p, false, lds, s) : launch_walk_k(walk_hot_dot_kernel<1, false>, p, false, lds, s);
        if (!wide && p.spec_rows && !GBNNS_HOT1_SPEC) return launch_walk_k(walk_hot_spec_kernel, p, false, lds, s);
        return wide ? launch_walk_k(walk_hotw_kernel, p, false, lds, s) : launch_walk_k(walk_hot_kernel, p, false, lds, s);
    }
    if (p.ef <= kHot2MaxEf) {
        if (metric == 1) return wide ? launch_walk_k(walk_hot_dot_kernel<2, true>, p, false, lds, s) : launch_walk_k(walk_hot_dot_kernel<2, false>, p, false, lds, s);
        return wide ? launch_walk_k(walk_hotw2_kernel, p, false, lds, s) : launch_walk_k(walk_hot2_kernel, p, false, lds, s);
    }
    if (metric == 1) return wide ? launch_walk_k(walk_hot_dot_big_kernel<true>, p, false, lds, s) : launch_walk_k(walk_hot_dot_big_kernel<false>, p, false, lds, s);
    return wide ? launch_walk_k(walk_hotw_big_kernel, p, false, lds, s) : launch_walk_k(walk_hot_big_kernel, p, false, lds, s);
}

}  // namespace gbnns
