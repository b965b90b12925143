// walk_common.h -- device code shared by every walk / re-rank kernel of the two-stage graph search (gfx950): small helpers,
// the reference-order distances, the exact LDS visited sets (4-byte slots, packed 24-bit ids, 16-bit quotient form), the
// pair-form re-rank core and the fused re-rank, the sorted result list, tie lists, selection / offer / result writers.
// (Until round 4 all of this was one 4 900-line kernels.hip compiled three times.)
//
// Arithmetic contract: every distance / dot product is evaluated in IEEE binary32 with the operation
// order of the reference's SSE/AVX source (support_func.h:107-163): separate mul and add (compiled with
// -ffp-contract=off), 4 resp. 8 independent running sums, the reference's horizontal-sum order, correctly rounded sqrt
// and divide.  Wavefront = 64 lanes; all walk / re-rank kernels use one 64-thread workgroup (= one wavefront) per query,
// so cross-lane traffic goes through ballots / shuffles / DPP and wave-private LDS.
#pragma once

#include "kernels.h"

namespace gbnns {

namespace {

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Query that work item `b` of a first pass runs: b itself, or -- deep batches (WalkParams::order) -- the b-th query in
// locality order, so that the wavefronts resident together walk neighbouring regions of the graph and find each
// other's rows in the L2 / Infinity Cache.  Queries are independent: the order changes nothing but the time.
template <typename P>
__device__ __forceinline__ uint32_t walk_query_of(const P& p, uint32_t b) {
    return p.order ? (uint32_t)__builtin_amdgcn_readfirstlane((int)p.order[b]) : b;
}

// One wavefront per workgroup: the barrier degenerates to a wave-local fence that orders LDS /
// global traffic between lanes of the wave.
// GBNNS_WAVE_LOCAL_SYNC (walk_coop.hip: workgroups of TWO wavefronts, each running its own code path around explicit
// s_barriers): a real barrier here would pair up with the other wavefront's unrelated ones, so the fence is spelled out --
// LDS operations of one wavefront are performed in issue order, what is needed is that the compiler keeps that order and
// that outstanding LDS / scalar traffic has landed before the next access.
#ifdef GBNNS_WAVE_LOCAL_SYNC
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}
#else
__device__ __forceinline__ void wave_sync() { __syncthreads(); }
#endif

// Monotone float -> u32 map (a < b  <=>  fkey(a) < fkey(b)); -0 and +0 map to the same key, as
// they compare equal in the reference's std::pair<float,int> ordering.
__device__ __forceinline__ uint32_t fkey(float x) {
    x = x + 0.0f;
    const uint32_t b = __float_as_uint(x);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
// same map for a value that cannot be -0 (a sum of squares): no canonicalising add
__device__ __forceinline__ uint32_t fkey_sumsq(float x) {
    const uint32_t b = __float_as_uint(x);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
    const uint32_t b = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
    return __uint_as_float(b);
}
// The key merges -0 and +0 (they compare equal in the reference's heaps).  A zero distance the reference
// reports is +0 for L2 (a sum of squares from +0) and -0 for the negative dot product (-(+0): the running sums
// start at +0 and +0 + -0 = +0, so the sum itself is never -0): `zero_bits` restores that sign on output.
__device__ __forceinline__ float fkey_inv_out(uint32_t k, uint32_t zero_bits) {
    return k == 0x80000000u ? __uint_as_float(zero_bits) : fkey_inv(k);
}

// Result-list entry: [63:32] fkey(dist) | [31:1] id | [0] expanded.  Ascending u64 order ==
// ascending (dist, id) pair order of the reference's result heap (search_function.h:50).
__device__ __forceinline__ uint64_t make_key(uint32_t dk, uint32_t id) {
    return ((uint64_t)dk << 32) | ((uint64_t)id << 1);
}
__device__ __forceinline__ uint32_t key_id(uint64_t k) { return (uint32_t)(k & 0xFFFFFFFFu) >> 1; }
__device__ __forceinline__ uint32_t key_hi(uint64_t k) { return (uint32_t)(k >> 32); }

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
    const uint32_t lo = __shfl((int)(uint32_t)v, src);
    const uint32_t hi = __shfl((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

// ------------------------------------------------------------------------------------------
// distances (support_func.h:107-128 L2Metric::Dist, :131-163 Angular::Dist)
// `a` is read with 16-B vector loads (rows are padded to a multiple of 4 floats with zeros);
// `b` likewise.  `dim` is the TRUE dimension: the L2 form drops dim%4 tail dims, the dot form
// performs the optional 4-wide and masked steps exactly when the reference does (the zero
// padding plays the role of masked_read's zeros).
// ------------------------------------------------------------------------------------------

template <typename PA, typename PB>
__device__ __forceinline__ float l2_ordered(PA a, PB b, uint32_t dim) {
    const uint32_t steps = dim >> 2;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    uint32_t t = 0;
    for (; t + 4 <= steps; t += 4) {
        const float4 a0 = a[t], a1 = a[t + 1], a2 = a[t + 2], a3 = a[t + 3];
        const float4 b0 = b[t], b1 = b[t + 1], b2 = b[t + 2], b3 = b[t + 3];
        float e;
        e = a0.x - b0.x; s0 = s0 + e * e;  e = a0.y - b0.y; s1 = s1 + e * e;
        e = a0.z - b0.z; s2 = s2 + e * e;  e = a0.w - b0.w; s3 = s3 + e * e;
        e = a1.x - b1.x; s0 = s0 + e * e;  e = a1.y - b1.y; s1 = s1 + e * e;
        e = a1.z - b1.z; s2 = s2 + e * e;  e = a1.w - b1.w; s3 = s3 + e * e;
        e = a2.x - b2.x; s0 = s0 + e * e;  e = a2.y - b2.y; s1 = s1 + e * e;
        e = a2.z - b2.z; s2 = s2 + e * e;  e = a2.w - b2.w; s3 = s3 + e * e;
        e = a3.x - b3.x; s0 = s0 + e * e;  e = a3.y - b3.y; s1 = s1 + e * e;
        e = a3.z - b3.z; s2 = s2 + e * e;  e = a3.w - b3.w; s3 = s3 + e * e;
    }
    for (; t < steps; ++t) {
        const float4 av = a[t];
        const float4 bv = b[t];
        float e;
        e = av.x - bv.x; s0 = s0 + e * e;  e = av.y - bv.y; s1 = s1 + e * e;
        e = av.z - bv.z; s2 = s2 + e * e;  e = av.w - bv.w; s3 = s3 + e * e;
    }
    return ((s0 + s1) + s2) + s3;
}

// Compile-time step count (d_low = 32 -> STEPS = 8): all row loads are issued up front.
template <int STEPS, typename PA, typename PB>
__device__ __forceinline__ float l2_ordered_fixed(PA a, PB b) {
    float4 av[STEPS];
#pragma unroll
    for (int t = 0; t < STEPS; ++t) av[t] = a[t];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int t = 0; t < STEPS; ++t) {
        const float4 bv = b[t];
        float e;
        e = av[t].x - bv.x; s0 = s0 + e * e;  e = av[t].y - bv.y; s1 = s1 + e * e;
        e = av[t].z - bv.z; s2 = s2 + e * e;  e = av[t].w - bv.w; s3 = s3 + e * e;
    }
    return ((s0 + s1) + s2) + s3;
}

template <typename PA, typename PB>
__device__ __forceinline__ float negdot_ordered(PA a, PB b, uint32_t dim) {
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, c4 = 0.f, c5 = 0.f, c6 = 0.f, c7 = 0.f;
    const uint32_t n8 = dim >> 3;
    for (uint32_t s = 0; s < n8; ++s) {
        const float4 a0 = a[2 * s], a1 = a[2 * s + 1];
        const float4 b0 = b[2 * s], b1 = b[2 * s + 1];
        c0 = c0 + a0.x * b0.x; c1 = c1 + a0.y * b0.y; c2 = c2 + a0.z * b0.z; c3 = c3 + a0.w * b0.w;
        c4 = c4 + a1.x * b1.x; c5 = c5 + a1.y * b1.y; c6 = c6 + a1.z * b1.z; c7 = c7 + a1.w * b1.w;
    }
    float m0 = c4 + c0, m1 = c5 + c1, m2 = c6 + c2, m3 = c7 + c3;
    uint32_t t = 2 * n8;
    uint32_t rem = dim & 7;
    if (rem >= 4) {
        const float4 av = a[t], bv = b[t];
        m0 = m0 + av.x * bv.x; m1 = m1 + av.y * bv.y; m2 = m2 + av.z * bv.z; m3 = m3 + av.w * bv.w;
        ++t;
        rem -= 4;
    }
    if (rem > 0) {  // masked step: padding lanes hold zeros, 0*0 = +0 is still added
        const float4 av = a[t], bv = b[t];
        m0 = m0 + av.x * bv.x; m1 = m1 + av.y * bv.y; m2 = m2 + av.z * bv.z; m3 = m3 + av.w * bv.w;
    }
    return -((m0 + m1) + (m2 + m3));
}

template <int METRIC, typename PA, typename PB>
__device__ __forceinline__ float metric_dist(PA a, PB b, uint32_t dim) {
    if constexpr (METRIC == 1) return negdot_ordered(a, b, dim);
    else return l2_ordered(a, b, dim);
}

// ------------------------------------------------------------------------------------------
// visited set (visited_list_pool.h): exact hash set of node ids in LDS, 4-slot buckets
// ------------------------------------------------------------------------------------------
// Returns true when `id` was not in the set (and is now).  A bucket (16 B) is read with one
// ds_read_b128: the id is present iff it is found in a bucket of its probe sequence before a
// bucket with an empty slot; a new id claims the first empty slot of that bucket with a CAS (a
// lane of the same wavefront may win the slot in the same instruction: then the bucket is read
// again).  Ids offered concurrently are distinct (rows are de-duplicated at index creation).
__device__ __forceinline__ bool visited_claim(uint32_t* hash, uint32_t nbuckets, uint32_t id, bool valid,
                                              unsigned int* dbg_iters = nullptr) {
    // Wave-uniform loop over "some lane still probing" (scalar branch, no per-lane loop masks);
    // ids are < 2^31 (gbnns_index_create), so only an empty slot (0xFFFFFFFF) has its sign bit set,
    // and slots of a bucket fill in order: the number of occupied slots is 4 + the sum of the signs.
    uint32_t b = __umulhi(id * 0x9E3779B1u, nbuckets);
    bool fresh = false, active = valid;
    do {
        if (dbg_iters) *dbg_iters += 1;  // diagnostic builds only (constant-folded away otherwise)
        if (active) {
            const uint4 e = *reinterpret_cast<const uint4*>(hash + 4u * b);
            const uint32_t differ = min(min(e.x ^ id, e.y ^ id), min(e.z ^ id, e.w ^ id));
            const int full = 4 + ((int)e.x >> 31) + ((int)e.y >> 31) + ((int)e.z >> 31) + ((int)e.w >> 31);
            const bool absent = differ != 0u;
            const bool claim = absent & (full < 4);
            // One CAS for every probing lane, no nested divergence: lanes that do not claim compare
            // against a value no slot ever holds (ids < 2^31) and change nothing.  A lane of this
            // wavefront may win the slot in the same instruction: the loser looks at the bucket again.
            const uint32_t old = atomicCAS(hash + 4u * b + ((uint32_t)full & 3u), claim ? kInvalidId : 0xFFFFFFFEu, id);
            const bool won = claim & (old == kInvalidId);
            fresh = won;
            active = absent & !won;
            const uint32_t nx = (b + 1u == nbuckets) ? 0u : b + 1u;
            b = (full >= 4) ? nx : b;
        }
    } while (__ballot(active));
    return fresh;
}

// Hand-scheduled form of visited_claim for the register-list kernels (the walk is instruction-issue
// bound and the compiler's version of the probe loop spends half of its ~50 instructions per iteration
// on lane-mask bookkeeping).  Same table, same protocol; returns the wave-uniform mask of the lanes
// whose id was new.  `lds_base` = LDS byte address of the table.  A lane that loses the slot race
// to another lane of the wavefront retries the following slots of the bucket straight away (the
// occupant cannot be its own id: ids offered together are distinct) and only re-reads the bucket
// when they run out.  e0..e3 need a contiguous register quad, hence the fixed v[68:71].
__device__ __forceinline__ uint64_t visited_claim_mask(uint32_t lds_base, uint32_t nbuckets, uint32_t id, uint64_t valid) {
    const uint32_t end = lds_base + (nbuckets << 4);
    const uint32_t mulc = 0x9E3779B1u;
    uint32_t basev = lds_base, neg1 = 0xFFFFFFFFu, addr;
    uint64_t fresh, act, sv;
    uint32_t t0, t1, t2;
    asm volatile(
        "v_mul_lo_u32 %[t0], %[id], %[mulc]\n\t"
        "s_mov_b64 %[fresh], 0\n\t"
        "v_mul_hi_u32 %[t0], %[t0], %[nb]\n\t"      // bucket = mulhi(id * C, nbuckets)
        "v_lshl_add_u32 %[addr], %[t0], 4, %[basev]\n\t"
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[valid]\n"
        "1:\n\t"
        "ds_read_b128 v[68:71], %[addr]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_xor_b32 %[t0], v68, %[id]\n\t"
        "v_xor_b32 %[t1], v69, %[id]\n\t"
        "v_xor_b32 %[t2], v70, %[id]\n\t"
        "v_min3_u32 %[t0], %[t0], %[t1], %[t2]\n\t"
        "v_xor_b32 %[t2], v71, %[id]\n\t"
        "v_ashrrev_i32 v68, 31, v68\n\t"
        "v_ashrrev_i32 v69, 31, v69\n\t"
        "v_ashrrev_i32 v70, 31, v70\n\t"
        "v_min_u32 %[t0], %[t0], %[t2]\n\t"          // 0 <=> id is in the bucket
        "v_ashrrev_i32 %[t2], 31, v71\n\t"
        "v_add3_u32 %[t1], v68, v69, v70\n\t"
        "v_cmp_ne_u32 vcc, 0, %[t0]\n\t"
        "v_add3_u32 %[t1], %[t1], %[t2], 4\n\t"       // occupied slots (they fill in order)
        "s_and_b64 exec, exec, vcc\n\t"                // lanes that found their id are done
        "s_cbranch_execz 9f\n\t"
        "s_mov_b64 %[act], exec\n\t"
        "v_lshl_add_u32 %[t2], %[t1], 2, %[addr]\n"    // first empty slot
        "2:\n\t"
        "v_cmp_gt_u32 vcc, 4, %[t1]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"                // lanes with a slot left to try
        "s_cbranch_execz 3f\n\t"
        "ds_cmpst_rtn_b32 %[t0], %[t2], %[neg1], %[id]\n\t"
        "v_add_u32 %[t1], 1, %[t1]\n\t"
        "v_add_u32 %[t2], 4, %[t2]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u32 vcc, -1, %[t0]\n\t"              // won the slot
        "s_or_b64 %[fresh], %[fresh], vcc\n\t"
        "s_andn2_b64 %[act], %[act], vcc\n\t"
        "s_andn2_b64 exec, exec, vcc\n\t"              // losers: next slot
        "s_cbranch_execnz 2b\n"
        "3:\n\t"
        "s_mov_b64 exec, %[act]\n\t"                   // still absent and unplaced: their bucket is full
        "s_cbranch_execz 9f\n\t"
        "v_add_u32 %[addr], 16, %[addr]\n\t"
        "v_cmp_eq_u32 vcc, %[end], %[addr]\n\t"
        "v_cndmask_b32 %[addr], %[addr], %[basev], vcc\n\t"
        "s_branch 1b\n"
        "9:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [fresh] "=&s"(fresh), [act] "=&s"(act), [sv] "=&s"(sv), [t0] "=&v"(t0), [t1] "=&v"(t1),
          [t2] "=&v"(t2), [addr] "=&v"(addr)
        : [id] "v"(id), [valid] "s"(valid), [end] "s"(end), [basev] "v"(basev), [neg1] "v"(neg1), [mulc] "s"(mulc),
          [nb] "s"(nbuckets)
        : "vcc", "memory", "v68", "v69", "v70", "v71");
    return fresh;
}

// Packed form of the table (n < 2^24, register-list kernels): a 16-byte bucket holds five 24-bit ids (bits
// 24k .. 24k+23, all-ones = empty) and, in its top byte, the number of slots handed out -- 3.2 bytes per id, so
// more wavefronts fit a CU.  Present iff found in a bucket of the probe sequence before a bucket with a free
// slot; a new id takes the slot number an atomic add on the counter returns (unique per lane: no
// compare-and-swap, no retry inside a bucket; a number >= 5 means the bucket filled up meanwhile -> next
// bucket; at most 4 + 64 additions per bucket ever, the byte cannot wrap) and writes its three bytes.
__device__ __forceinline__ uint64_t visited_claim_mask_packed(uint32_t lds_base, uint32_t nbuckets, uint32_t id, uint64_t valid) {
    const uint32_t end = lds_base + (nbuckets << 4);
    const uint32_t mulc = 0x9E3779B1u;
    uint32_t basev = lds_base, inc = 1u << 24, addr;
    uint64_t fresh, act, sv;
    uint32_t t0, t1, t2;
    asm volatile(
        "v_mul_lo_u32 %[t0], %[id], %[mulc]\n\t"
        "s_mov_b64 %[fresh], 0\n\t"
        "v_mul_hi_u32 %[t0], %[t0], %[nb]\n\t"
        "v_lshl_add_u32 %[addr], %[t0], 4, %[basev]\n\t"
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[valid]\n"
        "1:\n\t"
        "ds_read_b128 v[68:71], %[addr]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bfe_u32 v64, v68, 0, 24\n\t"
        "v_alignbit_b32 v65, v69, v68, 24\n\t"
        "v_alignbit_b32 v66, v70, v69, 16\n\t"
        "v_lshrrev_b32 v67, 8, v70\n\t"
        "v_bfe_u32 %[t1], v71, 0, 24\n\t"
        "v_bfe_u32 v65, v65, 0, 24\n\t"
        "v_bfe_u32 v66, v66, 0, 24\n\t"
        "v_xor_b32 v64, v64, %[id]\n\t"
        "v_xor_b32 v65, v65, %[id]\n\t"
        "v_xor_b32 v66, v66, %[id]\n\t"
        "v_xor_b32 v67, v67, %[id]\n\t"
        "v_xor_b32 %[t1], %[t1], %[id]\n\t"
        "v_min3_u32 v64, v64, v65, v66\n\t"
        "v_min3_u32 v64, v64, v67, %[t1]\n\t"               // 0 <=> id is in the bucket
        "v_lshrrev_b32 %[t1], 24, v71\n\t"                  // slots handed out
        "v_cmp_ne_u32 vcc, 0, v64\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz 9f\n\t"
        "s_mov_b64 %[act], exec\n\t"
        "v_cmp_gt_u32 vcc, 5, %[t1]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz 3f\n\t"
        "ds_add_rtn_u32 %[t0], %[addr], %[inc] offset:12\n\t"
        "v_lshrrev_b32 %[t2], 8, %[id]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_lshrrev_b32 %[t0], 24, %[t0]\n\t"
        "v_cmp_gt_u32 vcc, 5, %[t0]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz 3f\n\t"
        "v_mad_u32_u24 %[t0], %[t0], 3, %[addr]\n\t"
        "ds_write_b8 %[t0], %[id]\n\t"
        "ds_write_b8 %[t0], %[t2] offset:1\n\t"
        "ds_write_b8_d16_hi %[t0], %[id] offset:2\n\t"
        "s_or_b64 %[fresh], %[fresh], exec\n\t"
        "s_andn2_b64 %[act], %[act], exec\n"
        "3:\n\t"
        "s_mov_b64 exec, %[act]\n\t"
        "s_cbranch_execz 9f\n\t"
        "v_add_u32 %[addr], 16, %[addr]\n\t"
        "v_cmp_eq_u32 vcc, %[end], %[addr]\n\t"
        "v_cndmask_b32 %[addr], %[addr], %[basev], vcc\n\t"
        "s_branch 1b\n"
        "9:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [fresh] "=&s"(fresh), [act] "=&s"(act), [sv] "=&s"(sv), [t0] "=&v"(t0), [t1] "=&v"(t1),
          [t2] "=&v"(t2), [addr] "=&v"(addr)
        : [id] "v"(id), [valid] "s"(valid), [end] "s"(end), [basev] "v"(basev), [inc] "v"(inc), [mulc] "s"(mulc),
          [nb] "s"(nbuckets)
        : "vcc", "memory", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71");
    return fresh;
}

// Quotient form of the table (see GBNNS_VS_ASM further down for the layout and the protocol; this is the same code
// outside hot_expand, for the generic two-list kernel): the lanes of `valid` claim `id`; returns the lanes whose id was
// new, `overflowed` = lanes whose probe sequence ran out (stash_claim takes those).  `ctl` = WalkParams::vs_shr.
__device__ __forceinline__ uint64_t visited_claim_mask_quotient(uint32_t lds_base, uint32_t nbuckets, uint32_t id, uint64_t valid, uint32_t ctl,
                                                              uint64_t& overflowed) {
    const uint32_t end = lds_base + (nbuckets << 4);
    uint32_t basev = lds_base, addr, t0, t1, t2, mulc;
    uint64_t fresh, act, sv, ovf;
    asm volatile(
        "s_bfe_u32 %[mulc], %[shr], 0x50008\n\t"
        "s_lshl_b32 %[mulc], 0x9E3779B1, %[mulc]\n\t"
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[valid]\n\t"
        "v_mul_lo_u32 %[t0], %[id], %[mulc]\n\t"
        "s_mov_b64 %[fresh], 0\n\t"
        "s_mov_b64 %[ovf], 0\n\t"
        "s_lshl_b32 %[mulc], %[nb], 4\n\t"                  // the table's bytes
        "v_mul_hi_u32 %[t1], %[t0], %[nb]\n\t"
        "v_mul_lo_u32 %[t0], %[t0], %[nb]\n\t"
        "v_lshl_add_u32 %[addr], %[t1], 4, %[basev]\n\t"
        "v_lshrrev_b32 %[t0], %[shr], %[t0]\n\t"
        "v_lshl_or_b32 %[t2], %[t0], 16, %[t0]\n"
        "5:\n\t"
        "ds_read_b128 v[68:71], %[addr]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_xor_b32 v64, v68, %[t2]\n\t"
        "v_xor_b32 v65, v69, %[t2]\n\t"
        "v_xor_b32 v66, v70, %[t2]\n\t"
        "v_xor_b32 v67, v71, %[t2]\n\t"
        "v_pk_min_u16 v64, v64, v65\n\t"
        "v_pk_min_u16 v66, v66, v67\n\t"
        "v_bfe_u32 %[t1], v71, 16, 12\n\t"
        "v_pk_min_u16 v64, v64, v66\n\t"
        "v_mad_u32_u16 v64, v64, v64, 0 op_sel:[0,1,0,0]\n\t"
        "v_cmp_ne_u32 vcc, 0, v64\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz 9f\n\t"
        "s_mov_b64 %[act], exec\n\t"
        "v_cmp_gt_u32 vcc, 7, %[t1]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz 6f\n\t"
        "v_mov_b32 %[t1], 0x10000\n\t"
        "ds_add_rtn_u32 %[t0], %[addr], %[t1] offset:12\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bfe_u32 %[t0], %[t0], 16, 12\n\t"
        "v_cmp_gt_u32 vcc, 7, %[t0]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz 6f\n\t"
        "v_lshl_add_u32 %[t0], %[t0], 1, %[addr]\n\t"
        "ds_write_b16 %[t0], %[t2]\n\t"
        "s_or_b64 %[fresh], %[fresh], exec\n\t"
        "s_andn2_b64 %[act], %[act], exec\n"
        "6:\n\t"
        "s_mov_b64 exec, %[act]\n\t"
        "s_cbranch_execz 9f\n\t"
        "v_and_b32 %[t0], 7, %[t2]\n\t"
        "v_lshl_add_u32 %[t0], %[t0], 4, 16\n\t"
        "v_add_u32 %[addr], %[addr], %[t0]\n\t"
        "s_bfe_u32 vcc_lo, %[shr], 0x10010\n\t"
        "s_lshl_b32 vcc_lo, 0x10001000, vcc_lo\n\t"
        "v_add_u32 %[t2], vcc_lo, %[t2]\n\t"
        "v_cmp_le_u32 vcc, %[end], %[addr]\n\t"
        "v_subrev_u32 %[t0], %[mulc], %[addr]\n\t"
        "v_cndmask_b32 %[addr], %[addr], %[t0], vcc\n\t"
        "s_and_b32 vcc_lo, %[shr], 0xF0000000\n\t"          // the probe-number field alone (the low bits of ctl hold shifts and flags)
        "v_cmp_gt_u32 vcc, vcc_lo, %[t2]\n\t"
        "s_andn2_b64 %[act], exec, vcc\n\t"
        "s_or_b64 %[ovf], %[ovf], %[act]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execnz 5b\n"
        "9:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [fresh] "=&s"(fresh), [act] "=&s"(act), [sv] "=&s"(sv), [ovf] "=&s"(ovf), [mulc] "=&s"(mulc), [t0] "=&v"(t0), [t1] "=&v"(t1),
          [t2] "=&v"(t2), [addr] "=&v"(addr)
        : [id] "v"(id), [valid] "s"(valid), [end] "s"(end), [basev] "v"(basev), [shr] "s"(ctl), [nb] "s"(nbuckets)
        : "vcc", "scc", "memory", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71");
    overflowed = ovf;
    return fresh;
}

// Initial state of a packed table of `nbuckets` buckets holding `entry` (every lane calls; no sync inside).
__device__ __forceinline__ void packed_table_init(uint32_t* hash, uint32_t nbuckets, uint32_t entry, int lane) {
    for (uint32_t i = lane; i < nbuckets * 4u; i += 64) hash[i] = (i & 3u) == 3u ? 0x00FFFFFFu : 0xFFFFFFFFu;
}
__device__ __forceinline__ void packed_table_put_first(uint32_t* hash, uint32_t nbuckets, uint32_t entry) {
    const uint32_t b = __umulhi(entry * 0x9E3779B1u, nbuckets);
    hash[4u * b] = 0xFF000000u | entry;   // slot 0 (slot 1's low byte stays empty)
    hash[4u * b + 3u] = 0x01FFFFFFu;      // one slot handed out
}
// The same two for the quotient form of the table (GBNNS_VS_ASM: seven 16-bit entries + 0xF000 | count per bucket).
// Its last kStashBuckets x 16 bytes are not buckets but an exact list of up to kStashIds ids whose probe sequences
// ran out (stash_claim): rare -- a 10 000-query batch at ef = 140 sees a handful -- but each one would otherwise cost
// a hand-over, i.e. a retry launch behind the batch.
constexpr uint32_t kStashBuckets = 4, kStashIds = kStashBuckets * 4 - 1;  // (the last word counts them)
// The lanes of `mo` (whose probe sequences ran out) look their ids up in the stash behind the table's `nbuckets`
// buckets and append the new ones; those are added to `claimed`.  false: the stash is full -- hand the query over.
__device__ __forceinline__ bool stash_claim(uint32_t hash_lds, uint32_t nbuckets, uint64_t mo, uint32_t nb, uint64_t& claimed, int lane) {
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32* const stash = (lds_u32*)(size_t)(hash_lds + 16u * nbuckets);  // (hash_lds: the table's LDS byte address)
    int stash_n = (int)stash[kStashIds];
    while (mo) {
        const int l = __ffsll((unsigned long long)mo) - 1;
        mo &= mo - 1;
        const uint32_t id = (uint32_t)__builtin_amdgcn_readlane((int)nb, l);
        const bool hit = lane < stash_n && stash[lane] == id;  // (kStashIds <= 64)
        if (__ballot(hit) != 0) continue;
        if (stash_n == (int)kStashIds) return false;
        if (lane == 0) {
            stash[stash_n] = id;
            stash[kStashIds] = (uint32_t)stash_n + 1u;
        }
        stash_n += 1;
        claimed |= 1ull << l;
        wave_sync();
    }
    return true;
}
__device__ __forceinline__ void quotient_table_init(uint32_t* hash, uint32_t nbuckets, int lane) {
    for (uint32_t i = lane; i < nbuckets * 4u; i += 64) hash[i] = (i & 3u) == 3u ? 0xF000FFFFu : 0xFFFFFFFFu;
    if (lane == 0) hash[nbuckets * 4u + kStashIds] = 0u;  // the stash behind the buckets is empty
}
__device__ __forceinline__ void quotient_table_put_first(uint32_t* hash, uint32_t nbuckets, uint32_t entry, uint32_t shr) {
    const uint32_t h = entry * (0x9E3779B1u << ((shr >> 8) & 31u));
    const uint32_t b = __umulhi(h, nbuckets);
    hash[4u * b] = 0xFFFF0000u | ((h * nbuckets) >> (shr & 31u));  // slot 0, displacement 0
    hash[4u * b + 3u] = 0xF001FFFFu;                       // one slot handed out
}

// ------------------------------------------------------------------------------------------
// re-rank, pair form (search_function.h:105-125 getRealNearest) -- shared by rerank_pair_kernel and by the
// walk kernels that re-rank their own query at the end of its walk
// ------------------------------------------------------------------------------------------
// L2 metric, dim % 8 == 0: lanes 2i / 2i+1 share candidate i's row and take its even / odd 16-byte steps,
// so the two lanes of a pair read 32 contiguous bytes per load.  The re-rank was bound by the CU's
// vector-memory path (one cache-line access per 16-B load when a lane streams a row alone, DESIGN.md
// section 5.1); pairs cost that path 1.4x less.  The running sums hop between the two lanes once per step
// (DPP quad_perm 1,0,3,2): step 2k is added in the even lane on top of the odd lane's sums, step 2k+1 in
// the odd lane on top of the even lane's -- the reference's order 0, 1, 2, ...
// Winner = strict minimum in pop order  <=>  min over (distance, pop index).  Returns the pop index of the
// winner (wave-uniform), -1 for an empty list.  `id_at(r)` gives the id of pop index r (called by all lanes).
struct RerankSrc {
    const float* q;      // [nq x qstride] original-space queries
    uint32_t qstride;
    const float* db;     // [n x dstride]
    uint32_t dstride, dim, n;
};

__device__ __forceinline__ float dpp_swap_pair(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, false));
}

// Negative-dot form (Angular::Dist, support_func.h:131-163, dim % 8 == 0): the reference keeps 8 running
// sums (k mod 8); the even lane of a pair owns sums 0..3 (first 16 bytes of every 32-byte step), the odd lane
// sums 4..7 -- independent chains, folded once at the end: m_j = c_{j+4} + c_j (one DPP add per j, in the
// even lane), then -((m0 + m1) + (m2 + m3)).
__device__ __forceinline__ float dpp_from_odd(float x) {  // even lane <- its odd partner (quad_perm 1,1,3,3)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xF5, 0xf, 0xf, false));
}

// DEEP > 8 (L2 only): that many 16-byte loads in flight per lane before the first one is consumed.  A lone wavefront
// streams its candidates' rows at (bytes in flight) / latency: 200 rows of 960 floats (GIST, ef = 200) take it 0.095 ms
// with 8 loads in flight and 0.070 ms with 24 (rocprofv3, one-query launches).  The generic wide-row walk kernels and
// the stand-alone kernel have the registers for 24 (same operations, same order).
// `pass0` / `pass_step` (walk_coop.hip: two wavefronts share a query's candidates): this wavefront takes the 32-candidate passes
// pass0, pass0 + pass_step, ...; `best_key` (optional) receives min (fkey(dist) << 32 | pop index) over them, all-ones when it
// saw none -- the minimum over the wavefronts' keys is the minimum over all candidates.
template <int METRIC, int DEEP = 8, typename IdAt>
__device__ __forceinline__ int rerank_pairs_core(const RerankSrc& a, uint32_t qi, int cnt, float* qf, int lane, IdAt id_at,
                                                 int pass0 = 0, int pass_step = 1, uint64_t* best_key = nullptr) {
    const uint32_t half = (uint32_t)lane & 1u, slot = (uint32_t)lane >> 1;
    const float4* qs = reinterpret_cast<const float4*>(qf);
    for (uint32_t i = lane; i < a.dstride; i += 64)
        qf[i] = (i < a.dim) ? a.q[(size_t)qi * a.qstride + i] : 0.f;
    wave_sync();
    const uint32_t pairs = a.dim >> 3;  // steps / 2
    uint64_t bestk = ~0ull;
    for (int base = 32 * pass0; base < cnt; base += 32 * pass_step) {
        const int r = base + (int)slot;
        const bool valid = r < cnt;
        uint32_t id = id_at(valid ? r : base);  // lanes beyond the list redo the first row (discarded)
        id = id < a.n ? id : 0u;                // (never dereference an id outside the table)
        const float4* row = reinterpret_cast<const float4*>(a.db + (size_t)id * a.dstride) + half;
        const float4* qh = qs + half;
        if constexpr (METRIC == 1) {
            float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;  // this lane's four of the eight running sums
            uint32_t k = 0;
            for (; k + 8 <= pairs; k += 8) {
                float4 rv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) rv[j] = row[2 * (k + j)];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 qv = qh[2 * (k + j)];
                    c0 = c0 + rv[j].x * qv.x; c1 = c1 + rv[j].y * qv.y; c2 = c2 + rv[j].z * qv.z; c3 = c3 + rv[j].w * qv.w;
                }
            }
            for (; k < pairs; ++k) {
                const float4 rv = row[2 * k];
                const float4 qv = qh[2 * k];
                c0 = c0 + rv.x * qv.x; c1 = c1 + rv.y * qv.y; c2 = c2 + rv.z * qv.z; c3 = c3 + rv.w * qv.w;
            }
            const float m0 = dpp_from_odd(c0) + c0, m1 = dpp_from_odd(c1) + c1;  // c_{j+4} + c_j, valid in the even lane
            const float m2 = dpp_from_odd(c2) + c2, m3 = dpp_from_odd(c3) + c3;
            const float dv = -((m0 + m1) + (m2 + m3));
            if (valid && !half) {
                const uint64_t kv = ((uint64_t)fkey(dv) << 32) | (uint32_t)r;
                bestk = kv < bestk ? kv : bestk;
            }
            continue;
        }
        float u0, u1, u2, u3, v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        uint32_t k = 0;
        if constexpr (DEEP > 8) {
            for (; k + DEEP <= pairs; k += DEEP) {
                float4 rv[DEEP];
#pragma unroll
                for (int j = 0; j < DEEP; ++j) rv[j] = row[2 * (k + j)];
#pragma unroll
                for (int j = 0; j < DEEP; ++j) {
                    const float4 qv = qh[2 * (k + j)];
                    float e;
                    e = rv[j].x - qv.x; const float p0 = e * e;
                    e = rv[j].y - qv.y; const float p1 = e * e;
                    e = rv[j].z - qv.z; const float p2 = e * e;
                    e = rv[j].w - qv.w; const float p3 = e * e;
                    u0 = dpp_swap_pair(v0) + p0; u1 = dpp_swap_pair(v1) + p1; u2 = dpp_swap_pair(v2) + p2; u3 = dpp_swap_pair(v3) + p3;
                    v0 = dpp_swap_pair(u0) + p0; v1 = dpp_swap_pair(u1) + p1; v2 = dpp_swap_pair(u2) + p2; v3 = dpp_swap_pair(u3) + p3;
                }
            }
        }
        for (; k + 8 <= pairs; k += 8) {  // eight 16-B loads in flight per lane
            float4 rv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) rv[j] = row[2 * (k + j)];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 qv = qh[2 * (k + j)];
                float e;
                e = rv[j].x - qv.x; const float p0 = e * e;
                e = rv[j].y - qv.y; const float p1 = e * e;
                e = rv[j].z - qv.z; const float p2 = e * e;
                e = rv[j].w - qv.w; const float p3 = e * e;
                u0 = dpp_swap_pair(v0) + p0; u1 = dpp_swap_pair(v1) + p1; u2 = dpp_swap_pair(v2) + p2; u3 = dpp_swap_pair(v3) + p3;
                v0 = dpp_swap_pair(u0) + p0; v1 = dpp_swap_pair(u1) + p1; v2 = dpp_swap_pair(u2) + p2; v3 = dpp_swap_pair(u3) + p3;
            }
        }
        for (; k + 4 <= pairs; k += 4) {
            float4 rv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) rv[j] = row[2 * (k + j)];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 qv = qh[2 * (k + j)];
                float e;
                e = rv[j].x - qv.x; const float p0 = e * e;
                e = rv[j].y - qv.y; const float p1 = e * e;
                e = rv[j].z - qv.z; const float p2 = e * e;
                e = rv[j].w - qv.w; const float p3 = e * e;
                u0 = dpp_swap_pair(v0) + p0; u1 = dpp_swap_pair(v1) + p1; u2 = dpp_swap_pair(v2) + p2; u3 = dpp_swap_pair(v3) + p3;
                v0 = dpp_swap_pair(u0) + p0; v1 = dpp_swap_pair(u1) + p1; v2 = dpp_swap_pair(u2) + p2; v3 = dpp_swap_pair(u3) + p3;
            }
        }
        // (dim % 8 == 4 -- glove's 300: one more 16-byte step, the even lane's alone; the odd lane adds +0 to the sums it takes over, which
        // leaves a sum of squares as it is)
        const uint32_t pairs_t = pairs + ((a.dim >> 2) & 1u);
        for (; k < pairs_t; ++k) {
            const bool mine = k < pairs || !half;
            float4 rv = make_float4(0.f, 0.f, 0.f, 0.f), qv = rv;
            if (mine) { rv = row[2 * k]; qv = qh[2 * k]; }
            float e;
            e = rv.x - qv.x; const float p0 = e * e;
            e = rv.y - qv.y; const float p1 = e * e;
            e = rv.z - qv.z; const float p2 = e * e;
            e = rv.w - qv.w; const float p3 = e * e;
            u0 = dpp_swap_pair(v0) + p0; u1 = dpp_swap_pair(v1) + p1; u2 = dpp_swap_pair(v2) + p2; u3 = dpp_swap_pair(v3) + p3;
            v0 = dpp_swap_pair(u0) + p0; v1 = dpp_swap_pair(u1) + p1; v2 = dpp_swap_pair(u2) + p2; v3 = dpp_swap_pair(u3) + p3;
        }
        // the odd lane's v holds all steps: in the even lane `u` is the valid one, in the odd lane `v`
        const float dv = ((v0 + v1) + v2) + v3;
        if (valid && half) {
            const uint64_t kv = ((uint64_t)fkey(dv) << 32) | (uint32_t)r;
            bestk = kv < bestk ? kv : bestk;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t o = shfl_u64(bestk, lane ^ off);
        bestk = o < bestk ? o : bestk;
    }
    bestk = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(bestk >> 32)) << 32) |
            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bestk);
    if (best_key) *best_key = bestk;
    return (cnt > 0 && bestk != ~0ull) ? (int)(uint32_t)(bestk & 0xFFFFFFFFu) : -1;
}

// Fused re-rank at the end of a walk: the wavefront re-ranks its own query's candidates (pop index r =
// list rank kept-1-r) instead of leaving them to a second kernel -- the re-rank's memory-bound work then
// runs beside other wavefronts' walks and fills the slots the last "round" of a batch leaves idle.  The
// walk's LDS is dead by now and stages the original-space query.
template <int DEEP = 8, typename IdAtRank>
__device__ __forceinline__ void fused_rerank(const WalkParams& p, uint32_t qi, int kept, unsigned char* smem, int lane,
                                             IdAtRank id_at_rank) {
    RerankSrc a{p.rr_q, p.rr_qstride, p.rr_db, p.rr_dstride, p.rr_dim, p.rr_n};
    wave_sync();  // every lane is done with the walk's LDS
    int win;
    if (p.rr_metric == 1)
        win = rerank_pairs_core<1>(a, qi, kept, reinterpret_cast<float*>(smem), lane, [&](int r) { return id_at_rank(kept - 1 - r); });
    else
        win = rerank_pairs_core<0, DEEP>(a, qi, kept, reinterpret_cast<float*>(smem), lane, [&](int r) { return id_at_rank(kept - 1 - r); });
    const uint32_t ans = id_at_rank(win >= 0 ? kept - 1 - win : 0);
    if (lane == 0) p.rr_out[qi] = win >= 0 ? ans : kInvalidId;
}

// ------------------------------------------------------------------------------------------
// sorted result list (search_function.h:50 topResults) -- ascending u64 keys, capacity ef
// ------------------------------------------------------------------------------------------

// Inserts `nk`; when the list is full the largest key is evicted (returned through `evicted`).
// One top-down pass: every chunk of 64 keys is read once, keys >= nk are rewritten one slot up.
template <typename KP>
__device__ __forceinline__ int list_insert(KP keys, int& size, int ef, uint64_t nk,
                                           uint64_t& evicted, bool& did_evict, int lane) {
    did_evict = (size >= ef);  // (> ef only with several entry points: the list then stays one longer per extra entry)
    evicted = did_evict ? keys[size - 1] : 0ull;
    const int top = did_evict ? size - 1 : size;  // keys [0, top) may have to move
    int pos = 0;
    for (int base = (top - 1) & ~63; base >= 0; base -= 64) {
        const int idx = base + lane;
        const uint64_t v = (idx < top) ? keys[idx] : ~0ull;
        const bool lt = v < nk;
        const uint64_t m = __ballot(lt);
        if (idx < top && !lt) keys[idx + 1] = v;
        if (m) {
            pos = base + __popcll(m);
            break;
        }
    }
    if (lane == 0) keys[pos] = nk;
    if (!did_evict) ++size;
    wave_sync();
    return pos;
}

// ------------------------------------------------------------------------------------------
// the beam walk
// ------------------------------------------------------------------------------------------

struct WalkState {
    int size;       // entries in the result list
    int tsize;      // entries in the tie list
    int first_un;   // every list entry below this index is expanded
    int hops;
    int dist_calc;
    int edges;      // neighbour ids read (for the algorithmic-bytes figure)
};

// Tie list: result-list entries that were evicted UNEXPANDED while their distance still equals
// the current worst distance.  The reference keeps every evicted entry in its unbounded candidate
// heap (search_function.h:55,65-69) and expands such an entry when it surfaces, because the stop
// test is strict (`cand.dist > worst.dist`, :67).  Entries whose distance exceeds the worst
// distance can never be expanded again (the worst distance only decreases), so only exact ties
// need to be kept; the list is flushed whenever the worst distance strictly decreases.

// Two containers for it.  TieList: a small array of evicted keys (LDS; the fast kernels hand a query over when it
// overflows).  TieBits: one bit per node in global memory (general kernel: exact for any number of ties at
// n / 8 bytes per wavefront slot) with the word range that may hold bits; ids in the set are distinct (an
// unexpanded entry was claimed in the visited set of the current entry point's walk exactly once).
struct TieList {
    uint64_t* a;
    int cap;
    __device__ __forceinline__ bool push(uint64_t ev, int& tsize, int lane) {
        if (tsize >= cap) return false;
        if (lane == 0) a[tsize] = ev;
        tsize += 1;
        wave_sync();
        return true;
    }
    __device__ __forceinline__ void clear(int& tsize, int) { tsize = 0; }
    // id + 1 of the largest id in the set (0 = empty); `pos` = its slot
    __device__ __forceinline__ uint32_t max_plus1(int tsize, int lane, int& pos) const {
        uint32_t tbest = 0;
        pos = -1;
        for (int base = 0; base < tsize; base += 64) {
            const int idx = base + lane;
            uint32_t v = (idx < tsize) ? key_id(a[idx]) + 1u : 0u;
            int w = idx;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {  // wave arg-max (ids are distinct)
                const uint32_t ov = (uint32_t)__shfl_xor((int)v, off);
                const int ow = __shfl_xor(w, off);
                if (ov > v) { v = ov; w = ow; }
            }
            v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
            w = __builtin_amdgcn_readfirstlane(w);
            if (v > tbest) { tbest = v; pos = w; }
        }
        return tbest;
    }
    __device__ __forceinline__ void remove(int pos, uint32_t, int& tsize, int lane) {
        if (lane == 0) a[pos] = a[tsize - 1];  // unordered remove
        tsize -= 1;
        wave_sync();
    }
};

struct TieBits {
    uint32_t* bm;       // [ceil(n / 32)] words, all zero whenever the set is empty
    uint32_t lo, hi;    // words that may hold bits: lo..hi (lo > hi: none)
    __device__ __forceinline__ void reset_range() { lo = 0xFFFFFFFFu; hi = 0u; }
    __device__ __forceinline__ bool push(uint64_t ev, int& tsize, int lane) {
        const uint32_t id = key_id(ev), w = id >> 5;
        if (lane == 0) bm[w] |= 1u << (id & 31u);
        lo = w < lo ? w : lo;
        hi = w > hi ? w : hi;
        tsize += 1;
        wave_sync();
        return true;
    }
    __device__ __forceinline__ void clear(int& tsize, int lane) {
        if (tsize > 0) {
            for (uint64_t w = (uint64_t)lo + lane; w <= hi; w += 64) bm[w] = 0u;
            wave_sync();
        }
        tsize = 0;
        reset_range();
    }
    __device__ __forceinline__ uint32_t max_plus1(int tsize, int lane, int& pos) const {
        pos = -1;
        if (tsize <= 0) return 0u;
        for (int64_t base = hi; base >= (int64_t)lo; base -= 64) {  // from the top word down, 64 words per pass
            const int64_t w = base - lane;
            const uint32_t v = w >= (int64_t)lo ? bm[w] : 0u;
            const uint64_t m = __ballot(v != 0u);
            if (m) {
                const int l = __ffsll((unsigned long long)m) - 1;
                const uint32_t vv = (uint32_t)__shfl((int)v, l);
                pos = 0;
                return (uint32_t)(base - l) * 32u + (31u - (uint32_t)__clz((int)vv)) + 1u;
            }
        }
        return 0u;
    }
    __device__ __forceinline__ void remove(int, uint32_t id, int& tsize, int lane) {
        if (lane == 0) bm[id >> 5] &= ~(1u << (id & 31u));
        tsize -= 1;
        wave_sync();
    }
};

// Picks the next node to expand (closest unexpanded; ties -> LARGEST id, because the candidate
// heap is keyed (-dist, id)).  Returns false when nothing is left (the reference's loop exit).
template <typename KP, typename TP>
__device__ __forceinline__ bool select_candidate(KP keys, TP& tie, WalkState& st, uint32_t& node,
                                                 int lane) {
    int p = -1, best = -1;
    uint32_t hi_p = 0;
    for (int base = st.first_un & ~63; base < st.size; base += 64) {
        const int idx = base + lane;
        const uint64_t kv = (idx < st.size) ? keys[idx] : ~0ull;
        const bool un = (idx < st.size) && !(kv & 1ull);
        if (p < 0) {
            const uint64_t m = __ballot(un);
            if (!m) continue;
            const int pl = __ffsll((unsigned long long)m) - 1;
            p = base + pl;
            hi_p = (uint32_t)__shfl((int)key_hi(kv), pl);
        }
        const bool same = (idx < st.size) && key_hi(kv) == hi_p;
        const uint64_t ms = __ballot(same && un && idx >= p);
        if (ms) best = base + 63 - __clzll((long long)ms);
        const uint64_t mall = __ballot(same);
        if (!((mall >> 63) & 1ull)) break;  // run of equal distances ends inside this chunk
    }
    st.first_un = (p < 0) ? st.size : p;

    if (st.tsize > 0) {
        const uint32_t worst_hi = key_hi(keys[st.size - 1]);
        if (p < 0 || hi_p == worst_hi) {
            // all tie entries sit at the worst distance: the largest id among them competes
            int tpos;
            const uint32_t tbest = tie.max_plus1(st.tsize, lane, tpos);  // id + 1, 0 = none
            const uint32_t tmax = tbest - 1u;
            const uint32_t lid = (best >= 0) ? key_id(keys[best]) : 0u;
            if (tpos >= 0 && (best < 0 || tmax > lid)) {
                node = tmax;
                tie.remove(tpos, tmax, st.tsize, lane);
                return true;
            }
        }
    }
    if (best < 0) return false;
    node = key_id(keys[best]);
    if (lane == 0) keys[best] = keys[best] | 1ull;
    wave_sync();
    return true;
}

// Result of offering one (dist, id) to the result list with the reference's rule
// (search_function.h:31-37): insert when worst.dist > dist || size < ef (strict, distance only),
// then evict the largest pair if size > ef.  Returns false if the tie list overflowed.
template <typename KP, typename TP>
__device__ __forceinline__ bool offer(KP keys, TP& tie, WalkState& st, int ef,
                                      uint32_t dk, uint32_t id, int lane) {
    if (st.size >= ef && !(dk < key_hi(keys[st.size - 1]))) return true;
    uint64_t ev;
    bool did;
    const int pos = list_insert(keys, st.size, ef, make_key(dk, id), ev, did, lane);
    if (pos < st.first_un) st.first_un = pos;
    if (did) {
        const uint32_t nw = key_hi(keys[st.size - 1]);
        if (key_hi(ev) == nw) {
            if (!(ev & 1ull) && !tie.push(ev, st.tsize, lane)) return false;
        } else {
            tie.clear(st.tsize, lane);
        }
    }
    return true;
}

// Writes the trimmed result list in POP order (worst -> best), as the reference's heap would be
// drained by getRealNearest (search_function.h:109-122).
template <typename KP>
__device__ __forceinline__ void write_results(const WalkParams& p, uint32_t qi, KP keys,
                                              const WalkState& st, int lane) {
    const int kept = st.size < p.k ? st.size : p.k;
    for (int r = lane; r < (int)p.cand_stride; r += 64) {
        uint32_t id = kInvalidId;
        float dv = __builtin_inff();
        if (r < kept) {
            const uint64_t kv = keys[kept - 1 - r];
            id = key_id(kv);
            dv = fkey_inv_out(key_hi(kv), p.zero_dist_bits);
        }
        p.cand[(size_t)qi * p.cand_stride + r] = id;
        if (p.cand_dist) p.cand_dist[(size_t)qi * p.cand_stride + r] = dv;
    }
    if (lane == 0) {
        p.count[qi] = kept;
        p.hops[qi] = st.hops;
        p.dist_calc[qi] = st.dist_calc;
        atomicMax(p.max_dc, (uint32_t)st.dist_calc);
        if (p.edges) p.edges[qi] = st.edges;
        // PLAIN answer = topk.top() after trimming the heap to k (search_function.h:174-181): the k-th best
        if (p.best) p.best[qi] = kept > 0 ? key_id(keys[kept - 1]) : kInvalidId;
    }
}

// An entry id outside the index (device buffers are not validated on the host, host buffers are): no row of it may
// be touched.  The query gets an empty result -- answer 0xFFFFFFFF, no candidates, zero counters (gbnns.h).
__device__ __forceinline__ void write_bad_entry(const WalkParams& p, uint32_t qi, int lane) {
    for (int r = lane; r < (int)p.cand_stride; r += 64) {
        p.cand[(size_t)qi * p.cand_stride + r] = kInvalidId;
        if (p.cand_dist) p.cand_dist[(size_t)qi * p.cand_stride + r] = __builtin_inff();
    }
    if (lane == 0) {
        p.count[qi] = 0;
        p.hops[qi] = 0;
        p.dist_calc[qi] = 0;
        if (p.edges) p.edges[qi] = 0;
        if (p.best) p.best[qi] = kInvalidId;
        if (p.rr_db) p.rr_out[qi] = kInvalidId;
    }
}

template <int METRIC, int STEPS, typename QP>
__device__ __forceinline__ float walk_dist(QP qs, const float* row, uint32_t dim) {
    const float4* r4 = reinterpret_cast<const float4*>(row);
    if constexpr (METRIC == 0 && STEPS > 0) return l2_ordered_fixed<STEPS>(r4, qs);  // row loads first
    else return metric_dist<METRIC>(qs, r4, dim);
}

}  // namespace

}  // namespace gbnns
