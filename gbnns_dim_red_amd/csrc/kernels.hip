// kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the two-stage graph search.
//
// Arithmetic contract: every distance / dot product is evaluated in IEEE binary32 with
// the operation order of the reference's SSE/AVX source (support_func.h:107-163): separate mul and
// add (this file is compiled with -ffp-contract=off), 4 resp. 8 independent running sums, the
// reference's horizontal-sum order, correctly rounded sqrt and divide.  That makes every float the
// kernels produce bit-identical to the reference built with strict flags, and therefore every
// compare-driven decision (beam order, visited set, hops, dist_calc, final ids) identical.
//
// Wavefront = 64 lanes.  All walk/re-rank kernels use one 64-thread workgroup (= one wavefront)
// per query, so cross-lane traffic goes through ballots/shuffles and wave-private LDS.

#include "kernels.h"

// This file is compiled three times (csrc/Makefile, -DGBNNS_TU=0/1/2) so that the many walk-kernel instantiations
// build in parallel: unit 0 holds every launcher and the L2 walks over generic / 128-byte rows, unit 1 the
// dot-metric walks, unit 2 the L2 walks over 192- and 256-byte rows.  Templates are instantiated where they are used.
#ifndef GBNNS_TU
#define GBNNS_TU 0
#endif

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
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

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
// when they run out.  e0..e3 need a contiguous register quad, hence the fixed v[92:95].
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
        "ds_read_b128 v[92:95], %[addr]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_xor_b32 %[t0], v92, %[id]\n\t"
        "v_xor_b32 %[t1], v93, %[id]\n\t"
        "v_xor_b32 %[t2], v94, %[id]\n\t"
        "v_min3_u32 %[t0], %[t0], %[t1], %[t2]\n\t"
        "v_xor_b32 %[t2], v95, %[id]\n\t"
        "v_ashrrev_i32 v92, 31, v92\n\t"
        "v_ashrrev_i32 v93, 31, v93\n\t"
        "v_ashrrev_i32 v94, 31, v94\n\t"
        "v_min_u32 %[t0], %[t0], %[t2]\n\t"          // 0 <=> id is in the bucket
        "v_ashrrev_i32 %[t2], 31, v95\n\t"
        "v_add3_u32 %[t1], v92, v93, v94\n\t"
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
        : "vcc", "memory", "v92", "v93", "v94", "v95");
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
        "ds_read_b128 v[92:95], %[addr]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bfe_u32 v88, v92, 0, 24\n\t"
        "v_alignbit_b32 v89, v93, v92, 24\n\t"
        "v_alignbit_b32 v90, v94, v93, 16\n\t"
        "v_lshrrev_b32 v91, 8, v94\n\t"
        "v_bfe_u32 %[t1], v95, 0, 24\n\t"
        "v_bfe_u32 v89, v89, 0, 24\n\t"
        "v_bfe_u32 v90, v90, 0, 24\n\t"
        "v_xor_b32 v88, v88, %[id]\n\t"
        "v_xor_b32 v89, v89, %[id]\n\t"
        "v_xor_b32 v90, v90, %[id]\n\t"
        "v_xor_b32 v91, v91, %[id]\n\t"
        "v_xor_b32 %[t1], %[t1], %[id]\n\t"
        "v_min3_u32 v88, v88, v89, v90\n\t"
        "v_min3_u32 v88, v88, v91, %[t1]\n\t"               // 0 <=> id is in the bucket
        "v_lshrrev_b32 %[t1], 24, v95\n\t"                  // slots handed out
        "v_cmp_ne_u32 vcc, 0, v88\n\t"
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
        : "vcc", "memory", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95");
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
        "ds_read_b128 v[92:95], %[addr]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_xor_b32 v88, v92, %[t2]\n\t"
        "v_xor_b32 v89, v93, %[t2]\n\t"
        "v_xor_b32 v90, v94, %[t2]\n\t"
        "v_xor_b32 v91, v95, %[t2]\n\t"
        "v_pk_min_u16 v88, v88, v89\n\t"
        "v_pk_min_u16 v90, v90, v91\n\t"
        "v_bfe_u32 %[t1], v95, 16, 12\n\t"
        "v_pk_min_u16 v88, v88, v90\n\t"
        "v_mad_u32_u16 v88, v88, v88, 0 op_sel:[0,1,0,0]\n\t"
        "v_cmp_ne_u32 vcc, 0, v88\n\t"
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
        : "vcc", "scc", "memory", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95");
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
template <int METRIC, int DEEP = 8, typename IdAt>
__device__ __forceinline__ int rerank_pairs_core(const RerankSrc& a, uint32_t qi, int cnt, float* qf, int lane, IdAt id_at) {
    const uint32_t half = (uint32_t)lane & 1u, slot = (uint32_t)lane >> 1;
    const float4* qs = reinterpret_cast<const float4*>(qf);
    for (uint32_t i = lane; i < a.dstride; i += 64)
        qf[i] = (i < a.dim) ? a.q[(size_t)qi * a.qstride + i] : 0.f;
    wave_sync();
    const uint32_t pairs = a.dim >> 3;  // steps / 2
    uint64_t bestk = ~0ull;
    for (int base = 0; base < cnt; base += 32) {
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
        for (; k < pairs; ++k) {
            const float4 rv = row[2 * k];
            const float4 qv = qh[2 * k];
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
    return cnt > 0 ? (int)(uint32_t)(bestk & 0xFFFFFFFFu) : -1;
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

// ---- fast kernel: result list, tie list, visited hash set and the query all live in LDS -------
//
// LDS layout (dynamic): [keys: ef_pad x u64][tie: kTieCap x u64][q: dstride x f32][hash: cap x u32]
// The visited set is an open-addressing hash set of node ids (exact: an id is "visited" iff it
// was inserted).  A query that would exceed hash_limit entries, or whose tie list overflows, is
// appended to the hand-over list and re-run from scratch by the general kernel.

// BITMAP: the visited set is one bit per node in HBM (`bitmap`, private to this wavefront's slot, cleared here per
// query) instead of the LDS table -- for large ef, where the table of a 10 000-distance walk would leave four
// wavefronts per CU: the LDS then holds the result list, the tie list and the query only.
template <int METRIC, int STEPS, bool PACKED, bool BITMAP = false>
__device__ __forceinline__ void walk_fast_one(const WalkParams& p, uint32_t qi, unsigned char* smem,
                                              uint32_t* ovf_count, uint32_t* ovf_list, uint32_t* bitmap = nullptr) {
    const int lane = lane_id();
    const int ef = p.ef;
    const int ef_pad = (ef + 63) & ~63;
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
    TieList tie{keys + ef_pad, kTieCap};
    float* qf = reinterpret_cast<float*>(tie.a + kTieCap);
    uint32_t* hash = reinterpret_cast<uint32_t*>(qf + p.dstride);
    const float4* qs = reinterpret_cast<const float4*>(qf);
    const uint32_t cap = p.hash_cap;  // any size: slot = mulhi(id * C, cap)
    // PACKED (n < 2^24): five 24-bit ids per 16-byte bucket (visited_claim_mask_packed), else 4-byte slots
    const uint32_t nbuckets = PACKED ? cap / 5u : cap >> 2;
    const uint32_t hash_lds = (uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(hash));

    if constexpr (BITMAP) {
        for (uint32_t i = lane; i < p.bitmap_words; i += 64) bitmap[i] = 0u;
    } else if constexpr (PACKED) packed_table_init(hash, nbuckets, 0u, lane);
    else for (uint32_t i = lane; i < cap; i += 64) hash[i] = kInvalidId;
    for (uint32_t i = lane; i < p.dstride; i += 64)
        qf[i] = (i < p.dim) ? p.q[(size_t)qi * p.qstride + i] : 0.f;
    wave_sync();

    WalkState st;
    st.size = 0; st.tsize = 0; st.first_un = 0; st.hops = 0; st.dist_calc = 1; st.edges = 0;

    const uint32_t entry = p.entries ? p.entries[qi] : 0u;
    if (entry >= p.n) { write_bad_entry(p, qi, lane); return; }
    {
        const float d0 = walk_dist<METRIC, STEPS>(qs, p.db + (size_t)entry * p.dstride, p.dim);
        if (lane == 0) {
            keys[0] = make_key(fkey(d0), entry);
            if constexpr (BITMAP) bitmap[entry >> 5] = 1u << (entry & 31u);
            else if constexpr (PACKED) packed_table_put_first(hash, nbuckets, entry);
            else hash[4u * __umulhi(entry * 0x9E3779B1u, nbuckets)] = entry;  // first slot of its bucket
        }
        st.size = 1;
        wave_sync();
    }

    bool handed_over = false;
    uint32_t node;
    // makeStep (search_function.h:15-40) over one adjacency row; `found` = something was inserted (:34)
    auto make_step = [&](const uint32_t* row, uint32_t stride, bool& found) {
        for (uint32_t c = 0; c < stride; c += 64) {
            const uint32_t nb = (c + lane < stride) ? row[c + lane] : kInvalidId;
            const bool valid = nb != kInvalidId;
            const uint64_t mv = __ballot(valid);
            if (!mv) break;
            if constexpr (!BITMAP)
                if ((uint32_t)st.dist_calc + 64u > p.hash_limit) { handed_over = true; break; }
            st.edges += __popcll(mv);
            bool fresh;
            if constexpr (BITMAP) {
                fresh = false;
                if (valid) {
                    const uint32_t bit = 1u << (nb & 31u);
                    fresh = !(atomicOr(&bitmap[nb >> 5], bit) & bit);
                }
            } else if constexpr (PACKED) fresh = __builtin_amdgcn_inverse_ballot_w64(visited_claim_mask_packed(hash_lds, nbuckets, nb, mv));
            else fresh = visited_claim(hash, nbuckets, nb, valid);
            uint32_t dk = 0xFFFFFFFFu;
            if (fresh) dk = fkey(walk_dist<METRIC, STEPS>(qs, p.db + (size_t)nb * p.dstride, p.dim));
            const uint64_t mf = __ballot(fresh);
            st.dist_calc += __popcll(mf);
            // reference order: neighbours are offered one by one in list order
            const uint32_t worst0 = key_hi(keys[st.size - 1]);
            uint64_t m = __ballot(fresh && (st.size < ef || dk < worst0));
            if (m) found = true;  // the first of them is inserted whatever the others do
            while (m) {
                const int l = __ffsll((unsigned long long)m) - 1;
                m &= m - 1;
                const uint32_t dl = (uint32_t)__shfl((int)dk, l);
                const uint32_t il = (uint32_t)__shfl((int)nb, l);
                if (!offer(keys, tie, st, ef, dl, il, lane)) { handed_over = true; break; }
            }
            if (handed_over) break;
        }
    };
    while (select_candidate(keys, tie, st, node, lane)) {
        bool found = false;
        if (p.aux_ell && (uint32_t)st.hops < p.hops_bound)  // search_function.h:73-80
            make_step(p.aux_ell + (size_t)node * p.aux_stride, p.aux_stride, found);
        if (!handed_over && !(found && p.llf))               // :82-89
            make_step(p.ell + (size_t)node * p.ell_stride, p.ell_stride, found);
        if (handed_over) break;
        st.hops += 1;
    }

    if (handed_over) {
        if (lane == 0) {
            const uint32_t slot = atomicAdd(ovf_count, 1u);
            ovf_list[slot] = qi;
        }
        return;
    }
    write_results(p, qi, keys, st, lane);
}

// First pass: one query per workgroup (= wavefront).  Retry pass: persistent wavefronts, one per CU
// with the largest visited set LDS allows, re-run the queries the first pass handed over; what
// still does not fit goes to the general kernel.
template <typename F>
__device__ __forceinline__ void retry_loop(const WalkParams& p, F&& run) {
    const uint32_t total = *p.ovf_count;
    while (true) {
        uint32_t w = 0;
        if (lane_id() == 0) w = atomicAdd(p.r_cursor, 1u);
        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
        if (w >= total) break;
        run(p.ovf_list[w]);
        wave_sync();
    }
}

template <int METRIC, int STEPS, bool RETRY, bool PACKED>
__global__ __launch_bounds__(64) void walk_fast_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (RETRY) {
        retry_loop(p, [&](uint32_t qi) { walk_fast_one<METRIC, STEPS, PACKED>(p, qi, smem, p.ovf2_count, p.ovf2_list); });
    } else {
        walk_fast_one<METRIC, STEPS, PACKED>(p, walk_query_of(p, blockIdx.x), smem, p.ovf_count, p.ovf_list);
    }
}

// First pass for large ef: persistent wavefronts (as many as the LDS holds result lists), each with its own
// visited bitmap in HBM, pulling query indices from a counter.
template <int METRIC, int STEPS>
__global__ __launch_bounds__(64) void walk_bitmap_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* bitmap = p.fp_bitmap + (size_t)blockIdx.x * p.bitmap_words;
    while (true) {
        uint32_t w = 0;
        if (lane_id() == 0) w = atomicAdd(p.fp_cursor, 1u);
        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
        if (w >= p.nq) break;
        walk_fast_one<METRIC, STEPS, false, true>(p, walk_query_of(p, w), smem, p.ovf_count, p.ovf_list, bitmap);
        wave_sync();
    }
}

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
// Uses v[88:95] as scratch (contiguous pairs are needed for the packed sums).
#define GBNNS_P_SUB(T)                                                                  \
    "v_pk_add_f32 %[pa" #T "], %[a" #T "], %[qa" #T "] neg_lo:[0,1] neg_hi:[0,1]\n\t" \
    "v_pk_add_f32 %[pb" #T "], %[b" #T "], %[qb" #T "] neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define GBNNS_P_MUL(T)                                           \
    "v_pk_mul_f32 %[pa" #T "], %[pa" #T "], %[pa" #T "]\n\t" \
    "v_pk_mul_f32 %[pb" #T "], %[pb" #T "], %[pb" #T "]\n\t"
#define GBNNS_P_ACC(T)                                         \
    "v_pk_add_f32 v[88:89], v[88:89], %[pa" #T "]\n\t"      \
    "v_pk_add_f32 v[90:91], v[90:91], %[pb" #T "]\n\t"
template <typename QP>
__device__ __forceinline__ float l2_pair_from_regs(const RowRegs<4>& r, QP qh) {
    f32x2 pa0, pb0, pa1, pb1, pa2, pb2, pa3, pb3;  // squared differences of this lane's four steps
    float d;
#define GBNNS_Q(T)                                                                               \
    [a##T] "v"(f32x2{r.v[T].x, r.v[T].y}), [b##T] "v"(f32x2{r.v[T].z, r.v[T].w}),                \
    [qa##T] "v"(f32x2{qh[T].x, qh[T].y}), [qb##T] "v"(f32x2{qh[T].z, qh[T].w})
    asm(GBNNS_P_SUB(0) GBNNS_P_SUB(1) GBNNS_P_MUL(0) GBNNS_P_MUL(1) GBNNS_P_SUB(2) GBNNS_P_SUB(3) GBNNS_P_MUL(2) GBNNS_P_MUL(3)
        "v_pk_add_f32 v[92:93], %[pa0], %[pa1]\n\t"      // even lane: steps 0..3 (0 + e*e == e*e)
        "v_pk_add_f32 v[94:95], %[pb0], %[pb1]\n\t"
        "v_pk_add_f32 v[92:93], v[92:93], %[pa2]\n\t"
        "v_pk_add_f32 v[94:95], v[94:95], %[pb2]\n\t"
        "v_pk_add_f32 v[92:93], v[92:93], %[pa3]\n\t"
        "v_pk_add_f32 v[94:95], v[94:95], %[pb3]\n\t"
        "s_nop 1\n\t"                                    // VALU write -> DPP read of the same register
        "v_mov_b32_dpp v88, v92 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp v89, v93 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp v90, v94 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp v91, v95 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
        GBNNS_P_ACC(0) GBNNS_P_ACC(1) GBNNS_P_ACC(2) GBNNS_P_ACC(3)   // odd lane: steps 4..7 on top
        "v_add_f32 %[d], v88, v89\n\t"
        "v_add_f32 %[d], %[d], v90\n\t"
        "v_add_f32 %[d], %[d], v91"
        : [d] "=&v"(d), [pa0] "=&v"(pa0), [pb0] "=&v"(pb0), [pa1] "=&v"(pa1), [pb1] "=&v"(pb1), [pa2] "=&v"(pa2),
          [pb2] "=&v"(pb2), [pa3] "=&v"(pa3), [pb3] "=&v"(pb3)
        : GBNNS_Q(0), GBNNS_Q(1), GBNNS_Q(2), GBNNS_Q(3)
        : "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95");
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
    float4 e[H];
#pragma unroll
    for (int t = 0; t < H; ++t) {
        const float4 a = r.v[t], b = qh[t];
        const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, dw = a.w - b.w;
        e[t] = make_float4(dx * dx, dy * dy, dz * dz, dw * dw);
    }
    float4 sa = e[0];
#pragma unroll
    for (int t = 1; t < H; ++t) sa = make_float4(sa.x + e[t].x, sa.y + e[t].y, sa.z + e[t].z, sa.w + e[t].w);
    auto from_even = [](float v) {  // lanes 2i and 2i + 1 <- lane 2i   (quad_perm [0, 0, 2, 2])
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xA0, 0xf, 0xf, false));
    };
    float4 sb = make_float4(from_even(sa.x), from_even(sa.y), from_even(sa.z), from_even(sa.w));
#pragma unroll
    for (int t = 0; t < H; ++t) sb = make_float4(sb.x + e[t].x, sb.y + e[t].y, sb.z + e[t].z, sb.w + e[t].w);
    return ((sb.x + sb.y) + sb.z) + sb.w;
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
              [nb2] "=&s"(nb2), [hr] "=&s"(hr), [t0] "=&v"(t0), [fm] "=&s"(fm), [node] "=&s"(node), [pred] "=&s"(pred)
            : [flo] "v"(F.lo[0]), [fhi] "v"(F.hi[0]), [h1] "s"(rfl(h1)), [n1] "s"(rfl(n1)), [h2] "s"(rfl(h2)), [n2] "s"(rfl(n2)), [ts] "s"(rfl((uint32_t)tsize))
            : "vcc", "scc");
        h2k = hr;
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

// AUX: the auxiliary-graph walk (search_function.h:73-89): a hop expands the node's auxiliary row first (while
// hops < hops_bound), then -- unless llf and that step inserted something -- its main row.
// BITMAP: visited set = one bit per node in HBM (`bitmap`, this wavefront's slot), see walk_bitmap_kernel.
template <int METRIC, int STEPS, bool OFF32, int R, bool ONE_CHUNK = false, bool AUX = false, bool BITMAP = false>
__device__ __forceinline__ void walk_reg_one(const WalkParams& p, uint32_t qi, unsigned char* smem,
                                             uint32_t* ovf_count, uint32_t* ovf_list, uint32_t* bitmap = nullptr) {
    static_assert(!(AUX && ONE_CHUNK), "auxiliary rows have their own length");
#ifdef GBNNS_STAMPS
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned int probe_iters = 0;  // (the hand-scheduled probe does not count its iterations)
    unsigned int hist[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // [0..7] survivors per hop (0,1,2,3,4,5-8,9-16,17+), [8] merges, [9] merge fallbacks, [10] sequential offers, [11] fast selects
    STAMP(t_begin)
#endif
    constexpr bool kEarlyLoad = (STEPS > 0);  // speculative row loads (METRIC 1 is instantiated with STEPS 0 or 8 only)
    // 128-byte rows: two lanes per neighbour (lane = 2 * slot + half), 32 adjacency slots per pass
    constexpr bool kPair = (STEPS == 8) || ((STEPS == 12 || STEPS == 16) && METRIC == 0);  // 128-byte rows; 192- / 256-byte rows with L2
    constexpr bool kAlt = (STEPS == 8 && METRIC == 1);        // dot metric: even / odd 16-B pieces instead of halves
    constexpr int kQSteps = kPair ? STEPS / 2 : STEPS;        // 16-B steps of the row one lane holds
    constexpr uint32_t kRowBytes = (uint32_t)STEPS * 16u;
    constexpr uint32_t kChunk = kPair ? 32u : 64u;           // adjacency slots per pass
    constexpr uint64_t kSlotLanes = kPair ? 0x5555555555555555ull : ~0ull;  // lanes that own a slot
    const int lane = lane_id();
    const uint32_t slot = kPair ? (uint32_t)lane >> 1 : (uint32_t)lane;    // adjacency slot of this lane
    const uint32_t half = kPair ? (uint32_t)lane & 1u : 0u;
    const int ef = p.ef;
    uint64_t* tie = reinterpret_cast<uint64_t*>(smem);
    uint64_t* stage = tie + kRegTieCap;  // scatter buffer of the batch merge
    float* qf = reinterpret_cast<float*>(stage + reg_stage_slots(R));
    uint32_t* hash = reinterpret_cast<uint32_t*>(qf + p.dstride);
    const float4* qs = reinterpret_cast<const float4*>(qf);
    const uint32_t cap = p.hash_cap;  // any size: slot = mulhi(id * C, cap)
    const uint32_t hash_lds = (uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(hash));

    // OFF32 instantiations serve "compact" indexes (tables < 4 GiB and n < 2^24, walk_off32): 32-bit byte
    // offsets and the packed visited set (24-bit ids, five per 16-byte bucket)
    constexpr bool packed = OFF32;
    // (first pass of a compact index: the host may ask for the quotient form of the table -- p.vs_shr, GBNNS_VS_ASM)
    const uint32_t vs_shr = (packed && !BITMAP && !AUX) ? p.vs_shr : 0u;
    const uint32_t nbuckets = vs_shr ? cap / 7u - kStashBuckets : (packed ? cap / 5u : cap >> 2);
    if constexpr (BITMAP) { for (uint32_t i = lane; i < p.bitmap_words; i += 64) bitmap[i] = 0u; }
    else if (vs_shr) quotient_table_init(hash, nbuckets, lane);
    else if constexpr (packed) packed_table_init(hash, nbuckets, 0u, lane);
    else for (uint32_t i = lane; i < cap; i += 64) hash[i] = kInvalidId;
    for (uint32_t i = lane; i < p.dstride; i += 64)
        qf[i] = (i < p.dim) ? p.q[(size_t)qi * p.qstride + i] : 0.f;
    wave_sync();

    // the query stays in registers (every lane holds all of it): the occupancy scan shows the walk is
    // issue-bound from ~14 wavefronts/CU, so the registers cost nothing and each hop saves 8 LDS reads
    RowRegs<kQSteps> qreg;
    if constexpr (kEarlyLoad) {
#pragma unroll
        for (int t = 0; t < kQSteps; ++t) qreg.v[t] = kAlt ? qs[2 * t + half] : qs[kQSteps * half + t];
    }

    RegList<R> L;  // this lane's R list entries
    L.clear();
    int size = 1, tsize = 0, hops = 0, dist_calc = 1, edges = 0;
    uint32_t worst;                                // hi of lane size-1 (wave-uniform)
    const uint32_t entry = p.entries ? p.entries[qi] : 0u;
    if (entry >= p.n) { write_bad_entry(p, qi, lane); return; }
    {
        const float d0 = walk_dist<METRIC, STEPS>(qs, row_ptr<OFF32>(p.db, entry, p.dstride), p.dim);
        worst = fkey(d0);
        if (lane == 0) {
            L.hi[0] = worst;
            L.lo[0] = entry << 1;
            if constexpr (BITMAP) bitmap[entry >> 5] = 1u << (entry & 31u);
            else if (vs_shr) quotient_table_put_first(hash, nbuckets, entry, vs_shr);
            else if constexpr (packed) packed_table_put_first(hash, nbuckets, entry);
            else hash[4u * __umulhi(entry * 0x9E3779B1u, nbuckets)] = entry;  // first slot of its bucket
        }
        wave_sync();
    }

    int status = 0;  // 0 = walking, 1 = finished, 2 = handed over to the general kernel
    // Adjacency prefetch: when a node is picked, the row of the entry that will be picked next IF
    // this expansion inserts nothing closer is requested too.  The load stays in flight behind this
    // hop's vector gathers (loads retire in order), so a correct guess removes one of the two
    // dependent memory round trips of the next hop; a wrong guess costs one 128-B row.
    uint32_t pf_node = kInvalidId, pf_val = kInvalidId;
    while (true) {
        STAMP(t0)
        // ---- next node to expand: closest unexpanded entry, ties -> largest id -------------
        uint64_t mu[R];
        int p1 = -1, p2 = -1;  // ranks of the two closest unexpanded entries
        uint32_t node = 0, pred = kInvalidId;
        bool picked = false;
        if constexpr (R == 1) {
            // one list register: straight branches to the rare path (the kernel is instruction-issue
            // bound -- a flag-and-merge formulation costs ~20 more scalar instructions per hop)
            mu[0] = __ballot(!(L.lo[0] & 1u)) & RegList<R>::lane_mask(0, ef);
            // one select + one branch; the empty asm keeps the compiler from folding it back into
            // `mu == 0 || tsize != 0`, which it evaluates with five 64-bit mask instructions
            uint64_t fastm = tsize == 0 ? mu[0] : 0ull;
            asm("" : "+s"(fastm));
            if (fastm == 0) goto slow_select;
            {
                const int q1 = __ffsll((unsigned long long)fastm) - 1;
                const uint64_t m2 = clear_bit64(fastm, q1);
                p1 = q1;
                if (m2) {
                    const int q2 = __ffsll((unsigned long long)m2) - 1;
                    if (readlane_u32(L.hi[0], q1) == readlane_u32(L.hi[0], q2)) goto slow_select;
                    pred = readlane_u32(L.lo[0], q2) >> 1;
                }
                node = readlane_u32(L.lo[0], q1) >> 1;
                if (lane == q1) L.lo[0] |= 1u;
#ifdef GBNNS_STAMPS
                hist[11] += 1;
#endif
                goto have_node;
            }
        slow_select:
            p1 = mu[0] ? __ffsll((unsigned long long)mu[0]) - 1 : -1;
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                mu[r] = __ballot(!(L.lo[r] & 1u)) & RegList<R>::lane_mask(r, ef);
                uint64_t m = mu[r];
                if (p1 < 0 && m) {
                    p1 = r * 64 + __ffsll((unsigned long long)m) - 1;
                    m &= m - 1;
                }
                if (p1 >= 0 && p2 < 0 && m) p2 = r * 64 + __ffsll((unsigned long long)m) - 1;
            }
            if (p1 >= 0 && tsize == 0) {
                // common case: the two closest unexpanded entries have different distances
                if (p2 >= 0) {
                    if (L.hi_at(p1) != L.hi_at(p2)) {
                        picked = true;
                        node = L.lo_at(p1) >> 1;
                        pred = L.lo_at(p2) >> 1;
                        L.mark_expanded(p1, lane);
                    }
                } else {
                    picked = true;
                    node = L.lo_at(p1) >> 1;
                    L.mark_expanded(p1, lane);
                }
            }
        }
        if (!picked) {
            // rare: equal-distance run among the unexpanded entries, a non-empty tie list, or the end
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
            bool from_tie = false;
            if (tsize > 0 && (best < 0 || hi_p == worst)) {
                // tie entries all sit at the worst distance: the largest id among them competes
                uint32_t v = (lane < tsize) ? key_id(tie[lane]) + 1u : 0u;
                int w = lane;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const uint32_t ov = (uint32_t)__shfl_xor((int)v, off);
                    const int ow = __shfl_xor(w, off);
                    if (ov > v) { v = ov; w = ow; }
                }
                // every lane now holds the same (v, w); tell the compiler so (keeps loop state scalar)
                v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
                w = __builtin_amdgcn_readfirstlane(w);
                const uint32_t lid = (best >= 0) ? (L.lo_at(best) >> 1) : 0u;
                if (best < 0 || v - 1u > lid) {
                    from_tie = true;
                    node = v - 1u;
                    if (lane == 0) tie[w] = tie[tsize - 1];
                    tsize -= 1;
                    wave_sync();
                }
            }
            if (!from_tie) {
                if (best < 0) { status = 1; break; }
                node = L.lo_at(best) >> 1;
                L.mark_expanded(best, lane);
            }
        }
    have_node:
        STAMP(t1)
        STAMP_ADD(0, t0, t1)

        // ---- adjacency row of `node` (prefetched or loaded now), then the prefetch for the next hop
        const uint32_t* row = reinterpret_cast<const uint32_t*>(
            row_ptr<OFF32>(reinterpret_cast<const float*>(p.ell), node, p.ell_stride));
        uint32_t nb0;
        if (node == pf_node) nb0 = pf_val;
        else nb0 = (slot < p.ell_stride) ? row[slot] : kInvalidId;
        // consume nb0 BEFORE issuing the prefetch: the wait for a (conditionally issued) row load
        // must not also cover the younger prefetch load
        const uint64_t mv0 = __ballot(nb0 != kInvalidId);
        STAMP(t2)
        STAMP_ADD(1, t1, t2)
        pf_node = pred;
        if (pred != kInvalidId)
            pf_val = (slot < p.ell_stride) ? reinterpret_cast<const uint32_t*>(row_ptr<OFF32>(reinterpret_cast<const float*>(p.ell), pred, p.ell_stride))[slot] : kInvalidId;
        STAMP(t3)
        STAMP_ADD(2, t2, t3)

        // ---- expand: neighbours in list order, 64 per pass --------------------------------------
        // ONE_CHUNK (rows of at most 64 slots): a single pass, the loop and its bookkeeping fold away
        // AUX: two rows per hop -- g = 0 the auxiliary row (makeStep :73-80), g = 1 the main row (:82-89)
        bool found = false;  // makeStep's flag (:34); only read when AUX
        for (int g = AUX ? ((uint32_t)hops < p.hops_bound ? 0 : 1) : 1; g < 2; ++g) {
        const bool is_aux = AUX && g == 0;
        if (AUX && g == 1 && found && p.llf) break;
        const uint32_t* grow = row;
        uint32_t gstride = p.ell_stride;
        if (is_aux) {
            grow = reinterpret_cast<const uint32_t*>(row_ptr<OFF32>(reinterpret_cast<const float*>(p.aux_ell), node, p.aux_stride));
            gstride = p.aux_stride;
        }
        for (uint32_t c = 0; c < (ONE_CHUNK ? kChunk : gstride); c += kChunk) {
            uint32_t nb = nb0;
            uint64_t mv = mv0;
            if (c || is_aux) {
                nb = (c + slot < gstride) ? grow[c + slot] : kInvalidId;
                mv = __ballot(nb != kInvalidId);
            }
            if (!mv) break;
            if constexpr (!BITMAP)
                if ((uint32_t)dist_calc + 64u > p.hash_limit) { status = 2; break; }
            const bool valid = nb != kInvalidId;
            edges += __popcll(mv & kSlotLanes);
            // row loads go out before the visited test: its LDS round trips overlap the memory latency
            // (rows of already-visited neighbours are fetched in vain -- we are not bandwidth bound)
            RowRegs<kQSteps> rr;
            uint32_t roff = 0;  // row byte offset, kept live past the loads (see below)
            if constexpr (kEarlyLoad) {
                // (see walk_reg_big_one: every lane loads, empty slots read row 0; measured: the pair form gains in the one-pass
                // hop only, 12- / 16-step rows one lane each wherever their 48 / 64 row registers would be carried around the loop)
                constexpr bool kAllLanes = (kPair && ONE_CHUNK) || (!kPair && kQSteps >= 12);
                const uint32_t nbl = kAllLanes ? (valid ? nb : 0u) : nb;
                const bool ld = kAllLanes || valid;
                if constexpr (OFF32) {
                    roff = kPair ? nbl * kRowBytes + half * (kAlt ? 16u : kRowBytes / 2u) : nbl * (p.dstride * 4u);
                    const float* rp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.db) + roff);
                    if constexpr (kAlt) { if (ld) load_row_alt(rr, rp); }
                    else { if (ld) load_row<kQSteps>(rr, rp); }
                } else {
                    const float* rp = row_ptr<OFF32>(p.db, nbl, p.dstride) + half * (kAlt ? 4u : kRowBytes / 8u);
                    if constexpr (kAlt) { if (ld) load_row_alt(rr, rp); }
                    else { if (ld) load_row<kQSteps>(rr, rp); }
                }
            }
            // pair form: the even lane of a pair tests / claims the id, the odd lane ends up with the distance
            uint64_t mclaimed;
            if constexpr (BITMAP) {
                bool fr = false;
                if (valid && (!kPair || half == 0u)) {  // the lane that owns the slot tests and sets the bit
                    const uint32_t bit = 1u << (nb & 31u);
                    fr = !(atomicOr(&bitmap[nb >> 5], bit) & bit);
                }
                mclaimed = __ballot(fr);
            } else if (vs_shr) {
                uint64_t movf;
                mclaimed = visited_claim_mask_quotient(hash_lds, nbuckets, nb, mv & kSlotLanes, vs_shr, movf);
                if (__builtin_expect(movf != 0, 0)) {
                    if (!stash_claim(hash_lds, nbuckets, movf, nb, mclaimed, lane)) { status = 2; break; }
                }
            } else if constexpr (packed) mclaimed = visited_claim_mask_packed(hash_lds, nbuckets, nb, mv & kSlotLanes);
            else mclaimed = visited_claim_mask(hash_lds, nbuckets, nb, mv & kSlotLanes);
            const uint64_t mfresh = kPair ? (mclaimed << 1) : mclaimed;
            const bool fresh = __builtin_amdgcn_inverse_ballot_w64(mfresh);
            STAMP(t4)
            STAMP_ADD(3, t3, t4)
            uint32_t dk = 0xFFFFFFFFu;
            if constexpr (kEarlyLoad) {
                if constexpr (kAlt) {
                    const uint32_t kd = fkey(dot_pair_from_regs(rr, qreg.v));  // all lanes; odd lanes hold distances
                    dk = fresh ? kd : 0xFFFFFFFFu;
                } else if constexpr (kPair && STEPS == 8) {
                    const uint32_t kd = fkey_sumsq(l2_pair_from_regs(rr, qreg.v));  // all lanes; odd lanes hold distances
                    dk = fresh ? kd : 0xFFFFFFFFu;
                } else if constexpr (kPair) {
                    const uint32_t kd = fkey_sumsq(l2_pair_from_regs_wide<kQSteps>(rr, qreg.v));
                    dk = fresh ? kd : 0xFFFFFFFFu;
                } else if constexpr (STEPS == 8) {
                    if (fresh) dk = fkey_sumsq(l2_from_regs8(rr, qreg.v));
                } else {
                    if (fresh) dk = fkey(l2_from_regs<STEPS>(rr, qreg.v));
                }
                // The address register must not double as a load destination: if it does, the next
                // hop's address computation has to wait for every load in flight (vmcnt(0)), which
                // serialises the adjacency prefetch with the gather (tools/check_isa.sh).
                asm volatile("" ::"v"(roff));
            } else {
                if (fresh) dk = fkey(walk_dist<METRIC, STEPS>(qs, row_ptr<OFF32>(p.db, nb, p.dstride), p.dim));
            }
            dist_calc += __popcll(mfresh);
            const bool offer_it = fresh && (size < ef || dk < worst);
            uint64_t m = __ballot(offer_it);
            if (AUX && m) found = true;  // the first of them is inserted whatever happens to the others
            STAMP(t5)
            STAMP_ADD(4, t4, t5)
            // several survivors: merge them in one pass (falls through to the sequential offers on a
            // boundary tie); reference order = one by one in list order (search_function.h:31-37)
#ifdef GBNNS_STAMPS
            {
                const int ns_ = __popcll(m);
                hist[ns_ <= 4 ? ns_ : (ns_ <= 8 ? 5 : (ns_ <= 16 ? 6 : 7))] += 1;
            }
#endif
            {
                if ((m & (m - 1)) != 0) {
                    bool merged;
                    if constexpr (R == 1) merged = reg_merge(m, offer_it, dk, nb, L, size, worst, tsize, stage, ef, lane);
                    else merged = reg_merge_multi<R>(m, offer_it, dk, nb, L, size, worst, tsize, stage, ef, lane);
                    if (merged) {
                        m = 0;
#ifdef GBNNS_STAMPS
                        hist[8] += 1;
                    } else {
                        hist[9] += 1;
#endif
                    }
                }
            }
#ifdef GBNNS_STAMPS
            hist[10] += __popcll(m);
#endif
            while (m) {
                const int l = __ffsll((unsigned long long)m) - 1;
                m &= m - 1;
                if (!reg_offer<R>(readlane_u32(dk, l), readlane_u32(nb, l) << 1, L, size, worst, tsize, tie, ef, lane)) {
                    status = 2;
                    break;
                }
            }
            STAMP(t6)
            STAMP_ADD(5, t5, t6)
            if (status) break;
        }
        if (status) break;
        }
        if (status) break;
        hops += 1;
    }
#ifdef GBNNS_STAMPS
    {
        STAMP(t_end)
        seg[6] = t_end - t_begin;
        if (lane == 0 && p.stamps)
        {
            for (int i = 0; i < 7; ++i) atomicAdd(p.stamps + i, seg[i]);
            for (int i = 0; i < 12; ++i) atomicAdd(p.stamps + 8 + i, (unsigned long long)hist[i]);
            atomicAdd(p.stamps + 20, (unsigned long long)probe_iters);
        }
    }
#endif

    if (status == 2) {
        if (lane == 0) {
            const uint32_t slot = atomicAdd(ovf_count, 1u);
            ovf_list[slot] = qi;
        }
        return;
    }
    reg_write_results<R>(p, qi, L, size, hops, dist_calc, edges, lane);
    if (p.rr_db) {
        const int kept = size < p.k ? size : p.k;
        fused_rerank(p, qi, kept, smem, lane, [&](int rank) { return reg_id_at_rank<R>(L, rank); });
    }
}

// ---- generic walk for 128 < ef <= 1024: walk_reg_one's hop around the two-list result structure (BigList) ---------------
//
// Every shape the hot instances do not take (256-byte rows, the dot metric, adjacency rows of more than one pass,
// auxiliary graphs, large indexes, the HBM-bitmap first pass): same expansion as walk_reg_one, but the result list is
// the base list in LDS + the front list in one register, so that selection and insertion cost what they cost at
// ef <= 64 whatever ef is (the R-register lists spent 28 % of a hop selecting and 30 % inserting at ef = 300).
// LDS: [BigList: big_list_fixed_bytes(ef)][query: dstride floats][visited set | (BITMAP) re-rank scratch].
// ONE_PASS: adjacency rows of one pass (the host checks ell_stride), no auxiliary graph -- the hop is straight-line code.
template <int METRIC, int STEPS, bool OFF32, bool AUX = false, bool BITMAP = false, bool ONE_PASS = false>
__device__ __forceinline__ void walk_reg_big_one(const WalkParams& p, uint32_t qi, unsigned char* smem,
                                                 uint32_t* ovf_count, uint32_t* ovf_list, uint32_t* bitmap = nullptr) {
    constexpr bool kEarlyLoad = (STEPS > 0);
    constexpr bool kPair = (STEPS == 8) || ((STEPS == 12 || STEPS == 16) && METRIC == 0);  // 128-byte rows, and 192- / 256-byte rows with L2: two lanes per neighbour
    constexpr bool kAlt = (STEPS == 8 && METRIC == 1);
    constexpr int kQSteps = kPair ? STEPS / 2 : STEPS;        // row steps (16 bytes) per lane
    constexpr uint32_t kRowBytes = (uint32_t)STEPS * 16u;
    constexpr uint32_t kChunk = kPair ? 32u : 64u;
    constexpr uint64_t kSlotLanes = kPair ? 0x5555555555555555ull : ~0ull;
    const int lane = lane_id();
#ifdef GBNNS_STAMPS  // diagnostic build: cycles per segment of the hop (tools/stamps.py)
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    STAMP(t_begin)
    unsigned long long t_prev = t_begin;
#endif
    const uint32_t slot = kPair ? (uint32_t)lane >> 1 : (uint32_t)lane;
    const uint32_t half = kPair ? (uint32_t)lane & 1u : 0u;
    const int ef = p.ef;
    BigList B;
    B.init(smem, ef);
    float* qf = reinterpret_cast<float*>(smem + big_list_fixed_bytes(ef));
    unsigned char* after_q = reinterpret_cast<unsigned char*>(qf + p.dstride);  // visited set, or re-rank scratch
    uint32_t* hash = reinterpret_cast<uint32_t*>(after_q);
    const float4* qs = reinterpret_cast<const float4*>(qf);
    const uint32_t cap = p.hash_cap;
    const uint32_t hash_lds = (uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(hash));
    constexpr bool packed = OFF32;
    // (first pass of a compact index: the host may ask for the quotient form of the table -- p.vs_shr, GBNNS_VS_ASM)
    const uint32_t vs_shr = (packed && !BITMAP && !AUX) ? p.vs_shr : 0u;
    const uint32_t nbuckets = vs_shr ? cap / 7u - kStashBuckets : (packed ? cap / 5u : cap >> 2);
    if constexpr (BITMAP) {
        // 16 bytes per lane and store (n / 8 bytes per query: 150 KB at n = 1.2 M); not unrolled: the unrolled form's address
        // registers were the kernel's register peak
        uint4* b4 = reinterpret_cast<uint4*>(bitmap);
        const uint32_t n4 = p.bitmap_words >> 2;
#pragma clang loop unroll(disable)
        for (uint32_t i = lane; i < n4; i += 64) b4[i] = make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t i = (n4 << 2) + lane; i < p.bitmap_words; i += 64) bitmap[i] = 0u;
    }
    else if (vs_shr) quotient_table_init(hash, nbuckets, lane);
    else if constexpr (packed) packed_table_init(hash, nbuckets, 0u, lane);
    else for (uint32_t i = lane; i < cap; i += 64) hash[i] = kInvalidId;
    for (uint32_t i = lane; i < p.dstride; i += 64)
        qf[i] = (i < p.dim) ? p.q[(size_t)qi * p.qstride + i] : 0.f;
    wave_sync();
    RowRegs<kQSteps> qreg;
    if constexpr (kEarlyLoad) {
#pragma unroll
        for (int t = 0; t < kQSteps; ++t) qreg.v[t] = kAlt ? qs[2 * t + half] : qs[kQSteps * half + t];
    }

    int hops = 0, dist_calc = 1, edges = 0;
    // (readfirstlane: every lane computes the same entry id / distance; the compiler must know they are wave-uniform)
    const uint32_t entry = (uint32_t)__builtin_amdgcn_readfirstlane((int)(p.entries ? p.entries[qi] : 0u));
    if (entry >= p.n) { write_bad_entry(p, qi, lane); return; }
    {
        uint32_t k0;
        if constexpr (kPair) {
            // the entry's distance in the pair form, on the query registers (lanes 0 / 1 would do; every pair computes it):
            // the one-lane form reads the whole row and the whole query into registers and was the kernel's register peak
            RowRegs<kQSteps> er;
            const float* rp = row_ptr<OFF32>(p.db, entry, p.dstride) + half * (kAlt ? 4u : kRowBytes / 8u);
            if constexpr (kAlt) load_row_alt(er, rp);
            else load_row<kQSteps>(er, rp);
            uint32_t kd;
            if constexpr (kAlt) kd = fkey(dot_pair_from_regs(er, qreg.v));
            else if constexpr (STEPS == 8) kd = fkey_sumsq(l2_pair_from_regs(er, qreg.v));
            else kd = fkey_sumsq(l2_pair_from_regs_wide<kQSteps>(er, qreg.v));
            k0 = readlane_u32(kd, 1);  // odd lanes hold the distance
        } else {
            k0 = fkey(walk_dist<METRIC, STEPS>(qs, row_ptr<OFF32>(p.db, entry, p.dstride), p.dim));
        }
        B.worst = B.fworst = (uint32_t)__builtin_amdgcn_readfirstlane((int)k0);
        B.F.hi[0] = lane == 0 ? B.worst : B.F.hi[0];
        B.F.lo[0] = lane == 0 ? entry << 1 : B.F.lo[0];
        if (lane == 0) {
            if constexpr (BITMAP) bitmap[entry >> 5] = 1u << (entry & 31u);
            else if (vs_shr) quotient_table_put_first(hash, nbuckets, entry, vs_shr);
            else if constexpr (packed) packed_table_put_first(hash, nbuckets, entry);
            else hash[4u * __umulhi(entry * 0x9E3779B1u, nbuckets)] = entry;
        }
        wave_sync();
    }

    int status = 0;  // 0 = walking, 1 = finished, 2 = handed over
    uint32_t pf_node = kInvalidId, pf_val = kInvalidId;
    while (true) {
        uint32_t node, pred, h2;
        STAMP(t0)
        STAMP_ADD(7, t_prev, t0)
        if (!B.select(node, pred, h2, lane)) { status = 1; break; }
        STAMP(t1)
        STAMP_ADD(0, t0, t1)
        // ---- adjacency row of `node` (prefetched or loaded now), then the prefetch for the next hop
        const uint32_t* row = reinterpret_cast<const uint32_t*>(
            row_ptr<OFF32>(reinterpret_cast<const float*>(p.ell), node, p.ell_stride));
        uint32_t nb0;
        if (node == pf_node) nb0 = pf_val;
        else nb0 = (slot < p.ell_stride) ? row[slot] : kInvalidId;
        const uint64_t mv0 = __ballot(nb0 != kInvalidId);  // consumed BEFORE the prefetch is issued
        STAMP(t2)
        STAMP_ADD(1, t1, t2)
        pf_node = pred;
        if (pred != kInvalidId)
            pf_val = (slot < p.ell_stride) ? reinterpret_cast<const uint32_t*>(row_ptr<OFF32>(reinterpret_cast<const float*>(p.ell), pred, p.ell_stride))[slot] : kInvalidId;

        bool found = false;  // makeStep's flag (:34); only read when AUX
        for (int g = AUX ? ((uint32_t)hops < p.hops_bound ? 0 : 1) : 1; g < 2; ++g) {
        const bool is_aux = AUX && g == 0;
        if (AUX && g == 1 && found && p.llf) break;
        const uint32_t* grow = row;
        uint32_t gstride = p.ell_stride;
        if (is_aux) {
            grow = reinterpret_cast<const uint32_t*>(row_ptr<OFF32>(reinterpret_cast<const float*>(p.aux_ell), node, p.aux_stride));
            gstride = p.aux_stride;
        }
        for (uint32_t c = 0; c < (ONE_PASS ? 1u : gstride); c += kChunk) {
            uint32_t nb = nb0;
            uint64_t mv = mv0;
            if (c || is_aux) {
                nb = (c + slot < gstride) ? grow[c + slot] : kInvalidId;
                mv = __ballot(nb != kInvalidId);
            }
            if (!mv) break;
            if constexpr (!BITMAP)
                if ((uint32_t)dist_calc + 64u > p.hash_limit) { status = 2; break; }
            const bool valid = nb != kInvalidId;
            STAMP(t3)
            if (c == 0 && !is_aux) { STAMP_ADD(2, t2, t3) }
            edges += __popcll(mv & kSlotLanes);
            RowRegs<kQSteps> rr;
            uint32_t roff = 0;
            if constexpr (kEarlyLoad) {
                // Rows of 12 / 16 steps (48 / 64 registers per lane): EVERY lane loads -- empty slots read row 0, all of them
                // the same lines -- so that the row registers are defined by this pass alone.  Loaded under `if (valid)`
                // the other lanes keep "the previous value", the compiler carries 64 registers around the hop loop and
                // copies them twice per hop (measured in the code object: 2 x 32 v_mov_b64 per hop on 256-byte rows).
                constexpr bool kAllLanes = kPair || kQSteps >= 12;
                const uint32_t nbl = kAllLanes ? (valid ? nb : 0u) : nb;
                const bool ld = kAllLanes || valid;
                if constexpr (OFF32) {
                    roff = kPair ? nbl * kRowBytes + half * (kAlt ? 16u : kRowBytes / 2u) : nbl * (p.dstride * 4u);
                    const float* rp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.db) + roff);
                    if constexpr (kAlt) { if (ld) load_row_alt(rr, rp); }
                    else { if (ld) load_row<kQSteps>(rr, rp); }
                } else {
                    const float* rp = row_ptr<OFF32>(p.db, nbl, p.dstride) + half * (kAlt ? 4u : kRowBytes / 8u);
                    if constexpr (kAlt) { if (ld) load_row_alt(rr, rp); }
                    else { if (ld) load_row<kQSteps>(rr, rp); }
                }
            }
            uint64_t mclaimed;
            if constexpr (BITMAP) {
                bool fr = false;
                if (valid && (!kPair || half == 0u)) {
                    const uint32_t bit = 1u << (nb & 31u);
                    fr = !(atomicOr(&bitmap[nb >> 5], bit) & bit);
                }
                mclaimed = __ballot(fr);
            } else if (vs_shr) {
                uint64_t movf;
                mclaimed = visited_claim_mask_quotient(hash_lds, nbuckets, nb, mv & kSlotLanes, vs_shr, movf);
                if (__builtin_expect(movf != 0, 0)) {
                    if (!stash_claim(hash_lds, nbuckets, movf, nb, mclaimed, lane)) { status = 2; break; }
                }
            } else if constexpr (packed) mclaimed = visited_claim_mask_packed(hash_lds, nbuckets, nb, mv & kSlotLanes);
            else mclaimed = visited_claim_mask(hash_lds, nbuckets, nb, mv & kSlotLanes);
            const uint64_t mfresh = kPair ? (mclaimed << 1) : mclaimed;
            STAMP(t4)
            STAMP_ADD(3, t3, t4)
            const bool fresh = __builtin_amdgcn_inverse_ballot_w64(mfresh);
            uint32_t dk = 0xFFFFFFFFu;
            if constexpr (kEarlyLoad) {
                if constexpr (kAlt) {
                    const uint32_t kd = fkey(dot_pair_from_regs(rr, qreg.v));
                    dk = fresh ? kd : 0xFFFFFFFFu;
                } else if constexpr (kPair && STEPS == 8) {
                    const uint32_t kd = fkey_sumsq(l2_pair_from_regs(rr, qreg.v));
                    dk = fresh ? kd : 0xFFFFFFFFu;
                } else if constexpr (kPair) {
                    const uint32_t kd = fkey_sumsq(l2_pair_from_regs_wide<kQSteps>(rr, qreg.v));
                    dk = fresh ? kd : 0xFFFFFFFFu;
                } else {
                    if (fresh) dk = fkey(l2_from_regs<STEPS>(rr, qreg.v));
                }
                asm volatile("" ::"v"(roff));  // the address register must not double as a load destination
            } else {
                if (fresh) dk = fkey(walk_dist<METRIC, STEPS>(qs, row_ptr<OFF32>(p.db, nb, p.dstride), p.dim));
            }
            dist_calc += __popcll(mfresh);
            const uint64_t m = (B.l + B.f < ef) ? mfresh : __ballot(fresh && dk < B.worst);
            STAMP(t5)
            STAMP_ADD(4, t4, t5)
            if (m) {
                if (AUX) found = true;  // the first of them is inserted whatever happens to the others
                // BigList::insert takes up to 32 survivors (its eviction step is one lane per split): the two halves of
                // a 64-slot pass go in one after the other -- same union, same rule (boundary ties fall back to the
                // sequential offers either way)
                const uint64_t ma = kPair ? m : (m & 0xFFFFFFFFull), mb = kPair ? 0ull : (m & 0xFFFFFFFF00000000ull);
                if (ma && !B.insert(ma, dk, nb, lane)) { status = 2; break; }
                if (!kPair && mb) {
                    const uint64_t mb2 = (B.l + B.f < ef) ? mb : (mb & __ballot(dk < B.worst));  // the first half may have lowered the bar
                    if (mb2 && !B.insert(mb2, dk, nb, lane)) { status = 2; break; }
                }
            }
            STAMP(t6)
            STAMP_ADD(5, t5, t6)
#ifdef GBNNS_STAMPS
            t_prev = t6;
#endif
        }
        if (status) break;
        }
        if (status) break;
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
        }
    }
#endif
    if (status == 2) {
        if (lane == 0) {
            const uint32_t s = atomicAdd(ovf_count, 1u);
            ovf_list[s] = qi;
        }
        return;
    }
    // (wide walked rows = the wide original rows of GIST: the re-rank keeps 24 loads in flight per lane, rerank_pairs_core)
    B.template finish<(STEPS >= 12 ? 24 : 8)>(p, qi, hops, dist_calc, edges, after_q, lane);
}

template <int METRIC, int STEPS, bool OFF32, bool RETRY, bool AUX = false, bool ONE_PASS = false>
__global__ __launch_bounds__(64) void walk_reg_big_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (RETRY) {
        retry_loop(p, [&](uint32_t qi) { walk_reg_big_one<METRIC, STEPS, OFF32, AUX>(p, qi, smem, p.ovf2_count, p.ovf2_list); });
    } else {
        walk_reg_big_one<METRIC, STEPS, OFF32, AUX, false, ONE_PASS>(p, walk_query_of(p, blockIdx.x), smem, p.ovf_count, p.ovf_list);
    }
}

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
#define GBNNS_HOT_END "\n8:\n\ts_mov_b64 exec, %[sv]"

#define GBNNS_HOT_CLOBBERS_40                                                                                           \
    "vcc", "scc", "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54",  \
        "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"

template <int METRIC = 0, bool QLDS = false, bool SPEC = !QLDS, typename QP>
__device__ __forceinline__ uint32_t hot_expand(const char* db_base, uint32_t roff, uint32_t nb, uint64_t valid, uint32_t lds_base,
                                               uint32_t nbuckets, QP qh, uint32_t qaddr, uint64_t& claimed, uint32_t shr, uint64_t& overflowed) {
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
    if constexpr (METRIC == 0) {
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
template <int R, bool WIDE = false, int METRIC = 0>
__device__ __forceinline__ void walk_hot_one(const WalkParams& p, uint32_t qi, unsigned char* smem) {
    const int lane = lane_id();
    const uint32_t slot = (uint32_t)lane >> 1, half = (uint32_t)lane & 1u;  // lane = 2 * adjacency slot + row half
    const int ef = p.ef;
    // R = 1 (QLDS): the query stays in LDS and hot_expand re-reads a lane's four pieces every hop -- 64 registers, 8 wavefronts
    // per SIMD; rows requested after the visited test.  R = 2: the query in registers, speculative row loads (rounds 1-3 layout).
    constexpr bool QLDS = R == 1 && GBNNS_HOT1_QLDS, SPEC = R == 1 ? GBNNS_HOT1_SPEC != 0 : true;
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
    uint32_t pf_node = kInvalidId, pf_val = kInvalidId;    // prefetch 1: the runner-up of the selection
    uint32_t pf2_node = kInvalidId, pf2_val = kInvalidId;  // prefetch 2: the closest new survivor (see below)
    uint32_t pf_valw = kInvalidId, pf2_valw = kInvalidId;  // WIDE: the rows' second halves

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
            const uint32_t kd = hot_expand<METRIC, QLDS, SPEC>(db_base, (nb << 7) + half * (METRIC == 0 ? 64u : 16u), nb, mv, hash_lds, nbuckets, qreg.v, qaddr, mclaimed, vs_shr, movf);
            if (__builtin_expect(movf != 0, 0)) {  // a probe sequence ran out (quotient form): the stash takes the id
                // (the two-pass instances are at their scalar-register budget -- tests/test_isa_contract.py -- and hand over)
                if constexpr (WIDE) return false;
                else if (!stash_claim(hash_lds, nbuckets, movf >> 1, nb, mclaimed, lane)) return false;  // (reported in the odd bits)
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
    reg_write_results<R>(p, qi, L, size, hops, dist_calc, edges, lane);
    if (p.rr_db) {
        const int kept = size < p.k ? size : p.k;
        fused_rerank(p, qi, kept, smem, lane, [&](int rank) { return reg_id_at_rank<R>(L, rank); });
    }
}

template <bool WIDE = false, int METRIC = 0>
__device__ __forceinline__ void walk_hot_big(const WalkParams& p, uint32_t qi, unsigned char* smem) {
    const int lane = lane_id();
#ifdef GBNNS_STAMPS  // diagnostic build: cycles per segment of the hop (tools/stamps.py)
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
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
                // (the two-pass instances are at their scalar-register budget -- tests/test_isa_contract.py -- and hand over)
                if constexpr (WIDE) return false;
                else if (!stash_claim(hash_lds, nbuckets, movf >> 1, nb, mclaimed, lane)) return false;  // (reported in the odd bits)
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
__global__ __launch_bounds__(64, GBNNS_HOT1_QLDS ? 8 : 7) void walk_hot_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef GBNNS_HOT1_CAP6  // experiment: 97+ scalar registers = 6 wavefronts per SIMD, leaving 128 vector registers per SIMD to other kernels
    asm volatile("" ::: "s96");
#endif
    walk_hot_one<1>(p, walk_query_of(p, blockIdx.x), smem);
}

// the same three for adjacency rows of 33 .. 64 slots (two expansion passes per hop)
__global__ __launch_bounds__(64, GBNNS_HOT1_QLDS ? 8 : 7) void walk_hotw_kernel(WalkParams p) {
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

template <int METRIC, int STEPS, bool OFF32, bool RETRY, int R, bool ONE_CHUNK = false, bool AUX = false>
__global__ __launch_bounds__(64) void walk_reg_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (RETRY) {
        retry_loop(p, [&](uint32_t qi) { walk_reg_one<METRIC, STEPS, OFF32, R, false, AUX>(p, qi, smem, p.ovf2_count, p.ovf2_list); });
    } else {
        walk_reg_one<METRIC, STEPS, OFF32, R, ONE_CHUNK, AUX>(p, walk_query_of(p, blockIdx.x), smem, p.ovf_count, p.ovf_list);
    }
}

// First pass with HBM visited bitmaps on register lists (128-byte rows, L2 or dot): persistent wavefronts.
template <int METRIC, int R>
__global__ __launch_bounds__(64) void walk_bitmap_reg_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* bitmap = p.fp_bitmap + (size_t)blockIdx.x * p.bitmap_words;
    while (true) {
        uint32_t w = 0;
        if (lane_id() == 0) w = atomicAdd(p.fp_cursor, 1u);
        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
        if (w >= p.nq) break;
        walk_reg_one<METRIC, 8, true, R, false, false, true>(p, walk_query_of(p, w), smem, p.ovf_count, p.ovf_list, bitmap);
        wave_sync();
    }
}

template <int METRIC, int STEPS = 8, bool ONE_PASS = false>
__global__ __launch_bounds__(64) void walk_bitmap_big_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* bitmap = p.fp_bitmap + (size_t)blockIdx.x * p.bitmap_words;
    while (true) {
        uint32_t w = 0;
        if (lane_id() == 0) w = atomicAdd(p.fp_cursor, 1u);
        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
        if (w >= p.nq) break;
        walk_reg_big_one<METRIC, STEPS, true, false, true, ONE_PASS>(p, walk_query_of(p, w), smem, p.ovf_count, p.ovf_list, bitmap);
        wave_sync();
    }
}

// Diagnostic kernel (tests only): runs one batch merge on a list / survivor set supplied by the host.
template <int R>
__global__ __launch_bounds__(64) void debug_merge_kernel(const uint64_t* entries, int size, const uint64_t* surv, int ef,
                                                         uint64_t* out, int* out_size) {
    __shared__ uint64_t stage[64 * R + 2];
    const int lane = lane_id();
    RegList<R> L;
    L.clear();
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (r * 64 + lane < size) {
            L.lo[r] = (uint32_t)entries[r * 64 + lane];
            L.hi[r] = (uint32_t)(entries[r * 64 + lane] >> 32);
        }
    const uint64_t sk = surv[lane];
    const bool is_surv = sk != ~0ull;
    const uint64_t m = __ballot(is_surv);
    uint32_t worst = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t t = readlane_u32(L.hi[r], (size - 1) & 63);
        if (((size - 1) >> 6) == r) worst = t;
    }
    int tsize = 0;
    bool ok;
    if constexpr (R == 1) ok = reg_merge(m, is_surv, (uint32_t)(sk >> 32), (uint32_t)sk >> 1, L, size, worst, tsize, stage, ef, lane);
    else ok = reg_merge_multi<R>(m, is_surv, (uint32_t)(sk >> 32), (uint32_t)sk >> 1, L, size, worst, tsize, stage, ef, lane);
#pragma unroll
    for (int r = 0; r < R; ++r) out[r * 64 + lane] = ((uint64_t)L.hi[r] << 32) | L.lo[r];
    if (lane == 0) {
        out_size[0] = size;
        out_size[1] = ok ? 1 : 0;
        out_size[2] = (int)worst;
    }
}

// ---- general kernel: exact for every input (any ef, any number of ties, any visited count) ----
//
// Persistent wavefronts pull query indices from the hand-over list.  Visited set = one bit per
// node in a per-slot global bitmap (cleared per query); result list and tie list in global
// memory (tie capacity n: every node can be in it at most once).

template <int METRIC>
__global__ __launch_bounds__(64) void walk_general_kernel(WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const uint32_t slot = blockIdx.x;
    float* qf = reinterpret_cast<float*>(smem);
    const float4* qs = reinterpret_cast<const float4*>(qf);
    uint32_t* bitmap = p.g_bitmap + (size_t)slot * 2u * p.bitmap_words;  // [visited bits][tie bits]
    const uint32_t n_ent = p.n_entries ? p.n_entries : 1u;
    uint64_t* keys = p.g_keys + (size_t)slot * ((size_t)p.ef + n_ent - 1u);  // one extra slot per extra entry point
    // tie set: one bit per node (second half of the slot's bitmap block); the host zeroes it when it allocates the
    // block, clear() leaves it all zero after every use
    TieBits tie{bitmap + p.bitmap_words, 0xFFFFFFFFu, 0u};
    const int ef = p.ef;
    const uint32_t total = p.all_general ? p.nq : *p.ovf2_count;
    if (slot == 0 && lane == 0 && total) atomicAdd(p.g_total, total);
    if (slot == 0 && lane < 7 && lane != 5 && p.next_ctrl) p.next_ctrl[lane] = 0u;  // control words of the next call (word 5 of block 0 is the persistent general-kernel total)

    while (true) {
        uint32_t w = 0;
        if (lane == 0) w = atomicAdd(p.g_cursor, 1u);
        w = (uint32_t)__shfl((int)w, 0);
        if (w >= total) break;
        const uint32_t qi = p.all_general ? w : p.ovf2_list[w];

        for (uint32_t i = lane; i < p.dstride; i += 64)
            qf[i] = (i < p.dim) ? p.q[(size_t)qi * p.qstride + i] : 0.f;
        wave_sync();

        {
            bool bad = false;  // an entry id outside the index: empty result, no row is touched
            for (uint32_t e = 0; e < n_ent; ++e) bad |= (p.entries ? p.entries[(size_t)qi * n_ent + e] : 0u) >= p.n;
            if (bad) {
                write_bad_entry(p, qi, lane);
                wave_sync();
                continue;
            }
        }
        WalkState st;
        st.size = 0; st.tsize = 0; st.first_un = 0; st.hops = 0; st.dist_calc = 1; st.edges = 0;
        // search_function.h:54-64: one walk per entry point -- fresh candidate set (every result so far counts as
        // expanded, the tie list is dropped) and fresh visited set; the result heap, hops and dist_calc carry
        // over; the entry's own distance is not counted and it is pushed without the size test (so the heap
        // stays one longer per extra entry point: makeStep pops once per push, :36-37)
        for (uint32_t e = 0; e < n_ent; ++e) {
        for (uint32_t i = lane; i < p.bitmap_words; i += 64) bitmap[i] = 0u;
        if (e > 0) {
            for (int i = lane; i < st.size; i += 64) keys[i] = keys[i] | 1ull;
            tie.clear(st.tsize, lane);
        }
        wave_sync();
        const uint32_t entry = p.entries ? p.entries[(size_t)qi * n_ent + e] : 0u;
        {
            const float d0 =
                metric_dist<METRIC>(qs, reinterpret_cast<const float4*>(p.db + (size_t)entry * p.dstride),
                                    p.dim);
            if (e == 0) {
                if (lane == 0) keys[0] = make_key(fkey(d0), entry);
                st.size = 1;
            } else {
                uint64_t ev;
                bool did;
                const int pos = list_insert(keys, st.size, 0x7FFFFFFF, make_key(fkey(d0), entry), ev, did, lane);
                st.first_un = pos;
            }
            if (lane == 0) bitmap[entry >> 5] = 1u << (entry & 31u);
            wave_sync();
        }
        uint32_t node;
        auto make_step = [&](const uint32_t* row, uint32_t stride, bool& found) {  // search_function.h:15-40
            for (uint32_t c = 0; c < stride; c += 64) {
                const uint32_t nb = (c + lane < stride) ? row[c + lane] : kInvalidId;
                const bool valid = nb != kInvalidId;
                const uint64_t mv = __ballot(valid);
                if (!mv) break;
                st.edges += __popcll(mv);
                bool fresh = false;
                if (valid) {
                    const uint32_t bit = 1u << (nb & 31u);
                    fresh = !(atomicOr(&bitmap[nb >> 5], bit) & bit);
                }
                uint32_t dk = 0xFFFFFFFFu;
                if (fresh)
                    dk = fkey(metric_dist<METRIC>(
                        qs, reinterpret_cast<const float4*>(p.db + (size_t)nb * p.dstride), p.dim));
                const uint64_t mf = __ballot(fresh);
                st.dist_calc += __popcll(mf);
                const uint32_t worst0 = key_hi(keys[st.size - 1]);
                uint64_t m = __ballot(fresh && (st.size < ef || dk < worst0));
                if (m) found = true;
                while (m) {
                    const int l = __ffsll((unsigned long long)m) - 1;
                    m &= m - 1;
                    const uint32_t dl = (uint32_t)__shfl((int)dk, l);
                    const uint32_t il = (uint32_t)__shfl((int)nb, l);
                    offer(keys, tie, st, ef, dl, il, lane);
                }
            }
        };
        while (select_candidate(keys, tie, st, node, lane)) {
            bool found = false;
            if (p.aux_ell && (uint32_t)st.hops < p.hops_bound)  // :73-80
                make_step(p.aux_ell + (size_t)node * p.aux_stride, p.aux_stride, found);
            if (!(found && p.llf))                                // :82-89
                make_step(p.ell + (size_t)node * p.ell_stride, p.ell_stride, found);
            st.hops += 1;
        }
        }  // entry points
        tie.clear(st.tsize, lane);  // leaves the tie bits all zero for the next query
        write_results(p, qi, keys, st, lane);
        if (p.rr_db) {
            const int kept = st.size < p.k ? st.size : p.k;
            fused_rerank(p, qi, kept, smem, lane, [&](int rank) { return key_id(keys[rank]); });
        }
        wave_sync();
    }
}

// ------------------------------------------------------------------------------------------
// re-rank (search_function.h:105-125 getRealNearest)
// ------------------------------------------------------------------------------------------
// One candidate per lane; each lane streams its own row with 16-B loads against the query staged
// in LDS.  Winner = strict minimum in pop order  <=>  min over (distance, pop index).

template <int METRIC>
__global__ __launch_bounds__(64) void rerank_kernel(RerankParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const uint32_t qi = blockIdx.x;
    float* qf = reinterpret_cast<float*>(smem);
    const float4* qs = reinterpret_cast<const float4*>(qf);
    for (uint32_t i = lane; i < p.dstride; i += 64)
        qf[i] = (i < p.dim) ? p.q[(size_t)qi * p.qstride + i] : 0.f;
    wave_sync();
    const int cnt = p.count[qi];
    const uint32_t* cand = p.cand + (size_t)qi * p.cand_stride;
    uint64_t bestk = ~0ull;
    for (int base = 0; base < cnt; base += 64) {
        const int r = base + lane;
        if (r < cnt) {
            uint32_t id = cand[r];
            id = id < p.n ? id : 0u;  // (never dereference an id outside the table)
            const float dv = metric_dist<METRIC>(
                reinterpret_cast<const float4*>(p.db + (size_t)id * p.dstride), qs, p.dim);
            const uint64_t kv = ((uint64_t)fkey(dv) << 32) | (uint32_t)r;
            bestk = kv < bestk ? kv : bestk;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t o = shfl_u64(bestk, lane ^ off);
        bestk = o < bestk ? o : bestk;
    }
    if (lane == 0) p.out[qi] = (cnt > 0) ? cand[(uint32_t)(bestk & 0xFFFFFFFFu)] : kInvalidId;
}

template <int METRIC>
__global__ __launch_bounds__(64) void rerank_pair_kernel(RerankParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const uint32_t qi = blockIdx.x;
    const int cnt = p.count[qi];
    const uint32_t* cand = p.cand + (size_t)qi * p.cand_stride;
    RerankSrc a{p.q, p.qstride, p.db, p.dstride, p.dim, p.n};
    const int win = (METRIC == 0 && p.dim >= 384u)
                        ? rerank_pairs_core<METRIC, 24>(a, qi, cnt, reinterpret_cast<float*>(smem), lane, [&](int r) { return cand[r]; })
                        : rerank_pairs_core<METRIC>(a, qi, cnt, reinterpret_cast<float*>(smem), lane, [&](int r) { return cand[r]; });
    if (lane == 0) p.out[qi] = (win >= 0) ? cand[win] : kInvalidId;
}

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

#if GBNNS_TU == 0
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
#endif  // GBNNS_TU == 0

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------

// Shape served by the walk_hot* kernels (first pass only): L2, 128-byte rows, adjacency rows of one 32-slot pass
// (walk_hotw*: 33 .. 64 slots, two passes), 32-bit byte offsets.
static bool walk_off32(const WalkParams& p) {  // "compact" index: every table the walk indexes is < 4 GiB, ids fit 24 bits
    return !p.force_wide && (uint64_t)p.n * p.dstride * 4 < (1ull << 32) && (uint64_t)p.n * p.ell_stride * 4 < (1ull << 32) &&
           (!p.aux_ell || (uint64_t)p.n * p.aux_stride * 4 < (1ull << 32)) && p.n <= 0xFFFFFFu;
}

#if GBNNS_TU == 0
// The LDS-list kernel serves ef beyond the register lists, and auxiliary-graph walks over tables >= 4 GiB.
bool walk_uses_lds_list(const WalkParams& p) { return p.ef > kRegListMaxEf || (p.aux_ell && !walk_off32(p)); }

bool walk_uses_hot(const WalkParams& p, int metric) {
    const bool off32 = walk_off32(p);
    return (metric == 0 || metric == 1) && p.dim == 32u && p.dstride == 32u && p.ef <= kBigMaxEf && p.ell_stride <= 64u && off32 && (!p.stamps_on || (p.ef > kHot2MaxEf && !getenv("GBNNS_STAMPS_GENERIC"))) &&
           !p.aux_ell;  // (off32 includes n < 2^24: its visited set stores 24-bit ids)
}

// LDS of one wavefront without the visited set.  Register kernels: tie list + merge buffer + query; the
// hot kernel stages the query inside the merge buffer (it lives in registers once the walk starts).
size_t walk_fast_lds_fixed_bytes(int ef, uint32_t dstride, bool hot, bool lds_list) {
    if (hot)  // tie list + merge buffer of 1 / 2 list registers; ef > 128: + the base list and the flush's flag bytes (walk_hot_big)
        return ef <= 64 ? (size_t)kRegTieCap * 8 + (size_t)kRegStageSlots * 8 + (GBNNS_HOT1_QLDS ? 128 : 0)   // (+ the query, re-read every hop)
                        : (ef <= kHot2MaxEf ? (size_t)kRegTieCap * 8 + (size_t)(64 * 2 + 2) * 8 : big_list_fixed_bytes(ef));
    if (ef <= kRegListMaxEf && !lds_list) {  // tie list + merge buffer (ranks 0..ef of the 1 / 2 / 4-register list) + query
        if (ef > kHot2MaxEf) return big_list_fixed_bytes(ef) + (size_t)dstride * 4;  // walk_reg_big_one
        const int regs = ef <= 64 ? 1 : 2;
        return (size_t)kRegTieCap * 8 + (size_t)(64 * regs + 2) * 8 + (size_t)dstride * 4;
    }
    const size_t ef_pad = ((size_t)ef + 63) & ~(size_t)63;
    return ef_pad * 8 + (size_t)kTieCap * 8 + (size_t)dstride * 4;
}

// Visited set of `entries` ids: 4-byte slots in 4-slot buckets; the hot kernel packs five 24-bit ids and a
// counter byte into each 16-byte bucket (3.2 bytes per id).
// The quotient form (hot first pass, small enough n: GBNNS_VS_ASM) packs seven 16-bit entries per bucket (2.29 bytes per id).
size_t walk_hash_bytes(uint32_t entries, int form) {
    return form == 2 ? (size_t)(entries / 7u) * 16 : form == 1 ? (size_t)(entries / 5u) * 16 : (size_t)entries * 4;
}
uint32_t walk_hash_entries(size_t bytes, int form) {
    return form == 2 ? (uint32_t)(bytes / 16) * 7u : form == 1 ? (uint32_t)(bytes / 16) * 5u : ((uint32_t)(bytes / 4) & ~3u);
}
int walk_hash_form(const WalkParams& p, bool) { return p.vs_shr ? 2 : (walk_uses_packed(p) ? 1 : 0); }  // (vs_shr is set only where the first-pass kernel reads it)
// First-pass kernels that know the quotient form: the walk_hot* family and the generic register-list / two-list kernels of a compact index.
bool walk_knows_quotient(const WalkParams& p, int metric) {
    return walk_uses_hot(p, metric) || (walk_off32(p) && !walk_uses_lds_list(p) && !p.aux_ell);
}

// Every LDS kernel packs its visited set when ids fit 24 bits (the register-list kernels: in their compact,
// 32-bit-offset instantiations).
bool walk_uses_packed(const WalkParams& p) { return walk_uses_lds_list(p) ? (p.n <= 0xFFFFFFu && !p.force_wide) : walk_off32(p); }

size_t walk_fast_lds_bytes(const WalkParams& p, bool hot) {
    return walk_fast_lds_fixed_bytes(p.ef, p.dstride, hot, walk_uses_lds_list(p)) + walk_hash_bytes(p.hash_cap, walk_hash_form(p, hot));
}

#endif  // GBNNS_TU == 0

template <typename K>
static hipError_t set_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return hipSuccess;
}

// Host function of the last first-pass walk kernel this thread launched (profiling: gbnns_profile.walk_kernel).
extern thread_local const void* g_walk_first_fn;
#if GBNNS_TU == 0
thread_local const void* g_walk_first_fn = nullptr;
const char* walk_first_pass_name(hipStream_t s) {
    return g_walk_first_fn ? hipKernelNameRefByPtr(g_walk_first_fn, s) : nullptr;
}
#endif

template <typename K>
static hipError_t launch_walk_k(K kernel, const WalkParams& p, bool retry, size_t lds, hipStream_t s) {
    hipError_t e = set_lds(kernel, lds);
    if (e != hipSuccess) return e;
    if (!retry) g_walk_first_fn = reinterpret_cast<const void*>(kernel);
    const unsigned grid = retry ? (unsigned)kRetrySlots : p.nq;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), lds, s, p);
    return hipGetLastError();
}

template <int METRIC, int STEPS, int R>
static hipError_t launch_reg_t(const WalkParams& p, bool retry, size_t lds, hipStream_t s) {
    // 32-bit byte offsets when both tables are < 4 GiB
    const bool off32 = walk_off32(p);
    if constexpr (R >= 4) {
        // ef > 128: base list in LDS + front list in one register (walk_reg_big_one), whatever the shape
        if constexpr (METRIC == 0 && STEPS == 8) {
            if (!retry && walk_uses_hot(p, METRIC))
                return p.ell_stride > 32u ? launch_walk_k(walk_hotw_big_kernel, p, false, walk_fast_lds_bytes(p, true), s)
                                          : launch_walk_k(walk_hot_big_kernel, p, false, walk_fast_lds_bytes(p, true), s);
        }
        if constexpr (METRIC == 1 && STEPS == 8) {
            if (!retry && walk_uses_hot(p, METRIC))
                return p.ell_stride > 32u ? launch_walk_k(walk_hot_dot_big_kernel<true>, p, false, walk_fast_lds_bytes(p, true), s)
                                          : launch_walk_k(walk_hot_dot_big_kernel<false>, p, false, walk_fast_lds_bytes(p, true), s);
        }
        if (p.aux_ell)
            return retry ? launch_walk_k(walk_reg_big_kernel<METRIC, STEPS, true, true, true>, p, true, lds, s)
                         : launch_walk_k(walk_reg_big_kernel<METRIC, STEPS, true, false, true>, p, false, lds, s);
        if (off32) {
            // the common shape (compact index, adjacency rows of one pass) gets the hop without the pass loop
            if (!retry && p.ell_stride <= ((STEPS == 8 || ((STEPS == 12 || STEPS == 16) && METRIC == 0)) ? 32u : 64u))  // pair form: 32 slots per pass
                return launch_walk_k(walk_reg_big_kernel<METRIC, STEPS, true, false, false, true>, p, false, lds, s);
            return retry ? launch_walk_k(walk_reg_big_kernel<METRIC, STEPS, true, true>, p, true, lds, s)
                         : launch_walk_k(walk_reg_big_kernel<METRIC, STEPS, true, false>, p, false, lds, s);
        }
        return retry ? launch_walk_k(walk_reg_big_kernel<METRIC, STEPS, false, true>, p, true, lds, s)
                     : launch_walk_k(walk_reg_big_kernel<METRIC, STEPS, false, false>, p, false, lds, s);
    } else {
    if (p.aux_ell)  // auxiliary-graph walk (32-bit offsets only; otherwise launch_fast_t took the LDS-list kernel)
        return retry ? launch_walk_k(walk_reg_kernel<METRIC, STEPS, true, true, R, false, true>, p, true, lds, s)
                     : launch_walk_k(walk_reg_kernel<METRIC, STEPS, true, false, R, false, true>, p, false, lds, s);
    if constexpr (R == 1) {
        // the common shape (ef <= 64, adjacency rows of one pass) gets a loop-free expansion;
        // 128-byte rows with L2 additionally the hand-laid-out hop of walk_hot_one
        if constexpr (METRIC == 0 && STEPS == 8) {
            if (!retry && walk_uses_hot(p, METRIC))
                return p.ell_stride > 32u ? launch_walk_k(walk_hotw_kernel, p, false, walk_fast_lds_bytes(p, true), s)
                                          : launch_walk_k(walk_hot_kernel, p, false, walk_fast_lds_bytes(p, true), s);
        }
        if constexpr (METRIC == 1 && STEPS == 8) {
            if (!retry && walk_uses_hot(p, METRIC))
                return p.ell_stride > 32u ? launch_walk_k(walk_hot_dot_kernel<1, true>, p, false, walk_fast_lds_bytes(p, true), s)
                                          : launch_walk_k(walk_hot_dot_kernel<1, false>, p, false, walk_fast_lds_bytes(p, true), s);
        }
        if (off32 && !retry && p.ell_stride <= ((STEPS == 8 || ((STEPS == 12 || STEPS == 16) && METRIC == 0)) ? 32u : 64u))  // pair form: 32 slots per pass
            return launch_walk_k(walk_reg_kernel<METRIC, STEPS, true, false, 1, true>, p, false, lds, s);
    }
    if constexpr (R == 2 && METRIC == 0 && STEPS == 8) {
        // the hot shape, 64 < ef <= 128: two list registers (measured: 0.85 ms against 0.93 ms with the two-list structure)
        if (!retry && walk_uses_hot(p, METRIC))
            return p.ell_stride > 32u ? launch_walk_k(walk_hotw2_kernel, p, false, walk_fast_lds_bytes(p, true), s)
                                      : launch_walk_k(walk_hot2_kernel, p, false, walk_fast_lds_bytes(p, true), s);
    }
    if constexpr (R == 2 && METRIC == 1 && STEPS == 8) {
        if (!retry && walk_uses_hot(p, METRIC))
            return p.ell_stride > 32u ? launch_walk_k(walk_hot_dot_kernel<2, true>, p, false, walk_fast_lds_bytes(p, true), s)
                                      : launch_walk_k(walk_hot_dot_kernel<2, false>, p, false, walk_fast_lds_bytes(p, true), s);
    }
    if (off32)
        return retry ? launch_walk_k(walk_reg_kernel<METRIC, STEPS, true, true, R>, p, true, lds, s)
                     : launch_walk_k(walk_reg_kernel<METRIC, STEPS, true, false, R>, p, false, lds, s);
    return retry ? launch_walk_k(walk_reg_kernel<METRIC, STEPS, false, true, R>, p, true, lds, s)
                 : launch_walk_k(walk_reg_kernel<METRIC, STEPS, false, false, R>, p, false, lds, s);
    }
}

// ef <= 64: one list register per lane (all row-length specialisations); ef <= 128 / 256: two / four
// registers (generic or 128-B-row distance); beyond that the list lives in LDS.
template <int METRIC, int STEPS>
static hipError_t launch_fast_t(const WalkParams& p, bool retry, hipStream_t s) {
    const size_t lds = walk_fast_lds_bytes(p, false);
    constexpr int kWideSteps = (STEPS == 8) ? 8 : 0;
    // 256-byte rows (d_low = 64, the GIST shape, whose efs start at 200): the 4- and 8-register lists keep the
    // unrolled distance with early row loads as well
    constexpr int kWideSteps48 = (STEPS == 8 || STEPS == 12 || STEPS == 16) ? STEPS : 0;
    if (walk_uses_lds_list(p)) {
        if (walk_uses_packed(p))
            return retry ? launch_walk_k(walk_fast_kernel<METRIC, STEPS, true, true>, p, true, lds, s)
                         : launch_walk_k(walk_fast_kernel<METRIC, STEPS, false, true>, p, false, lds, s);
        return retry ? launch_walk_k(walk_fast_kernel<METRIC, STEPS, true, false>, p, true, lds, s)
                     : launch_walk_k(walk_fast_kernel<METRIC, STEPS, false, false>, p, false, lds, s);
    }
    if (p.ef <= 64) return launch_reg_t<METRIC, STEPS, 1>(p, retry, lds, s);
    if (p.ef <= kHot2MaxEf) return launch_reg_t<METRIC, kWideSteps48, 2>(p, retry, lds, s);  // (12- / 16-step rows keep the unrolled distance)
    return launch_reg_t<METRIC, kWideSteps48, 4>(p, retry, lds, s);  // (R >= 4: the two-list kernels, one instance for every ef up to 512)
}

// the walk launchers of the other two compilation units of this file
hipError_t launch_walk_tu1(const WalkParams& p, bool retry, hipStream_t s);              // dot metric
hipError_t launch_walk_tu2(const WalkParams& p, int steps, bool retry, hipStream_t s);   // L2, 12 / 16 steps per row

#if GBNNS_TU == 1
hipError_t launch_walk_tu1(const WalkParams& p, bool retry, hipStream_t s) {
    if (p.dstride == p.dim && p.dim == 32) return launch_fast_t<1, 8>(p, retry, s);  // 128-byte rows: pair form
    return launch_fast_t<1, 0>(p, retry, s);
}
#endif

#if GBNNS_TU == 2
hipError_t launch_walk_tu2(const WalkParams& p, int steps, bool retry, hipStream_t s) {
    return steps == 12 ? launch_fast_t<0, 12>(p, retry, s) : launch_fast_t<0, 16>(p, retry, s);
}
#endif

#if GBNNS_TU == 0
static hipError_t launch_walk_any(const WalkParams& p, int metric, bool retry, hipStream_t s) {
    if (p.nq == 0) return hipSuccess;
    if (metric == 1) return launch_walk_tu1(p, retry, s);
    if (p.dstride == p.dim) {
        switch (p.dim) {
            case 32: return launch_fast_t<0, 8>(p, retry, s);
            case 48: return launch_walk_tu2(p, 12, retry, s);
            case 64: return launch_walk_tu2(p, 16, retry, s);
            default: break;
        }
    }
    return launch_fast_t<0, 0>(p, retry, s);
}

hipError_t launch_walk_fast(const WalkParams& p, int metric, hipStream_t s) { return launch_walk_any(p, metric, false, s); }
hipError_t launch_walk_retry(const WalkParams& p, int metric, hipStream_t s) { return launch_walk_any(p, metric, true, s); }

// The bitmap first pass runs the register-list (ef <= 128, L2) / two-list (128 < ef <= 1 024, both metrics) walk for
// 128-byte rows of a compact index, the two-list walk for 256-byte rows with L2, else the LDS-list walk.
bool walk_bitmap_uses_reg(const WalkParams& p, int metric) {
    const bool rows128 = p.dim == 32u && p.dstride == 32u, rows256 = p.dim == 64u && p.dstride == 64u && metric == 0 && p.ef > kHot2MaxEf;
    return (metric == 0 || p.ef > kHot2MaxEf) && p.ef <= kRegListMaxEf && (rows128 || rows256) && walk_off32(p) && !p.aux_ell;
}

// LDS of the bitmap first pass: result list (or merge buffer) + tie list + query (no visited table)
size_t walk_bitmap_lds_bytes(const WalkParams& p, int metric) {
    const bool reg = walk_bitmap_uses_reg(p, metric);
    // (two-list kernel: the re-rank query cannot overlay the base list it reads its candidates from)
    return walk_fast_lds_fixed_bytes(p.ef, p.dstride, false, !reg) + (reg && p.ef > kHot2MaxEf ? p.rr_reserve : 0u);
}

// Room the fused re-rank has for the original-space query (it is staged once the walk is over).
size_t walk_rr_room(const WalkParams& p, int metric, bool hot, bool bitmap_pass) {
    if (bitmap_pass) return walk_bitmap_uses_reg(p, metric) && p.ef > kHot2MaxEf ? (size_t)p.rr_reserve : walk_bitmap_lds_bytes(p, metric);
    if (p.ef > kHot2MaxEf && !walk_uses_lds_list(p)) return walk_hash_bytes(p.hash_cap, walk_hash_form(p, hot));  // two-list kernels: the visited-set area
    return walk_fast_lds_bytes(p, hot);
}

template <int R>
static hipError_t launch_bitmap_reg(const WalkParams& p, unsigned slots, size_t lds, hipStream_t s) {
    hipError_t e = set_lds(walk_bitmap_reg_kernel<0, R>, lds);
    if (e != hipSuccess) return e;
    g_walk_first_fn = reinterpret_cast<const void*>(walk_bitmap_reg_kernel<0, R>);
    hipLaunchKernelGGL((walk_bitmap_reg_kernel<0, R>), dim3(slots), dim3(64), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_walk_bitmap(const WalkParams& p, int metric, unsigned slots, hipStream_t s) {
    if (p.nq == 0) return hipSuccess;
    const size_t lds = walk_bitmap_lds_bytes(p, metric);
    if (walk_bitmap_uses_reg(p, metric)) {
        if (p.ef <= 64) return launch_bitmap_reg<1>(p, slots, lds, s);
        if (p.ef <= kHot2MaxEf) return launch_bitmap_reg<2>(p, slots, lds, s);
        // (adjacency rows of one 32-slot pass -- the common case -- take the instance without the pass loop)
        auto go = [&](auto kernel) -> hipError_t {
            hipError_t e = set_lds(kernel, lds);
            if (e != hipSuccess) return e;
            g_walk_first_fn = reinterpret_cast<const void*>(kernel);
            hipLaunchKernelGGL(kernel, dim3(slots), dim3(64), lds, s, p);
            return hipGetLastError();
        };
        const bool one = p.ell_stride <= 32u;
        if (p.dim == 64u) return one ? go(walk_bitmap_big_kernel<0, 16, true>) : go(walk_bitmap_big_kernel<0, 16, false>);  // 256-byte rows, L2 (pair form)
        if (metric == 1) return one ? go(walk_bitmap_big_kernel<1, 8, true>) : go(walk_bitmap_big_kernel<1, 8, false>);
        return one ? go(walk_bitmap_big_kernel<0, 8, true>) : go(walk_bitmap_big_kernel<0, 8, false>);
    }
    if (metric == 1) {
        hipError_t e = set_lds(walk_bitmap_kernel<1, 0>, lds);
        if (e != hipSuccess) return e;
        g_walk_first_fn = reinterpret_cast<const void*>(walk_bitmap_kernel<1, 0>);
        hipLaunchKernelGGL((walk_bitmap_kernel<1, 0>), dim3(slots), dim3(64), lds, s, p);
    } else if (p.dstride == p.dim && p.dim == 32) {
        hipError_t e = set_lds(walk_bitmap_kernel<0, 8>, lds);
        if (e != hipSuccess) return e;
        g_walk_first_fn = reinterpret_cast<const void*>(walk_bitmap_kernel<0, 8>);
        hipLaunchKernelGGL((walk_bitmap_kernel<0, 8>), dim3(slots), dim3(64), lds, s, p);
    } else {
        hipError_t e = set_lds(walk_bitmap_kernel<0, 0>, lds);
        if (e != hipSuccess) return e;
        g_walk_first_fn = reinterpret_cast<const void*>(walk_bitmap_kernel<0, 0>);
        hipLaunchKernelGGL((walk_bitmap_kernel<0, 0>), dim3(slots), dim3(64), lds, s, p);
    }
    return hipGetLastError();
}

hipError_t launch_walk_general(const WalkParams& p, int metric, hipStream_t s) {
    if (p.nq == 0) return hipSuccess;
    const size_t lds = std::max((size_t)p.dstride * 4, p.rr_db ? (size_t)p.rr_dstride * 4 : (size_t)0);
    if (metric == 1) {
        hipError_t e = set_lds(walk_general_kernel<1>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((walk_general_kernel<1>), dim3(kGeneralSlots), dim3(64), lds, s, p);
    } else {
        hipError_t e = set_lds(walk_general_kernel<0>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((walk_general_kernel<0>), dim3(kGeneralSlots), dim3(64), lds, s, p);
    }
    return hipGetLastError();
}

hipError_t launch_rerank(const RerankParams& p, int metric, hipStream_t s) {
    if (p.nq == 0) return hipSuccess;
    const size_t lds = (size_t)p.dstride * 4;
    const bool pairs = p.dim % 8 == 0 && p.dim > 0;  // pair form (both metrics)
    hipError_t e;
    if (metric == 1) {
        if (pairs) {
            e = set_lds(rerank_pair_kernel<1>, lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((rerank_pair_kernel<1>), dim3(p.nq), dim3(64), lds, s, p);
        } else {
            e = set_lds(rerank_kernel<1>, lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((rerank_kernel<1>), dim3(p.nq), dim3(64), lds, s, p);
        }
    } else if (pairs) {
        e = set_lds(rerank_pair_kernel<0>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((rerank_pair_kernel<0>), dim3(p.nq), dim3(64), lds, s, p);
    } else {
        e = set_lds(rerank_kernel<0>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((rerank_kernel<0>), dim3(p.nq), dim3(64), lds, s, p);
    }
    return hipGetLastError();
}

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
    if (aligned) {
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

hipError_t launch_gd_prune(const GdParams& p, int metric, hipStream_t s) {
    if (p.n == 0) return hipSuccess;
    if (metric == 1) hipLaunchKernelGGL((gd_prune_kernel<1>), dim3((unsigned)p.n), dim3(64), 0, s, p);
    else hipLaunchKernelGGL((gd_prune_kernel<0>), dim3((unsigned)p.n), dim3(64), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_debug_merge(int regs, const uint64_t* entries, int size, const uint64_t* surv, int ef, uint64_t* out,
                              int* out_size, hipStream_t s) {
    if (regs == 1) hipLaunchKernelGGL((debug_merge_kernel<1>), dim3(1), dim3(64), 0, s, entries, size, surv, ef, out, out_size);
    else if (regs == 2) hipLaunchKernelGGL((debug_merge_kernel<2>), dim3(1), dim3(64), 0, s, entries, size, surv, ef, out, out_size);
    else hipLaunchKernelGGL((debug_merge_kernel<4>), dim3(1), dim3(64), 0, s, entries, size, surv, ef, out, out_size);
    return hipGetLastError();
}

#endif  // GBNNS_TU == 0

}  // namespace gbnns
