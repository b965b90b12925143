// walk_coop.hip -- the two-wavefront walk for SMALL batches (round 6): one query per workgroup of TWO wavefronts.
//
// Why.  The reference's gist row searches 1 000 queries at a time (parameters_of_databases.txt:13-21).  One wavefront per query
// puts 1 000 wavefronts on 1 024 SIMDs: nothing hides a hop's chain -- select, adjacency row, visited test, row gather,
// distances, insertion, 5 200 cycles, every step waiting for the one before (profiles/r04_stamps.txt: select 15 %, visited 18 %,
// gather + distances 26 %, insertion 25 % of a hop) -- while the machine's vector ports idle at 18 %.  The chain has two halves that need
// not wait for each other: the LIST half (pick the next node, insert the survivors: needs the result lists) and the EXPANSION half
// (adjacency row, which ids are new, their rows, their distances: needs the query and the visited set).  And the node a hop will pick
// is known a hop ahead almost always: it is the runner-up of the last selection unless one of the hop's own new survivors is closer
// (measured on the gist shape: the prepared expansion is the right one in 0.995 of the hops).
//
// How.  Wavefront 0, the KEEPER, owns the two-list result structure (BigList, walk_lists.h); wavefront 1, the SCOUT, owns the query
// registers and is the only writer of the exact visited set.  Per hop:
//     keeper: selects `node` (search_function.h:65-71), posts (node, runner-up, runner-up distance)            -- barrier 1 --
//     scout:  if the expansion it prepared ahead is of `node`: nothing to do; else takes the prepared one's claims back out of the
//             visited set and expands `node` now                                                                 -- barrier 2 --
//     keeper: reads the scout's 32 (id, new?, distance) triples, counts dist_calc, inserts the survivors in stored-neighbour order
//             (makeStep, search_function.h:15-40), selects again ...
//     scout:  ... while it predicts the next node from what it has just handed over (its closest new id if that beats the runner-up,
//             else the runner-up) and expands THAT: adjacency row (requested a hop ahead), claims in the visited set, rows, distances.
// Exactness.  The visited set has one writer, and the claims it KEEPS are made in the reference's visit order: an expansion prepared
// ahead claims its ids right after the ids of the node before it -- exactly when the reference would, if the prediction holds.  If it
// does not hold, nothing else has touched the table since: every slot the prepared expansion won is cleared and its bucket's counter
// taken back (winners sit on top of their buckets; counter ticks of lanes that lost a race stay, as they do in the one-wavefront
// kernels), then the right node is expanded.  A prepared expansion never uses the stash (a probe sequence that runs out aborts it).
// The reference's arithmetic (pair form, support_func.h:107-128 order), visit order, hops, dist_calc, candidate lists and answers are
// those of the one-wavefront kernels, bit for bit; hand-overs (a visited set that fills up, a tie list that overflows) go to the same
// retry / general passes.
// At the end both wavefronts re-rank (getRealNearest, search_function.h:105-125): each takes every other 32-candidate pass, the strict
// `<` in pop order is the minimum over (distance key, pop index) of the two.
//
// Shapes: L2, walked rows of 128 / 192 / 256 bytes (d_low 32 / 48 / 64), 128 < ef <= 1 024, compact index, adjacency rows of one
// 32-slot pass, one entry point, no auxiliary graph.  The host takes it for a batch that runs alone and is resident at once in this form
// (search_core.cpp; knob "coop").
#define GBNNS_WAVE_LOCAL_SYNC 1
#define GBNNS_COOP_HINT 1
#include <algorithm>

#include "launch_util.h"
#include "walk_lists.h"

namespace gbnns {

namespace {

constexpr uint32_t kCoopDone = 0xFFFFFFFFu;      // mailbox: the walk is over (finished or handed over)
constexpr uint32_t kCoopNoNode = 0xFFFFFFFEu;    // no node (ids are < 2^24 here)

// both wavefronts: everything written to LDS before is visible after.  (No vmcnt wait: the scout's adjacency prefetches stay in flight
// across the barriers; neither wavefront stores to global memory inside the walk.)
__device__ __forceinline__ void coop_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// visited_claim_mask_quotient / _packed (walk_common.h: same table, same protocol, same code) with one more output: `won` = LDS byte
// address of the slot a lane's claim wrote (0: it wrote none) -- what it takes to take the claim back (coop_unclaim).
__device__ __forceinline__ uint64_t coop_claim_quotient(uint32_t lds_base, uint32_t nbuckets, uint32_t id, uint64_t valid, uint32_t ctl,
                                                        uint64_t& overflowed, uint32_t& won) {
    const uint32_t end = lds_base + (nbuckets << 4);
    uint32_t basev = lds_base, addr, t0, t1, t2, mulc, wad = 0u;
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
        "v_mov_b32 %[wad], %[t0]\n\t"                       // the slot this lane's claim wrote
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
        "s_and_b32 vcc_lo, %[shr], 0xF0000000\n\t"          // the probe-number field alone
        "v_cmp_gt_u32 vcc, vcc_lo, %[t2]\n\t"
        "s_andn2_b64 %[act], exec, vcc\n\t"
        "s_or_b64 %[ovf], %[ovf], %[act]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execnz 5b\n"
        "9:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [fresh] "=&s"(fresh), [act] "=&s"(act), [sv] "=&s"(sv), [ovf] "=&s"(ovf), [mulc] "=&s"(mulc), [t0] "=&v"(t0), [t1] "=&v"(t1),
          [t2] "=&v"(t2), [addr] "=&v"(addr), [wad] "+v"(wad)
        : [id] "v"(id), [valid] "s"(valid), [end] "s"(end), [basev] "v"(basev), [shr] "s"(ctl), [nb] "s"(nbuckets)
        : "vcc", "scc", "memory", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71");
    overflowed = ovf;
    won = wad;
    return fresh;
}
__device__ __forceinline__ uint64_t coop_claim_packed(uint32_t lds_base, uint32_t nbuckets, uint32_t id, uint64_t valid, uint32_t& won) {
    const uint32_t end = lds_base + (nbuckets << 4);
    const uint32_t mulc = 0x9E3779B1u;
    uint32_t basev = lds_base, inc = 1u << 24, addr, wad = 0u;
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
        "v_mov_b32 %[wad], %[t0]\n\t"                       // the slot this lane's claim wrote
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
          [t2] "=&v"(t2), [addr] "=&v"(addr), [wad] "+v"(wad)
        : [id] "v"(id), [valid] "s"(valid), [end] "s"(end), [basev] "v"(basev), [inc] "v"(inc), [mulc] "s"(mulc),
          [nb] "s"(nbuckets)
        : "vcc", "memory", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71");
    won = wad;
    return fresh;
}
// Takes the claims of one expansion back out of the table (every lane with won != 0): the slot reads "empty" again and its bucket's
// counter loses the tick.  Valid only while nothing else has been claimed since -- the winners then sit on top of their buckets.
__device__ __forceinline__ void coop_unclaim(uint32_t won, bool quotient) {
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    typedef __attribute__((address_space(3))) uint16_t lds_u16;
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    if (won) {
        lds_u32* const cnt = (lds_u32*)(size_t)((won & ~15u) + 12u);
        if (quotient) {
            *(lds_u16*)(size_t)won = (uint16_t)0xFFFFu;
            __hip_atomic_fetch_sub(cnt, 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else {
            lds_u8* const b = (lds_u8*)(size_t)won;
            b[0] = (uint8_t)0xFFu; b[1] = (uint8_t)0xFFu; b[2] = (uint8_t)0xFFu;
            __hip_atomic_fetch_sub(cnt, 1u << 24, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    }
    wave_sync();
}

// Mailbox words (LDS, in the two-list layout's query area: the scout reads its query from global memory into registers).
// Inside the walk the two wavefronts do not meet at barriers -- a barrier makes every hop as long as the slower wavefront's turn plus
// whatever it does afterwards.  The keeper POSTS a selection: four words (node / runner-up / its distance / hint) into the slot of the
// post's parity, then the post's number into kMbSeq.  The scout PUBLISHES an expansion: the buffer, then kMbReady = node | buffer << 30
// | stash-full << 31.  LDS performs a wavefront's operations in issue order, so whoever sees the second word sees the first.
//   * A node is expanded at most once in a walk and kMbReady only ever names the scout's LATEST expansion, so "kMbReady names the node
//     I selected" can only mean that node's current, claimed expansion.
//   * The keeper can post at most ONE selection the scout has not read: the post after that needs an expansion the scout prepares only
//     after reading -- so two post slots are enough, and a post the scout finds skipped (kMbSeq two ahead) was a hit: its prepared
//     expansion has been taken, the post in front of it is the one the keeper is waiting on.
// Three wavefronts (the scout split into a CLAIMER and a RANGER, see coop_agent3): the buffer of an expansion is a function of the post
// it answers -- post s is answered from buffer s & 1, whether expanded on demand or prepared after post s - 1 -- so that both agents
// write the halves of the same buffer, and each agent publishes PER BUFFER: kMbRdC + b / kMbRdR + b = the node whose claimer / ranger
// half sits in buffer b (the claimer's with stash-full << 31).  A word keeps naming its node until its owner rewrites that buffer, two
// posts later: an agent that lags a phase behind the other still finds the other's half of the expansion it is working on.
// kMbBestC: the ranger's re-rank result.
enum { kMbPost0 = 0, kMbPost1 = 4, kMbReady = 8, kMbSeq = 9, kMbK0 = 10, kMbKept = 11, kMbBestB = 12, kMbBestC = 14, kMbRdC = 16, kMbRdR = 18 };
typedef __attribute__((address_space(3))) uint32_t coop_lds_u32;
__device__ __forceinline__ uint32_t coop_peek(const uint32_t* w) {   // one LDS word, read now (never hoisted out of a polling loop)
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void coop_wait_for(const uint32_t* w, uint32_t value) {
    while (coop_peek(w) != value) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void coop_flag(uint32_t* w, uint32_t value, int lane) {   // everything this wavefront wrote to LDS before is visible with it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(w, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- three wavefronts per query: the scout's two halves as two wavefronts -----------------------------------------------------------
// Where the two-wavefront hop goes (profiles/r06_coop_stamps.txt, v8): the scout's expansion is 3 700 cycles -- adjacency words 480,
// claims 1 200, rows + distances 1 060, closest new id 540 -- against 2 900 of list work in the keeper: the scout is the hop.  Its claims
// and its distances do not need each other (rows are requested before the claims; a distance does not depend on the visited set), so
// they are two wavefronts here: the CLAIMER (ROLE 1, wavefront 1) owns the visited set and publishes, per neighbour slot, the id and
// whether it is new (even buffer words, kMbRdC); the RANGER (ROLE 2, wavefront 2) owns the query registers and publishes the distance
// of every stored neighbour, new or not (odd buffer words, kMbRdR).  The keeper takes an expansion when both words of the post's buffer
// name its node.  Both agents run the scout's loop -- read a post, expand its node now if it is not what they prepared, predict
// the next node, expand that ahead -- and predict from the same data (the closest new id of the expansion just consumed needs the
// claimer's flags and the ranger's distances: each waits for the other's half and reads the buffer), so they prepare the same node as a
// rule; when they do not (a half read while it was being rewritten), one of them expands on demand: predictions only ever cost time.
// Exactness is the two-wavefront argument: one writer of the visited set, claims kept in visit order, a wrong prepared expansion's
// claims taken back before anything else is claimed; distances are stateless.  Every wait gives up after kCoopSpinLimit polls (a
// protocol error, never seen): the keeper then hands the query to the retry pass, which is exact by itself.
constexpr uint32_t kCoopSpinLimit = 1u << 21;

template <int STEPS, int ROLE>
__device__ __forceinline__ void coop_agent3(const WalkParams& p, uint32_t* mbox, uint32_t* bufs, uint32_t* pfs, uint32_t* hash, uint32_t nbuckets,
                                            const RowRegs<STEPS / 2>& qreg, int lane) {
    static_assert(ROLE == 1 || ROLE == 2, "claimer, ranger");
    constexpr int kQSteps = STEPS / 2;
    constexpr uint32_t kRowBytes = (uint32_t)STEPS * 16u;
    constexpr bool kClaims = ROLE == 1;
    const uint32_t slot = (uint32_t)lane >> 1, half = (uint32_t)lane & 1u;
    const uint32_t vs_shr = p.vs_shr;
    const uint32_t hash_lds = (uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(hash));
    uint32_t* const mine = mbox + (kClaims ? kMbRdC : kMbRdR);          // [buffer]
    const uint32_t* const other = mbox + (kClaims ? kMbRdR : kMbRdC);
    uint32_t prepared = kCoopNoNode, pbuf = 0u, r_won = 0u;      // the expansion held ready (its buffer; the slots its claims wrote)
    uint32_t e_for = kCoopNoNode, e_cmin = 0xFFFFFFFFu, e_cnode = kCoopNoNode;   // closest new id of the expansion of node e_for
    uint32_t pfn0 = kCoopNoNode, pfn1 = kCoopNoNode, pfn2 = kCoopNoNode;
    const uint32_t* const ell = p.ell;
    const uint32_t aslot = slot < p.ell_stride ? slot : p.ell_stride - 1u;
    auto adjacency = [&](uint32_t node) -> uint32_t {
        const uint32_t* row = reinterpret_cast<const uint32_t*>(row_ptr<true>(reinterpret_cast<const float*>(ell), node, p.ell_stride));
        return (slot < p.ell_stride) ? row[slot] : kInvalidId;
    };
    const uint32_t pfs_lds = (uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(pfs));
    auto request_adjacency = [&](uint32_t k, uint32_t node) {   // node's adjacency words -> this agent's LDS slot k (wave-uniform arguments)
        const uint32_t* src = reinterpret_cast<const uint32_t*>(row_ptr<true>(reinterpret_cast<const float*>(ell), node, p.ell_stride)) + aslot;
        const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pfs_lds + 256u * k));
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };
    auto requested_slot = [&](uint32_t node) -> int { return node == pfn0 ? 0 : (node == pfn1 ? 1 : (node == pfn2 ? 2 : -1)); };
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto expand = [&](uint32_t node, uint32_t bufno, bool ahead, uint32_t ahead1, uint32_t ahead2) -> bool {
        uint32_t* const buf = bufs + 64 * bufno;
        uint32_t nb;
        const int have = requested_slot(node);
        if (have >= 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t wv = pfs[64 * have + lane];
            nb = (slot < p.ell_stride) ? wv : kInvalidId;
        } else {
            nb = adjacency(node);
        }
        const bool valid = nb != kInvalidId;
        const uint64_t mv = __ballot(valid);
        RowRegs<kQSteps> rr;
        uint32_t roff = 0;
        if constexpr (!kClaims) {   // every lane loads (empty slots: row 0), as in walk_reg_big_one
            const uint32_t nbl = valid ? nb : 0u;
            roff = nbl * kRowBytes + half * (kRowBytes / 2u);
            load_row<kQSteps>(rr, reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.db) + roff));
        }
        if (ahead1 != kInvalidId && ahead1 != node && requested_slot(ahead1) < 0) {
            pfn1 = ahead1;
            request_adjacency(1u, ahead1);
        }
        if (ahead2 < p.n && ahead2 != node && requested_slot(ahead2) < 0) {
            pfn2 = ahead2;
            request_adjacency(2u, ahead2);
        }
        if constexpr (kClaims) {
            bool stash_full = false;
            uint64_t mclaimed, movf = 0;
            uint32_t won;
            const uint32_t hl = uni(hash_lds), nbk = uni(nbuckets), ctl = uni(vs_shr);
            if (ctl) mclaimed = coop_claim_quotient(hl, nbk, nb, mv & 0x5555555555555555ull, ctl, movf, won);
            else mclaimed = coop_claim_packed(hl, nbk, nb, mv & 0x5555555555555555ull, won);
            if (__builtin_expect(movf != 0, 0)) {
                if (ahead) {   // a probe sequence ran out: not ahead of time (the stash cannot be undone)
                    coop_unclaim(won, true);
                    return false;
                }
                if (!stash_claim(hl, nbk, movf, nb, mclaimed, lane)) stash_full = true;   // the keeper hands the query over
            }
            const bool fresh = __builtin_amdgcn_inverse_ballot_w64(mclaimed);   // (even lanes)
            if (!half) buf[lane] = valid ? (nb | (fresh ? 0x80000000u : 0u)) : 0xFFFFFFFFu;
            coop_flag(mine + bufno, node | (stash_full ? 1u << 31 : 0u), lane);
            r_won = won;
        } else {
            (void)mv;
            uint32_t kd;
            if constexpr (STEPS == 8) kd = fkey_sumsq(l2_pair_from_regs(rr, qreg.v));
            else kd = fkey_sumsq(l2_pair_from_regs_wide<kQSteps>(rr, qreg.v));
            asm volatile("" ::"v"(roff));  // the address register must not double as a load destination
            if (half) buf[lane] = kd;
            coop_flag(mine + bufno, node, lane);
        }
        return true;
    };
    // the closest NEW id of the expansion in buffer `bufno` (both halves there), and its adjacency words requested now
    auto closest_new = [&](uint32_t bufno) {
        const uint64_t rv = reinterpret_cast<const uint64_t*>(bufs + 64 * bufno)[lane & 31];
        const uint32_t w0 = lane < 32 ? (uint32_t)rv : 0xFFFFFFFFu;
        const bool fresh = w0 != 0xFFFFFFFFu && (w0 >> 31) != 0u;
        uint32_t x = fresh ? (uint32_t)(rv >> 32) : 0xFFFFFFFFu;
        const uint32_t dkf = x;
        x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xf, 0xf, false));   // quad_perm 1,0,3,2
        x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xf, 0xf, false));   // quad_perm 2,3,0,1
        x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xf, 0xf, false));  // row_half_mirror
        x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xf, 0xf, false));  // row_mirror
        const uint32_t dmin = min(readlane_u32(x, 0), readlane_u32(x, 16));
        e_cmin = 0xFFFFFFFFu;
        e_cnode = kCoopNoNode;
        if (dmin != 0xFFFFFFFFu) {
            const uint64_t me = __ballot(fresh && dkf == dmin);
            if (me && (me & (me - 1)) == 0) {   // unique
                const uint32_t c = readlane_u32(w0 & 0x7FFFFFFFu, __ffsll((unsigned long long)me) - 1);
                if (c < p.n) {   // (a half read while its owner rewrites it can be anything: a prediction, checked)
                    e_cmin = dmin;
                    e_cnode = c;
                    if (requested_slot(c) < 0) {
                        pfn0 = c;
                        request_adjacency(0u, c);
                    }
                }
            }
        }
    };
    // waits until the other agent's half of (node, bufno) is there, then takes the closest new id; gives up when the keeper has moved on
    // (post number beyond `seen`: it consumed that expansion, the other agent may be rewriting the buffer)
    auto pair_up = [&](uint32_t node, uint32_t bufno, uint32_t seen) {
        uint32_t spins = 0;
        while (true) {
            if ((coop_peek(other + bufno) & 0x7FFFFFFFu) == node) {
                asm volatile("" ::: "memory");
                closest_new(bufno);
                e_for = node;
                return;
            }
            if (coop_peek(mbox + kMbSeq) > seen || ++spins > kCoopSpinLimit) return;
            __builtin_amdgcn_s_sleep(1);
        }
    };
    uint32_t expect = 0;
#ifdef GBNNS_STAMPS
    unsigned long long seg[5] = {0, 0, 0, 0, 0};
    unsigned n_hit = 0, n_miss = 0, n_noec = 0, n_gsurv = 0;
#endif
    while (true) {
        STAMP(a0)
        if (prepared != kCoopNoNode && e_for != prepared) pair_up(prepared, pbuf, expect);   // under the keeper's list work
        STAMP(a1)
        STAMP_ADD(3, a0, a1)
        uint32_t seq, spins = 0;
        while ((seq = coop_peek(mbox + kMbSeq)) <= expect) {
            if (++spins > kCoopSpinLimit) break;
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        if (seq <= expect) break;   // (gave up: the keeper's own wait hands the query over)
        if (seq != expect + 1u) prepared = kCoopNoNode;   // a post was skipped: it took the prepared expansion (see the mailbox notes)
        expect = seq;
        const uint4 post = *reinterpret_cast<const uint4*>(mbox + (((seq - 1u) & 1u) ? kMbPost1 : kMbPost0));
        const uint32_t node = uni(post.x), pred = uni(post.y), h2 = uni(post.z), hint = uni(post.w);
        if (node == kCoopDone) break;
        STAMP(a2)
        STAMP_ADD(0, a1, a2)
#ifdef GBNNS_STAMPS
        if (node != prepared) n_miss += 1; else n_hit += 1;
#endif
        const uint32_t bufno = seq & 1u;
        if (node != prepared) {   // not what was prepared (or nothing was): the keeper is waiting
            if constexpr (kClaims) {
                if (prepared != kCoopNoNode) coop_unclaim(r_won, vs_shr != 0u);
            }
            expand(node, bufno, false, pred, hint);
        }
        STAMP(a3)
        STAMP_ADD(1, a2, a3)
        if (e_for != node) pair_up(node, bufno, seq);
        STAMP(a4)
        STAMP_ADD(4, a3, a4)
        // the next node, probably: the closest new id just handed over if it beats the runner-up, else the runner-up
        uint32_t guess = pred == kInvalidId ? kCoopNoNode : pred;
        if (e_for == node && e_cmin < h2 && e_cnode != kCoopNoNode) guess = e_cnode;
#ifdef GBNNS_STAMPS
        if (e_for != node) n_noec += 1;
        if (e_for == node && e_cmin < h2 && e_cnode != kCoopNoNode) n_gsurv += 1;
#endif
        prepared = kCoopNoNode;
        if (guess != kCoopNoNode && guess < p.n) {
            pbuf = (seq + 1u) & 1u;
            if (expand(guess, pbuf, true, pred, hint)) prepared = guess;
        }
        STAMP(a5)
        STAMP_ADD(2, a4, a5)
    }
#ifdef GBNNS_STAMPS
    if (lane == 0 && p.stamps) {   // claimer: words 8 .. 15 (+ 25), ranger: 16 .. 23 (+ 24)
        unsigned long long* st = p.stamps + (kClaims ? 8 : 16);
        for (int i = 0; i < 5; ++i) atomicAdd(st + i, seg[i]);
        atomicAdd(st + 5, (unsigned long long)n_hit); atomicAdd(st + 6, (unsigned long long)n_miss); atomicAdd(st + 7, (unsigned long long)n_noec);
        atomicAdd(p.stamps + (kClaims ? 25 : 24), (unsigned long long)n_gsurv);
    }
#endif
}

// LDS: [BigList: big_list_fixed_bytes(ef)][mailbox: dstride floats][result buffers: 2 x 64 words][adjacency words requested ahead: 3 x 64]
//      [visited set | re-rank scratch]
// Result buffer word 2 s     = slot s's id | 0x80000000 when the id is NEW (the scout's claim won); all-ones: empty slot
//               word 2 s + 1 = its distance key (meaningful with the flag)
template <int STEPS, bool LATE, int NW>
// (registers: the attribute's upper bound is what the compiler pads the allocation to.  Three wavefronts per SIMD is what a batch of four
// workgroups per CU needs, but an allocation of exactly 512 / 3 leaves the dispatcher no slack -- a CU whose SIMDs are not filled in
// rotation cannot take its fourth workgroup, and 15 - 30 of 1 000 workgroups started only when others had ended, half a millisecond
// late: the three-wavefront form allows four per SIMD and leaves the per-CU count to the launch's LDS size, coop_lds_floor)
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(1, NW == 3 ? 4 : 2))) void walk_coop_kernel(WalkParams p) {
    static_assert(STEPS == 8 || STEPS == 12 || STEPS == 16, "pair-form rows of 128 / 192 / 256 bytes");
    static_assert(NW == 2 || (NW == 3 && !LATE), "keeper + scout, or keeper + claimer + ranger (whose rows never wait for the claims)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kQSteps = STEPS / 2;
    constexpr uint32_t kRowBytes = (uint32_t)STEPS * 16u;
    constexpr int kRerankLoads = (STEPS >= 12 && NW == 2) ? 24 : 8;   // row pieces in flight per lane (three wavefronts share 512 registers a lane)
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t qi = walk_query_of(p, blockIdx.x);
    const int ef = p.ef;
    STAMP(k_begin)
#ifdef GBNNS_STAMPS
#ifdef GBNNS_STAMPS_SPREAD
    constexpr bool kSpread = true;    // (EXTRA_DEFS=-DGBNNS_STAMPS_SPREAD: the launch's residency stamps for the two-wavefront form too, in place of its scout's words 24 .. 31)
#else
    constexpr bool kSpread = NW == 3;
#endif
    if (kSpread && threadIdx.x == 0 && p.stamps) {   // how many workgroups are alive at once (30: now, 31: the most)
        const unsigned long long alive = atomicAdd(p.stamps + 30, 1ull) + 1ull;
        atomicMax(p.stamps + 31, alive);
    }
    const unsigned long long r_begin = __builtin_amdgcn_s_memrealtime();   // (100 MHz, one clock for the whole device)
#endif
    uint32_t* const mbox = reinterpret_cast<uint32_t*>(smem + big_list_fixed_bytes(ef));
    uint32_t* const bufs = mbox + p.dstride;                          // 2 x 64 words
    uint32_t* const pfs = bufs + 128;                                 // 3 x 64 words: adjacency words requested ahead (scout; NW == 3: claimer, then the ranger's)
    uint32_t* const hash = pfs + 192 * (NW - 1);
    unsigned char* const after_q = reinterpret_cast<unsigned char*>(hash);
    const uint32_t cap = p.hash_cap;
    const uint32_t vs_shr = p.vs_shr;
    const uint32_t nbuckets = vs_shr ? cap / 7u - kStashBuckets : cap / 5u;
    const uint32_t hash_lds = (uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(hash));

    // ---- set-up: the table (both wavefronts), the entry's distance (scout) ---------------------------------------------
    {
        const uint32_t words = vs_shr ? (nbuckets + kStashBuckets) * 4u : nbuckets * 4u;
        const uint32_t empty3 = vs_shr ? 0xF000FFFFu : 0x00FFFFFFu;
        for (uint32_t i = threadIdx.x; i < words; i += 64 * NW) hash[i] = (i & 3u) == 3u ? empty3 : 0xFFFFFFFFu;
        if (threadIdx.x == 0) { mbox[kMbReady] = kCoopNoNode; mbox[kMbSeq] = 0u; }
        if (NW == 3 && threadIdx.x < 4) mbox[kMbRdC + threadIdx.x] = kCoopNoNode;   // (kMbRdC + 0 / 1, kMbRdR + 0 / 1)
    }
    const uint32_t entry = (uint32_t)__builtin_amdgcn_readfirstlane((int)(p.entries ? p.entries[qi] : 0u));
    if (entry >= p.n) {  // (both wavefronts take this exit)
        if (wave == 0) write_bad_entry(p, qi, lane);
        return;
    }
    const uint32_t slot = (uint32_t)lane >> 1, half = (uint32_t)lane & 1u;
    RowRegs<kQSteps> qreg;   // scout (NW == 3: ranger): this lane's half of the query
    if (wave == NW - 1) {
        const float4* q4 = reinterpret_cast<const float4*>(p.q + (size_t)qi * p.qstride);
#pragma unroll
        for (int t = 0; t < kQSteps; ++t) qreg.v[t] = q4[kQSteps * half + t];
        RowRegs<kQSteps> er;
        load_row<kQSteps>(er, row_ptr<true>(p.db, entry, p.dstride) + half * (kRowBytes / 8u));
        uint32_t kd;
        if constexpr (STEPS == 8) kd = fkey_sumsq(l2_pair_from_regs(er, qreg.v));
        else kd = fkey_sumsq(l2_pair_from_regs_wide<kQSteps>(er, qreg.v));
        const uint32_t k0 = readlane_u32(kd, 1);   // odd lanes hold the distance
        if (lane == 0) mbox[kMbK0] = k0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    coop_barrier();   // table initialised, the entry's distance posted

    if (wave == 0) {
        // =================================================== KEEPER ===================================================
        BigList B;
        B.init(smem, ef);
        {
            const uint32_t k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mbox[kMbK0]);
            B.worst = B.fworst = k0;
            B.F.hi[0] = lane == 0 ? k0 : B.F.hi[0];
            B.F.lo[0] = lane == 0 ? entry << 1 : B.F.lo[0];
            if (lane == 0) {   // (the one table write that is not the scout's: before its first access, barrier 1 of the first hop)
                if (vs_shr) {
                    quotient_table_put_first(hash, nbuckets, entry, vs_shr);
                    hash[nbuckets * 4u + kStashIds] = 0u;  // the stash behind the buckets is empty
                } else {
                    packed_table_put_first(hash, nbuckets, entry);
                }
            }
            wave_sync();
        }
        int hops = 0, dist_calc = 1, edges = 0, status = 0;  // status: 0 walking, 1 finished, 2 handed over
        uint32_t kseq = 0;
        B.hint3 = kInvalidId;
#ifdef GBNNS_STAMPS
        unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        STAMP(t_begin)
#endif
        while (true) {
            uint32_t node = 0, pred = kInvalidId, h2 = 0xFFFFFFFFu;
            STAMP(t0)
            if (!status && !B.select(node, pred, h2, lane)) status = 1;
            STAMP(t1)
            STAMP_ADD(0, t0, t1)
            if (lane == 0) *reinterpret_cast<uint4*>(mbox + ((kseq & 1u) ? kMbPost1 : kMbPost0)) = make_uint4(status ? kCoopDone : node, pred, h2, B.hint3);
            kseq += 1;
            coop_flag(mbox + kMbSeq, kseq, lane);   // posted (post number kseq sits in slot (kseq - 1) & 1)
            if (status) break;
            uint32_t rdy;     // the scout's expansion of `node` (prepared ahead: there already)
            if constexpr (NW == 2) {
                while (((rdy = coop_peek(mbox + kMbReady)) & 0x3FFFFFFFu) != node) __builtin_amdgcn_s_sleep(1);
            } else {          // the claimer's half and the ranger's in the buffer of this post
                const uint32_t b = kseq & 1u;
                uint32_t spins = 0;
                while (true) {
                    const uint32_t rc = coop_peek(mbox + kMbRdC + b), rr = coop_peek(mbox + kMbRdR + b);
                    if ((rc & 0x7FFFFFFFu) == node && rr == node) { rdy = b << 30 | (rc & 0x80000000u); break; }
                    if (++spins > kCoopSpinLimit) { rdy = 0x80000000u; break; }   // (never, unless the protocol is broken: hand the query over)
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            asm volatile("" ::: "memory");
            const uint32_t rb = rdy >> 30;   // bit 0: buffer, bit 1: the scout's stash is full
            STAMP(t2)
            STAMP_ADD(1, t1, t2)
            const uint64_t rv = reinterpret_cast<const uint64_t*>(bufs + 64 * (rb & 1u))[lane & 31];
            const uint32_t w0 = lane < 32 ? (uint32_t)rv : 0xFFFFFFFFu;
            const uint32_t dk = lane < 32 ? (uint32_t)(rv >> 32) : 0xFFFFFFFFu;
            const uint32_t nb = w0 & 0x7FFFFFFFu;
            const uint64_t mv = __ballot(w0 != 0xFFFFFFFFu);
            if (rb & 2u) status = 2;
            if (mv && !status) {
                // (the one-wavefront kernels test dist_calc + 64 before a pass's claims; here up to 32 ids of a prepared expansion are in
                // the table ahead of their count: the same fill bound with 32 more of margin -- a hand-over is exact at any point)
                if ((uint32_t)dist_calc + 96u > p.hash_limit) { status = 2; }
                else {
                    edges += __popcll(mv);
                    const uint64_t mclaimed = __ballot(w0 != 0xFFFFFFFFu && (w0 >> 31) != 0u);
                    STAMP(t3)
                    STAMP_ADD(2, t2, t3)
                    dist_calc += __popcll(mclaimed);
                    const bool fresh = __builtin_amdgcn_inverse_ballot_w64(mclaimed);
                    const uint64_t m = (B.l + B.f < ef) ? mclaimed : __ballot(fresh && dk < B.worst);
                    if (m && !B.insert(m, dk, nb, lane)) status = 2;
                    STAMP(t4)
                    STAMP_ADD(3, t3, t4)
                }
            }
            if (!status) hops += 1;
            // (a hand-over is posted as "done" by the next turn of the loop)
        }
#ifdef GBNNS_STAMPS
        {
            STAMP(t_end)
            if (lane == 0 && p.stamps) {
                for (int i = 0; i < 4; ++i) atomicAdd(p.stamps + i, seg[i]);
                atomicAdd(p.stamps + 6, t_end - t_begin);
                atomicAdd(p.stamps + 4, t_begin - k_begin);   // set-up
                atomicMax(p.stamps + 7, t_end - t_begin);     // the longest walk
            }
        }
#endif
        if (status == 2) {
            if (lane == 0) {
                const uint32_t s = atomicAdd(p.ovf_count, 1u);
                p.ovf_list[s] = qi;
                mbox[kMbKept] = 0xFFFFFFFFu;
            }
            coop_barrier();   // F: the scout learns there is nothing to re-rank
            return;
        }
        // ---- the outputs (flush, pop order), then both wavefronts re-rank
        WalkParams pw = p;
        pw.rr_db = nullptr;   // (BigList::finish would re-rank alone)
        B.template finish<8>(pw, qi, hops, dist_calc, edges, after_q, lane);
        const int kept = B.l < p.k ? B.l : p.k;
        if (lane == 0) mbox[kMbKept] = p.rr_db ? (uint32_t)kept : 0xFFFFFFFFu;
        coop_barrier();   // F: the base list is final
        if (!p.rr_db) return;
        RerankSrc a{p.rr_q, p.rr_qstride, p.rr_db, p.rr_dstride, p.rr_dim, p.rr_n};
        const uint64_t* b = B.base;
        uint64_t bestk = ~0ull;
        auto id_at = [&](int r) { return key_id(b[kept - 1 - r]); };
        if (p.rr_metric == 1) rerank_pairs_core<1>(a, qi, kept, reinterpret_cast<float*>(after_q), lane, id_at, 0, NW, &bestk);
        else rerank_pairs_core<0, kRerankLoads>(a, qi, kept, reinterpret_cast<float*>(after_q), lane, id_at, 0, NW, &bestk);
        coop_barrier();   // G: the other wavefronts' best keys are posted
        uint64_t other = *reinterpret_cast<const uint64_t*>(mbox + kMbBestB);
        if constexpr (NW == 3) {
            const uint64_t o2 = *reinterpret_cast<const uint64_t*>(mbox + kMbBestC);
            other = o2 < other ? o2 : other;
        }
        const uint64_t best = other < bestk ? other : bestk;
        const int win = (kept > 0 && best != ~0ull) ? (int)(uint32_t)best : -1;
        const uint32_t ans = key_id(b[win >= 0 ? kept - 1 - win : 0]);
        if (lane == 0) p.rr_out[qi] = win >= 0 ? ans : kInvalidId;
#ifdef GBNNS_STAMPS
        {
            STAMP(k_end)
            if (lane == 0 && p.stamps) {
                atomicAdd(p.stamps + 5, k_end - k_begin);      // the keeper's whole life
                if (kSpread) {
                    const unsigned long long r_end = __builtin_amdgcn_s_memrealtime();
                    atomicMax(p.stamps + 26, (1ull << 62) - r_begin);   // (the earliest start, from 2^62 down)
                    atomicMax(p.stamps + 27, r_begin);         // the latest start
                    atomicMax(p.stamps + 28, r_end);           // the latest end
                    atomicMax(p.stamps + 29, k_end - k_begin); // the longest life
                    atomicAdd(p.stamps + 30, ~0ull);           // (one fewer alive)
                }
            }
        }
#endif
        return;
    }

    // ====================================================== SCOUT ======================================================
    if constexpr (NW == 3) {
        if (wave == 1) coop_agent3<STEPS, 1>(p, mbox, bufs, pfs, hash, nbuckets, qreg, lane);
        else coop_agent3<STEPS, 2>(p, mbox, bufs, pfs + 192, hash, nbuckets, qreg, lane);
    } else {
    // The expansion it holds ready: of node `spec_node`; r_won = the slots its claims wrote (even lanes).
    // e_cmin / e_cnode: the closest new id of the last expansion handed over or prepared (all-ones / none), for the prediction.
    // Adjacency words requested ahead go straight into LDS (global_load_lds_dword: no register is written while the load is in flight,
    // so nothing the compiler does with registers can meet a half-arrived value; the compiler does not count these loads -- they are
    // waited for by hand, vmcnt(0), right before a slot is read): slot 0 = the last expansion's closest new id, slot 1 = the posted
    // runner-up, slot 2 = the keeper's hint (the entry after the runner-up: next hop's runner-up, probably).
    uint32_t spec_node = kCoopNoNode, r_won = 0u;
    uint32_t e_cmin = 0xFFFFFFFFu, e_cnode = kCoopNoNode;
    uint32_t pfn0 = kCoopNoNode, pfn1 = kCoopNoNode, pfn2 = kCoopNoNode;   // the nodes whose words the slots hold / will hold
#ifdef GBNNS_STAMPS
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned n_hit = 0, n_miss = 0, n_noguess = 0, n_guess_surv = 0, n_abort = 0;
#endif
    const uint32_t* const ell = p.ell;
    const uint32_t aslot = slot < p.ell_stride ? slot : p.ell_stride - 1u;   // (lanes beyond the row re-read its last word; masked on use)
    auto adjacency = [&](uint32_t node) -> uint32_t {
        const uint32_t* row = reinterpret_cast<const uint32_t*>(row_ptr<true>(reinterpret_cast<const float*>(ell), node, p.ell_stride));
        return (slot < p.ell_stride) ? row[slot] : kInvalidId;
    };
    const uint32_t pfs_lds = (uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(pfs));
    auto request_adjacency = [&](uint32_t k, uint32_t node) {   // node's adjacency words -> LDS slot k (wave-uniform arguments)
        const uint32_t* src = reinterpret_cast<const uint32_t*>(row_ptr<true>(reinterpret_cast<const float*>(ell), node, p.ell_stride)) + aslot;
        const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pfs_lds + 256u * k));
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };
    auto requested_slot = [&](uint32_t node) -> int { return node == pfn0 ? 0 : (node == pfn1 ? 1 : (node == pfn2 ? 2 : -1)); };
    // (ahead1 / ahead2: nodes whose adjacency words are worth requesting now -- the runner-up when it is not being expanded, the hint)
    auto expand = [&](uint32_t node, uint32_t bufno, bool prepared, uint32_t ahead1, uint32_t ahead2) -> bool {
        uint32_t* const buf = bufs + 64 * bufno;
        bool stash_full = false;
        uint32_t nb;
        const int have = requested_slot(node);
        if (have >= 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (requested a hop ago or more: landed, as a rule)
            const uint32_t wv = pfs[64 * have + lane];
            nb = (slot < p.ell_stride) ? wv : kInvalidId;
        } else {
            nb = adjacency(node);
        }
        STAMP(e0)
        const bool valid = nb != kInvalidId;
        const uint64_t mv = __ballot(valid);
        STAMP(e1)
        STAMP_ADD(3, e0, e1)
        RowRegs<kQSteps> rr;
        uint32_t roff = 0;
        auto request_rows = [&](bool want) {  // every lane loads (empty slots: row 0), as in walk_reg_big_one
            const uint32_t nbl = want ? nb : 0u;
            roff = nbl * kRowBytes + half * (kRowBytes / 2u);
            load_row<kQSteps>(rr, reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.db) + roff));
        };
        if constexpr (!LATE) request_rows(valid);
        // adjacency words for later hops, behind this hop's own loads (slot 1: the runner-up, slot 2: the hint)
        if (ahead1 != kInvalidId && ahead1 != node && requested_slot(ahead1) < 0) {
            pfn1 = ahead1;
            request_adjacency(1u, ahead1);
        }
        if (ahead2 < p.n && ahead2 != node && requested_slot(ahead2) < 0) {   // (the hint is unvalidated: never a row outside the table)
            pfn2 = ahead2;
            request_adjacency(2u, ahead2);
        }
        uint64_t mclaimed, movf = 0;
        uint32_t won;
        // (the scout carries a lot of wave-uniform state; where the compiler parks some of it in vector registers, the hand-written
        // blocks below still get scalars)
        auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
        const uint32_t hl = uni(hash_lds), nbk = uni(nbuckets), ctl = uni(vs_shr);
        if (ctl) mclaimed = coop_claim_quotient(hl, nbk, nb, mv & 0x5555555555555555ull, ctl, movf, won);
        else mclaimed = coop_claim_packed(hl, nbk, nb, mv & 0x5555555555555555ull, won);
        if (__builtin_expect(movf != 0, 0)) {
            if (prepared) {   // a probe sequence ran out: not ahead of time (the stash cannot be undone) -- take back what was claimed
                coop_unclaim(won, true);
                return false;
            }
            if (!stash_claim(hl, nbk, movf, nb, mclaimed, lane)) stash_full = true;   // the keeper hands the query over
        }
        const uint64_t mboth = mclaimed | (mclaimed << 1);
        STAMP(e2)
        STAMP_ADD(4, e1, e2)
        if constexpr (LATE) request_rows(__builtin_amdgcn_inverse_ballot_w64(mboth));
        uint32_t kd;
        if constexpr (STEPS == 8) kd = fkey_sumsq(l2_pair_from_regs(rr, qreg.v));
        else kd = fkey_sumsq(l2_pair_from_regs_wide<kQSteps>(rr, qreg.v));
        asm volatile("" ::"v"(roff));  // the address register must not double as a load destination
        const bool fresh = __builtin_amdgcn_inverse_ballot_w64(mboth);
        const uint32_t w0 = valid ? (nb | (fresh ? 0x80000000u : 0u)) : 0xFFFFFFFFu;
        buf[lane] = half ? kd : w0;
        coop_flag(mbox + kMbReady, node | bufno << 30 | (stash_full ? 1u << 31 : 0u), lane);   // published: the keeper may take it
        STAMP(e3)
        STAMP_ADD(5, e2, e3)
        r_won = won;
        // the closest new id (for the prediction), and its adjacency word requested now
        uint32_t x = (fresh && half) ? kd : 0xFFFFFFFFu;
        const uint32_t dkf = x;
        x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xf, 0xf, false));   // quad_perm 1,0,3,2
        x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xf, 0xf, false));   // quad_perm 2,3,0,1
        x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xf, 0xf, false));  // row_half_mirror
        x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xf, 0xf, false));  // row_mirror
        const uint32_t dmin = min(min(readlane_u32(x, 0), readlane_u32(x, 16)), min(readlane_u32(x, 32), readlane_u32(x, 48)));
        e_cmin = 0xFFFFFFFFu;
        e_cnode = kCoopNoNode;
        if (dmin != 0xFFFFFFFFu) {
            const uint64_t me = __ballot(dkf == dmin);
            if ((me & (me - 1)) == 0) {   // unique
                e_cmin = dmin;
                e_cnode = readlane_u32(nb, __ffsll((unsigned long long)me) - 1);
                if (requested_slot(e_cnode) < 0) {
                    pfn0 = e_cnode;
                    request_adjacency(0u, e_cnode);
                }
            }
        }
        STAMP(e4)
        STAMP_ADD(6, e3, e4)
        return true;
    };
#ifdef GBNNS_STAMPS
    unsigned it_no = 0;
    unsigned long long it_t[4] = {0, 0, 0, 0};
    STAMP(sc_begin)
#endif
    uint32_t expect = 0, cur = 0;
    while (true) {
        STAMP(s0)
#ifdef GBNNS_STAMPS
        if (it_no < 4) it_t[it_no] = s0 - sc_begin;
        it_no += 1;
#endif
        uint32_t seq;   // the keeper's next post -- or the one after it (see the mailbox notes: then the prepared expansion has been taken)
        while ((seq = coop_peek(mbox + kMbSeq)) <= expect) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        if (seq != expect + 1u) spec_node = kCoopNoNode;
        expect = seq;
        STAMP(s1)
        STAMP_ADD(0, s0, s1)
        const uint4 post = *reinterpret_cast<const uint4*>(mbox + (((seq - 1u) & 1u) ? kMbPost1 : kMbPost0));
        const uint32_t node = (uint32_t)__builtin_amdgcn_readfirstlane((int)post.x);
        const uint32_t pred = (uint32_t)__builtin_amdgcn_readfirstlane((int)post.y);
        const uint32_t h2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)post.z);
        const uint32_t hint = (uint32_t)__builtin_amdgcn_readfirstlane((int)post.w);
        if (node == kCoopDone) break;
#ifdef GBNNS_STAMPS
        if (node != spec_node) n_miss += 1; else n_hit += 1;
#endif
        if (node != spec_node) {   // not what was prepared (or nothing was): the keeper is waiting
            if (spec_node != kCoopNoNode) coop_unclaim(r_won, vs_shr != 0u);
            expand(node, cur, false, pred, hint);
        }
        STAMP(s2)
        STAMP_ADD(1, s1, s2)
        cur ^= 1u;   // buffer cur ^ 1 is the keeper's now; the next expansion goes to the other one (the keeper has finished with it: it
                     // read it before the selection this turn started with)
        // the next node, probably: the closest new id just handed over if it beats the runner-up, else the runner-up
        uint32_t guess = pred == kInvalidId ? kCoopNoNode : pred;
        if (e_cmin < h2 && e_cnode != kCoopNoNode) {
            guess = e_cnode;
#ifdef GBNNS_STAMPS
            n_guess_surv += 1;
#endif
        }
        STAMP(s3)
        spec_node = kCoopNoNode;
        if (guess != kCoopNoNode) {
            if (expand(guess, cur, true, pred, hint)) spec_node = guess;
#ifdef GBNNS_STAMPS
            else n_abort += 1;
#endif
        }
        STAMP(s4)
        STAMP_ADD(2, s3, s4)
#ifdef GBNNS_STAMPS
        if (guess == kCoopNoNode) n_noguess += 1;
#endif
    }
#ifdef GBNNS_STAMPS
    if (lane == 0 && p.stamps) {
        atomicAdd(p.stamps + 16, seg[0]); atomicAdd(p.stamps + 17, seg[1]); atomicAdd(p.stamps + 18, seg[2]);
        atomicAdd(p.stamps + 19, (unsigned long long)n_hit); atomicAdd(p.stamps + 20, (unsigned long long)n_miss);
        atomicAdd(p.stamps + 21, (unsigned long long)n_noguess); atomicAdd(p.stamps + 22, (unsigned long long)n_guess_surv);
        atomicAdd(p.stamps + 23, (unsigned long long)n_abort);
#ifndef GBNNS_STAMPS_SPREAD
        for (int i = 0; i < 4; ++i) atomicAdd(p.stamps + 28 + i, it_t[i]);   // 28 .. 31: when the scout's iterations 0 .. 3 began, since its loop started
        for (int i = 3; i < 7; ++i) atomicAdd(p.stamps + 21 + i, seg[i]);   // 24 .. 27: inside the expansions (adjacency word, claims, rows + distances, tail)
#endif
    }
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (adjacency words still in flight)
    coop_barrier();   // F
    const uint32_t keptw = (uint32_t)__builtin_amdgcn_readfirstlane((int)mbox[kMbKept]);
    if (keptw == 0xFFFFFFFFu) return;   // handed over, or no re-rank asked for
    {
        const int kept = (int)keptw;
        const uint64_t* b = reinterpret_cast<const uint64_t*>(smem) + kRegTieCap + kRegStageSlots;   // BigList::base
        RerankSrc a{p.rr_q, p.rr_qstride, p.rr_db, p.rr_dstride, p.rr_dim, p.rr_n};
        uint64_t bestk = ~0ull;
        auto id_at = [&](int r) { return key_id(b[kept - 1 - r]); };
        // (the same staging area as the keeper's: both wavefronts write the whole original-space query there -- the same values -- and
        // each reads what it has written itself, in its own issue order)
        float* qf = reinterpret_cast<float*>(after_q);
        if (p.rr_metric == 1) rerank_pairs_core<1>(a, qi, kept, qf, lane, id_at, wave, NW, &bestk);
        else rerank_pairs_core<0, kRerankLoads>(a, qi, kept, qf, lane, id_at, wave, NW, &bestk);
        if (lane == 0) *reinterpret_cast<uint64_t*>(mbox + (wave == 1 ? kMbBestB : kMbBestC)) = bestk;
    }
    coop_barrier();   // G
}

template <int STEPS>
hipError_t launch_coop_t(const WalkParams& p, size_t lds, hipStream_t s) {
    auto go = [&](auto kernel, unsigned threads) -> hipError_t {
        hipError_t e = set_lds(kernel, lds);
        if (e != hipSuccess) return e;
        g_walk_first_fn = reinterpret_cast<const void*>(kernel);
        hipLaunchKernelGGL(kernel, dim3(p.nq), dim3(threads), lds, s, p);
        return hipGetLastError();
    };
    if (p.coop >= 2) return go(walk_coop_kernel<STEPS, false, 3>, 192u);
    return p.late_rows ? go(walk_coop_kernel<STEPS, true, 2>, 128u) : go(walk_coop_kernel<STEPS, false, 2>, 128u);
}

}  // namespace

bool walk_coop_serves(const WalkParams& p, int metric) {
    const bool rows = p.dim == p.dstride && (p.dim == 32u || p.dim == 48u || p.dim == 64u);
    // (a visited set in the packed form: no id may look like a half-written slot -- see visited_test_mask_packed)
    return metric == 0 && rows && p.ef > kHot2MaxEf && p.ef <= kRegListMaxEf && walk_off32(p) && p.n < 0xFF0000u && !p.aux_ell && p.ell_stride <= 32u &&
           p.n_entries <= 1u;
}

hipError_t launch_walk_coop(const WalkParams& p, hipStream_t s) {
    if (p.nq == 0) return hipSuccess;
    if (!walk_coop_serves(p, 0)) return hipErrorInvalidValue;
    const size_t lds = std::max<size_t>(walk_fast_lds_bytes(p, false), p.coop_lds_floor);
    switch (p.dim) {
        case 32: return launch_coop_t<8>(p, lds, s);
        case 48: return launch_coop_t<12>(p, lds, s);
        default: return launch_coop_t<16>(p, lds, s);
    }
}

}  // namespace gbnns
