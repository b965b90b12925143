// Micro-benchmark (diagnostic): issue cost in cycles of single gfx950 instructions for one wavefront
// per SIMD.  16 independent copies per loop iteration, 2000 iterations, s_memtime around the loop.
// Build: hipcc --offload-arch=gfx950 -O2 -o instr_cost instr_cost.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP16S(A) A "\n\t" A "\n\t" A "\n\t" A "\n\t" A "\n\t" A "\n\t" A "\n\t" A "\n\t" A "\n\t" A "\n\t" A "\n\t" A "\n\t" A "\n\t" A "\n\t" A "\n\t" A

#define TEST(NAME, ASM, ...)                                                                   \
    __global__ void k_##NAME(unsigned long long* out, int iters) {                             \
        unsigned a0 = threadIdx.x, a1 = threadIdx.x * 3u + 1u, a2 = 7u, a3 = (threadIdx.x & 63u) * 4u;             \
        unsigned long long b0 = a0, b1 = a1 * 77ull, m0 = 0;                                   \
        float f0 = a0, f1 = 1.5f;                                                              \
        typedef float f2 __attribute__((ext_vector_type(2)));                                  \
        f2 p0 = {f0, f1}, p1 = {f1, f0}, p2 = {0.f, 0.f};                                      \
        int s0 = 3, s1 = iters;                                                                \
        unsigned long long su = (unsigned long long)iters * 77ull;                             \
        unsigned long long t0 = __builtin_readcyclecounter();                                  \
        for (int i = 0; i < iters; ++i) {                                                      \
            asm volatile(REP16S(ASM) : __VA_ARGS__);                                           \
        }                                                                                      \
        unsigned long long t1 = __builtin_readcyclecounter();                                  \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                       \
        if (a2 + a3 + (unsigned)b0 + (unsigned)m0 + (unsigned)f0 + (unsigned)p2.x + s0 == 0xdeadbeef) out[1] = a0 + a1 + b1 + s1 + su + (unsigned)p0.x + (unsigned)p1.y; \
    }

TEST(v_add_u32, "v_add_u32 %0, %1, %2", "=v"(a2) : "v"(a0), "v"(a1))
TEST(v_xor_b32, "v_xor_b32 %0, %1, %2", "=v"(a2) : "v"(a0), "v"(a1))
TEST(v_min3_u32, "v_min3_u32 %0, %1, %2, %1", "=v"(a2) : "v"(a0), "v"(a1))
TEST(v_add3_u32, "v_add3_u32 %0, %1, %2, %1", "=v"(a2) : "v"(a0), "v"(a1))
TEST(v_mul_lo_u32, "v_mul_lo_u32 %0, %1, %2", "=v"(a2) : "v"(a0), "v"(a1))
TEST(v_mul_hi_u32, "v_mul_hi_u32 %0, %1, %2", "=v"(a2) : "v"(a0), "v"(a1))
TEST(v_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %3", "=v"(b0) : "v"(a0), "v"(a1), "v"(b1) : "vcc")
TEST(v_lshl_add_u64, "v_lshl_add_u64 %0, %1, 2, %2", "=v"(b0) : "v"(b1), "v"(b1))
TEST(v_cmp_lt_u32, "v_cmp_lt_u32 vcc, %1, %2", "=v"(a2) : "v"(a0), "v"(a1) : "vcc")
TEST(v_cmp_lt_u64, "v_cmp_lt_u64 vcc, %1, %2", "=v"(a2) : "v"(b0), "v"(b1) : "vcc")
TEST(v_cmp_lt_u64_sgpr, "v_cmp_lt_u64 %0, %1, %2", "=s"(m0) : "v"(b0), "v"(b1))
TEST(v_cndmask, "v_cndmask_b32 %0, %1, %2, vcc", "=v"(a2) : "v"(a0), "v"(a1))
TEST(v_addc, "v_addc_co_u32 %0, vcc, 0, %1, vcc", "=v"(a2) : "v"(a0) : "vcc")
TEST(v_mov_dpp_wave_shr, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf", "+v"(a2) : "v"(a0))
TEST(v_mov_dpp_row_shr, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "+v"(a2) : "v"(a0))
TEST(v_readlane, "v_readlane_b32 %0, %1, 5", "=s"(s0) : "v"(a0))
TEST(v_readlane_s, "v_readlane_b32 %0, %1, %2", "=s"(s0) : "v"(a0), "s"(s1))
TEST(v_readfirstlane, "v_readfirstlane_b32 %0, %1", "=s"(s0) : "v"(a0))
TEST(v_add_f32, "v_add_f32 %0, %1, %2", "=v"(f0) : "v"(f1), "v"(f1))
TEST(v_mul_f32, "v_mul_f32 %0, %1, %2", "=v"(f0) : "v"(f1), "v"(f1))
TEST(v_pk_add_f32, "v_pk_add_f32 %0, %1, %2", "=v"(p2) : "v"(p0), "v"(p1))
TEST(v_pk_mul_f32, "v_pk_mul_f32 %0, %1, %2", "=v"(p2) : "v"(p0), "v"(p1))
TEST(v_pk_add_f32_neg, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]", "=v"(p2) : "v"(p0), "v"(p1))
TEST(v_ashrrev, "v_ashrrev_i32 %0, 31, %1", "=v"(a2) : "v"(a0))
TEST(v_lshl_add_u32, "v_lshl_add_u32 %0, %1, 4, %2", "=v"(a2) : "v"(a0), "v"(a1))
TEST(v_bitop3_b16, "v_bitop3_b16 %0, %1, 3, %2 bitop3:0xc8", "=v"(a2) : "v"(a0), "v"(a1))
TEST(s_add_u32, "s_add_u32 %0, %1, %1", "=s"(s0) : "s"(s1) : "scc")
TEST(s_and_b64, "s_and_b64 %0, %1, %1", "=s"(m0) : "s"(su) : "scc")
TEST(s_ff1_b64, "s_ff1_i32_b64 %0, %1", "=s"(s0) : "s"(su))
TEST(s_bcnt1_b64, "s_bcnt1_i32_b64 %0, %1", "=s"(s0) : "s"(su) : "scc")
TEST(s_nop0, "s_nop 0", "=s"(s0) : "s"(s1))
TEST(s_cmp, "s_cmp_lt_u32 %1, %1", "=s"(s0) : "s"(s1) : "scc")
TEST(v_cndmask_e64, "v_cndmask_b32_e64 %0, %1, %2, %3", "=v"(a2) : "v"(a0), "v"(a1), "s"(su))
TEST(v_cndmask_const, "v_cndmask_b32_e64 %0, 0, 1, %1", "=v"(a2) : "s"(su))
TEST(v_cmp_eq_sgpr_src, "v_cmp_eq_u32 vcc, %2, %1", "=v"(a2) : "v"(a0), "s"(s1) : "vcc")
TEST(v_add_sgpr_src, "v_add_u32 %0, %2, %1", "=v"(a2) : "v"(a0), "s"(s1))
TEST(v_mov_sgpr, "v_mov_b32 %0, %1", "=v"(a2) : "s"(s1))
TEST(ds_read_b32, "ds_read_b32 %0, %1", "=v"(a2) : "v"(a3))
TEST(ds_read_b128_wait, "ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)", "=v"(a2) : "v"(a3))
TEST(s_waitcnt, "s_waitcnt lgkmcnt(0)", "=s"(s0) : "s"(s1))
TEST(s_cbranch_nt, "s_cbranch_scc1 1f\n1:", "=s"(s0) : "s"(s1))
TEST(v_cmp_x, "v_cmp_lt_u32 %0, %1, %2", "=s"(m0) : "v"(a0), "v"(a1))
TEST(s_andn2_exec, "s_andn2_b64 %0, %1, exec", "=s"(m0) : "s"(su) : "scc")
TEST(s_mov_b64, "s_mov_b64 %0, %1", "=s"(m0) : "s"(su))
TEST(s_lshl_b64, "s_lshl_b64 %0, %1, 1", "=s"(m0) : "s"(su) : "scc")
// dependent chains (latency): each copy consumes the previous result
TEST(dep_v_add_u32, "v_add_u32 %0, %0, %1", "+v"(a2) : "v"(a0))
TEST(dep_v_pk_add_f32, "v_pk_add_f32 %0, %0, %1", "+v"(p2) : "v"(p0))
TEST(dep_v_add_f32, "v_add_f32 %0, %0, %1", "+v"(f0) : "v"(f1))
TEST(dep_s_add_u32, "s_add_u32 %0, %0, %1", "+s"(s0) : "s"(s1) : "scc")
TEST(dep_cmp_cndmask, "v_cmp_lt_u32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %1, %0, vcc", "+v"(a2) : "v"(a0) : "vcc")
TEST(dep_readlane_cmp, "v_readlane_b32 %0, %1, 3\n\ts_nop 1\n\tv_add_u32 %1, %0, %1", "+s"(s0), "+v"(a2) :)
TEST(dep_valu_salu, "v_cmp_lt_u32 vcc, %0, %3\n\ts_and_b64 %1, vcc, exec\n\ts_bcnt1_i32_b64 %2, %1\n\tv_add_u32 %0, %2, %0", "+v"(a2), "+s"(m0), "+s"(s0) : "v"(a0) : "vcc", "scc")

struct T { const char* name; void (*fn)(unsigned long long*, int); int per; };
#define E(NAME, PER) {#NAME, k_##NAME, PER}

int main(int argc, char** argv) {
    std::vector<T> tests = {
        E(v_add_u32,1), E(v_xor_b32,1), E(v_min3_u32,1), E(v_add3_u32,1), E(v_mul_lo_u32,1), E(v_mul_hi_u32,1), E(v_mad_u64_u32,1),
        E(v_lshl_add_u64,1), E(v_cmp_lt_u32,1), E(v_cmp_lt_u64,1), E(v_cmp_lt_u64_sgpr,1), E(v_cndmask,1), E(v_addc,1),
        E(v_mov_dpp_wave_shr,1), E(v_mov_dpp_row_shr,1), E(v_readlane,1), E(v_readlane_s,1), E(v_readfirstlane,1), E(v_add_f32,1), E(v_mul_f32,1),
        E(v_pk_add_f32,1), E(v_pk_mul_f32,1), E(v_pk_add_f32_neg,1), E(v_ashrrev,1), E(v_lshl_add_u32,1), E(v_bitop3_b16,1),
        E(s_add_u32,1), E(s_and_b64,1), E(s_ff1_b64,1), E(s_bcnt1_b64,1), E(s_nop0,1), E(s_cmp,1),
        E(v_cndmask_e64,1), E(v_cndmask_const,1), E(v_cmp_eq_sgpr_src,1), E(v_add_sgpr_src,1), E(v_mov_sgpr,1), E(ds_read_b32,1), E(ds_read_b128_wait,1), E(s_waitcnt,1), E(s_cbranch_nt,1), E(v_cmp_x,1), E(s_andn2_exec,1), E(s_mov_b64,1), E(s_lshl_b64,1),
        E(dep_v_add_u32,1), E(dep_v_pk_add_f32,1), E(dep_v_add_f32,1), E(dep_s_add_u32,1), E(dep_cmp_cndmask,1), E(dep_readlane_cmp,1), E(dep_valu_salu,1)};
    unsigned long long* d;
    if (hipMalloc(&d, 4096) != hipSuccess) return 1;
    const int iters = 2000;
    for (int waves = 1; waves <= 4; waves *= 2) {   // 1 or 2 wavefronts per SIMD
        printf("== %d wavefront(s) per SIMD (block of %d threads), cycles per instruction copy\n", waves, 256 * waves);
        for (auto& t : tests) {
            unsigned long long h = 0, best = ~0ull;
            for (int r = 0; r < 3; ++r) {
                hipLaunchKernelGGL(t.fn, dim3(1), dim3(256 * waves), 0, 0, d, iters);
                if (hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
                if (h < best) best = h;
            }
            printf("%-22s %7.2f\n", t.name, (double)best / (iters * 16.0));
        }
    }
    return 0;
}
