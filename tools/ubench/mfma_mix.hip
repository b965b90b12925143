// mfma_mix (diagnostic): can the matrix pipe deliver the projection's PRODUCTS while the vector pipe does its sums?
// v_mfma_f32_*x1_*b_f32 with C = 0 returns fma(a, b, +0) = the correctly rounded product a * b for every (row, column) pair of
// each block: an exact stand-in for v_pk_mul_f32.  The loops below issue one such instruction and the v_pk_add_f32 that would
// consume the PREVIOUS one's products (two product register sets), and report ns per product-and-sum pair and SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_mix.hip -o tools/ubench/mfma_mix ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CLOB_0_127 "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99"

#define ADD(ACC, P) "v_pk_add_f32 v[" #ACC ":" #ACC "+1], v[" #ACC ":" #ACC "+1], v[" #P ":" #P "+1]\n\t"
// 16 sums on accumulators v[64..95] from products v[P..P+31]
#define ADD16(P)                                                                                                            \
    ADD(64, P) ADD(66, P + 2) ADD(68, P + 4) ADD(70, P + 6) ADD(72, P + 8) ADD(74, P + 10) ADD(76, P + 12) ADD(78, P + 14)  \
    ADD(80, P + 16) ADD(82, P + 18) ADD(84, P + 20) ADD(86, P + 22) ADD(88, P + 24) ADD(90, P + 26) ADD(92, P + 28) ADD(94, P + 30)
#define ADD8(P) ADD(64, P) ADD(66, P + 2) ADD(68, P + 4) ADD(70, P + 6) ADD(72, P + 8) ADD(74, P + 10) ADD(76, P + 12) ADD(78, P + 14)
#define ADD2(P) ADD(64, P) ADD(66, P + 2)
// the same sums with single (not packed) instructions
#define SADD(ACC, P) "v_add_f32 v[" #ACC "], v[" #ACC "], v[" #P "]\n\t"
#define SADD8(A, P) SADD(A, P) SADD(A + 1, P + 1) SADD(A + 2, P + 2) SADD(A + 3, P + 3) SADD(A + 4, P + 4) SADD(A + 5, P + 5) SADD(A + 6, P + 6) SADD(A + 7, P + 7)
#define SADD16(P) SADD8(64, P) SADD8(72, P + 8)
#define SADD32(P) SADD8(64, P) SADD8(72, P + 8) SADD8(80, P + 16) SADD8(88, P + 24)

// MODE 0: 32x32x1_2b + 16 sums; 1: 16x16x1_4b + 8 sums; 2: 4x4x1_16b + 2 sums; 3: sums only (16); 4: 32x32x1_2b only;
// 5: v_pk_mul + v_pk_add (what the projection does today), 16 pairs
template <int MODE>
__global__ void mix_kernel(float* out, int iters) {
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0)
            asm volatile("v_mfma_f32_32x32x1_2b_f32 v[0:31], %0, %1, 0\n\t" ADD16(32) "v_mfma_f32_32x32x1_2b_f32 v[32:63], %0, %1, 0\n\t" ADD16(0)
                         :: "v"(a), "v"(b) : CLOB_0_127);
        else if (MODE == 1)
            asm volatile("v_mfma_f32_16x16x1_4b_f32 v[0:15], %0, %1, 0\n\t" ADD8(32) "v_mfma_f32_16x16x1_4b_f32 v[32:47], %0, %1, 0\n\t" ADD8(0)
                         "v_mfma_f32_16x16x1_4b_f32 v[0:15], %0, %1, 0\n\t" ADD8(32) "v_mfma_f32_16x16x1_4b_f32 v[32:47], %0, %1, 0\n\t" ADD8(0)
                         :: "v"(a), "v"(b) : CLOB_0_127);
        else if (MODE == 2)
            asm volatile("v_mfma_f32_4x4x1_16b_f32 v[0:3], %0, %1, 0\n\t" ADD2(32) "v_mfma_f32_4x4x1_16b_f32 v[32:35], %0, %1, 0\n\t" ADD2(0)
                         "v_mfma_f32_4x4x1_16b_f32 v[0:3], %0, %1, 0\n\t" ADD2(32) "v_mfma_f32_4x4x1_16b_f32 v[32:35], %0, %1, 0\n\t" ADD2(0)
                         "v_mfma_f32_4x4x1_16b_f32 v[0:3], %0, %1, 0\n\t" ADD2(32) "v_mfma_f32_4x4x1_16b_f32 v[32:35], %0, %1, 0\n\t" ADD2(0)
                         "v_mfma_f32_4x4x1_16b_f32 v[0:3], %0, %1, 0\n\t" ADD2(32) "v_mfma_f32_4x4x1_16b_f32 v[32:35], %0, %1, 0\n\t" ADD2(0)
                         "v_mfma_f32_4x4x1_16b_f32 v[0:3], %0, %1, 0\n\t" ADD2(32) "v_mfma_f32_4x4x1_16b_f32 v[32:35], %0, %1, 0\n\t" ADD2(0)
                         "v_mfma_f32_4x4x1_16b_f32 v[0:3], %0, %1, 0\n\t" ADD2(32) "v_mfma_f32_4x4x1_16b_f32 v[32:35], %0, %1, 0\n\t" ADD2(0)
                         "v_mfma_f32_4x4x1_16b_f32 v[0:3], %0, %1, 0\n\t" ADD2(32) "v_mfma_f32_4x4x1_16b_f32 v[32:35], %0, %1, 0\n\t" ADD2(0)
                         "v_mfma_f32_4x4x1_16b_f32 v[0:3], %0, %1, 0\n\t" ADD2(32) "v_mfma_f32_4x4x1_16b_f32 v[32:35], %0, %1, 0\n\t" ADD2(0)
                         :: "v"(a), "v"(b) : CLOB_0_127);
        else if (MODE == 6)
            asm volatile("v_mfma_f32_32x32x1_2b_f32 v[0:31], %0, %1, 0\n\t" SADD32(32) "v_mfma_f32_32x32x1_2b_f32 v[32:63], %0, %1, 0\n\t" SADD32(0)
                         :: "v"(a), "v"(b) : CLOB_0_127);
        else if (MODE == 7)
            asm volatile("v_mfma_f32_16x16x1_4b_f32 v[0:15], %0, %1, 0\n\t" SADD16(32) "v_mfma_f32_16x16x1_4b_f32 v[32:47], %0, %1, 0\n\t" SADD16(0)
                         "v_mfma_f32_16x16x1_4b_f32 v[0:15], %0, %1, 0\n\t" SADD16(32) "v_mfma_f32_16x16x1_4b_f32 v[32:47], %0, %1, 0\n\t" SADD16(0)
                         :: "v"(a), "v"(b) : CLOB_0_127);
        else if (MODE == 8)
            asm volatile(SADD32(32) SADD32(0) :: "v"(a), "v"(b) : CLOB_0_127);
        else if (MODE == 3)
            asm volatile(ADD16(32) ADD16(0) :: "v"(a), "v"(b) : CLOB_0_127);
        else if (MODE == 4)
            asm volatile("v_mfma_f32_32x32x1_2b_f32 v[0:31], %0, %1, 0\n\tv_mfma_f32_32x32x1_2b_f32 v[32:63], %0, %1, 0\n\t" :: "v"(a), "v"(b) : CLOB_0_127);
        else {
#define MUL(P) "v_pk_mul_f32 v[" #P ":" #P "+1], v[96:97], v[98:99]\n\t"
#define MA4(P) MUL(P) MUL(P + 2) MUL(P + 4) MUL(P + 6) ADD(64, P) ADD(66, P + 2) ADD(68, P + 4) ADD(70, P + 6)
            asm volatile(MA4(0) MA4(8) MA4(16) MA4(24) MA4(32) MA4(40) MA4(48) MA4(56) :: "v"(a), "v"(b) : CLOB_0_127);
        }
    }
    float s;
    asm volatile("v_add_f32 %0, v64, v65\n\tv_add_f32 %0, %0, v0" : "=v"(s) :: CLOB_0_127);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, double pairs_per_iter_per_wave, float* d_out) {
    for (int wps : {1, 2, 4, 5}) {
        const int iters = 20000, blocks = wps == 5 ? 512 : 256, threads = wps == 5 ? 640 : 256 * wps;  // (5: two blocks of 10 wavefronts per CU)
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(mix_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, d_out, 100);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mix_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, d_out, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double pairs = pairs_per_iter_per_wave * iters * (wps == 5 ? 5 : wps);  // product-and-sum pairs (lane level) one SIMD processed
        printf("  %-44s %d wave(s)/SIMD: %.3f ms, %.1f G multiply-adds/s per SIMD\n", name, wps, ms, pairs / (ms * 1e6));
    }
}

int main() {
    float* d_out;
    (void)hipMalloc(&d_out, 256 * 1024 * 4);
    run<5>("32 x (v_pk_mul_f32 + v_pk_add_f32): today", 64 * 64, d_out);
    run<0>("32x32x1_2b (C = 0) + 16 v_pk_add_f32", 2 * 2048, d_out);
    run<1>("16x16x1_4b (C = 0) + 8 v_pk_add_f32", 4 * 1024, d_out);
    run<2>("4x4x1_16b (C = 0) + 2 v_pk_add_f32", 16 * 256, d_out);
    run<6>("32x32x1_2b (C = 0) + 32 v_add_f32", 2 * 2048, d_out);
    run<7>("16x16x1_4b (C = 0) + 16 v_add_f32", 4 * 1024, d_out);
    run<8>("32 v_add_f32 alone (sums of 2048 products)", 2 * 2048, d_out);
    run<3>("16 v_pk_add_f32 alone (sums of 2048 products)", 2 * 2048, d_out);
    run<4>("32x32x1_2b alone", 2 * 2048, d_out);
    return 0;
}
