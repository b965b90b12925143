// Micro-benchmark (diagnostic): sustained instruction issue per CU for VALU / SALU mixes at 5
// wavefronts per SIMD (the walk kernel's residency).  Prints instructions per cycle and CU.
// Build: hipcc --offload-arch=gfx950 -O2 -o issue_mix issue_mix.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define R8(A) A A A A A A A A
template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned* out, int iters) {
    extern __shared__ unsigned char smem[];
    unsigned a = threadIdx.x, b = 3, c = 5, d = 7;
    unsigned s = iters, t = 1;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0)  // VALU only (independent)
            asm volatile(R8("v_add_u32 %0, %0, %1\n\tv_xor_b32 %2, %2, %3\n\t") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        else if (MODE == 1)  // SALU only
            asm volatile(R8("s_add_u32 %0, %0, %1\n\ts_xor_b32 %1, %1, %0\n\t") : "+s"(s), "+s"(t)::"scc");
        else if (MODE == 2)  // 1 VALU : 1 SALU
            asm volatile(R8("v_add_u32 %0, %0, %1\n\ts_add_u32 %2, %2, %3\n\t") : "+v"(a), "+v"(b), "+s"(s), "+s"(t)::"scc");
        else if (MODE == 3)  // dependent VALU chain
            asm volatile(R8("v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\t") : "+v"(a), "+v"(b));
        else  // 1 VALU : 1 SALU, both dependent chains
            asm volatile(R8("v_add_u32 %0, %0, %1\n\ts_add_u32 %2, %2, %2\n\t") : "+v"(a), "+v"(b), "+s"(s), "+s"(t)::"scc");
    }
    if (a + b + c + d + s + t == 0xdeadbeef) out[0] = a + smem[0];
}

template <int MODE>
void run(unsigned* out, const char* name, int waves_per_cu) {
    const int iters = 4000, grid = 256 * waves_per_cu;
    const size_t lds = 160 * 1024 / waves_per_cu / 512 * 512;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), lds, 0, out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), lds, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)grid * iters * 16;
    printf("%-34s %2d waves/CU: %7.3f ms  %6.3f instr/ns/CU  (= instr/cycle/CU at 1 GHz; / clock in GHz)\n", name, waves_per_cu, ms,
           instr / 256 / (ms * 1e6));
}

int main() {
    unsigned* out;
    (void)hipMalloc(&out, 64);
    for (int w : {4, 8, 20}) {
        if (w == 4) { run<0>(out, "VALU independent", 4); run<1>(out, "SALU", 4); run<2>(out, "VALU:SALU 1:1", 4); run<3>(out, "VALU dependent", 4); run<4>(out, "VALU:SALU 1:1 dependent", 4); }
        if (w == 8) { run<0>(out, "VALU independent", 8); run<1>(out, "SALU", 8); run<2>(out, "VALU:SALU 1:1", 8); run<3>(out, "VALU dependent", 8); run<4>(out, "VALU:SALU 1:1 dependent", 8); }
        if (w == 20) { run<0>(out, "VALU independent", 20); run<1>(out, "SALU", 20); run<2>(out, "VALU:SALU 1:1", 20); run<3>(out, "VALU dependent", 20); run<4>(out, "VALU:SALU 1:1 dependent", 20); }
    }
    return 0;
}
