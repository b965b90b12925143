// swap_probe (diagnostic): where v_permlane16_swap_b32 / v_permlane32_swap_b32 move the lanes of their two operands.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    const unsigned l = threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(100u + l, 200u + l, false, false);
    o[l] = r[0]; o[64 + l] = r[1];
    auto s = __builtin_amdgcn_permlane32_swap(100u + l, 200u + l, false, false);
    o[128 + l] = s[0]; o[192 + l] = s[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* n[] = {"permlane16_swap first ", "permlane16_swap second", "permlane32_swap first ", "permlane32_swap second"};
    for (int i = 0; i < 4; ++i) { printf("%s:", n[i]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[i * 64 + l]); printf("\n"); }
    return 0;
}
