// Which CUs does a stream created with hipExtStreamCreateWithCUMask use?  (GPU box.)
// For each mask: launches 8 192 one-wavefront blocks that spin ~20 us and record (XCC_ID, HW_ID); prints the distinct
// (xcc, se, sh, cu) places seen per XCC.  Used to lay out the complementary masks of the projection / walk streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void where_kernel(uint2* out, int spin) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) out[blockIdx.x] = make_uint2(xcc, hw);
}

static void run(const char* name, const std::vector<uint32_t>& mask, bool masked) {
    hipStream_t s;
    if (masked) {
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
        if (e != hipSuccess) { printf("%s: create -> %s\n", name, hipGetErrorString(e)); return; }
    } else CK(hipStreamCreate(&s));
    const int nb = 8192;
    uint2* d; CK(hipMalloc(&d, nb * sizeof(uint2)));
    std::vector<uint2> h(nb);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    where_kernel<<<nb, 64, 0, s>>>(d, 2000);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    where_kernel<<<nb, 64, 0, s>>>(d, 2000);  // 100 MHz wall clock: 2 000 ticks = 20 us
    CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipMemcpy(h.data(), d, nb * sizeof(uint2), hipMemcpyDeviceToHost));
    std::map<uint32_t, std::set<uint32_t>> per;
    for (auto& v : h) {
        const uint32_t xcc = v.x & 15u;
        const uint32_t cu = (v.y >> 8) & 15u, sh = (v.y >> 12) & 1u, se = (v.y >> 13) & 7u;
        per[xcc].insert(se << 8 | sh << 4 | cu);
    }
    size_t total = 0;
    printf("%-28s %.3f ms :", name, ms);
    for (auto& kv : per) { printf(" x%u:%zu", kv.first, kv.second.size()); total += kv.second.size(); }
    printf("  = %zu CUs\n", total);
    if (getenv("CUMASK_VERBOSE"))
        for (auto& kv : per) { printf("   xcc %u:", kv.first); for (uint32_t c : kv.second) printf(" %u.%u.%u", c >> 8, (c >> 4) & 15, c & 15); printf("\n"); }
    CK(hipFree(d)); CK(hipStreamDestroy(s));
}

int main() {
    std::vector<uint32_t> all(8, 0xffffffffu);
    run("plain stream", all, false);
    run("all 256 bits", all, true);
    { std::vector<uint32_t> m(8, 0); m[0] = 0xffffffffu; run("bits 0..31", m, true); }
    { std::vector<uint32_t> m(8, 0); m[0] = 0xffu; run("bits 0..7", m, true); }
    { std::vector<uint32_t> m(8, 0); m[0] = 0x1u; run("bit 0", m, true); }
    { std::vector<uint32_t> m(8, 0); m[0] = 0x100u; run("bit 8", m, true); }
    { std::vector<uint32_t> m(8, 0); m[7] = 0xffffffffu; run("bits 224..255", m, true); }
    { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 64; ++i) m[i >> 5] |= 1u << (i & 31); run("bits 0..63", m, true); }
    { std::vector<uint32_t> m(8, 0); for (int i = 64; i < 256; ++i) m[i >> 5] |= 1u << (i & 31); run("bits 64..255", m, true); }
    { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 48; ++i) m[i >> 5] |= 1u << (i & 31); run("bits 0..47", m, true); }
    { std::vector<uint32_t> m(8, 0); for (int i = 48; i < 256; ++i) m[i >> 5] |= 1u << (i & 31); run("bits 48..255", m, true); }
    return 0;
}
