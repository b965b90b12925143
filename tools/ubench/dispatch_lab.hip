// What does re-dispatch cost a launch of many one-wavefront workgroups?  (Round 6: should the walk kernels be persistent -- a resident
// wavefront per slot that pulls queries from a counter -- instead of one workgroup per query?)
//   hipcc --offload-arch=gfx950 -O2 -o dispatch_lab dispatch_lab.hip && ./dispatch_lab
// Kernel A: `n` workgroups of one wavefront with the walk kernel's footprint (64 VGPRs via v63, `lds` bytes of LDS), each busy for `us`
// microseconds (s_memrealtime spin), so that a launch of r full rounds of the 8 192 slots should take r * us.  Kernel B: 8 192 persistent
// wavefronts that take items from an atomic counter and stay busy `us` per item.  The difference is what the dispatcher costs per round.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ void busy(unsigned ticks) {   // 100 MHz ticks
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
}

__global__ __launch_bounds__(64) void per_item(unsigned ticks, unsigned* sink) {
    extern __shared__ unsigned char smem[];
    asm volatile("v_mov_b32 v63, 0" ::: "v63");
    smem[threadIdx.x] = 1;
    busy(ticks);
    if (threadIdx.x == 0 && smem[1] == 7) sink[0] = 1;
}

__global__ __launch_bounds__(64) void persistent(unsigned ticks, unsigned n, unsigned* ctr, unsigned* sink) {
    extern __shared__ unsigned char smem[];
    asm volatile("v_mov_b32 v63, 0" ::: "v63");
    while (true) {
        unsigned item = 0;
        if (threadIdx.x == 0) item = atomicAdd(ctr, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n) break;
        smem[threadIdx.x] = 1;
        busy(ticks);
    }
    if (threadIdx.x == 0 && smem[1] == 7) sink[0] = 1;
}

int main() {
    unsigned *ctr, *sink;
    CHECK(hipMalloc(&ctr, 4)); CHECK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t lds = 5120;
    const unsigned slots = 8192;
    for (unsigned us : {20u, 50u, 150u}) {
        for (unsigned n : {8192u, 10000u, 16384u, 81920u}) {
            float best_a = 1e9f, best_b = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(per_item, dim3(n), dim3(64), lds, 0, us * 100u, sink);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                best_a = ms < best_a ? ms : best_a;
                CHECK(hipMemset(ctr, 0, 4));
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(persistent, dim3(slots), dim3(64), lds, 0, us * 100u, n, ctr, sink);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best_b = ms < best_b ? ms : best_b;
            }
            const unsigned rounds = (n + slots - 1) / slots;
            std::printf("busy %3u us, %6u items (%2u rounds of %u slots): ideal %7.1f us; a workgroup per item %7.1f us; persistent %7.1f us\n", us, n, rounds, slots,
                        (double)rounds * us, best_a * 1000.0, best_b * 1000.0);
        }
    }
    return 0;
}
