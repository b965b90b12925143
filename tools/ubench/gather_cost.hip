// Micro-benchmark (diagnostic): throughput of random 128-byte row gathers from a table, by how the
// lanes of a wavefront share a row.  LPR lanes per row (1 = one lane streams its whole row with 8
// 16-B loads, 2 = pair, 4 = quad, 8 = octet: one 16-B load per lane), ROWS rows per iteration.
// One wavefront per workgroup, 20 wavefronts per CU resident by default (LDS-limited like the walk kernel; argv[2]).
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_cost gather_cost.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int LPR, int ROWS>
__global__ __launch_bounds__(64) void gather(const float4* __restrict__ tab, uint32_t nrows, int iters, float* out) {
    extern __shared__ unsigned char smem[];  // only to bound occupancy
    const int lane = threadIdx.x;
    constexpr int LOADS = 8 / LPR;                 // 16-B loads per lane per row
    constexpr int ACTIVE = ROWS * LPR > 64 ? 64 : ROWS * LPR;
    constexpr int PASSES = (ROWS * LPR + 63) / 64;
    uint32_t h = (blockIdx.x * 64u + lane / LPR) * 2654435761u + 12345u;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            h = h * 1664525u + 1013904223u;
            const uint32_t row = __umulhi(h, nrows);
            if (lane < ACTIVE) {
                const float4* r = tab + (size_t)row * 8 + (lane % LPR) * LOADS;
                float4 v[LOADS];
#pragma unroll
                for (int t = 0; t < LOADS; ++t) v[t] = r[t];
#pragma unroll
                for (int t = 0; t < LOADS; ++t) { acc.x += v[t].x; acc.y += v[t].y; acc.z += v[t].z; acc.w += v[t].w; }
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x + smem[0];
}

static size_t g_lds = 8 * 1024;  // 20 wavefronts per CU (argv[2]: LDS bytes per wavefront; 5120 -> 32 per CU, 6400 -> 25)

template <int LPR, int ROWS>
void run(const float4* tab, uint32_t nrows, float* out, const char* name) {
    const int iters = 200, grid = 256 * 40;
    const size_t lds = g_lds;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((gather<LPR, ROWS>), dim3(grid), dim3(64), lds, 0, tab, nrows, 20, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((gather<LPR, ROWS>), dim3(grid), dim3(64), lds, 0, tab, nrows, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double rows = (double)grid * iters * ROWS;
    printf("%-28s lanes/row %d rows/iter %2d : %8.3f ms  %7.2f Grows/s  %6.2f TB/s\n", name, LPR, ROWS, ms, rows / ms * 1e-6,
           rows * 128 / ms * 1e-9);
}

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 128;
    if (argc > 2) g_lds = (size_t)atoi(argv[2]);
    const uint32_t nrows = (uint32_t)(mb * 1024 * 1024 / 128);
    float4* tab; float* out;
    hipMalloc(&tab, (size_t)nrows * 128);
    hipMalloc(&out, 64);
    hipMemset(tab, 0, (size_t)nrows * 128);
    printf("table %zu MB (%u rows of 128 B), %zu B of LDS per wavefront = %zu wavefronts per CU\n", mb, nrows, g_lds,
           (size_t)(163840 / ((g_lds + 1279) / 1280 * 1280)) > 32 ? (size_t)32 : (size_t)(163840 / ((g_lds + 1279) / 1280 * 1280)));
    run<1, 16>(tab, nrows, out, "lane per row");
    run<2, 16>(tab, nrows, out, "pair per row");
    run<4, 16>(tab, nrows, out, "quad per row");
    run<8, 16>(tab, nrows, out, "octet per row");
    run<1, 32>(tab, nrows, out, "lane per row");
    run<2, 32>(tab, nrows, out, "pair per row");
    run<4, 32>(tab, nrows, out, "quad per row");
    run<8, 32>(tab, nrows, out, "octet per row");
    run<1, 64>(tab, nrows, out, "lane per row");
    return 0;
}
