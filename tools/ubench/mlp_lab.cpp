// mlp_lab (diagnostic): the one-launch projection (csrc/mlp_net.hip) against the three per-layer launches (csrc/mlp.hip) on
// one random net -- outputs compared bit for bit, both timed with HIP events -- and a few issue-rate loops for the packed
// f32 instructions the projection is made of.
// Build (from the repo root, after `make -C gbnns_dim_red_amd/csrc`):
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -x hip -Igbnns_dim_red_amd/csrc tools/ubench/mlp_lab.cpp gbnns_dim_red_amd/csrc/build/mlp.o \
//         gbnns_dim_red_amd/csrc/build/mlp_net.o -o tools/ubench/mlp_lab
// Run on the GPU box: tools/ubench/mlp_lab [nq d d_hidden d_low [force_a]]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <random>
#include <vector>

#include "kernels.h"

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(2);                                                               \
        }                                                                          \
    } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

// issue-rate loops: MODE 0 = independent v_pk_mul_f32, 1 = independent v_mul_f32, 2 = pk_mul -> pk_add pairs (8 chains),
// 3 = v_mul_f32 -> v_add_f32 pairs (16 chains), 4 = pk_mul x4 then pk_add x4 (software pipelined)
template <int MODE>
__global__ void rate_kernel(float* out, int iters, unsigned long long* cyc) {
    f2 a[8], b = {1.0000001f, 0.9999999f}, c = {threadIdx.x * 1e-9f, 1e-9f};
    for (int i = 0; i < 8; ++i) a[i] = f2{1.f + i, 2.f + i};
    f2 t[8];
    for (int i = 0; i < 8; ++i) t[i] = f2{0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
        } else if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    asm volatile("v_pk_mul_f32 %1, %2, %3\n\ts_nop 0\n\tv_pk_add_f32 %0, %0, %1" : "+v"(a[i]), "+v"(t[0]) : "v"(b), "v"(c));
        } else if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    asm volatile("v_mul_f32 %1, %2, %3\n\tv_add_f32 %0, %0, %1" : "+v"(a[i].x), "+v"(t[0].x) : "v"(b.x), "v"(c.x));
        } else if (MODE == 5 || MODE == 6) {
            // both 64-bit sources in the same pair of register banks (v[..] index mod 4 equal) / in different ones
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (MODE == 5)
                    asm volatile("v_pk_mul_f32 v[40:41], v[4:5], v[8:9]\n\tv_pk_mul_f32 v[42:43], v[12:13], v[8:9]\n\t"
                                 "v_pk_mul_f32 v[44:45], v[16:17], v[8:9]\n\tv_pk_mul_f32 v[46:47], v[20:21], v[8:9]\n\t"
                                 "v_pk_mul_f32 v[48:49], v[6:7], v[10:11]\n\tv_pk_mul_f32 v[50:51], v[14:15], v[10:11]\n\t"
                                 "v_pk_mul_f32 v[52:53], v[18:19], v[10:11]\n\tv_pk_mul_f32 v[54:55], v[22:23], v[10:11]" ::
                                     : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
                else
                    asm volatile("v_pk_mul_f32 v[40:41], v[4:5], v[10:11]\n\tv_pk_mul_f32 v[42:43], v[12:13], v[10:11]\n\t"
                                 "v_pk_mul_f32 v[44:45], v[16:17], v[10:11]\n\tv_pk_mul_f32 v[46:47], v[20:21], v[10:11]\n\t"
                                 "v_pk_mul_f32 v[48:49], v[6:7], v[8:9]\n\tv_pk_mul_f32 v[50:51], v[14:15], v[8:9]\n\t"
                                 "v_pk_mul_f32 v[52:53], v[18:19], v[8:9]\n\tv_pk_mul_f32 v[54:55], v[22:23], v[8:9]" ::
                                     : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
            }
        } else if (MODE == 7) {
#pragma unroll
            for (int r = 0; r < 40; ++r) {
                asm volatile(
                    "v_pk_mul_f32 %4, %8, %9\n\tv_pk_mul_f32 %5, %8, %9\n\tv_pk_mul_f32 %6, %8, %9\n\tv_pk_mul_f32 %7, %8, %9\n\t"
                    "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %7"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3])
                    : "v"(b), "v"(c));
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                asm volatile(
                    "v_pk_mul_f32 %4, %8, %9\n\tv_pk_mul_f32 %5, %8, %9\n\tv_pk_mul_f32 %6, %8, %9\n\tv_pk_mul_f32 %7, %8, %9\n\t"
                    "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %7"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3])
                    : "v"(b), "v"(c));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y + t[i].x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// 32 packed instructions (4 x mul, 4 x add, ...) with NR ds_read_b128 among them (conflict-free: lane * 16 bytes)
template <int NR>
__global__ void mix_kernel(float* out, int iters, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 4];
    f2 a[4], t[4], b = {1.0000001f, 0.9999999f}, c = {threadIdx.x * 1e-9f, 1e-9f};
    for (int i = 0; i < 4; ++i) { a[i] = f2{1.f + i, 2.f + i}; t[i] = f2{0.f, 0.f}; }
    for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
    float4 r0 = {0, 0, 0, 0}, r1 = r0, r2 = r0, r3 = r0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r < NR) {
                if (r == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(addr));
                if (r == 1) asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(r1) : "v"(addr));
                if (r == 2) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(r2) : "v"(addr));
                if (r == 3) asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(r3) : "v"(addr));
            }
            asm volatile(
                "v_pk_mul_f32 %4, %8, %9\n\tv_pk_mul_f32 %5, %8, %9\n\tv_pk_mul_f32 %6, %8, %9\n\tv_pk_mul_f32 %7, %8, %9\n\t"
                "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %7"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3])
                : "v"(b), "v"(c));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = r0.x + r1.y + r2.z + r3.w;
    for (int i = 0; i < 4; ++i) s += a[i].x + a[i].y + t[i].x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NR>
static void mix(float* d_out, unsigned long long* d_cyc) {
    for (int wps : {2, 4}) {
        const int iters = 20000, blocks = 256, threads = 256 * wps;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(mix_kernel<NR>, dim3(blocks), dim3(threads), 0, 0, d_out, 100, d_cyc);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(mix_kernel<NR>, dim3(blocks), dim3(threads), 0, 0, d_out, iters, d_cyc);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double per_simd = (double)iters * 32 * wps;
        printf("  32 packed instructions + %d ds_read_b128, %d wave(s)/SIMD: %.3f ns per packed instruction and SIMD = %.1f G lane-ops/s per SIMD\n",
               NR, wps, ms * 1e6 / per_simd, per_simd * 128 / (ms * 1e6));
    }
}

template <int MODE>
static void rate(const char* name, int instr_per_iter, float* d_out, unsigned long long* d_cyc) {
    // wps wavefronts per SIMD: blocks of 256 wps threads, one per CU; 8 = two blocks of 1 024 threads per CU
    for (int wps : {1, 2, 4, 8}) {
        const int iters = MODE == 7 ? 2000 : 20000;
        const int blocks = wps == 8 ? 512 : 256, threads = wps == 8 ? 1024 : 256 * wps;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, d_out, 100, d_cyc);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, d_out, iters, d_cyc);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> c(blocks);
        CK(hipMemcpy(c.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost));
        double m = 0;
        for (auto v : c) m += (double)v;
        m /= blocks;
        const double per_simd = (double)iters * instr_per_iter * wps;  // instructions one SIMD executed
        printf("  %-38s %d wave(s)/SIMD: %.2f counter ticks per instruction and SIMD (one wave: %.2f); wall %.3f ms = %.3f ns per "
               "instruction and SIMD = %.1f G lane-ops/s per SIMD\n",
               name, wps, m / per_simd, m / ((double)iters * instr_per_iter), ms, ms * 1e6 / per_simd,
               per_simd * 64 * (MODE == 1 || MODE == 3 ? 1 : 2) / (ms * 1e6));
    }
}

int main(int argc, char** argv) {
    const uint32_t nq = argc > 1 ? atoi(argv[1]) : 10000, d = argc > 2 ? atoi(argv[2]) : 128, dh = argc > 3 ? atoi(argv[3]) : 256,
                   dl = argc > 4 ? atoi(argv[4]) : 32;
    const int force_a = argc > 5 ? atoi(argv[5]) : 0, variant = argc > 6 ? atoi(argv[6]) : 0;
    const int reps = 200;
    auto pad8 = [](uint32_t v) { return (v + 15u) & ~15u; };  // rows padded to 16 floats (the one-launch kernel reads whole 16-float blocks)
    const uint32_t din[3] = {d, dh, dh}, dout[3] = {dh, dh, dl};
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> x((size_t)nq * d);
    for (auto& v : x) v = nd(rng);
    std::vector<float> w[3], bias[3];
    uint32_t ws[3];
    for (int l = 0; l < 3; ++l) {
        ws[l] = pad8(din[l]);
        w[l].assign((size_t)dout[l] * ws[l], 0.f);
        bias[l].resize(dout[l]);
        const float sc = 1.f / std::sqrt((float)din[l]);
        for (uint32_t o = 0; o < dout[l]; ++o) {
            for (uint32_t k = 0; k < din[l]; ++k) w[l][(size_t)o * ws[l] + k] = nd(rng) * sc;
            bias[l][o] = nd(rng) * 0.1f;
        }
    }
    float *dx, *dw[3], *db[3], *h1, *h2, *o_ref, *o_net;
    const uint32_t ostride = (dl + 3u) & ~3u;
    CK(hipMalloc(&dx, x.size() * 4));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    for (int l = 0; l < 3; ++l) {
        CK(hipMalloc(&dw[l], w[l].size() * 4));
        CK(hipMemcpy(dw[l], w[l].data(), w[l].size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&db[l], bias[l].size() * 4));
        CK(hipMemcpy(db[l], bias[l].data(), bias[l].size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&h1, (size_t)nq * dh * 4));
    CK(hipMalloc(&h2, (size_t)nq * dh * 4));
    CK(hipMalloc(&o_ref, (size_t)nq * ostride * 4));
    CK(hipMalloc(&o_net, (size_t)nq * ostride * 4));
    CK(hipMemset(o_ref, 0xFF, (size_t)nq * ostride * 4));
    CK(hipMemset(o_net, 0xEE, (size_t)nq * ostride * 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    auto per_layer = [&](int small) {
        gbnns::LayerParams p{};
        p.small_footprint = small;
        p.x = dx; p.xstride = d; p.w = dw[0]; p.wstride = ws[0]; p.bias = db[0]; p.out = h1; p.ostride = dh; p.nq = nq; p.din = d;
        p.dout = dh; p.relu = 1;
        CK(gbnns::launch_mlp_layer(p, s));
        p.x = h1; p.xstride = dh; p.w = dw[1]; p.wstride = ws[1]; p.bias = db[1]; p.out = h2; p.din = dh;
        CK(gbnns::launch_mlp_layer(p, s));
        p.x = h2; p.w = dw[2]; p.wstride = ws[2]; p.bias = db[2]; p.out = o_ref; p.ostride = ostride; p.dout = dl; p.relu = 0;
        p.normalize = 1;
        CK(gbnns::launch_mlp_layer(p, s));
    };
    gbnns::NetLaunch n{};
    n.x = dx; n.xstride = d; n.nq = nq; n.out = o_net; n.ostride = ostride; n.force_a = force_a; (void)variant;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    n.cus = prop.multiProcessorCount;
    for (int l = 0; l < 3; ++l) { n.w[l] = dw[l]; n.wstride[l] = ws[l]; n.bias[l] = db[l]; n.din[l] = din[l]; n.dout[l] = dout[l]; }
    printf("net %u x %u -> %u -> %u -> %u on %s (%d CUs); one-launch kernel serves it: %d\n", nq, d, dh, dh, dl, prop.name, n.cus,
           (int)gbnns::mlp_net_serves(n));

    auto timeit = [&](const char* name, auto&& fn) {
        for (int i = 0; i < 20; ++i) fn();
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) fn();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %-40s %.2f us per projection\n", name, ms * 1000.f / reps);
    };
    if (getenv("MLP_LAB_SLAB")) {  // layer by layer: mlp.hip's kernels against the slab kernel; bit comparison of the slab net
        const int fa = getenv("MLP_LAB_SLAB_A") ? atoi(getenv("MLP_LAB_SLAB_A")) : 0;
        auto layer_params = [&](int l, float* out_last) {
            gbnns::LayerParams p{};
            p.nq = nq; p.w = dw[l]; p.wstride = ws[l]; p.bias = db[l]; p.din = din[l]; p.dout = dout[l];
            p.x = l == 0 ? dx : (l == 1 ? h1 : h2); p.xstride = din[l];
            p.out = l == 0 ? h1 : (l == 1 ? h2 : out_last); p.ostride = l == 2 ? ostride : dh;
            p.relu = l < 2; p.normalize = l == 2;
            return p;
        };
        per_layer(0);
        for (int l = 0; l < 3; ++l) {
            char name[64];
            snprintf(name, sizeof name, "layer %d, mlp.hip", l + 1);
            timeit(name, [&] { CK(gbnns::launch_mlp_layer(layer_params(l, o_ref), s)); });
            snprintf(name, sizeof name, "layer %d, slab kernel", l + 1);
            timeit(name, [&] { CK(gbnns::launch_mlp_slab(layer_params(l, o_net), n.cus, s, fa)); });
        }
        per_layer(0);
        for (int l = 0; l < 3; ++l) CK(gbnns::launch_mlp_slab(layer_params(l, o_net), n.cus, s, fa));
        CK(hipStreamSynchronize(s));
        std::vector<uint32_t> a((size_t)nq * ostride), b((size_t)nq * ostride);
        CK(hipMemcpy(a.data(), o_ref, a.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), o_net, b.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t q = 0; q < nq; ++q)
            for (uint32_t c = 0; c < dl; ++c) bad += a[q * ostride + c] != b[q * ostride + c];
        printf("  slab net: outputs that differ in a bit: %zu of %zu\n", bad, (size_t)nq * dl);
    }
    timeit("three launches (mlp.hip)", [&] { per_layer(0); });
    timeit("three launches, small-footprint hidden", [&] { per_layer(1); });
    if (gbnns::mlp_net_serves(n)) {
        timeit("one launch (mlp_net.hip)", [&] { CK(gbnns::launch_mlp_net(n, s)); });
        per_layer(0);
        CK(hipStreamSynchronize(s));
        std::vector<uint32_t> a((size_t)nq * ostride), b((size_t)nq * ostride);
        CK(hipMemcpy(a.data(), o_ref, a.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), o_net, b.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0, first = (size_t)-1;
        for (size_t i = 0; i < a.size(); ++i)
            if (a[i] != b[i]) { if (!bad) first = i; ++bad; }
        printf("  outputs that differ in a bit: %zu of %zu", bad, a.size());
        if (bad) {
            float fa, fb;
            memcpy(&fa, &a[first], 4); memcpy(&fb, &b[first], 4);
            printf(" (first at query %zu column %zu: %.9g against %.9g)", first / ostride, first % ostride, fa, fb);
        }
        printf("\n");
    }
    if (getenv("MLP_LAB_STAMPS") && gbnns::mlp_net_serves(n)) {   // needs a library built with EXTRA_DEFS=-DGBNNS_NET_STAMPS
        const int nb = 1024;
        unsigned long long* d_st;
        CK(hipMalloc(&d_st, nb * 40 * 8));
        CK(hipMemset(d_st, 0, nb * 40 * 8));
        n.stamps = d_st;
        CK(gbnns::launch_mlp_net(n, s));
        CK(hipStreamSynchronize(s));
        std::vector<unsigned long long> st(nb * 40);
        CK(hipMemcpy(st.data(), d_st, nb * 40 * 8, hipMemcpyDeviceToHost));
        double sum[8] = {0}; int cnt = 0; unsigned long long t0min = ~0ull, t6max = 0;
        for (int b = 0; b < nb; ++b) {
            if (!st[b * 8 + 6]) continue;
            ++cnt;
            t0min = std::min(t0min, st[b * 8]); t6max = std::max(t6max, st[b * 8 + 6]);
            for (int i = 1; i <= 6; ++i) sum[i] += (double)(st[b * 8 + i] - st[b * 8 + i - 1]);
        }
        const char* names[] = {"", "stage x", "layer 1", "barrier", "layer 2 (+barrier)", "layer 3 (+barrier)", "normalise + store"};
        printf("  phases of wave 0, mean over %d blocks (10 ns ticks):", cnt);
        for (int i = 1; i <= 6; ++i) printf(" %s %.2f us;", names[i], sum[i] / cnt / 100.0);
        printf(" first start -> last end %.2f us\n", (double)(t6max - t0min) / 100.0);
        const int nwv = 8;
        printf("  layer 2 by wavefront (k loops / folds + outputs, us, sums over the layer's passes):");
        for (int wv = 0; wv < nwv; ++wv) {
            double ml = 0, ep = 0;
            for (int b = 0; b < cnt; ++b) { ml += (double)st[8 * nb + 2 * (b * nwv + wv)]; ep += (double)st[8 * nb + 2 * (b * nwv + wv) + 1]; }
            printf(" %d: %.2f / %.2f;", wv, ml / cnt / 100.0, ep / cnt / 100.0);
        }
        printf("\n");
        n.stamps = nullptr;
    }
    if (!getenv("MLP_LAB_NO_RATES")) {
        float* d_out;
        unsigned long long* d_cyc;
        CK(hipMalloc(&d_out, 512 * 1024 * 4));
        CK(hipMalloc(&d_cyc, 512 * 8));
        printf("issue rates (256 blocks, one per CU):\n");
        rate<0>("v_pk_mul_f32, independent", 32, d_out, d_cyc);
        rate<1>("v_mul_f32, independent", 32, d_out, d_cyc);
        rate<2>("v_pk_mul_f32 ; s_nop 0 ; v_pk_add_f32", 32, d_out, d_cyc);
        rate<3>("v_mul_f32 ; v_add_f32", 32, d_out, d_cyc);
        rate<4>("4 x v_pk_mul_f32 ; 4 x v_pk_add_f32", 32, d_out, d_cyc);
        rate<7>("40 x (4 x pk_mul ; 4 x pk_add): 2.5 KB loop body", 320, d_out, d_cyc);
        mix<0>(d_out, d_cyc); mix<1>(d_out, d_cyc); mix<2>(d_out, d_cyc); mix<3>(d_out, d_cyc); mix<4>(d_out, d_cyc);
        rate<5>("v_pk_mul_f32, sources in the same banks", 32, d_out, d_cyc);
        rate<6>("v_pk_mul_f32, sources in different banks", 32, d_out, d_cyc);
    }
    return 0;
}
