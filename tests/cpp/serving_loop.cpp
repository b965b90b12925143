// serving_loop.cpp -- the C ABI's "batches in flight" contract driven from C++ (no Python, no torch): a serving loop that
// rotates three sets of device buffers through gbnns_search_ex(GBNNS_FLAG_DEFER_JOIN, defer_depth = 3) on one HIP
// stream, consumes batch i-2 on that stream right after call i (a device-to-device copy enqueued on the same stream: it
// must see complete answers WITHOUT any host synchronisation), and compares every batch with the answers of plain calls.
// Built with hipcc by tests/test_gpu_parity.py::test_serving_loop_cpp.
//   serving_loop <n> <d> <d_low> <d_hidden> <n_q> <batches> <ef>      (random data: only self-consistency is checked here;
//                                                                      the values themselves are checked by the Python tests)
#include "../../include/gbnns.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { if ((x) != hipSuccess) { std::fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); return 3; } } while (0)
#define GB(x) do { if ((x) != GBNNS_OK) { std::fprintf(stderr, "gbnns: %s (%s:%d)\n", gbnns_last_error(), __FILE__, __LINE__); return 4; } } while (0)

int main(int argc, char** argv) {
    if (argc != 8) return 2;
    const uint64_t n = std::strtoull(argv[1], nullptr, 10);
    const uint32_t d = std::atoi(argv[2]), dl = std::atoi(argv[3]), dh = std::atoi(argv[4]);
    const uint64_t nq = std::strtoull(argv[5], nullptr, 10);
    const int batches = std::atoi(argv[6]), ef = std::atoi(argv[7]);
    std::mt19937 gen(7);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<float> db(n * d), db_low(n * dl), l1((size_t)dh * (d + 1)), l2((size_t)dh * (dh + 1)), l3((size_t)dl * (dh + 1));
    for (auto* v : {&db, &db_low, &l1, &l2, &l3}) for (float& x : *v) x = u(gen);
    // a ring with a few chords: every node reachable, degree 6
    std::vector<uint64_t> off(n + 1);
    std::vector<uint32_t> nbr;
    for (uint64_t i = 0; i < n; ++i) {
        off[i] = nbr.size();
        for (uint64_t s : {1ull, 2ull, 7ull, 31ull, 257ull, 4099ull}) nbr.push_back((uint32_t)((i + s) % n));
    }
    off[n] = nbr.size();
    gbnns_index_desc desc = {};
    desc.struct_size = sizeof desc; desc.metric = GBNNS_METRIC_L2; desc.mem_kind = GBNNS_MEM_HOST; desc.n = n; desc.d = d;
    desc.d_low = dl; desc.d_hidden = dh; desc.db = db.data(); desc.db_low = db_low.data(); desc.graph_offsets = off.data();
    desc.graph_nbrs = nbr.data(); desc.net_l1 = l1.data(); desc.net_l2 = l2.data(); desc.net_l3 = l3.data();
    gbnns_index* ix = nullptr;
    GB(gbnns_index_create(&desc, &ix));

    hipStream_t s;
    CK(hipStreamCreate(&s));
    constexpr int kDepth = 3;
    float* q_dev[kDepth];
    uint32_t *ids_dev[kDepth], *seen_dev;
    for (int b = 0; b < kDepth; ++b) {
        CK(hipMalloc((void**)&q_dev[b], nq * d * 4));
        CK(hipMalloc((void**)&ids_dev[b], nq * 4));
    }
    CK(hipMalloc((void**)&seen_dev, (size_t)batches * nq * 4));
    std::vector<std::vector<float>> q(batches, std::vector<float>(nq * d));
    for (auto& v : q) for (float& x : v) x = u(gen);

    // reference answers: plain calls, host buffers
    std::vector<std::vector<uint32_t>> want(batches, std::vector<uint32_t>(nq));
    for (int i = 0; i < batches; ++i) GB(gbnns_search_batch(ix, q[i].data(), nq, ef, nullptr, want[i].data(), nullptr, nullptr, nullptr));

    gbnns_search_args a = {};
    a.struct_size = sizeof a; a.mode = GBNNS_MODE_NET; a.ef = ef; a.k = ef; a.mem_kind = GBNNS_MEM_DEVICE; a.n_q = nq;
    a.stream = s; a.flags = GBNNS_FLAG_DEFER_JOIN; a.defer_depth = kDepth;
    for (int i = 0; i < batches; ++i) {
        const int b = i % kDepth;
        // the set's previous user (batch i-3) was joined by call i-1 and consumed right after it: stream order protects the reuse
        CK(hipMemcpyAsync(q_dev[b], q[i].data(), nq * d * 4, hipMemcpyHostToDevice, s));
        a.queries = q_dev[b];
        a.out_ids = ids_dev[b];
        GB(gbnns_search_ex(ix, &a));  // batch i released; s now waits for batch i-2
        if (i >= kDepth - 1) {
            const int j = i - (kDepth - 1);
            CK(hipMemcpyAsync(seen_dev + (size_t)j * nq, ids_dev[j % kDepth], nq * 4, hipMemcpyDeviceToDevice, s));
        }
    }
    GB(gbnns_index_join(ix));  // s waits for what is still in flight
    for (int j = batches - (kDepth - 1) < 0 ? 0 : batches - (kDepth - 1); j < batches; ++j)
        CK(hipMemcpyAsync(seen_dev + (size_t)j * nq, ids_dev[j % kDepth], nq * 4, hipMemcpyDeviceToDevice, s));
    std::vector<uint32_t> seen((size_t)batches * nq);
    CK(hipMemcpyAsync(seen.data(), seen_dev, seen.size() * 4, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    size_t bad = 0;
    for (int i = 0; i < batches; ++i)
        for (uint64_t k = 0; k < nq; ++k) bad += seen[(size_t)i * nq + k] != want[i][k];
    // ---- host batches in flight: page-locked buffers, GBNNS_MEM_HOST + GBNNS_FLAG_DEFER_JOIN, gbnns_index_wait ----------
    // Three sets of page-locked buffers; after call i the host waits until all but the two newest batches have
    // finished, reads batch i-2's ids (and hop counts) from host memory and refills that set with batch i+1.
    float* q_pin[kDepth];
    uint32_t* ids_pin[kDepth];
    int32_t* hops_pin[kDepth];
    for (int b = 0; b < kDepth; ++b) {
        CK(hipHostMalloc((void**)&q_pin[b], nq * d * 4, hipHostMallocDefault));
        CK(hipHostMalloc((void**)&ids_pin[b], nq * 4, hipHostMallocDefault));
        CK(hipHostMalloc((void**)&hops_pin[b], nq * 4, hipHostMallocDefault));
    }
    std::vector<std::vector<int32_t>> want_hops(batches, std::vector<int32_t>(nq));
    for (int i = 0; i < batches; ++i)
        GB(gbnns_search_batch(ix, q[i].data(), nq, ef, nullptr, want[i].data(), want_hops[i].data(), nullptr, nullptr));
    gbnns_search_args h = {};
    h.struct_size = sizeof h; h.mode = GBNNS_MODE_NET; h.ef = ef; h.k = ef; h.mem_kind = GBNNS_MEM_HOST; h.n_q = nq;
    h.stream = s; h.flags = GBNNS_FLAG_DEFER_JOIN; h.defer_depth = kDepth;
    size_t bad_host = 0;
    auto check = [&](int j) {
        for (uint64_t k = 0; k < nq; ++k)
            bad_host += ids_pin[j % kDepth][k] != want[j][k] || hops_pin[j % kDepth][k] != want_hops[j][k];
    };
    for (int i = 0; i < batches; ++i) {
        const int b = i % kDepth;
        std::copy(q[i].begin(), q[i].end(), q_pin[b]);  // (set b's previous batch, i-3, was waited for after call i-1)
        for (uint64_t k = 0; k < nq; ++k) ids_pin[b][k] = 0xffffffffu;
        h.queries = q_pin[b]; h.out_ids = ids_pin[b]; h.out_hops = hops_pin[b];
        GB(gbnns_search_ex(ix, &h));
        GB(gbnns_index_wait(ix, kDepth - 1));
        if (i >= kDepth - 1) check(i - (kDepth - 1));
    }
    GB(gbnns_index_wait(ix, 0));
    for (int j = batches - (kDepth - 1) < 0 ? 0 : batches - (kDepth - 1); j < batches; ++j) check(j);
    // a pageable buffer among them: the flag is ignored, the call is the plain synchronous one
    std::vector<uint32_t> ids_pageable(nq, 0xffffffffu);
    h.queries = q[0].data(); h.out_ids = ids_pageable.data(); h.out_hops = nullptr;
    GB(gbnns_search_ex(ix, &h));
    size_t bad_pageable = 0;
    for (uint64_t k = 0; k < nq; ++k) bad_pageable += ids_pageable[k] != want[0][k];
    GB(gbnns_index_join(ix));
    CK(hipStreamSynchronize(s));
    GB(gbnns_index_destroy(ix));
    std::printf("serving_loop batches %d depth %d mismatches %zu host_mismatches %zu pageable_mismatches %zu\n", batches, kDepth, bad, bad_host,
                bad_pageable);
    return (bad || bad_host || bad_pageable) ? 1 : 0;
}
