// serving_loop_multi.cpp -- the C ABI's query-sharded replicas driven from C++ (no Python, no torch): gbnns_multi_create over
// the device list given on the command line (a one-GPU box: "0,0" or "0,0,0": the replicas share the card, each with its own
// handle, host thread and stream), a serving loop of host batches through gbnns_multi_search_ex -- every replica writes
// its block of ids / hops / dist_calc straight into the caller's arrays -- compared with ONE gbnns_search_ex call over the
// whole batch on a single handle; and, where the replicas sit on distinct devices, the device-block form
// (gbnns_multi_search_device: one RCCL all-gather of the ids per batch) against the same answers.
// Built with hipcc by tests/test_gpu_parity.py::test_serving_loop_multi_cpp.
//   serving_loop_multi <devices, e.g. 0,0> <n> <d> <d_low> <d_hidden> <n_q> <batches> <ef>
// The reference loop this shards: search/search_function.h:152 (#pragma omp parallel for over the queries), :346-387.
#include "../../include/gbnns.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <set>
#include <vector>

#define CK(x) do { if ((x) != hipSuccess) { std::fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); return 3; } } while (0)
#define GB(x) do { if ((x) != GBNNS_OK) { std::fprintf(stderr, "gbnns: %s / %s (%s:%d)\n", gbnns_last_error(), gbnns_multi_last_error(), __FILE__, __LINE__); return 4; } } while (0)

int main(int argc, char** argv) {
    if (argc != 9) return 2;
    std::vector<int32_t> devices;
    for (char* t = std::strtok(argv[1], ","); t; t = std::strtok(nullptr, ",")) devices.push_back(std::atoi(t));
    const uint64_t n = std::strtoull(argv[2], nullptr, 10);
    const uint32_t d = std::atoi(argv[3]), dl = std::atoi(argv[4]), dh = std::atoi(argv[5]);
    const uint64_t nq = std::strtoull(argv[6], nullptr, 10);
    const int batches = std::atoi(argv[7]), ef = std::atoi(argv[8]);
    std::mt19937 gen(11);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<float> db(n * d), db_low(n * dl), l1((size_t)dh * (d + 1)), l2((size_t)dh * (dh + 1)), l3((size_t)dl * (dh + 1));
    for (auto* v : {&db, &db_low, &l1, &l2, &l3}) for (float& x : *v) x = u(gen);
    std::vector<uint64_t> off(n + 1);
    std::vector<uint32_t> nbr;
    for (uint64_t i = 0; i < n; ++i) {  // a ring with a few chords: every node reachable, degree 6
        off[i] = nbr.size();
        for (uint64_t s : {1ull, 2ull, 7ull, 31ull, 257ull, 4099ull}) nbr.push_back((uint32_t)((i + s) % n));
    }
    off[n] = nbr.size();
    gbnns_index_desc desc = {};
    desc.struct_size = sizeof desc; desc.metric = GBNNS_METRIC_L2; desc.mem_kind = GBNNS_MEM_HOST; desc.n = n; desc.d = d;
    desc.d_low = dl; desc.d_hidden = dh; desc.db = db.data(); desc.db_low = db_low.data(); desc.graph_offsets = off.data();
    desc.graph_nbrs = nbr.data(); desc.net_l1 = l1.data(); desc.net_l2 = l2.data(); desc.net_l3 = l3.data();
    desc.device = devices[0];
    gbnns_index* one = nullptr;
    GB(gbnns_index_create(&desc, &one));
    gbnns_multi* multi = nullptr;
    GB(gbnns_multi_create(&desc, devices.data(), (int32_t)devices.size(), &multi));
    const int R = gbnns_multi_size(multi);
    if (R != (int)devices.size()) { std::fprintf(stderr, "replicas %d != %zu\n", R, devices.size()); return 5; }
    // the blocks tile the batch, in order, sizes within one of each other
    uint64_t expect_lo = 0;
    for (int r = 0; r < R; ++r) {
        uint64_t lo, hi;
        gbnns_shard_bounds(nq, R, r, &lo, &hi);
        if (lo != expect_lo || hi < lo || hi - lo > nq / R + 1) { std::fprintf(stderr, "bad block %d\n", r); return 5; }
        expect_lo = hi;
    }
    if (expect_lo != nq) return 5;

    std::vector<std::vector<float>> q(batches, std::vector<float>(nq * d));
    for (auto& v : q) for (float& x : v) x = u(gen);
    size_t mism = 0, mism_counts = 0;
    std::vector<std::vector<uint32_t>> want(batches, std::vector<uint32_t>(nq));
    for (int b = 0; b < batches; ++b) {
        std::vector<int32_t> hops1(nq), dc1(nq), hopsm(nq, -1), dcm(nq, -1);
        std::vector<uint32_t> idsm(nq, 0xFFFFFFFEu);
        gbnns_search_args a = {};
        a.struct_size = sizeof a; a.mode = GBNNS_MODE_NET; a.ef = ef; a.k = ef; a.mem_kind = GBNNS_MEM_HOST; a.n_q = nq;
        a.queries = q[b].data(); a.out_ids = want[b].data(); a.out_hops = hops1.data(); a.out_dist_calc = dc1.data();
        GB(gbnns_search_ex(one, &a));
        a.out_ids = idsm.data(); a.out_hops = hopsm.data(); a.out_dist_calc = dcm.data();
        GB(gbnns_multi_search_ex(multi, &a));
        for (uint64_t i = 0; i < nq; ++i) {
            mism += idsm[i] != want[b][i];
            mism_counts += hopsm[i] != hops1[i] || dcm[i] != dc1[i];
        }
    }
    // device blocks + one RCCL all-gather per batch: only where every replica has a device of its own
    long dev_mism = -1;
    const bool distinct = std::set<int32_t>(devices.begin(), devices.end()).size() == devices.size();
    if (distinct && R > 1) {
        dev_mism = 0;
        std::vector<float*> qd(R);
        std::vector<uint32_t*> od(R);
        std::vector<const float*> qc(R);
        for (int b = 0; b < batches; ++b) {
            for (int r = 0; r < R; ++r) {
                uint64_t lo, hi;
                gbnns_shard_bounds(nq, R, r, &lo, &hi);
                CK(hipSetDevice(gbnns_multi_device_of(multi, r)));
                if (b == 0) {
                    CK(hipMalloc((void**)&qd[r], (hi - lo ? hi - lo : 1) * d * 4));
                    CK(hipMalloc((void**)&od[r], nq * 4));
                }
                CK(hipMemcpy(qd[r], q[b].data() + lo * d, (hi - lo) * d * 4, hipMemcpyHostToDevice));
                qc[r] = qd[r];
            }
            gbnns_search_args a = {};
            a.struct_size = sizeof a; a.mode = GBNNS_MODE_NET; a.ef = ef; a.k = ef;
            GB(gbnns_multi_search_device(multi, &a, nq, qc.data(), nullptr, od.data()));
            GB(gbnns_multi_synchronize(multi));
            for (int r = 0; r < R; ++r) {  // every replica holds the whole answer vector
                std::vector<uint32_t> got(nq);
                CK(hipSetDevice(gbnns_multi_device_of(multi, r)));
                CK(hipMemcpy(got.data(), od[r], nq * 4, hipMemcpyDeviceToHost));
                for (uint64_t i = 0; i < nq; ++i) dev_mism += got[i] != want[b][i];
            }
        }
    }
    std::printf("replicas %d batches %d mismatches %zu count_mismatches %zu device_form_mismatches %ld\n", R, batches, mism, mism_counts, dev_mism);
    GB(gbnns_multi_destroy(multi));
    GB(gbnns_index_destroy(one));
    return mism == 0 && mism_counts == 0 && dev_mism <= 0 ? 0 : 1;
}
