// graph_build.cpp -- host-side construction of the search graph from a kNN graph:
// the "GD" pruning of support_func.h:521-575 (hnswlikeGD, need_const_degree = false) followed by
// the reverse-edge pass of support_func.h:402-445 (addReverseEdgesForGD), which is what
// prepare_graph.cpp:70 runs (M = 30, reverse = true) to produce the graph final_test.cpp walks.
// The reference builder is host C++ with OpenMP; so is this one (flat CSR in, flat CSR out).
//
// Distances use the scalar 4-lane order of L2Metric::Dist / 8-lane order of Angular::Dist; this
// file is compiled with -ffp-contract=off so comparisons see the same floats as the reference.

#include "../../include/gbnns.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

float dist_l2(const float* a, const float* b, uint32_t d) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const uint32_t steps = d / 4;
    for (uint32_t t = 0; t < steps; ++t)
        for (int j = 0; j < 4; ++j) {
            const float e = a[4 * t + j] - b[4 * t + j];
            s[j] = s[j] + e * e;
        }
    return ((s[0] + s[1]) + s[2]) + s[3];
}

float dist_negdot(const float* a, const float* b, uint32_t d) {
    float c[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint32_t k = 0;
    for (; k + 8 <= d; k += 8)
        for (int l = 0; l < 8; ++l) c[l] = c[l] + a[k + l] * b[k + l];
    float m[4];
    for (int j = 0; j < 4; ++j) m[j] = c[j + 4] + c[j];
    if (d - k >= 4) {
        for (int j = 0; j < 4; ++j) m[j] = m[j] + a[k + j] * b[k + j];
        k += 4;
    }
    if (d - k > 0)
        for (uint32_t j = 0; j < 4; ++j) {
            const float av = (k + j < d) ? a[k + j] : 0.f;
            const float bv = (k + j < d) ? b[k + j] : 0.f;
            m[j] = m[j] + av * bv;
        }
    return -((m[0] + m[1]) + (m[2] + m[3]));
}

struct Scored {
    uint32_t id;
    float dist;
};

}  // namespace

// One node of hnswlikeGD (support_func.h:528-563): score the kNN candidates, drop (near-)duplicates of i,
// std::sort by distance (the reference's own call on the same records, so equal distances end up in the same
// order), greedy pruning until M are kept, then the M/2 nearest are always linked.  g has room for 2M ids.
static bool prune_node(uint64_t i, const uint64_t* knn_offsets, const uint32_t* knn_nbrs, const float* ds, uint64_t n,
                       uint32_t d, int M, float (*dist)(const float*, const float*, uint32_t), std::vector<Scored>& sc,
                       uint32_t* g, uint32_t& deg_out) {
    const float eps = 1e-10f;  // support_func.h:41-43
    const float* pi = ds + (size_t)i * d;
    bool ok = true;
    sc.clear();
    for (uint64_t j = knn_offsets[i]; j < knn_offsets[i + 1]; ++j) {
        const uint32_t c = knn_nbrs[j];
        if (c >= n) {
            ok = false;
            continue;
        }
        const float dc = dist(pi, ds + (size_t)c * d, d);
        if (dc > eps) sc.push_back(Scored{c, dc});  // :535 drops (near-)duplicates of i
    }
    // :540 -- std::sort on distance only; same call on the same records as the reference
    std::sort(sc.begin(), sc.end(), [](const Scored& a, const Scored& b) { return a.dist < b.dist; });
    uint32_t m = 0;
    deg_out = 0;
    if (sc.empty()) return ok;
    g[m++] = sc[0].id;
    for (size_t j = 1; j < sc.size(); ++j) {
        // keep candidate j only if it is closer to i than to every neighbour kept so far
        const float* pj = ds + (size_t)sc[j].id * d;
        bool keep = true;
        for (uint32_t l = 0; l < m && keep; ++l)
            keep = !(dist(pj, pi, d) + eps > dist(pj, ds + (size_t)g[l] * d, d));
        if (keep) g[m++] = sc[j].id;
        if ((int)m == M) break;  // :555
    }
    // :559-563 -- the M/2 nearest are always linked
    for (int j = 0; j < M / 2 && j < (int)sc.size(); ++j)
        if (std::find(g, g + m, sc[j].id) == g + m) g[m++] = sc[j].id;
    deg_out = m;
    return ok;
}

// Everything after the per-node pruning: nodes the device left to the host (deg == 0xFFFFFFFF: equal distances in
// the candidate list, where only std::sort itself gives the reference's order; or lists the kernel does not take),
// the serial, order-dependent reverse-edge pass (support_func.h:417-442), and the CSR arrays.
// adj: [n x 2M], deg: [n].  Internal to the library (called by gbnns_build_graph_gd and gbnns_build_graph_gd_device).
extern "C" int gbnns_internal_gd_finish(const uint64_t* knn_offsets, const uint32_t* knn_nbrs, const float* ds,
                                        uint64_t n, uint32_t d, int M, int metric, int reverse, int threads,
                                        uint32_t* adj, uint32_t* deg, uint64_t* host_nodes, uint64_t** out_offsets,
                                        uint32_t** out_nbrs) {
    float (*dist)(const float*, const float*, uint32_t) = metric == GBNNS_METRIC_NEG_DOT ? dist_negdot : dist_l2;
    const uint32_t cap = 2u * (uint32_t)M;  // no list ever exceeds max(M + M/2, 2M) = 2M entries
    int bad = 0;
    uint64_t on_host = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel reduction(+ : on_host)
    {
        std::vector<Scored> sc;
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < (int64_t)n; ++i) {
            if (deg[i] != 0xFFFFFFFFu) continue;
            on_host += 1;
            uint32_t m = 0;
            if (!prune_node((uint64_t)i, knn_offsets, knn_nbrs, ds, n, d, M, dist, sc, adj + (size_t)i * cap, m)) bad = 1;
            deg[i] = m;
        }
    }
    if (host_nodes) *host_nodes = on_host;
    if (bad) return GBNNS_ERR_INVALID;
    if (reverse) {
        // serial and order dependent, like the reference (:417-442)
        std::vector<uint32_t> indeg(n, 0);
        for (uint64_t i = 0; i < n; ++i)
            for (uint32_t j = 0; j < deg[i]; ++j) indeg[adj[(size_t)i * cap + j]]++;
        for (uint64_t i = 0; i < n; ++i) {
            int budget = std::min(M - (int)indeg[i], M / 2);
            if (budget <= 0) continue;
            for (uint32_t j = 0; j < deg[i]; ++j) {
                const uint32_t c = adj[(size_t)i * cap + j];
                if (deg[c] >= cap) continue;
                uint32_t* gc = adj + (size_t)c * cap;
                if (std::find(gc, gc + deg[c], (uint32_t)i) != gc + deg[c]) continue;
                gc[deg[c]++] = (uint32_t)i;
                if (--budget <= 0) break;
            }
        }
    }
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; ++i) total += deg[i];
    uint64_t* off = (uint64_t*)std::malloc((n + 1) * sizeof(uint64_t));
    uint32_t* nb = (uint32_t*)std::malloc(std::max<uint64_t>(total, 1) * sizeof(uint32_t));
    if (!off || !nb) {
        std::free(off);
        std::free(nb);
        return GBNNS_ERR_OOM;
    }
    uint64_t p = 0;
    for (uint64_t i = 0; i < n; ++i) {
        off[i] = p;
        std::memcpy(nb + p, adj + (size_t)i * cap, (size_t)deg[i] * 4);
        p += deg[i];
    }
    off[n] = p;
    *out_offsets = off;
    *out_nbrs = nb;
    return GBNNS_OK;
}

extern "C" int gbnns_build_graph_gd(const uint64_t* knn_offsets, const uint32_t* knn_nbrs,
                                    const float* ds, uint64_t n, uint32_t d, int M, int metric,
                                    int reverse, int threads, uint64_t** out_offsets,
                                    uint32_t** out_nbrs) {
    // M >= 2: with M = 1 the reference's `size == M` stop test (:555) can be stepped over.
    if (!knn_offsets || !knn_nbrs || !ds || !out_offsets || !out_nbrs || M < 2 || n == 0)
        return GBNNS_ERR_INVALID;
    *out_offsets = nullptr;
    *out_nbrs = nullptr;
    const uint32_t cap = 2u * (uint32_t)M;
    std::vector<uint32_t> adj((size_t)n * cap);
    std::vector<uint32_t> deg(n, 0xFFFFFFFFu);  // every node on the host
    return gbnns_internal_gd_finish(knn_offsets, knn_nbrs, ds, n, d, M, metric, reverse, threads, adj.data(), deg.data(),
                                    nullptr, out_offsets, out_nbrs);
}
