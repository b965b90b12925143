// support_func.h -- host-side primitives of the drop-in, mirroring the names final_test.cpp and
// search_function.h use from the reference's search/support_func.h:
//   Net (:45-49), Metric / L2Metric / Angular (:87-163), findGraphAverageDegree (:166),
//   readXvec / writeXvec / loadXvecs (:176-228), writeEdges / loadEdges (:205-249),
//   splitString / addMapFromStr / readSearchParams / getVectorFromString (:578-621),
//   hnswlikeGD (:521-575), GetLowQueryFromNet (:645-658); the remaining graph utilities, the synthetic-data helpers
//   and the mmap / bvecs loaders live in graph_utils.h, included at the end of this file.
// Like the reference's header it holds non-inline definitions: include it from one .cpp only.
//
// The Metric classes here are plain scalar code in the reference's operation order (4 running
// sums for L2, 8 for the dot product); they are used on the host only for the harness's scoring
// rule (search_function.h:394-399) and to tell the device which metric a Metric* stands for.
// Compile host code with -ffp-contract=off.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/gbnns.h"

using namespace std;

struct Net {
    vector<float> layerFirst;   // d_hidden rows of [W(d) | b]
    vector<float> layerSecond;  // d_hidden rows of [W(d_hidden) | b]
    vector<float> layerFinal;   // d_low rows of [W(d_hidden) | b]
};

class Metric {
public:
    virtual float Dist(const float* x, const float* y, size_t d) = 0;
    virtual int gbnnsMetric() const = 0;  // which device metric this object stands for
    virtual ~Metric() {}
};

// squared L2 over the first 4*floor(d/4) dims, 4 running sums, ((s0+s1)+s2)+s3
class L2Metric : public Metric {
public:
    float Dist(const float* a, const float* b, size_t d) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (size_t t = 0; t + 4 <= d; t += 4)
            for (int j = 0; j < 4; ++j) {
                const float e = a[t + j] - b[t + j];
                s[j] = s[j] + e * e;
            }
        return ((s[0] + s[1]) + s[2]) + s[3];
    }
    int gbnnsMetric() const { return GBNNS_METRIC_L2; }
};

// negative dot product, 8 running sums folded 8->4, optional 4-wide and masked tail steps
class Angular : public Metric {
public:
    float Dist(const float* x, const float* y, size_t d) {
        float c[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        size_t k = 0;
        for (; k + 8 <= d; k += 8)
            for (int l = 0; l < 8; ++l) c[l] = c[l] + x[k + l] * y[k + l];
        float m[4];
        for (int j = 0; j < 4; ++j) m[j] = c[j + 4] + c[j];
        if (d - k >= 4) {
            for (int j = 0; j < 4; ++j) m[j] = m[j] + x[k + j] * y[k + j];
            k += 4;
        }
        if (d > k)
            for (size_t j = 0; j < 4; ++j) {
                const float xv = k + j < d ? x[k + j] : 0.f;
                const float yv = k + j < d ? y[k + j] : 0.f;
                m[j] = m[j] + xv * yv;
            }
        return -((m[0] + m[1]) + (m[2] + m[3]));
    }
    int gbnnsMetric() const { return GBNNS_METRIC_NEG_DOT; }
};

int findGraphAverageDegree(vector<vector<uint32_t>>& graph) {
    double total = 0;
    for (const auto& row : graph) total += row.size();
    return graph.empty() ? 0 : (int)(total / graph.size());
}

// ---- xvecs: per row [int32 dim][dim * sizeof(T) bytes] -------------------------------------
template <typename T>
void readXvec(std::ifstream& in, T* data, const size_t d, const size_t n = 1) {
    for (size_t i = 0; i < n; ++i) {
        uint32_t dim = 0;
        in.read(reinterpret_cast<char*>(&dim), sizeof dim);
        if (!in || dim != d) {  // same fatal behaviour as the reference (:182-188)
            std::cout << "file error\n";
            std::cout << "dim " << dim << ", d " << d << std::endl;
            std::cout << "our fault\n";
            exit(1);
        }
        in.read(reinterpret_cast<char*>(data + i * d), d * sizeof(T));
    }
}

template <typename T>
void writeXvec(std::ofstream& out, T* data, const size_t d, const size_t n = 1) {
    const uint32_t dim = (uint32_t)d;
    for (size_t i = 0; i < n; ++i) {
        out.write(reinterpret_cast<const char*>(&dim), sizeof dim);
        out.write(reinterpret_cast<const char*>(data + i * d), d * sizeof(T));
    }
}

template <typename T>
vector<T> loadXvecs(string dataPath, const size_t d, const size_t n = 1) {
    vector<T> data(n * d);
    std::ifstream in(dataPath.c_str(), std::ios::binary);
    readXvec<T>(in, data.data(), d, n);
    return data;
}

// ---- edge lists: per node [uint32 size][size * uint32 ids] ---------------------------------
void writeEdges(string location, const std::vector<std::vector<uint32_t>>& edges) {
    std::cout << "Saving edges to " << location << std::endl;
    std::ofstream out(location.c_str(), std::ios::binary);
    for (const auto& row : edges) {
        const uint32_t size = (uint32_t)row.size();
        out.write(reinterpret_cast<const char*>(&size), sizeof size);
        out.write(reinterpret_cast<const char*>(row.data()), sizeof(uint32_t) * size);
    }
}

vector<std::vector<uint32_t>> loadEdges(string location, uint32_t n, string edges_name) {
    std::vector<std::vector<uint32_t>> edges(n);
    std::ifstream in(location.c_str(), std::ios::binary);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t size = 0;
        in.read(reinterpret_cast<char*>(&size), sizeof size);
        if (!in) {  // the reference reads on without checking; a short file is fatal here
            std::cout << "file error\n" << location << ": truncated at node " << i << std::endl;
            exit(1);
        }
        edges[i].resize(size);
        in.read(reinterpret_cast<char*>(edges[i].data()), sizeof(uint32_t) * size);
    }
    cout << edges_name + " " << findGraphAverageDegree(edges) << endl;
    return edges;
}

// ---- parameters_of_databases.txt: "<dataset> <key> <value>" lines ---------------------------
std::vector<string> splitString(const string& str, char delimiter) {
    std::vector<string> tokens;
    std::istringstream ss(str);
    string tok;
    while (std::getline(ss, tok, delimiter)) tokens.push_back(tok);
    return tokens;
}

std::map<string, string> addMapFromStr(string str, std::map<string, string> paramsMap, string globalKey) {
    const std::vector<string> parts = splitString(str, ' ');
    if (parts.size() == 3 && parts[0] == globalKey) paramsMap[parts[1]] = parts[2];
    return paramsMap;
}

std::map<string, string> readSearchParams(string fileName, string databaseName) {
    std::map<string, string> paramsMap;
    std::ifstream file(fileName);
    string line;
    while (std::getline(file, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        paramsMap = addMapFromStr(line, paramsMap, databaseName);
    }
    return paramsMap;
}

vector<int> getVectorFromString(string str) {
    vector<int> values;
    for (const string& tok : splitString(str, ',')) values.push_back(atoi(tok.c_str()));
    return values;
}

// ---- CSR view of an adjacency in the reference's vector<vector<uint32_t>> form ---------------
struct GbnnsCsr {
    std::vector<uint64_t> offsets;
    std::vector<uint32_t> nbrs;
};

inline GbnnsCsr gbnnsToCsr(const vector<vector<uint32_t>>& graph) {
    GbnnsCsr csr;
    csr.offsets.resize(graph.size() + 1, 0);
    for (size_t i = 0; i < graph.size(); ++i) csr.offsets[i + 1] = csr.offsets[i] + graph[i].size();
    csr.nbrs.reserve(csr.offsets.back());
    for (const auto& row : graph) csr.nbrs.insert(csr.nbrs.end(), row.begin(), row.end());
    return csr;
}

inline void gbnnsDie(const char* what) {
    std::cerr << "gbnns: " << what << ": " << gbnns_last_error() << std::endl;
    exit(2);
}

// hnswlikeGD (reference support_func.h:521-575): prunes a kNN graph into the search graph (the
// "GD" rule + the M/2 nearest + optional reverse edges).  Same signature; the per-node pruning runs on the
// device when there is one (gbnns_build_graph_gd_device: nodes with equal candidate distances and the reverse
// pass are finished on the host, so the graph is the host builder's bit for bit), else -- or with
// GBNNS_GD_HOST=1 -- on the host (gbnns_build_graph_gd, OpenMP, reference operation order).
// need_const_degree = true pads the lists to 2M afterwards (getConstantDegreeForGD, :466-485, graph_utils.h).
vector<vector<uint32_t>> getConstantDegreeForGD(vector<vector<uint32_t>>& graph, const float* ds,
                                                vector<vector<uint32_t>>& gd_graph, int M, size_t N, size_t d,
                                                Metric* metric);

vector<vector<uint32_t>> hnswlikeGD(vector<vector<uint32_t>>& graph, const float* ds, int M, size_t N, size_t d,
                                    Metric* metric, bool reverse, bool need_const_degree) {
    const GbnnsCsr knn = gbnnsToCsr(graph);
    uint64_t* off = nullptr;
    uint32_t* nbr = nullptr;
    const char* force_host = getenv("GBNNS_GD_HOST");
    const char* dev = getenv("GBNNS_DEVICE");
    int rc;
    if (gbnns_device_count() > 0 && !(force_host && atoi(force_host))) {
        uint64_t on_host = 0;
        rc = gbnns_build_graph_gd_device(dev ? atoi(dev) : 0, knn.offsets.data(), knn.nbrs.data(), ds, N, (uint32_t)d, M,
                                         metric->gbnnsMetric(), reverse ? 1 : 0, 0, &off, &nbr, &on_host);
    } else {
        rc = gbnns_build_graph_gd(knn.offsets.data(), knn.nbrs.data(), ds, N, (uint32_t)d, M, metric->gbnnsMetric(),
                                  reverse ? 1 : 0, 0, &off, &nbr);
    }
    if (rc) {
        std::cerr << "gbnns: graph builder failed (bad ids, M < 2 or out of memory): " << gbnns_last_error() << std::endl;
        exit(2);
    }
    vector<vector<uint32_t>> out(N);
    for (size_t i = 0; i < N; ++i) out[i].assign(nbr + off[i], nbr + off[i + 1]);
    gbnns_free(off);
    gbnns_free(nbr);
    if (need_const_degree) out = getConstantDegreeForGD(graph, ds, out, M, N, d, metric);  // :570-572
    return out;
}

// GetLowQueryFromNet: one query through the 3-layer net on the device (gbnns_project).  Same
// signature as support_func.h:645-646; `zeros`, `ang`, `l2` are accepted for compatibility.
// A throw-away single-vector index carries the net; batch callers use performNetTest instead.
void GetLowQueryFromNet(const Net* net, const float* query, vector<float>& ans, const float* zeros,
                        size_t d, size_t d_hidden, size_t d_hidden_2, size_t d_low, Metric* ang,
                        Metric* l2) {
    (void)zeros; (void)ang; (void)l2;
    if (d_hidden_2 != d_hidden) {
        std::cerr << "gbnns: GetLowQueryFromNet needs d_hidden_2 == d_hidden" << std::endl;
        exit(2);
    }
    static const Net* cached_net = nullptr;
    static gbnns_index* cached = nullptr;
    if (cached_net != net) {
        if (cached) gbnns_index_destroy(cached);
        const uint64_t off[2] = {0, 0};
        const uint32_t none = 0;
        std::vector<float> low(d_low, 0.f);
        gbnns_index_desc desc = {};
        desc.struct_size = sizeof desc;
        desc.n = 1;
        desc.d = (uint32_t)d;
        desc.d_low = (uint32_t)d_low;
        desc.d_hidden = (uint32_t)d_hidden;
        desc.db = query;
        desc.db_low = low.data();
        desc.graph_offsets = off;
        desc.graph_nbrs = &none;
        desc.net_l1 = net->layerFirst.data();
        desc.net_l2 = net->layerSecond.data();
        desc.net_l3 = net->layerFinal.data();
        if (gbnns_index_create(&desc, &cached)) gbnnsDie("GetLowQueryFromNet");
        cached_net = net;
    }
    ans.resize(d_low);
    if (gbnns_project(cached, query, 1, ans.data(), GBNNS_MEM_HOST, nullptr)) gbnnsDie("gbnns_project");
}

#include "graph_utils.h"
