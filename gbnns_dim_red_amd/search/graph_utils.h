// graph_utils.h -- the rest of the reference's search/support_func.h that surrounds the search path: graph
// utilities used by prepare_graph.cpp / naive_test.cpp / performSyntheticTests, memory-mapped and byte-vector
// loaders for the real datasets (dim_red/data.py:69-78), and getTruth on the device.  Included at the end of
// support_func.h (one translation unit, like everything in this directory).  Host code in the reference's
// operation order; Metric::Dist here is the scalar form defined in support_func.h.
//
//   Neighbor (:51-66), createUniformData (:252-267), getTruth (:270-290), cutKNNbyThreshold / cutKNNbyK / cutKL
//   (:292-364), findGraphMaxDegree (:366-375), checkFileExistence (:377-380), mergeGraph (:383-399),
//   checkConstDegree / getConstantDegreeForGD / fillGraphToConstantDegree (:448-518).
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstring>

struct Neighbor {
    uint32_t number;
    float dist;
    size_t operator()(const Neighbor& n) const { return std::hash<uint32_t>()(n.number); }
};
// ordered by distance alone: equal distances compare equivalent (std::set keeps the first, std::sort leaves their
// order to the library -- as in the reference)
bool operator<(const Neighbor& x, const Neighbor& y) { return x.dist < y.dist; }

// ---- memory-mapped xvecs / bvecs (dim_red/data.py:69-78 mmap_fvecs / mmap_bvecs) ----------------------------
// A read-only view of a file of fixed-dimension records: [int32 dim][dim x T].  The big files of the real
// datasets (bigann_base.bvecs: 128-byte vectors behind a 4-byte header) are not copied into a vector first.
template <typename T>
class MappedXvecs {
    const unsigned char* base_ = nullptr;
    size_t bytes_ = 0, n_ = 0, d_ = 0, stride_ = 0;

public:
    MappedXvecs() {}
    explicit MappedXvecs(const string& path) { open(path); }
    MappedXvecs(const MappedXvecs&) = delete;
    MappedXvecs& operator=(const MappedXvecs&) = delete;
    ~MappedXvecs() { close(); }
    bool open(const string& path) {
        close();
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size < 4) {
            ::close(fd);
            return false;
        }
        void* p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (p == MAP_FAILED) return false;
        base_ = static_cast<const unsigned char*>(p);
        bytes_ = (size_t)st.st_size;
        int32_t dim = 0;
        std::memcpy(&dim, base_, 4);
        d_ = dim > 0 ? (size_t)dim : 0;
        stride_ = 4 + d_ * sizeof(T);
        n_ = d_ ? bytes_ / stride_ : 0;
        if (d_ == 0 || n_ * stride_ != bytes_) {  // not a whole number of records of that dimension
            close();
            return false;
        }
        return true;
    }
    void close() {
        if (base_) munmap(const_cast<unsigned char*>(base_), bytes_);
        base_ = nullptr;
        bytes_ = n_ = d_ = stride_ = 0;
    }
    bool good() const { return base_ != nullptr; }
    size_t size() const { return n_; }
    size_t dim() const { return d_; }
    const T* row(size_t i) const { return reinterpret_cast<const T*>(base_ + i * stride_ + 4); }
    // rows [first, first + count) as float (the search path works on binary32): checks every record's header like
    // readXvec does ("file error" + exit(1) on a mismatch)
    vector<float> toFloat(size_t first, size_t count, size_t d) const {
        if (!good() || d != d_ || first + count > n_) {
            std::cout << "file error\n";
            std::cout << "dim " << d_ << ", d " << d << std::endl;
            std::cout << "our fault\n";
            exit(1);
        }
        vector<float> out(count * d);
        for (size_t i = 0; i < count; ++i) {
            int32_t dim = 0;
            std::memcpy(&dim, base_ + (first + i) * stride_, 4);
            if ((size_t)dim != d) {
                std::cout << "file error\n";
                std::cout << "dim " << dim << ", d " << d << std::endl;
                std::cout << "our fault\n";
                exit(1);
            }
            const T* r = row(first + i);
            for (size_t j = 0; j < d; ++j) out[i * d + j] = (float)r[j];
        }
        return out;
    }
};

// The base / query vectors of a dataset, whatever container they ship in: <prefix>.fvecs (loadXvecs, the
// reference's only form) or, when that file is absent, <prefix>.bvecs (bigann-style byte vectors, converted to
// float like dim_red/data.py does before training).  GBNNS_MMAP=1 reads the fvecs through a mapping as well.
vector<float> loadVectorsAny(const string& prefix, const size_t d, const size_t n) {
    const string fv = prefix + ".fvecs", bv = prefix + ".bvecs";
    const bool want_map = getenv("GBNNS_MMAP") && atoi(getenv("GBNNS_MMAP"));
    if (std::ifstream(fv.c_str()).good()) {
        if (!want_map) return loadXvecs<float>(fv, d, n);
        MappedXvecs<float> m(fv);
        return m.toFloat(0, n, d);
    }
    if (std::ifstream(bv.c_str()).good()) {
        MappedXvecs<uint8_t> m(bv);
        return m.toFloat(0, n, d);
    }
    return loadXvecs<float>(fv, d, n);  // reports the missing file the way the reference does
}

// ---- graph utilities ---------------------------------------------------------------------------------------
vector<vector<uint32_t>> cutKNNbyThreshold(vector<vector<uint32_t>>& knn, vector<float>& ds, float thr, int N, int d,
                                           Metric* metric) {
    vector<vector<uint32_t>> kept(N);
    for (int i = 0; i < N; ++i) {
        const float* pi = ds.data() + (size_t)i * d;
        for (uint32_t cur : knn[i])
            if (metric->Dist(pi, ds.data() + (size_t)cur * d, d) < thr) kept[i].push_back(cur);
    }
    return kept;
}

vector<vector<uint32_t>> cutKNNbyK(vector<vector<uint32_t>>& knn, const float* ds, int knn_size, int N, int d,
                                   Metric* metric) {
    vector<vector<uint32_t>> kept(N);
    bool warned = false;
    for (int i = 0; i < N; ++i) {
        vector<Neighbor> neigs;
        const float* pi = ds + (size_t)i * d;
        for (uint32_t cur : knn[i]) neigs.push_back(Neighbor{cur, metric->Dist(pi, ds + (size_t)cur * d, d)});
        if (!warned && knn_size > (int)knn[i].size()) {
            cout << "Size knn less than you want" << endl;
            cout << knn[i].size() << endl;
            warned = true;
        }
        std::sort(neigs.begin(), neigs.end());
        const size_t take = std::min<size_t>((size_t)std::max(knn_size, 0), knn[i].size());
        for (size_t j = 0; j < take; ++j) kept[i].push_back(neigs[j].number);
    }
    return kept;
}

vector<vector<uint32_t>> cutKL(vector<vector<uint32_t>>& kl, int l, int N, vector<vector<uint32_t>>& knn) {
    vector<vector<uint32_t>> kept(N);
    for (int i = 0; i < N; ++i) {
        if (l > (int)kl[i].size()) {
            cout << "Graph have less edges that you want" << endl;
            exit(1);
        }
        vector<uint32_t> shuffled = kl[i];
        std::random_shuffle(shuffled.begin(), shuffled.end());
        for (size_t it = 0; (int)kept[i].size() < l && it < shuffled.size(); ++it)
            if (std::find(knn[i].begin(), knn[i].end(), shuffled[it]) == knn[i].end()) kept[i].push_back(shuffled[it]);
    }
    return kept;
}

int findGraphMaxDegree(vector<vector<uint32_t>>& graph) {
    size_t best = 0;
    for (const auto& row : graph) best = std::max(best, row.size());
    return (int)best;
}

inline bool checkFileExistence(string name) {
    std::ifstream f(name.c_str());
    return f.good();
}

vector<vector<uint32_t>> mergeGraph(vector<vector<uint32_t>>& graph_f, vector<vector<uint32_t>>& graph_s) {
    const size_t n = graph_f.size();
    vector<vector<uint32_t>> merged(n);
    for (size_t i = 0; i < n; ++i) {
        merged[i] = graph_f[i];
        for (uint32_t v : graph_s[i])
            if (std::find(merged[i].begin(), merged[i].end(), v) == merged[i].end()) merged[i].push_back(v);
    }
    return merged;
}

void checkConstDegree(vector<vector<uint32_t>>& graph) {
    if (graph.empty()) return;
    const size_t degree = graph[0].size();
    for (const auto& row : graph)
        if (row.size() != degree) {
            cout << " Graph degree is not constant " << endl;
            return;
        }
}

// Pads every GD list shorter than 2M with the node's kNN candidates it does not hold yet, in list order, SKIPPING
// the first candidate (the reference's loop starts at j = 1, support_func.h:473).
vector<vector<uint32_t>> getConstantDegreeForGD(vector<vector<uint32_t>>& graph, const float* ds,
                                                vector<vector<uint32_t>>& gd_graph, int M, size_t N, size_t d,
                                                Metric* metric) {
    (void)ds; (void)d; (void)metric;
    const size_t want = 2 * (size_t)M;
    for (size_t i = 0; i < N; ++i) {
        if (gd_graph[i].size() >= want) continue;
        for (size_t j = 1; j < graph[i].size(); ++j) {
            if (std::find(gd_graph[i].begin(), gd_graph[i].end(), graph[i][j]) != gd_graph[i].end()) continue;
            gd_graph[i].push_back(graph[i][j]);
            if (gd_graph[i].size() == want) break;
        }
    }
    checkConstDegree(gd_graph);
    return gd_graph;
}

vector<vector<uint32_t>> fillGraphToConstantDegree(vector<vector<uint32_t>>& graph, vector<vector<uint32_t>>& wide_graph,
                                                   int degree_needed) {
    const size_t N = graph.size();
    const int max_degree = findGraphMaxDegree(graph);
    if (degree_needed < max_degree) degree_needed = max_degree;
    for (size_t i = 0; i < N; ++i) {
        if ((int)graph[i].size() >= degree_needed) continue;
        for (uint32_t v : wide_graph[i]) {
            if (std::find(graph[i].begin(), graph[i].end(), v) != graph[i].end()) continue;
            graph[i].push_back(v);
            if ((int)graph[i].size() == degree_needed) break;
        }
    }
    checkConstDegree(graph);
    return graph;
}

// ---- synthetic sphere data + brute-force ground truth (the leftover experiment harness, :252-290) ------------
vector<float> createUniformData(int N, int d, std::mt19937 random_gen) {
    vector<float> ds((size_t)N * d);
    std::normal_distribution<float> norm_distr(0, 1);
    vector<float> point(d);
    for (int i = 0; i < N; ++i) {
        float sq = 0;
        for (int j = 0; j < d; ++j) {
            point[j] = norm_distr(random_gen);
            sq += point[j] * point[j];
        }
        const float norm = (float)pow(sq, 0.5);  // pow(float, double) -> double, stored to a float, as in the reference
        for (int j = 0; j < d; ++j) ds[(size_t)i * d + j] = point[j] / norm;
    }
    return ds;
}

// getTruth: strict minimum of Dist(ds_j, query_i) over ascending j.  The n_q x N distance scan runs on the device
// (gbnns_exact_knn, k = 1, the reference's arithmetic: same ids bit for bit); no host fallback.
vector<uint32_t> getTruth(vector<float> ds, vector<float> query, int N, int d, int N_q, Metric* metric) {
    vector<uint32_t> truth(N_q);
    if (N_q == 0) return truth;
    const char* dev = getenv("GBNNS_DEVICE");
    if (gbnns_exact_knn(dev ? atoi(dev) : 0, ds.data(), (uint64_t)N, query.data(), (uint64_t)N_q, (uint32_t)d, 1,
                        metric->gbnnsMetric(), -1, truth.data(), nullptr, GBNNS_MEM_HOST, nullptr)) {
        std::cerr << "gbnns: getTruth: " << gbnns_last_error() << std::endl;
        exit(2);
    }
    return truth;
}
