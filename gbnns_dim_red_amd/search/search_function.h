// search_function.h -- drop-in for the reference's search/search_function.h on MI355X.
//
// Same free functions, parameter order and types as the reference, so search/final_test.cpp
// compiles against this header unchanged apart from its data paths:
//   TripleResult (:7-12), getOneSearchResults (:43-47), getRealNearest (:105-107),
//   performTest (:128-134), performRealTests (:291-295), performNetTest (:319-325),
//   performRealNetTests (:411-415).
// What differs is where the work happens: the per-query loop inside the StopW region of
// performTest / performNetTest (:151-188, :346-387) is ONE gbnns_search_ex call for the whole
// batch (host buffers in, answers out, inside the same timed region); scoring and the result
// line are produced exactly as the reference does (:389-407).
//
// use_second_graph = true / llf / hops_bound (auxiliary "long link" graph, :73-89; naive_test.cpp:103-105)
// are served by the device too: the auxiliary graph is attached to the index
// (gbnns_index_set_aux_graph) and the call sets GBNNS_FLAG_AUX_GRAPH / GBNNS_FLAG_LLF.
// Several entry points per query (:54) are served too (same count for every query of a batch; exact, on the
// general kernel -- no driver of the reference uses them).
// makeStep (:15-40) keeps its signature for callers that drive a walk step by step on host containers; the batch
// functions never call it (on the device the step is part of the walk kernels).
// Besides the result line, every sweep point is appended as one JSON object to <output_txt>.json (queries/s,
// recall, hops, dist_calc, ef, devices) -- the machine-readable sidecar SURVEY.md section 5 asks for.
#pragma once

#include <cstring>
#include <queue>
#include <set>
#include <tuple>

#include "support_classes.h"
#include "visited_list_pool.h"

using namespace std;

struct TripleResult {
    priority_queue<pair<float, int>> topk;
    int hops;
    int dist_calc;
    int degree;
};

// makeStep (search_function.h:15-40): offers the not-yet-visited neighbours of one node, in list order, to the two
// heaps with the reference's strict rule.  Host containers in, host containers out -- a compatibility entry for
// code that steps a walk itself; it is NOT how this build searches (getOneSearchResults and the perform*
// functions run whole walks on the device, where this step lives inside the walk kernels).
void makeStep(vector<uint32_t>& graph_level, const float* query, const float* db,
              priority_queue<pair<float, int>>& topResults, priority_queue<std::pair<float, int>>& candidateSet,
              Metric* metric, uint32_t d, int& query_dist_calc, bool& found, int& ef, int& k, VisitedList* vl) {
    (void)k;
    for (size_t j = 0; j < graph_level.size(); ++j) {
        const int id = (int)graph_level[j];
        if (vl->mass[id] == vl->curV) continue;
        vl->mass[id] = vl->curV;
        const float dist = metric->Dist(query, db + (size_t)id * d, d);
        query_dist_calc++;
        if (topResults.top().first > dist || (int)topResults.size() < ef) {
            candidateSet.emplace(-dist, id);
            found = true;
            topResults.emplace(dist, id);
            if ((int)topResults.size() > ef) topResults.pop();
        }
    }
}

// ---- device index cache ----------------------------------------------------------------------
// The reference passes raw vectors to every call; the device copy is created on first use and
// reused while the same buffers are passed again (performRealNetTests calls performNetTest once
// per ef with identical data).  The key holds the buffers' addresses, their sizes and a fingerprint of a strided
// sample of their contents, so a caller that refills a vector in place gets a fresh device copy.
typedef std::tuple<const void*, const void*, const void*, const void*, size_t, size_t, size_t, int, uint64_t> GbnnsKey;

inline std::map<GbnnsKey, gbnns_index*>& gbnnsCache() {
    static std::map<GbnnsKey, gbnns_index*> cache;
    return cache;
}

// Several devices (GBNNS_DEVICES=0,1,...): the handle of the cache is replica 0 of a gbnns_multi; this map finds
// the multi handle of such a replica (single-device handles are absent from it).
inline std::map<gbnns_index*, gbnns_multi*>& gbnnsMultiOf() {
    static std::map<gbnns_index*, gbnns_multi*> m;
    return m;
}

// auxiliary graph (by address) currently attached to a handle
inline std::map<gbnns_index*, const void*>& gbnnsAttached() {
    static std::map<gbnns_index*, const void*> m;
    return m;
}

// devices the drop-in searches on: GBNNS_DEVICES (comma list; "all" = every visible device), else GBNNS_DEVICE, else 0
inline vector<int32_t> gbnnsDevices() {
    vector<int32_t> devs;
    if (const char* list = getenv("GBNNS_DEVICES")) {
        if (string(list) == "all") {
            for (int i = 0; i < gbnns_device_count(); ++i) devs.push_back(i);
        } else {
            for (const string& tok : splitString(list, ','))
                if (!tok.empty()) devs.push_back((int32_t)atoi(tok.c_str()));
        }
    }
    if (devs.empty()) {
        const char* dev = getenv("GBNNS_DEVICE");
        devs.push_back(dev ? atoi(dev) : 0);
    }
    return devs;
}

inline uint64_t gbnnsFingerprint(const float* p, size_t count) {
    // FNV-1a over <= 4096 evenly spaced elements (bit patterns) and the length
    uint64_t h = 1469598103934665603ull ^ (uint64_t)count;
    if (!p || count == 0) return h;
    const size_t step = count / 4096 + 1;
    for (size_t i = 0; i < count; i += step) {
        uint32_t b;
        std::memcpy(&b, p + i, 4);
        h = (h ^ b) * 1099511628211ull;
    }
    return h;
}

inline void gbnnsForget(gbnns_index* ix);   // (the warm-up record of a handle that is being destroyed: below, with gbnnsFirstUse)

inline gbnns_index* gbnnsIndexFor(vector<vector<uint32_t>>& graph, const float* db, size_t n, size_t d,
                                  const float* db_low, size_t d_low, const Net* net, size_t d_hidden,
                                  Metric* metric) {
    uint64_t fp = gbnnsFingerprint(db, n * d) ^ (gbnnsFingerprint(db_low, n * d_low) << 1);
    size_t edges = 0;
    for (size_t i = 0; i < graph.size(); i += graph.size() / 1024 + 1) edges = edges * 31 + graph[i].size() + (graph[i].empty() ? 0 : graph[i][0]);
    fp ^= (uint64_t)edges * 0x9E3779B97F4A7C15ull;
    if (net) fp ^= gbnnsFingerprint(net->layerFirst.data(), net->layerFirst.size()) + gbnnsFingerprint(net->layerFinal.data(), net->layerFinal.size());
    const GbnnsKey key(db, db_low, &graph, net, n, d, d_low, metric->gbnnsMetric(), fp);
    auto it = gbnnsCache().find(key);
    if (it != gbnnsCache().end()) return it->second;
    // same buffers, other contents (a caller refilled a vector in place): the old device copy is dead weight
    for (auto old = gbnnsCache().begin(); old != gbnnsCache().end();) {
        const GbnnsKey& k = old->first;
        if (std::get<0>(k) == std::get<0>(key) && std::get<1>(k) == std::get<1>(key) && std::get<2>(k) == std::get<2>(key) &&
            std::get<3>(k) == std::get<3>(key) && std::get<7>(k) == std::get<7>(key)) {
            gbnnsAttached().erase(old->second);
            gbnnsForget(old->second);
            auto mi = gbnnsMultiOf().find(old->second);
            if (mi != gbnnsMultiOf().end()) {
                gbnns_multi_destroy(mi->second);
                gbnnsMultiOf().erase(mi);
            } else {
                gbnns_index_destroy(old->second);
            }
            old = gbnnsCache().erase(old);
        } else {
            ++old;
        }
    }
    const GbnnsCsr csr = gbnnsToCsr(graph);
    gbnns_index_desc desc = {};
    desc.struct_size = sizeof desc;
    desc.metric = metric->gbnnsMetric();
    desc.mem_kind = GBNNS_MEM_HOST;
    desc.n = n;
    desc.d = (uint32_t)d;
    desc.db = db;
    if (db_low) {
        desc.d_low = (uint32_t)d_low;
        desc.db_low = db_low;
    }
    desc.graph_offsets = csr.offsets.data();
    desc.graph_nbrs = csr.nbrs.data();
    if (net) {
        desc.d_hidden = (uint32_t)d_hidden;
        desc.net_l1 = net->layerFirst.data();
        desc.net_l2 = net->layerSecond.data();
        desc.net_l3 = net->layerFinal.data();
    }
    const vector<int32_t> devs = gbnnsDevices();
    gbnns_index* ix = nullptr;
    if (devs.size() > 1) {
        // one replica per device, the query loop (:152 / :346) is cut into contiguous blocks (gbnns_multi_search_ex)
        gbnns_multi* multi = nullptr;
        if (gbnns_multi_create(&desc, devs.data(), (int32_t)devs.size(), &multi)) {
            std::cerr << "gbnns_multi_create: " << gbnns_multi_last_error() << std::endl;
            exit(1);
        }
        ix = gbnns_multi_replica(multi, 0);
        gbnnsMultiOf()[ix] = multi;
    } else {
        desc.device = devs[0];
        if (gbnns_index_create(&desc, &ix)) gbnnsDie("gbnns_index_create");
    }
    gbnnsCache()[key] = ix;
    return ix;
}

// The reference's (auxiliary_graph, use_second_graph, llf, hops_bound) quadruple of one call.  llf has no
// effect without use_second_graph (:82), exactly as in the reference.
struct GbnnsAux {
    vector<vector<uint32_t>>* graph;
    bool use;
    bool llf;
    uint32_t hops_bound;
};

// Attaches `aux.graph` to the index unless it already is (keyed on the vector's address, like the index
// cache) and returns the flags of the search call.
inline uint32_t gbnnsAttachAux(gbnns_index* ix, const GbnnsAux& aux) {
    if (!aux.use) return 0;
    std::map<gbnns_index*, const void*>& attached = gbnnsAttached();
    if (attached[ix] != (const void*)aux.graph) {
        const GbnnsCsr csr = gbnnsToCsr(*aux.graph);
        auto mi = gbnnsMultiOf().find(ix);
        if (mi != gbnnsMultiOf().end()) {
            if (gbnns_multi_set_aux_graph(mi->second, csr.offsets.data(), csr.nbrs.data())) gbnnsDie("gbnns_multi_set_aux_graph");
        } else if (gbnns_index_set_aux_graph(ix, csr.offsets.data(), csr.nbrs.data())) {
            gbnnsDie("gbnns_index_set_aux_graph");
        }
        attached[ix] = (const void*)aux.graph;
    }
    return GBNNS_FLAG_AUX_GRAPH | (aux.llf ? GBNNS_FLAG_LLF : 0u);
}

// Entry points as the flat [n_q x m] array of the C ABI.  The reference takes a ragged vector per query
// (search_function.h:54); every driver passes exactly one, the device path takes any m as long as it is the same
// for all queries of a batch (it exits with a message otherwise -- never a silent difference).
inline vector<uint32_t> gbnnsEntries(const vector<vector<uint32_t>>& inter_points, size_t n_q, uint32_t& m) {
    m = inter_points.empty() ? 1u : (uint32_t)inter_points[0].size();
    if (m == 0) {
        std::cerr << "gbnns: a query without an entry point" << std::endl;
        exit(2);
    }
    vector<uint32_t> e(n_q * m, 0);
    for (size_t i = 0; i < n_q && i < inter_points.size(); ++i) {
        if (inter_points[i].size() != m) {
            std::cerr << "gbnns: the device path needs the same number of entry points for every query" << std::endl;
            exit(2);
        }
        for (uint32_t j = 0; j < m; ++j) e[i * m + j] = inter_points[i][j];
    }
    return e;
}

// One batch on the device.  mode NET / LOWQ: two-stage with ef = recheck_size;
// mode PLAIN: walk in the space of the index's `db` with (ef, k), answer = top of the heap trimmed to k (:174-181).
// Page-locks a host buffer for the guard's lifetime (gbnns_host_pin): the buffers the timed device calls read and
// write.  Set up before the StopW region, where the reference builds its VisitedListPool (:333); a buffer that cannot
// be pinned is simply used as it is.
struct GbnnsPin {
    void* p;
    GbnnsPin(const void* ptr, size_t bytes) : p(const_cast<void*>(ptr)) {
        if (!p || bytes == 0 || gbnns_host_pin(p, bytes) != GBNNS_OK) p = nullptr;
    }
    ~GbnnsPin() { if (p) gbnns_host_unpin(p); }
    GbnnsPin(const GbnnsPin&) = delete;
    GbnnsPin& operator=(const GbnnsPin&) = delete;
};

inline void gbnnsBatch(gbnns_index* ix, int mode, const float* queries, const float* queries_low,
                       size_t n_q, int ef, int k, const vector<uint32_t>& entries, vector<uint32_t>& ans,
                       vector<int32_t>& hops, vector<int32_t>& dist_calc, const GbnnsAux& aux, uint32_t n_entries = 1,
                       vector<int32_t>* edges = nullptr) {
    gbnns_search_args a = {};
    a.n_entries = n_entries;
    a.flags = gbnnsAttachAux(ix, aux);
    a.hops_bound = aux.hops_bound;
    a.struct_size = sizeof a;
    a.mode = mode;
    a.ef = ef;
    a.k = k;
    a.mem_kind = GBNNS_MEM_HOST;
    a.n_q = n_q;
    a.queries = queries;
    a.queries_low = queries_low;
    a.entry_ids = entries.data();
    a.out_ids = ans.data();
    a.out_hops = hops.data();
    a.out_dist_calc = dist_calc.data();
    if (edges) a.out_edges = edges->data();  // neighbour ids read: feeds the sidecar's byte count
    auto mi = gbnnsMultiOf().find(ix);
    if (mi != gbnnsMultiOf().end()) {  // GBNNS_DEVICES: contiguous query blocks, one per replica / device
        if (gbnns_multi_search_ex(mi->second, &a)) {
            std::cerr << "gbnns_multi_search_ex: " << gbnns_multi_last_error() << std::endl;
            exit(1);
        }
    } else if (gbnns_search_ex(ix, &a)) {
        gbnnsDie("gbnns_search_ex");
    }
}

// Untimed batches in front of a timed region, the same for every line of a sweep (round 6; until then only the FIRST beam of a sweep got
// them, and the low-dimensional-only branch none):
//   * first use of a handle in a mode: one batch with the arguments of the timed ones.  It is where the handle's workspaces (sized by
//     n_q), its stream and the kernels' code objects come into being -- 2.5 ms once per handle, which would otherwise sit inside the first
//     repeat of a sweep's first beam and halve that line's rate (final_test at full size: 14.6 M queries/s on the ef = 1 line, 26.8 M on the
//     next).  The reference does its own set-up at this very spot, outside its StopW region: `new VisitedListPool(1, n)`
//     (search_function.h:142-144, :331-333);
//   * first use of a (handle, mode, beam): one more batch -- the library sizes a beam's visited sets from the walks it has seen, so every
//     beam's timed repeats now run on measured sizes, not only the first beam's.
// The record is keyed by handle and forgotten when gbnnsIndexFor destroys the handle (a new handle at the same address starts over); the
// results sidecar says how many untimed batches a line had (`warmup_batches`).  GBNNS_NO_WARMUP=1 leaves everything inside the timed
// region.  Host code of the drop-in is single-threaded, like the reference's drivers.
inline std::set<std::tuple<gbnns_index*, int, int>>& gbnnsUsed() {
    static std::set<std::tuple<gbnns_index*, int, int>> used;   // (handle, mode, ef); ef = -1: the handle + mode entry
    return used;
}
inline void gbnnsForget(gbnns_index* ix) {
    auto& used = gbnnsUsed();
    for (auto it = used.begin(); it != used.end();) it = std::get<0>(*it) == ix ? used.erase(it) : std::next(it);
}
inline int gbnnsFirstUse(gbnns_index* ix, int mode, const float* queries, const float* queries_low, size_t n_q, int ef, int k,
                         const vector<uint32_t>& entries, vector<uint32_t>& ans, vector<int32_t>& hops, vector<int32_t>& dist_calc,
                         const GbnnsAux& aux, uint32_t n_entries, vector<int32_t>* edges) {
    static const bool off = getenv("GBNNS_NO_WARMUP") && atoi(getenv("GBNNS_NO_WARMUP")) != 0;
    if (off) return 0;
    int batches = 0;
    if (gbnnsUsed().insert(std::make_tuple(ix, mode, -1)).second) batches += 1;
    if (gbnnsUsed().insert(std::make_tuple(ix, mode, ef)).second) batches += 1;
    for (int i = 0; i < batches; ++i)
        gbnnsBatch(ix, mode, queries, queries_low, n_q, ef, k, entries, ans, hops, dist_calc, aux, n_entries, edges);
    return batches;
}

// ---- per-query entry points (single-query device calls; the batch functions below are the fast
// path) ------------------------------------------------------------------------------------------

TripleResult getOneSearchResults(const float* query, const float* db, uint32_t N, uint32_t d,
                                 vector<vector<uint32_t>>& main_graph, vector<vector<uint32_t>>& auxiliary_graph,
                                 int ef, int k, vector<uint32_t>& inter_points, Metric* metric,
                                 VisitedListPool* visitedlistpool, bool use_second_graph, bool llf,
                                 uint32_t hops_bound) {
    (void)visitedlistpool;
    const GbnnsAux aux = {&auxiliary_graph, use_second_graph, llf, hops_bound};
    if (inter_points.empty()) {
        std::cerr << "gbnns: a query without an entry point" << std::endl;
        exit(2);
    }
    gbnns_index* ix = gbnnsIndexFor(main_graph, db, N, d, nullptr, 0, nullptr, 0, metric);
    const int kept = k < ef ? k : ef;
    vector<uint32_t> cand(kept);
    vector<float> cdist(kept);
    uint32_t best = 0;
    int32_t hops = 0, dc = 0;
    gbnns_search_args a = {};
    a.struct_size = sizeof a;
    a.mode = GBNNS_MODE_PLAIN;
    a.ef = ef;
    a.k = kept;
    a.mem_kind = GBNNS_MEM_HOST;
    a.n_q = 1;
    a.queries = query;
    a.entry_ids = inter_points.data();
    a.n_entries = (uint32_t)inter_points.size();
    a.out_ids = &best;
    a.out_hops = &hops;
    a.out_dist_calc = &dc;
    a.out_cand = cand.data();
    a.out_cand_dist = cdist.data();
    a.flags = gbnnsAttachAux(ix, aux);
    a.hops_bound = hops_bound;
    if (gbnns_search_ex(ix, &a)) gbnnsDie("gbnns_search_ex");
    TripleResult r;
    for (int i = 0; i < kept; ++i)
        if (cand[i] != 0xFFFFFFFFu) r.topk.emplace(cdist[i], (int)cand[i]);
    r.hops = hops;
    r.dist_calc = dc;
    r.degree = 0;
    return r;
}

int getRealNearest(const float* point_q, int k, int d, int d_low, priority_queue<pair<float, int>>& topk,
                   vector<float>& ds, Metric* metric) {
    (void)k; (void)d_low;
    vector<uint32_t> ids;
    while (!topk.empty()) {  // pop order: worst -> best, as the reference walks the heap (:109-122)
        ids.push_back((uint32_t)topk.top().second);
        topk.pop();
    }
    static vector<vector<uint32_t>> no_graph;  // re-rank needs no adjacency
    if (no_graph.size() != ds.size() / d) no_graph.assign(ds.size() / d, vector<uint32_t>());
    gbnns_index* ix = gbnnsIndexFor(no_graph, ds.data(), ds.size() / d, d, nullptr, 0, nullptr, 0, metric);
    uint32_t out = 0;
    const int32_t count = (int32_t)ids.size();
    if (gbnns_rerank(ix, point_q, 1, ids.data(), (uint32_t)ids.size(), &count, &out, GBNNS_MEM_HOST, nullptr))
        gbnnsDie("gbnns_rerank");
    return (int)out;
}

// ---- scoring + result line (search_function.h:190-209 / :389-407) ------------------------------
inline void gbnnsScore(const vector<uint32_t>& ans, vector<float>& ds, vector<uint32_t>& truth, int d,
                       int n_q, int n_tr, Metric* metric, float& acc) {
    for (int i = 0; i < n_q; ++i) {
        acc += ans[i] == truth[(size_t)i * n_tr];
        // SIFT duplicate rule: the 2nd ground-truth id also counts when it is an exact duplicate
        const float* first = ds.data() + (size_t)d * truth[(size_t)i * n_tr];
        const float* second = ds.data() + (size_t)d * truth[(size_t)i * n_tr + 1];
        const float dist = metric->Dist(first, second, d);
        if (dist == 0 and truth[(size_t)i * n_tr] != truth[(size_t)i * n_tr + 1])
            acc += ans[i] == truth[(size_t)i * n_tr + 1];
    }
}

// What one query moves, SURVEY.md section 8(d): low-dim (walked) rows gathered + neighbour ids read + row offsets +
// the query and its result, plus -- two-stage -- the re-ranked original-space rows, the original query and the answer.
struct GbnnsTraffic {
    double walk_dist_calc;  // mean distances of the walk (without the "+ recheck_size" the harness adds, :362)
    double edges;           // mean neighbour ids read
    int walked_dim;         // dimension of the walked space
    bool rerank;            // original-space re-rank of `ef` candidates
    int warmup_batches;     // untimed batches in front of this line's timed region (gbnnsFirstUse)
};

// one JSON object per sweep point, appended to <output_txt>.json
inline void gbnnsSidecar(const char* output_txt, const string& graph_name, int ef, int k, int recheck_size, double acc,
                         double hops, double dist_calc, double sec_per_query, int num_exp, int n_q, int n, int d, int d_low,
                         const GbnnsTraffic& t) {
    if (!output_txt) return;
    std::ofstream js((string(output_txt) + ".json").c_str(), std::ios_base::app);
    const vector<int32_t> devs = gbnnsDevices();
    const double walk_bytes = t.walk_dist_calc * 4.0 * t.walked_dim + 4.0 * t.edges + 8.0 * hops + 4.0 * t.walked_dim + 4.0 * ef;
    const double rerank_bytes = t.rerank ? (double)ef * 4.0 * d + 4.0 * d + 4.0 * ef + 4.0 : 0.0;
    const double gbps = sec_per_query > 0 ? (walk_bytes + rerank_bytes) / sec_per_query * 1e-9 : 0.0;
    const double peak_gbps = 8000.0 * (double)devs.size();  // MI355X HBM3E: 8 TB/s per device
    js << "{\"graph_type\": \"" << graph_name << "\", \"ef\": " << ef << ", \"k\": " << k << ", \"recheck_size\": " << recheck_size
       << ", \"n\": " << n << ", \"n_q\": " << n_q << ", \"d\": " << d << ", \"d_low\": " << d_low
       << ", \"repeats\": " << num_exp << ", \"recall_at_1\": " << acc << ", \"mean_hops\": " << hops
       << ", \"mean_dist_calc\": " << dist_calc << ", \"sec_per_query\": " << sec_per_query
       << ", \"queries_per_s\": " << (sec_per_query > 0 ? 1.0 / sec_per_query : 0.0)
       << ", \"timed_region\": \"host buffers in, answers out (H2D + kernels + D2H), as search_function.h:346-387\""
       << ", \"warmup_batches\": " << t.warmup_batches
       << ", \"algorithmic_bytes_per_query\": " << (walk_bytes + rerank_bytes)
       << ", \"algorithmic_bytes_walk_part\": " << walk_bytes << ", \"mean_edges_read\": " << t.edges
       << ", \"GBps\": " << gbps << ", \"roofline_peak_GBps\": " << peak_gbps
       << ", \"roofline_frac\": " << gbps / peak_gbps
       << ", \"roofline_note\": \"algorithmic bytes (SURVEY 8d) over the WHOLE timed region incl. PCIe copies; the kernel-level figure is bench.py's roofline.frac\""
       << ", \"devices\": [";
    for (size_t i = 0; i < devs.size(); ++i) js << (i ? ", " : "") << devs[i];
    js << "], \"backend\": \"libgbnns_hip (gfx950)\"}" << std::endl;
}

inline void gbnnsReport(std::ofstream& outfile, const string& graph_name, float acc, long long hops,
                        long long dist_calc, float work_time, int num_exp, int n_q) {
    // same expression shapes as the reference: float / int, integer / integer, float / double
    const long long denom = (long long)num_exp * n_q;
    cout << "graph_type " << graph_name << " acc " << acc / (num_exp * n_q) << " hops " << hops / denom
         << " dist_calc " << dist_calc / denom << " work_time " << work_time / (num_exp * 1e6 * n_q) << endl;
    outfile << "graph_type " << graph_name << " acc " << acc / (num_exp * n_q) << " hops " << hops / denom
            << " dist_calc " << dist_calc / denom << " work_time " << work_time / (num_exp * 1e6 * n_q) << endl;
}

// ---- batch harness ---------------------------------------------------------------------------

void performTest(vector<vector<uint32_t>>& knn_graph, vector<vector<uint32_t>>& kl_graph, vector<float>& ds,
                 vector<float>& queries, vector<float>& ds_low, vector<float>& queries_low,
                 vector<uint32_t>& truth, int n, int d, int d_low, int n_q, int n_tr, int ef, int k,
                 string graph_name, Metric* metric, const char* output_txt,
                 vector<vector<uint32_t>> inter_points, bool use_second_graph, bool llf, uint32_t hops_bound,
                 int dist_calc_boost, int recheck_size, int number_exper, int number_of_threads) {
    (void)number_of_threads;
    const GbnnsAux aux = {&kl_graph, use_second_graph, llf, hops_bound};
    std::ofstream outfile;
    outfile.open(output_txt, std::ios_base::app);

    // which device call reproduces the reference's branch (:158-182)
    gbnns_index* ix;
    int mode, run_ef, run_k;
    const float* q_main = queries.data();
    const float* q_low = nullptr;
    if (d != d_low && recheck_size > 0) {
        ix = gbnnsIndexFor(knn_graph, ds.data(), n, d, ds_low.data(), d_low, nullptr, 0, metric);
        mode = GBNNS_MODE_LOWQ; run_ef = recheck_size; run_k = recheck_size;
        q_low = queries_low.data();
    } else if (d != d_low) {
        ix = gbnnsIndexFor(knn_graph, ds_low.data(), n, d_low, nullptr, 0, nullptr, 0, metric);
        mode = GBNNS_MODE_PLAIN; run_ef = ef; run_k = k;
        q_main = queries_low.data();
    } else {
        ix = gbnnsIndexFor(knn_graph, ds.data(), n, d, nullptr, 0, nullptr, 0, metric);
        mode = GBNNS_MODE_PLAIN; run_ef = ef; run_k = k;
    }
    uint32_t n_entries = 1;
    const vector<uint32_t> entries = gbnnsEntries(inter_points, n_q, n_entries);

    long long hops = 0;
    long long dist_calc = 0 + (long long)dist_calc_boost * n_q;
    float acc = 0;
    float work_time = 0;
    int num_exp = 0;
    vector<int32_t> q_hops(n_q), q_dc(n_q), q_edges(n_q);
    long long walk_dc = 0, edges = 0;
    vector<uint32_t> ans(n_q);
    const GbnnsPin pin_q(q_main, (size_t)n_q * (mode == GBNNS_MODE_LOWQ || d == d_low ? d : d_low) * sizeof(float)),
        pin_ql(q_low, (size_t)n_q * d_low * sizeof(float)), pin_e(entries.data(), entries.size() * sizeof(uint32_t)),
        pin_a(ans.data(), ans.size() * 4), pin_h(q_hops.data(), q_hops.size() * 4), pin_d(q_dc.data(), q_dc.size() * 4),
        pin_g(q_edges.data(), q_edges.size() * 4);
    const int warmup_batches = gbnnsFirstUse(ix, mode, q_main, q_low, n_q, run_ef, run_k, entries, ans, q_hops, q_dc, aux, n_entries, &q_edges);
    for (int v = 0; v < number_exper; ++v) {
        num_exp += 1;
        StopW stopw = StopW();
        gbnnsBatch(ix, mode, q_main, q_low, n_q, run_ef, run_k, entries, ans, q_hops, q_dc, aux, n_entries, &q_edges);
        work_time += stopw.getElapsedTimeMicro();
        for (int i = 0; i < n_q; ++i) {
            hops += q_hops[i];
            dist_calc += q_dc[i] + (mode == GBNNS_MODE_LOWQ ? recheck_size : 0);
            walk_dc += q_dc[i];
            edges += q_edges[i];
        }
        gbnnsScore(ans, ds, truth, d, n_q, n_tr, metric, acc);
    }
    gbnnsReport(outfile, graph_name, acc, hops, dist_calc, work_time, num_exp, n_q);
    const double per = (double)num_exp * n_q;
    const GbnnsTraffic traffic = {walk_dc / per, edges / per, d != d_low ? d_low : d, mode == GBNNS_MODE_LOWQ, warmup_batches};
    gbnnsSidecar(output_txt, graph_name, run_ef, run_k, recheck_size, acc / per, (double)hops / per, (double)dist_calc / per,
                 work_time / (num_exp * 1e6 * n_q), num_exp, n_q, n, d, d_low, traffic);
}

inline vector<vector<uint32_t>> gbnnsInterPoints(int n, int n_q, std::mt19937& random_gen, const string& graph_name) {
    // "hnsw*" graphs start from node 0, everything else from a uniformly random node (:297-307)
    vector<vector<uint32_t>> inter_points(n_q);
    const int mult = graph_name.substr(0, 4) == "hnsw" ? 0 : 1;
    uniform_int_distribution<int> uniform_distr(0, n - 1);
    for (int j = 0; j < n_q; ++j) inter_points[j].push_back(uniform_distr(random_gen) * mult);
    return inter_points;
}

void performRealTests(int n, int d, int d_low, int n_q, int n_tr, vector<int> efs, std::mt19937 random_gen,
                      vector<vector<uint32_t>>& main_graph, vector<vector<uint32_t>>& kl, vector<float>& db,
                      vector<float>& queries, vector<float>& db_low, vector<float>& queries_low,
                      vector<uint32_t>& truth, const char* output_txt, Metric* metric, string graph_name,
                      bool use_second_graph, bool llf, int number_exper, int number_of_threads) {
    const vector<vector<uint32_t>> inter_points = gbnnsInterPoints(n, n_q, random_gen, graph_name);
    const uint32_t hops_bound = 50;
    for (size_t i = 0; i < efs.size(); ++i)
        performTest(main_graph, kl, db, queries, db_low, queries_low, truth, n, d, d_low, n_q, n_tr, efs[i], 1,
                    graph_name, metric, output_txt, inter_points, use_second_graph, llf, hops_bound, 0, efs[i],
                    number_exper, number_of_threads);
}

// performSyntheticTests (search_function.h:214-287): the leftover synthetic-sphere harness (d in {3, 5, 9, 17}; no
// caller in the reference).  Same sweep tables; each point is one performTest call on a kNN graph cut to k_coeff
// neighbours.  The reference leaves every query WITHOUT an entry point (the lines that draw one are commented out,
// :219-223), which makes its getOneSearchResults read an empty heap; here the queries start at node 0 -- the only
// deviation, stated on stdout.
void performSyntheticTests(int n, int d, int n_q, int n_tr, std::mt19937 random_gen, vector<vector<uint32_t>>& knn,
                           vector<vector<uint32_t>>& kl, vector<float>& db, vector<float>& queries,
                           vector<uint32_t>& truth, const char* output_txt, Metric* metric, string graph_name,
                           bool use_second_graph, bool llf, bool beam_search) {
    (void)random_gen;
    cout << "performSyntheticTests: entry point = node 0 for every query (the reference passes none)" << endl;
    vector<vector<uint32_t>> inter_points(n_q, vector<uint32_t>(1, 0u));
    vector<int> ef_coeff, k_coeff;
    uint32_t hops_bound = 11;
    const int recheck_size = -1;
    const int knn_size = findGraphAverageDegree(knn);
    if (beam_search) k_coeff.assign(6, knn_size);
    else ef_coeff.assign(6, 1);
    vector<int>& swept = beam_search ? ef_coeff : k_coeff;
    if (d == 3) {
        swept = beam_search ? vector<int>{10, 15, 20, 25, 30} : vector<int>{12, 14, 16, 18, 20};
        hops_bound = 11;
    } else if (d == 5) {
        swept = beam_search ? vector<int>{7, 10, 15, 22, 25, 30} : vector<int>{15, 20, 25, 30, 40, 60};
        hops_bound = 7;
    } else if (d == 9) {
        swept = beam_search ? vector<int>{5, 8, 15, 25, 30, 35} : vector<int>{60, 100, 150, 200, 250, 300};
        hops_bound = 5;
    } else if (d == 17) {
        swept = beam_search ? vector<int>{10, 40, 70, 100, 130, 160} : vector<int>{750, 1000, 1250, 1500, 1750, 2000};
        hops_bound = 4;
    }
    const size_t exp_size = std::min(ef_coeff.size(), k_coeff.size());
    for (size_t i = 0; i < exp_size; ++i) {
        vector<vector<uint32_t>> knn_cur = cutKNNbyK(knn, db.data(), k_coeff[i], n, d, metric);
        performTest(knn_cur, kl, db, queries, db, queries, truth, n, d, d, n_q, n_tr, ef_coeff[i], 1, graph_name, metric,
                    output_txt, inter_points, use_second_graph, llf, hops_bound, 0, recheck_size, 1, 1);
    }
}

void performNetTest(vector<vector<uint32_t>>& knn_graph, vector<vector<uint32_t>>& kl_graph, vector<float>& ds,
                    vector<float>& queries, vector<float>& ds_low, const Net* net, size_t d_hidden,
                    vector<uint32_t>& truth, int n, int d, int d_low, int n_q, int n_tr, int ef, int k,
                    string graph_name, Metric* metric, const char* output_txt,
                    vector<vector<uint32_t>> inter_points, bool use_second_graph, bool llf, uint32_t hops_bound,
                    int dist_calc_boost, int recheck_size, int number_exper, int number_of_threads) {
    (void)number_of_threads;
    const GbnnsAux aux = {&kl_graph, use_second_graph, llf, hops_bound};
    std::ofstream outfile;
    outfile.open(output_txt, std::ios_base::app);

    // branches of the reference's loop body (:353-381)
    const bool two_stage = d != d_low && recheck_size > 0;
    const bool low_only = d != d_low && !two_stage;
    gbnns_index* ix = (d != d_low)
                          ? gbnnsIndexFor(knn_graph, ds.data(), n, d, ds_low.data(), d_low, net, d_hidden, metric)
                          : gbnnsIndexFor(knn_graph, ds.data(), n, d, nullptr, 0, nullptr, 0, metric);
    gbnns_index* ix_low = low_only ? gbnnsIndexFor(knn_graph, ds_low.data(), n, d_low, nullptr, 0, nullptr, 0, metric)
                                   : nullptr;
    uint32_t n_entries = 1;
    const vector<uint32_t> entries = gbnnsEntries(inter_points, n_q, n_entries);

    long long hops = 0;
    long long dist_calc = 0 + (long long)dist_calc_boost * n_q;
    float acc = 0;
    float work_time = 0;
    int num_exp = 0;
    vector<int32_t> q_hops(n_q), q_dc(n_q), q_edges(n_q);
    long long walk_dc = 0, edges = 0;
    vector<float> q_low;
    if (low_only) q_low.resize((size_t)n_q * d_low);
    vector<uint32_t> ans(n_q);
    const GbnnsPin pin_q(queries.data(), queries.size() * sizeof(float)), pin_ql(q_low.data(), q_low.size() * sizeof(float)),
        pin_e(entries.data(), entries.size() * sizeof(uint32_t)), pin_a(ans.data(), ans.size() * 4),
        pin_h(q_hops.data(), q_hops.size() * 4), pin_d(q_dc.data(), q_dc.size() * 4), pin_g(q_edges.data(), q_edges.size() * 4);
    int warmup_batches = 0;
    if (two_stage) {
        warmup_batches = gbnnsFirstUse(ix, GBNNS_MODE_NET, queries.data(), nullptr, n_q, recheck_size, recheck_size, entries, ans, q_hops, q_dc,
                                       aux, n_entries, &q_edges);
    } else if (low_only) {   // (the projection + a plain walk over the low-dimensional index: both handles get their first use here)
        if (gbnns_project(ix, queries.data(), n_q, q_low.data(), GBNNS_MEM_HOST, nullptr)) gbnnsDie("gbnns_project");
        warmup_batches = gbnnsFirstUse(ix_low, GBNNS_MODE_PLAIN, q_low.data(), nullptr, n_q, ef, k, entries, ans, q_hops, q_dc, aux, n_entries,
                                       &q_edges);
    } else {
        warmup_batches = gbnnsFirstUse(ix, GBNNS_MODE_PLAIN, queries.data(), nullptr, n_q, ef, k, entries, ans, q_hops, q_dc, aux, n_entries,
                                       &q_edges);
    }
    for (int v = 0; v < number_exper; ++v) {
        num_exp += 1;
        StopW stopw = StopW();
        if (two_stage) {
            gbnnsBatch(ix, GBNNS_MODE_NET, queries.data(), nullptr, n_q, recheck_size, recheck_size, entries, ans,
                       q_hops, q_dc, aux, n_entries, &q_edges);
        } else if (low_only) {
            if (gbnns_project(ix, queries.data(), n_q, q_low.data(), GBNNS_MEM_HOST, nullptr)) gbnnsDie("gbnns_project");
            gbnnsBatch(ix_low, GBNNS_MODE_PLAIN, q_low.data(), nullptr, n_q, ef, k, entries, ans, q_hops, q_dc, aux, n_entries,
                       &q_edges);
        } else {
            gbnnsBatch(ix, GBNNS_MODE_PLAIN, queries.data(), nullptr, n_q, ef, k, entries, ans, q_hops, q_dc, aux, n_entries,
                       &q_edges);
        }
        work_time += stopw.getElapsedTimeMicro();
        for (int i = 0; i < n_q; ++i) {
            hops += q_hops[i];
            dist_calc += q_dc[i] + (two_stage ? recheck_size : 0);
            walk_dc += q_dc[i];
            edges += q_edges[i];
        }
        gbnnsScore(ans, ds, truth, d, n_q, n_tr, metric, acc);
    }
    gbnnsReport(outfile, graph_name, acc, hops, dist_calc, work_time, num_exp, n_q);
    const double per = (double)num_exp * n_q;
    const GbnnsTraffic traffic = {walk_dc / per, edges / per, d != d_low ? d_low : d, two_stage, warmup_batches};
    gbnnsSidecar(output_txt, graph_name, two_stage ? recheck_size : ef, two_stage ? recheck_size : k, recheck_size,
                 acc / per, (double)hops / per, (double)dist_calc / per, work_time / (num_exp * 1e6 * n_q), num_exp, n_q, n, d,
                 d_low, traffic);
}

void performRealNetTests(int n, int d, int d_low, int n_q, int n_tr, vector<int> efs, std::mt19937 random_gen,
                         vector<vector<uint32_t>>& main_graph, vector<vector<uint32_t>>& kl, vector<float>& db,
                         vector<float>& queries, vector<float>& db_low, const Net* net, size_t d_hidden,
                         vector<uint32_t>& truth, const char* output_txt, Metric* metric, string graph_name,
                         bool use_second_graph, bool llf, int number_exper, int number_of_threads) {
    const vector<vector<uint32_t>> inter_points = gbnnsInterPoints(n, n_q, random_gen, graph_name);
    const uint32_t hops_bound = 50;
    for (size_t i = 0; i < efs.size(); ++i)
        performNetTest(main_graph, kl, db, queries, db_low, net, d_hidden, truth, n, d, d_low, n_q, n_tr, efs[i], 1,
                       graph_name, metric, output_txt, inter_points, use_second_graph, llf, hops_bound, 0, efs[i],
                       number_exper, number_of_threads);
}
