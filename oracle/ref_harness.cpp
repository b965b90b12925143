// ref_harness.cpp -- thin extern "C" shim around the *compiled reference* headers.
//
// TEST INFRASTRUCTURE ONLY (see oracle/README.md).  This file contains no search logic of its
// own: it #includes /root/reference/search/search_function.h (found through -I, the reference
// sources are never copied into this repository) and exposes the reference's own functions
// through a flat C ABI so that tests/golden/make_golden.py can capture golden vectors and
// bench.py can time the reference on host cores (cpu_baseline.kind = "reference").
// Built by oracle/Makefile into oracle/_ref/libgbnns_ref.so (git-ignored) only where
// /root/reference exists.
//
// Flags (oracle/Makefile): -O2 -std=c++11 -fopenmp -mavx2 -mfma -ffp-contract=off -w, which
// compiles the reference's intrinsics to exactly the operations written in its source
// (SURVEY.md section 8c).

#include "search_function.h"  // reference: pulls support_classes.h, support_func.h, visited_list_pool.h

#include <cstdint>

namespace {

Metric* pick_metric(int metric) {
    static L2Metric l2;
    static Angular ang;
    return metric == 1 ? (Metric*)&ang : (Metric*)&l2;
}

vector<vector<uint32_t>> csr_to_lists(const uint64_t* off, const uint32_t* nbr, uint64_t n) {
    vector<vector<uint32_t>> g(n);
    for (uint64_t i = 0; i < n; ++i) g[i].assign(nbr + off[i], nbr + off[i + 1]);
    return g;
}

struct RefGraph {
    vector<vector<uint32_t>> lists;
};

// getRealNearest takes `vector<float>& ds` (search_function.h:106), so the harness must own a
// std::vector copy of the base set.  It is cached on (pointer, length) so that a caller timing
// ref_search_batch can warm it first (ref_prepare_db_cache) and not pay for the copy.
vector<float> ds_cache;
const float* ds_key = nullptr;
size_t ds_len = 0;

void ensure_ds_cache(const float* db_ptr, uint64_t n, int d) {
    if (ds_key != db_ptr || ds_len != (size_t)n * d) {
        ds_cache.assign(db_ptr, db_ptr + (size_t)n * d);
        ds_key = db_ptr;
        ds_len = (size_t)n * d;
    }
}

}  // namespace

extern "C" {

float ref_l2(const float* a, const float* b, uint64_t d) {
    L2Metric m;
    return m.Dist(a, b, d);
}

float ref_negdot(const float* a, const float* b, uint64_t d) {
    Angular m;
    return m.Dist(a, b, d);
}

void* ref_graph_create(const uint64_t* off, const uint32_t* nbr, uint64_t n) {
    RefGraph* g = new RefGraph;
    g->lists = csr_to_lists(off, nbr, n);
    return g;
}

void ref_graph_destroy(void* g) { delete (RefGraph*)g; }

// GetLowQueryFromNet (support_func.h:645) per query, exactly as performNetTest calls it (:354-355).
void ref_project(const float* l1, const float* l2, const float* l3, const float* q, float* out,
                 uint64_t nq, int d, int dh, int dlow) {
    Net net;
    net.layerFirst.assign(l1, l1 + (size_t)dh * (d + 1));
    net.layerSecond.assign(l2, l2 + (size_t)dh * (dh + 1));
    net.layerFinal.assign(l3, l3 + (size_t)dlow * (dh + 1));
    Angular ang;
    L2Metric l2m;
    vector<float> zeros(dlow);
    for (uint64_t i = 0; i < nq; ++i) {
        vector<float> low(dlow);
        GetLowQueryFromNet(&net, q + i * d, low, zeros.data(), d, dh, dh, dlow, &ang, &l2m);
        for (int j = 0; j < dlow; ++j) out[i * dlow + j] = low[j];
    }
}

// getOneSearchResults (search_function.h:43) per query; result heap dumped in pop order.
// aux_graph != NULL: use_second_graph = true with that auxiliary graph, llf and hops_bound as given.
void ref_walk_aux(const float* q, uint64_t nq, const float* db, uint64_t n, int d, void* graph,
                  int ef, int k, const uint32_t* entries, int n_entries, int metric,
                  uint32_t* out_ids, float* out_dists, int32_t* out_count, int32_t* out_hops,
                  int32_t* out_dist_calc, int threads, void* aux_graph, int llf, uint32_t hops_bound) {
    RefGraph* g = (RefGraph*)graph;
    RefGraph* ga = aux_graph ? (RefGraph*)aux_graph : g;
    const bool use2 = aux_graph != nullptr;
    VisitedListPool* pool = new VisitedListPool(1, n);
    Metric* m = pick_metric(metric);
    const int stride = k < ef ? k : ef;
    omp_set_num_threads(threads > 0 ? threads : 1);
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < (int64_t)nq; ++i) {
        vector<uint32_t> ep;
        if (entries) ep.assign(entries + i * n_entries, entries + (i + 1) * n_entries);
        else ep.push_back(0);
        TripleResult r = getOneSearchResults(q + i * d, db, n, d, g->lists, ga->lists, ef, k, ep, m,
                                             pool, use2, llf != 0, hops_bound);
        if (out_hops) out_hops[i] = r.hops;
        if (out_dist_calc) out_dist_calc[i] = r.dist_calc;
        int c = 0;
        while (!r.topk.empty()) {
            if (out_ids) out_ids[i * stride + c] = (uint32_t)r.topk.top().second;
            if (out_dists) out_dists[i * stride + c] = r.topk.top().first;
            r.topk.pop();
            ++c;
        }
        if (out_count) out_count[i] = c;
        for (int s = c; s < stride; ++s) {
            if (out_ids) out_ids[i * stride + s] = 0xFFFFFFFFu;
            if (out_dists) out_dists[i * stride + s] = INFINITY;
        }
    }
    delete pool;
}

void ref_walk(const float* q, uint64_t nq, const float* db, uint64_t n, int d, void* graph,
              int ef, int k, const uint32_t* entries, int n_entries, int metric,
              uint32_t* out_ids, float* out_dists, int32_t* out_count, int32_t* out_hops,
              int32_t* out_dist_calc, int threads) {
    ref_walk_aux(q, nq, db, n, d, graph, ef, k, entries, n_entries, metric, out_ids, out_dists,
                 out_count, out_hops, out_dist_calc, threads, nullptr, 0, 50);
}

// The per-query body of performNetTest (search_function.h:348-385) / performTest (:153-186)
// calling the reference's own functions; same modes as gbo_search_batch in gbnns_oracle.cpp.
void ref_search_batch_aux(int mode, const float* queries, const float* q_low_in, uint64_t nq,
                          const float* db_ptr, const float* db_low, uint64_t n, int d, int dlow,
                          int dh, const float* l1, const float* l2, const float* l3, void* graph,
                          int ef, int k, const uint32_t* entries, int metric, uint32_t* out_ids,
                          int32_t* out_hops, int32_t* out_dist_calc, int threads, void* aux_graph,
                          int llf, uint32_t hops_bound) {
    RefGraph* g = (RefGraph*)graph;
    RefGraph* ga = aux_graph ? (RefGraph*)aux_graph : g;
    const bool use2 = aux_graph != nullptr;
    VisitedListPool* pool = new VisitedListPool(1, n);
    Metric* m = pick_metric(metric);
    Net net;
    if (mode == 0) {
        net.layerFirst.assign(l1, l1 + (size_t)dh * (d + 1));
        net.layerSecond.assign(l2, l2 + (size_t)dh * (dh + 1));
        net.layerFinal.assign(l3, l3 + (size_t)dlow * (dh + 1));
    }
    if (mode != 2) ensure_ds_cache(db_ptr, n, d);
    Angular ang;
    L2Metric l2m;
    vector<float> zeros(dlow > 0 ? dlow : 1);
    omp_set_num_threads(threads > 0 ? threads : 1);
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < (int64_t)nq; ++i) {
        vector<uint32_t> ep(1, entries ? entries[i] : 0u);
        const float* point_q = queries + i * d;
        TripleResult r;
        if (mode == 2) {
            r = getOneSearchResults(point_q, db_ptr, n, d, g->lists, ga->lists, ef, k, ep, m, pool,
                                    use2, llf != 0, hops_bound);
            while ((int)r.topk.size() > k) r.topk.pop();
            out_ids[i] = r.topk.top().second;
            if (out_hops) out_hops[i] = r.hops;
            if (out_dist_calc) out_dist_calc[i] = r.dist_calc;
            continue;
        }
        vector<float> low(dlow);
        const float* ql;
        if (mode == 0) {
            GetLowQueryFromNet(&net, point_q, low, zeros.data(), d, dh, dh, dlow, &ang, &l2m);
            ql = low.data();
        } else {
            ql = q_low_in + i * dlow;
        }
        r = getOneSearchResults(ql, db_low, n, dlow, g->lists, ga->lists, ef, ef, ep, m, pool,
                                use2, llf != 0, hops_bound);
        out_ids[i] = getRealNearest(point_q, k, d, dlow, r.topk, ds_cache, m);
        if (out_hops) out_hops[i] = r.hops;
        if (out_dist_calc) out_dist_calc[i] = r.dist_calc + ef;
    }
    delete pool;
}

void ref_search_batch(int mode, const float* queries, const float* q_low_in, uint64_t nq,
                      const float* db_ptr, const float* db_low, uint64_t n, int d, int dlow,
                      int dh, const float* l1, const float* l2, const float* l3, void* graph,
                      int ef, int k, const uint32_t* entries, int metric, uint32_t* out_ids,
                      int32_t* out_hops, int32_t* out_dist_calc, int threads) {
    ref_search_batch_aux(mode, queries, q_low_in, nq, db_ptr, db_low, n, d, dlow, dh, l1, l2, l3, graph,
                         ef, k, entries, metric, out_ids, out_hops, out_dist_calc, threads, nullptr, 0,
                         50);
}

void ref_prepare_db_cache(const float* db_ptr, uint64_t n, int d) { ensure_ds_cache(db_ptr, n, d); }

// hnswlikeGD (support_func.h:521) on a kNN graph given in CSR; result fetched in a second call.
static vector<vector<uint32_t>> g_gd_result;

uint64_t ref_hnswlike_gd(const uint64_t* koff, const uint32_t* knbr, const float* ds, int M,
                         uint64_t n, int d, int metric, int reverse, int threads) {
    vector<vector<uint32_t>> knn = csr_to_lists(koff, knbr, n);
    omp_set_num_threads(threads > 0 ? threads : 1);
    g_gd_result = hnswlikeGD(knn, ds, M, n, d, pick_metric(metric), reverse != 0, false);
    uint64_t total = 0;
    for (auto& l : g_gd_result) total += l.size();
    return total;
}

void ref_hnswlike_gd_fetch(uint64_t* out_off, uint32_t* out_nbr) {
    uint64_t p = 0;
    for (size_t i = 0; i < g_gd_result.size(); ++i) {
        out_off[i] = p;
        for (uint32_t v : g_gd_result[i]) out_nbr[p++] = v;
    }
    out_off[g_gd_result.size()] = p;
    g_gd_result.clear();
}

// Runs the reference's own harness end to end (performRealNetTests, search_function.h:411) so
// that the result line it appends to `output_txt` can be compared with the drop-in's.
void ref_perform_real_net_tests(int n, int d, int d_low, int n_q, int n_tr, const int* efs,
                                int n_efs, void* graph, const float* db, const float* queries,
                                const float* db_low, const float* l1, const float* l2,
                                const float* l3, int d_hidden, const uint32_t* truth,
                                const char* output_txt, const char* graph_name, int number_exper,
                                int number_of_threads) {
    RefGraph* g = (RefGraph*)graph;
    vector<int> efv(efs, efs + n_efs);
    vector<float> dbv(db, db + (size_t)n * d), qv(queries, queries + (size_t)n_q * d),
        dblv(db_low, db_low + (size_t)n * d_low);
    vector<uint32_t> tv(truth, truth + (size_t)n_q * n_tr);
    Net net;
    net.layerFirst.assign(l1, l1 + (size_t)d_hidden * (d + 1));
    net.layerSecond.assign(l2, l2 + (size_t)d_hidden * (d_hidden + 1));
    net.layerFinal.assign(l3, l3 + (size_t)d_low * (d_hidden + 1));
    L2Metric l2m;
    std::mt19937 rng(1);
    performRealNetTests(n, d, d_low, n_q, n_tr, efv, rng, g->lists, g->lists, dbv, qv, dblv, &net,
                        d_hidden, tv, output_txt, &l2m, graph_name, false, false, number_exper,
                        number_of_threads);
}

// aux_graph != NULL: the reference's `kl` argument with use_second_graph = true and the given llf
// (naive_test.cpp:102-105); seed = state of the mt19937 the harness receives by value.
void ref_perform_real_tests_aux(int n, int d, int d_low, int n_q, int n_tr, const int* efs, int n_efs,
                                void* graph, const float* db, const float* queries,
                                const float* db_low, const float* queries_low, const uint32_t* truth,
                                const char* output_txt, const char* graph_name, int number_exper,
                                int number_of_threads, void* aux_graph, int llf, uint32_t seed) {
    RefGraph* g = (RefGraph*)graph;
    RefGraph* ga = aux_graph ? (RefGraph*)aux_graph : g;
    vector<int> efv(efs, efs + n_efs);
    vector<float> dbv(db, db + (size_t)n * d), qv(queries, queries + (size_t)n_q * d),
        dblv(db_low, db_low + (size_t)n * d_low),
        qlv(queries_low, queries_low + (size_t)n_q * d_low);
    vector<uint32_t> tv(truth, truth + (size_t)n_q * n_tr);
    L2Metric l2m;
    std::mt19937 rng(seed);
    performRealTests(n, d, d_low, n_q, n_tr, efv, rng, g->lists, ga->lists, dbv, qv, dblv, qlv, tv,
                     output_txt, &l2m, graph_name, aux_graph != nullptr, llf != 0, number_exper,
                     number_of_threads);
}

void ref_perform_real_tests(int n, int d, int d_low, int n_q, int n_tr, const int* efs, int n_efs,
                            void* graph, const float* db, const float* queries,
                            const float* db_low, const float* queries_low, const uint32_t* truth,
                            const char* output_txt, const char* graph_name, int number_exper,
                            int number_of_threads) {
    ref_perform_real_tests_aux(n, d, d_low, n_q, n_tr, efs, n_efs, graph, db, queries, db_low, queries_low,
                               truth, output_txt, graph_name, number_exper, number_of_threads, nullptr, 0, 1);
}

// getTruth (support_func.h:270): brute-force nearest base row of every query, the reference's own code.
void ref_get_truth(const float* base, uint64_t n, const float* queries, uint64_t nq, int d, int metric,
                   uint32_t* out, int threads) {
    vector<float> ds(base, base + (size_t)n * d), qv(queries, queries + (size_t)nq * d);
    omp_set_num_threads(threads > 0 ? threads : 1);
    vector<uint32_t> t = getTruth(ds, qv, (int)n, d, (int)nq, pick_metric(metric));
    for (uint64_t i = 0; i < nq; ++i) out[i] = t[i];
}

int ref_max_threads() { return omp_get_max_threads(); }

// One call of the reference's makeStep (search_function.h:15-40) on caller-supplied state: the heaps as (key, id)
// pairs, the ids already marked visited.  Results: both heaps in pop order, dist_calc, found, number of marked ids.
void ref_make_step(const float* db, uint64_t n, int d, const float* query, const uint32_t* nb, int n_nb, int ef,
                   const float* top_key, const uint32_t* top_id, int n_top, const float* cand_key, const uint32_t* cand_id,
                   int n_cand, const uint32_t* visited, int n_vis, int metric, int32_t* out_info, float* out_top_key,
                   uint32_t* out_top_id, float* out_cand_key, uint32_t* out_cand_id) {
    priority_queue<pair<float, int>> top, cand;
    for (int i = 0; i < n_top; ++i) top.emplace(top_key[i], (int)top_id[i]);
    for (int i = 0; i < n_cand; ++i) cand.emplace(cand_key[i], (int)cand_id[i]);
    VisitedListPool* pool = new VisitedListPool(1, n);
    VisitedList* vl = pool->getFreeVisitedList();
    for (int i = 0; i < n_vis; ++i) vl->mass[visited[i]] = vl->curV;
    vector<uint32_t> level(nb, nb + n_nb);
    int dist_calc = 0, k = 1;
    bool found = false;
    makeStep(level, query, db, top, cand, pick_metric(metric), (uint32_t)d, dist_calc, found, ef, k, vl);
    int marked = 0;
    for (uint64_t i = 0; i < n; ++i) marked += vl->mass[i] == vl->curV;
    out_info[0] = dist_calc; out_info[1] = found ? 1 : 0; out_info[2] = (int)top.size(); out_info[3] = (int)cand.size();
    out_info[4] = marked;
    for (int i = 0; !top.empty(); ++i, top.pop()) { out_top_key[i] = top.top().first; out_top_id[i] = (uint32_t)top.top().second; }
    for (int i = 0; !cand.empty(); ++i, cand.pop()) { out_cand_key[i] = cand.top().first; out_cand_id[i] = (uint32_t)cand.top().second; }
    pool->releaseVisitedList(vl);
    delete pool;
}

// ---- graph utilities around the search path (golden vectors for the drop-in's graph_utils.h / support_classes.h) ----
// Each call leaves its adjacency-list result in g_gd_result; ref_hnswlike_gd_fetch copies it out as CSR.
static uint64_t keep_lists(vector<vector<uint32_t>> lists) {
    g_gd_result = lists;
    uint64_t total = 0;
    for (auto& l : g_gd_result) total += l.size();
    return total;
}

// hnswlikeGD with need_const_degree = true (support_func.h:570-572 -> getConstantDegreeForGD :466-485)
uint64_t ref_hnswlike_gd_const(const uint64_t* koff, const uint32_t* knbr, const float* ds, int M, uint64_t n, int d,
                               int metric, int reverse) {
    vector<vector<uint32_t>> knn = csr_to_lists(koff, knbr, n);
    omp_set_num_threads(1);
    return keep_lists(hnswlikeGD(knn, ds, M, n, d, pick_metric(metric), reverse != 0, true));
}

uint64_t ref_cut_knn_by_k(const uint64_t* koff, const uint32_t* knbr, const float* ds, int knn_size, uint64_t n, int d,
                          int metric) {
    vector<vector<uint32_t>> knn = csr_to_lists(koff, knbr, n);
    omp_set_num_threads(1);
    return keep_lists(cutKNNbyK(knn, ds, knn_size, (int)n, d, pick_metric(metric)));
}

uint64_t ref_cut_knn_by_threshold(const uint64_t* koff, const uint32_t* knbr, const float* ds, float thr, uint64_t n,
                                  int d, int metric) {
    vector<vector<uint32_t>> knn = csr_to_lists(koff, knbr, n);
    vector<float> v(ds, ds + n * d);
    omp_set_num_threads(1);
    return keep_lists(cutKNNbyThreshold(knn, v, thr, (int)n, d, pick_metric(metric)));
}

uint64_t ref_merge_graph(const uint64_t* aoff, const uint32_t* anbr, const uint64_t* boff, const uint32_t* bnbr,
                         uint64_t n) {
    vector<vector<uint32_t>> a = csr_to_lists(aoff, anbr, n), b = csr_to_lists(boff, bnbr, n);
    omp_set_num_threads(1);
    return keep_lists(mergeGraph(a, b));
}

uint64_t ref_fill_const_degree(const uint64_t* aoff, const uint32_t* anbr, const uint64_t* boff, const uint32_t* bnbr,
                               uint64_t n, int degree_needed) {
    vector<vector<uint32_t>> a = csr_to_lists(aoff, anbr, n), b = csr_to_lists(boff, bnbr, n);
    omp_set_num_threads(1);
    return keep_lists(fillGraphToConstantDegree(a, b, degree_needed));
}

// KLgraph builders (support_classes.h:38-175) with a generator seeded `seed`, one thread (the reference shares the
// generator between OpenMP threads: only the single-thread sequence is defined).  which: 0 BuildByNumber,
// 1 BuildByNumberCustom (sqrtN candidates), 2 BuildByDist.
uint64_t ref_kl_build(int which, int l, const float* ds, uint64_t n, int d, uint64_t sqrtN, uint32_t seed, int metric) {
    vector<float> v(ds, ds + n * d);
    std::mt19937 gen(seed);
    omp_set_num_threads(1);
    KLgraph kl;
    if (which == 0) kl.BuildByNumber(l, v, n, d, gen, pick_metric(metric));
    else if (which == 1) kl.BuildByNumberCustom(l, v, n, d, sqrtN, gen, pick_metric(metric));
    else kl.BuildByDist(l, v, n, d, gen, pick_metric(metric));
    return keep_lists(kl.longmatrixNN);
}

void ref_create_uniform_data(int n, int d, uint32_t seed, float* out) {
    std::mt19937 gen(seed);
    vector<float> v = createUniformData(n, d, gen);
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
}

}  // extern "C"
