// gbnns_oracle.cpp -- CPU restatement ("oracle") of the reference two-stage graph search.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product path: only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
// only as the checker / reported CPU baseline -- never as the thing measured or shipped.
//
// Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
// tests/golden/*.npz, which were captured from the *compiled reference* (oracle/_ref, built from
// /root/reference/search/*.h by oracle/Makefile with strict-IEEE flags) by
// tests/golden/make_golden.py.
//
// Arithmetic spec (DESIGN.md "Arithmetic contract"): IEEE-754 binary32, source order of the
// reference, one rounding per operation -- no FMA contraction, no re-association, correctly
// rounded sqrt and divide, subnormals kept.  Build with -O2 -ffp-contract=off (see Makefile).
// Everything is written with scalar code and explicit accumulators, no intrinsics.
//
// Each function cites the reference lines (relative to /root/reference/search/) it restates.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <queue>
#include <utility>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

enum Metric : int { kL2 = 0, kNegDot = 1 };

// support_func.h:107-128  L2Metric::Dist.
// Four running sums (the four SSE lanes); lane j owns dims j, j+4, j+8, ...; only the first
// 4*floor(d/4) dims take part; the horizontal sum is left-associated: ((s0+s1)+s2)+s3.
inline float l2_dist(const float* a, const float* b, size_t d) {
    const size_t steps = d >> 2;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (size_t t = 0; t < steps; ++t) {
        const float e0 = a[4 * t + 0] - b[4 * t + 0];
        const float e1 = a[4 * t + 1] - b[4 * t + 1];
        const float e2 = a[4 * t + 2] - b[4 * t + 2];
        const float e3 = a[4 * t + 3] - b[4 * t + 3];
        s0 = s0 + e0 * e0;
        s1 = s1 + e1 * e1;
        s2 = s2 + e2 * e2;
        s3 = s3 + e3 * e3;
    }
    return ((s0 + s1) + s2) + s3;
}

// support_func.h:131-163  Angular::Dist  (+ masked_read :70-84).
// Eight running sums over floor(d/8) steps; fold hi half onto lo half (m[j] = acc[j+4]+acc[j]);
// one optional 4-wide step; one optional masked step (missing lanes contribute 0*0 = +0, which is
// still *added*); then hadd,hadd = (m0+m1)+(m2+m3); result negated.
inline float negdot_dist(const float* x, const float* y, size_t d) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    while (d >= 8) {
        for (int l = 0; l < 8; ++l) acc[l] = acc[l] + x[l] * y[l];
        x += 8;
        y += 8;
        d -= 8;
    }
    float m[4];
    for (int j = 0; j < 4; ++j) m[j] = acc[j + 4] + acc[j];
    if (d >= 4) {
        for (int j = 0; j < 4; ++j) m[j] = m[j] + x[j] * y[j];
        x += 4;
        y += 4;
        d -= 4;
    }
    if (d > 0) {
        for (int j = 0; j < 4; ++j) {
            const float xv = (size_t)j < d ? x[j] : 0.f;
            const float yv = (size_t)j < d ? y[j] : 0.f;
            m[j] = m[j] + xv * yv;
        }
    }
    const float lo = m[0] + m[1];
    const float hi = m[2] + m[3];
    return -(lo + hi);
}

inline float metric_dist(int metric, const float* a, const float* b, size_t d) {
    return metric == kNegDot ? negdot_dist(a, b, d) : l2_dist(a, b, d);
}

// support_func.h:624-633  computeNetLayer.  `out` starts at zero (fresh std::vector in the
// caller, :648/:651, and the caller-provided `ans` in performNetTest :354); per neuron:
//   out -= Angular.Dist(row, in)   (i.e. 0 - (-(dot)))
//   out += bias                    (bias = last element of the row, step = d_in+1)
//   if (activation && out < 0) out = 0
inline void net_layer(const float* layer, const float* in, float* out, bool relu, int d_in,
                      int d_out) {
    const size_t step = (size_t)d_in + 1;
    for (int i = 0; i < d_out; ++i) {
        const float* row = layer + (size_t)i * step;
        float v = 0.f;
        v = v - negdot_dist(row, in, (size_t)d_in);
        v = v + row[step - 1];
        if (relu && v < 0.f) v = 0.f;
        out[i] = v;
    }
}

// support_func.h:636-642 normalizeVector + :645-658 GetLowQueryFromNet.
// norm = sqrt(L2Metric.Dist(y, zeros, d_low))  -> ignores the d_low % 4 tail dims; then every
// one of the d_low outputs is divided by it.
inline void project_one(const float* l1, const float* l2, const float* l3, const float* q,
                        float* out, int d, int dh, int dlow, float* h1, float* h2,
                        const float* zeros) {
    net_layer(l1, q, h1, true, d, dh);
    net_layer(l2, h1, h2, true, dh, dh);
    net_layer(l3, h2, out, false, dh, dlow);
    float norm = l2_dist(out, zeros, (size_t)dlow);
    norm = std::sqrt(norm);
    for (int i = 0; i < dlow; ++i) out[i] = out[i] / norm;
}

typedef std::pair<float, int> Entry;  // search_function.h:50,55  priority_queue<pair<float,int>>

// Exact visited set.  The reference uses an epoch-stamped uint16 array per thread
// (visited_list_pool.h:8-31); any exact set gives identical behaviour, this one is u32-stamped.
struct Visited {
    std::vector<uint32_t> stamp;
    uint32_t cur = 0;
    void start(size_t n) {
        if (stamp.size() != n) {
            stamp.assign(n, 0);
            cur = 0;
        }
        if (++cur == 0) {
            std::fill(stamp.begin(), stamp.end(), 0u);
            cur = 1;
        }
    }
    bool test_and_set(uint32_t id) {
        if (stamp[id] == cur) return true;
        stamp[id] = cur;
        return false;
    }
};

struct WalkOut {
    std::priority_queue<Entry> top;  // max-heap on (dist, id)
    int hops = 0;
    int dist_calc = 0;
};

// search_function.h:43-102 getOneSearchResults + :15-40 makeStep, with use_second_graph=false
// (the only mode final_test.cpp uses, :85,:88).  Graph is CSR (order of each list preserved).
//  - dist_calc starts at 1 (:52) although one entry distance is computed per entry point;
//  - per entry point: fresh candidate heap and fresh visited epoch, shared result heap (:54-64);
//  - loop: closest candidate (max of (-dist, id): ties -> LARGEST id); stop if its distance is
//    strictly greater than the current worst result (:66-67); expand all neighbours in stored
//    order (:23-39): skip visited, else mark, distance, dist_calc++, insert when
//    worst.dist > dist || size < ef (strict, distance only), evict the largest pair if size > ef;
//  - afterwards trim to k (:96-98).
//
// Auxiliary graph (use_second_graph = true, :73-80; naive_test.cpp:103-105 passes the KL graph with
// llf = true): while num_hops < hops_bound the node's auxiliary list is expanded first (same makeStep);
// `found` (makeStep's flag, :34) says whether that step inserted anything; the main list is expanded
// unless llf && found (:82).  aux_off == NULL restates use_second_graph = false.
struct AuxGraph {
    const uint64_t* off = nullptr;
    const uint32_t* nbr = nullptr;
    bool llf = false;
    uint32_t hops_bound = 50;
};

void walk_one(const float* q, const float* db, size_t n, int d, const uint64_t* off,
              const uint32_t* nbr, int ef, int k, const uint32_t* entries, int n_entries,
              int metric, Visited& vis, WalkOut& out, const AuxGraph& aux = AuxGraph()) {
    out.top = std::priority_queue<Entry>();
    out.dist_calc = 1;
    out.hops = 0;
    for (int e = 0; e < n_entries; ++e) {
        std::priority_queue<Entry> cand;  // keyed (-dist, id)
        const uint32_t ep = entries[e];
        const float d0 = metric_dist(metric, q, db + (size_t)ep * d, (size_t)d);
        out.top.emplace(d0, (int)ep);
        cand.emplace(-d0, (int)ep);
        vis.start(n);
        vis.test_and_set(ep);
        while (!cand.empty()) {
            const Entry c = cand.top();
            if (-c.first > out.top.top().first) break;
            cand.pop();
            const uint32_t node = (uint32_t)c.second;
            bool found = false;
            auto make_step = [&](const uint64_t* o, const uint32_t* nb) {  // :15-40
                for (uint64_t j = o[node]; j < o[node + 1]; ++j) {
                    const uint32_t v = nb[j];
                    if (vis.test_and_set(v)) continue;
                    const float dv = metric_dist(metric, q, db + (size_t)v * d, (size_t)d);
                    out.dist_calc++;
                    if (out.top.top().first > dv || (int)out.top.size() < ef) {
                        cand.emplace(-dv, (int)v);
                        found = true;
                        out.top.emplace(dv, (int)v);
                        if ((int)out.top.size() > ef) out.top.pop();
                    }
                }
            };
            if (aux.off && (uint32_t)out.hops < aux.hops_bound) make_step(aux.off, aux.nbr);  // :73-80
            if (!(found && aux.llf) || !aux.off) make_step(off, nbr);                         // :82-89
            out.hops++;
        }
    }
    while ((int)out.top.size() > k) out.top.pop();
}

// search_function.h:105-125 getRealNearest: pops worst->best, keeps the strict minimum of the
// exact distance Dist(db_row, q) -- on ties the entry popped EARLIER wins.
uint32_t rerank_ids(const float* q, int d, const uint32_t* ids, int count, const float* db,
                    int metric) {
    uint32_t best = ids[0];
    float best_d = metric_dist(metric, db + (size_t)ids[0] * d, q, (size_t)d);
    for (int i = 1; i < count; ++i) {
        const float di = metric_dist(metric, db + (size_t)ids[i] * d, q, (size_t)d);
        if (di < best_d) {
            best_d = di;
            best = ids[i];
        }
    }
    return best;
}

}  // namespace

extern "C" {

float gbo_l2(const float* a, const float* b, uint64_t d) { return l2_dist(a, b, (size_t)d); }
float gbo_negdot(const float* a, const float* b, uint64_t d) {
    return negdot_dist(a, b, (size_t)d);
}

// Batched GetLowQueryFromNet: q [nq x d] -> out [nq x d_low].
void gbo_project(const float* l1, const float* l2, const float* l3, const float* q, float* out,
                 uint64_t nq, int d, int dh, int dlow, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        std::vector<float> h1(dh), h2(dh), zeros(dlow, 0.f);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < (int64_t)nq; ++i)
            project_one(l1, l2, l3, q + (size_t)i * d, out + (size_t)i * dlow, d, dh, dlow,
                        h1.data(), h2.data(), zeros.data());
    }
}

// Batched getOneSearchResults.  Per query i: entry points entries[i*n_entries ..] (NULL -> 0).
// out_ids/out_dists (optional, [nq x k_out], k_out = min(k, ef)) receive the trimmed result heap
// in POP order (worst -> best), padded with 0xFFFFFFFF / +inf; out_count its size.
// aux_off != NULL: the auxiliary-graph walk (see walk_one).
void gbo_walk_aux(const float* q, uint64_t nq, const float* db, uint64_t n, int d,
                  const uint64_t* off, const uint32_t* nbr, int ef, int k, const uint32_t* entries,
                  int n_entries, int metric, uint32_t* out_ids, float* out_dists, int32_t* out_count,
                  int32_t* out_hops, int32_t* out_dist_calc, int threads, const uint64_t* aux_off,
                  const uint32_t* aux_nbr, int llf, uint32_t hops_bound) {
    AuxGraph aux;
    aux.off = aux_off; aux.nbr = aux_nbr; aux.llf = llf != 0; aux.hops_bound = hops_bound;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    const int stride = k < ef ? k : ef;
#pragma omp parallel
    {
        Visited vis;
        WalkOut w;
        const uint32_t zero = 0;
#pragma omp for schedule(dynamic, 16)
        for (int64_t i = 0; i < (int64_t)nq; ++i) {
            const uint32_t* ep = entries ? entries + (size_t)i * n_entries : &zero;
            walk_one(q + (size_t)i * d, db, (size_t)n, d, off, nbr, ef, k, ep,
                     entries ? n_entries : 1, metric, vis, w, aux);
            if (out_hops) out_hops[i] = w.hops;
            if (out_dist_calc) out_dist_calc[i] = w.dist_calc;
            int c = 0;
            while (!w.top.empty()) {
                if (out_ids) out_ids[(size_t)i * stride + c] = (uint32_t)w.top.top().second;
                if (out_dists) out_dists[(size_t)i * stride + c] = w.top.top().first;
                w.top.pop();
                ++c;
            }
            if (out_count) out_count[i] = c;
            for (int r = c; r < stride; ++r) {
                if (out_ids) out_ids[(size_t)i * stride + r] = 0xFFFFFFFFu;
                if (out_dists) out_dists[(size_t)i * stride + r] = INFINITY;
            }
        }
    }
}

void gbo_walk(const float* q, uint64_t nq, const float* db, uint64_t n, int d,
              const uint64_t* off, const uint32_t* nbr, int ef, int k, const uint32_t* entries,
              int n_entries, int metric, uint32_t* out_ids, float* out_dists, int32_t* out_count,
              int32_t* out_hops, int32_t* out_dist_calc, int threads) {
    gbo_walk_aux(q, nq, db, n, d, off, nbr, ef, k, entries, n_entries, metric, out_ids, out_dists,
                 out_count, out_hops, out_dist_calc, threads, nullptr, nullptr, 0, 50);
}

// Batched getRealNearest over candidate lists in pop order ([nq x stride], count[i] valid).
void gbo_rerank(const float* q, uint64_t nq, int d, const uint32_t* cand, int stride,
                const int32_t* count, const float* db, int metric, uint32_t* out_ids,
                int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < (int64_t)nq; ++i)
        out_ids[i] = rerank_ids(q + (size_t)i * d, d, cand + (size_t)i * stride,
                                count ? count[i] : stride, db, metric);
}

// The timed per-query body of performNetTest (search_function.h:348-385) / performTest
// (:153-186), batched.  mode 0: two-stage with the MLP (d != d_low, recheck = ef > 0);
// mode 1: two-stage with precomputed low-dim queries `q_low` (performTest :159-164);
// mode 2: plain walk in the space of `db` with (ef, k) and ans = top of the heap after trimming
//         to k (:174-181; also :165-172).
// hops / dist_calc are per query; for modes 0/1 dist_calc[i] includes the "+ recheck_size" the
// harness adds (:362 / :164).  threads = 1 reproduces final_test.cpp:71.
void gbo_search_batch_aux(int mode, const float* queries, const float* q_low_in, uint64_t nq,
                          const float* db, const float* db_low, uint64_t n, int d, int dlow, int dh,
                          const float* l1, const float* l2, const float* l3, const uint64_t* off,
                          const uint32_t* nbr, int ef, int k, const uint32_t* entries, int metric,
                          uint32_t* out_ids, int32_t* out_hops, int32_t* out_dist_calc,
                          int threads, const uint64_t* aux_off, const uint32_t* aux_nbr, int llf,
                          uint32_t hops_bound) {
    AuxGraph aux;
    aux.off = aux_off; aux.nbr = aux_nbr; aux.llf = llf != 0; aux.hops_bound = hops_bound;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        Visited vis;
        WalkOut w;
        std::vector<float> h1(dh > 0 ? dh : 1), h2(dh > 0 ? dh : 1), ql(dlow > 0 ? dlow : 1),
            zeros(dlow > 0 ? dlow : 1, 0.f);
        std::vector<uint32_t> ids;
        const uint32_t zero = 0;
#pragma omp for schedule(dynamic, 16)
        for (int64_t i = 0; i < (int64_t)nq; ++i) {
            const uint32_t* ep = entries ? entries + i : &zero;
            const float* qi = queries + (size_t)i * d;
            if (mode == 2) {
                walk_one(qi, db, (size_t)n, d, off, nbr, ef, k, ep, 1, metric, vis, w, aux);
                out_ids[i] = (uint32_t)w.top.top().second;
                if (out_hops) out_hops[i] = w.hops;
                if (out_dist_calc) out_dist_calc[i] = w.dist_calc;
                continue;
            }
            const float* qlow;
            if (mode == 0) {
                project_one(l1, l2, l3, qi, ql.data(), d, dh, dlow, h1.data(), h2.data(),
                            zeros.data());
                qlow = ql.data();
            } else {
                qlow = q_low_in + (size_t)i * dlow;
            }
            walk_one(qlow, db_low, (size_t)n, dlow, off, nbr, ef, ef, ep, 1, metric, vis, w, aux);
            ids.clear();
            while (!w.top.empty()) {
                ids.push_back((uint32_t)w.top.top().second);
                w.top.pop();
            }
            out_ids[i] = rerank_ids(qi, d, ids.data(), (int)ids.size(), db, metric);
            if (out_hops) out_hops[i] = w.hops;
            if (out_dist_calc) out_dist_calc[i] = w.dist_calc + ef;
        }
    }
}

void gbo_search_batch(int mode, const float* queries, const float* q_low_in, uint64_t nq,
                      const float* db, const float* db_low, uint64_t n, int d, int dlow, int dh,
                      const float* l1, const float* l2, const float* l3, const uint64_t* off,
                      const uint32_t* nbr, int ef, int k, const uint32_t* entries, int metric,
                      uint32_t* out_ids, int32_t* out_hops, int32_t* out_dist_calc,
                      int threads) {
    gbo_search_batch_aux(mode, queries, q_low_in, nq, db, db_low, n, d, dlow, dh, l1, l2, l3, off, nbr,
                         ef, k, entries, metric, out_ids, out_hops, out_dist_calc, threads, nullptr,
                         nullptr, 0, 50);
}

// support_func.h:521-575 hnswlikeGD (need_const_degree=false) + :402-445 addReverseEdgesForGD.
// Input kNN lists in CSR; output adjacency returned through two calls: first with out_nbr=NULL to
// get per-node degrees into out_deg, then with out_nbr sized sum(deg).  Implementation keeps the
// result in a static so the second call only copies.
static std::vector<std::vector<uint32_t>> g_gd;

static const float kEps = 1e-10f;  // support_func.h:41-43 getEps()

struct Nb {
    uint32_t id;
    float dist;
};

uint64_t gbo_hnswlike_gd(const uint64_t* koff, const uint32_t* knbr, const float* ds, int M,
                         uint64_t n, int d, int metric, int reverse, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    g_gd.assign(n, std::vector<uint32_t>());
    const int edge = M / 2;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        std::vector<Nb> nb;
        const float* pi = ds + (size_t)i * d;
        for (uint64_t j = koff[i]; j < koff[i + 1]; ++j) {
            const float di = metric_dist(metric, pi, ds + (size_t)knbr[j] * d, (size_t)d);
            if (di > kEps) nb.push_back(Nb{knbr[j], di});
        }
        // std::sort with operator< on dist only (:63-66, :540).  Not stable: the order of
        // equal-distance candidates is whatever libstdc++'s introsort yields for this sequence;
        // the same call on the same 8-byte {id, dist} records reproduces it.
        std::sort(nb.begin(), nb.end(), [](const Nb& a, const Nb& b) { return a.dist < b.dist; });
        std::vector<uint32_t>& g = g_gd[i];
        if (nb.empty()) continue;  // reference would read nb[0] out of bounds (:541)
        g.push_back(nb[0].id);
        for (size_t j = 1; j < nb.size(); ++j) {
            const float* pp = ds + (size_t)nb[j].id * d;
            bool good = true;
            for (size_t l = 0; l < g.size(); ++l) {
                const float* pa = ds + (size_t)g[l] * d;
                if (metric_dist(metric, pp, pi, (size_t)d) + kEps >
                    metric_dist(metric, pp, pa, (size_t)d)) {
                    good = false;
                    break;
                }
            }
            if (good) g.push_back(nb[j].id);
            if ((int)g.size() == M) break;
        }
        for (int j = 0; j < edge && j < (int)nb.size(); ++j)
            if (std::find(g.begin(), g.end(), nb[j].id) == g.end()) g.push_back(nb[j].id);
    }
    if (reverse) {
        // :417-442 -- serial, order dependent.
        std::vector<uint32_t> rev_count(n, 0);
        for (uint64_t i = 0; i < n; ++i)
            for (uint32_t v : g_gd[i]) rev_count[v]++;
        for (uint64_t i = 0; i < n; ++i) {
            const int upper = M - (int)rev_count[i];
            int thr = std::min(upper, M / 2);
            if (thr > 0) {
                for (size_t j = 0; j < g_gd[i].size(); ++j) {
                    const uint32_t c = g_gd[i][j];
                    if ((int)g_gd[c].size() < 2 * M) {
                        if (std::find(g_gd[c].begin(), g_gd[c].end(), (uint32_t)i) ==
                            g_gd[c].end()) {
                            g_gd[c].push_back((uint32_t)i);
                            if (--thr <= 0) break;
                        }
                    }
                }
            }
        }
    }
    uint64_t total = 0;
    for (auto& g : g_gd) total += g.size();
    return total;
}

void gbo_hnswlike_gd_fetch(uint64_t* out_off, uint32_t* out_nbr) {
    uint64_t p = 0;
    for (size_t i = 0; i < g_gd.size(); ++i) {
        out_off[i] = p;
        for (uint32_t v : g_gd[i]) out_nbr[p++] = v;
    }
    out_off[g_gd.size()] = p;
    g_gd.clear();
    g_gd.shrink_to_fit();
}

// Exact brute-force nearest neighbours: getTruth (support_func.h:270-290) generalised to k results.
// getTruth keeps the strict minimum of Dist(base_j, query) over ascending j, i.e. the smallest
// (distance, id) pair; here the k smallest pairs in ascending pair order.  self_offset >= 0: query i is
// base row i + self_offset and is skipped.  Missing results (k > rows) are 0xFFFFFFFF / +inf.
void gbo_exact_knn(const float* base, uint64_t n, const float* queries, uint64_t nq, int d, int k,
                   int metric, int64_t self_offset, uint32_t* out_ids, float* out_dist, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        std::vector<std::pair<float, uint32_t>> all;
#pragma omp for schedule(dynamic, 4)
        for (int64_t i = 0; i < (int64_t)nq; ++i) {
            all.clear();
            for (uint64_t j = 0; j < n; ++j) {
                if (self_offset >= 0 && j == (uint64_t)i + (uint64_t)self_offset) continue;
                float dv = metric_dist(metric, base + (size_t)j * d, queries + (size_t)i * d, (size_t)d);
                dv = dv + 0.0f;  // -0 and +0 compare equal in getTruth's `<`; canonical form for the pair order
                all.emplace_back(dv, (uint32_t)j);
            }
            const size_t kk = std::min<size_t>((size_t)k, all.size());
            std::partial_sort(all.begin(), all.begin() + kk, all.end());
            for (int e = 0; e < k; ++e) {
                const bool have = (size_t)e < kk;
                out_ids[(size_t)i * k + e] = have ? all[e].second : 0xFFFFFFFFu;
                if (out_dist) out_dist[(size_t)i * k + e] = have ? all[e].first : INFINITY;
            }
        }
    }
}

int gbo_max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
