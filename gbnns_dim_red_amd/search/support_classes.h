// support_classes.h -- StopW, the stopwatch the reference harness times its query loop with
// (search/support_classes.h:9-24), and KLgraph, the random "long link" auxiliary graphs naive_test.cpp walks
// with use_second_graph = true (:27-175; built at naive_test.cpp:78-88 when the files are missing).
#pragma once

#include <chrono>
#include <set>
#include <unordered_set>

#include "support_func.h"

class StopW {
    std::chrono::steady_clock::time_point begin_;

public:
    StopW() : begin_(std::chrono::steady_clock::now()) {}
    // whole microseconds, returned as float like the reference (:16-19)
    float getElapsedTimeMicro() {
        const auto now = std::chrono::steady_clock::now();
        return (float)std::chrono::duration_cast<std::chrono::microseconds>(now - begin_).count();
    }
    void reset() { begin_ = std::chrono::steady_clock::now(); }
};

// Long-link graphs: every node gets L links drawn by RANK among a set of candidates sorted by distance, rank r
// with probability proportional to 1 / (r + 1).  Host code (the builders are sequences of draws from one
// std::mt19937 and std::set / std::unordered_set operations; the distance work is sqrt(N) per node).  The
// reference runs the node loop under `#pragma omp parallel for` with the generator shared between the threads,
// so its output is only defined for one thread -- that single-thread sequence is what these loops reproduce:
// same distributions, same draw order, same containers, hence (same libstdc++) the same graph for a given seed.
class KLgraph {
public:
    int L;
    vector<vector<uint32_t>> longmatrixNN;
    void BuildByNumber(int l, vector<float> dataset, size_t N, size_t d, std::mt19937 random_gen, Metric* metric);
    void BuildByNumberCustom(int l, vector<float> dataset, size_t N, size_t d, size_t sqrtN, std::mt19937 random_gen,
                             Metric* metric);
    void BuildByDist(int l, vector<float> dataset, size_t N, size_t d, std::mt19937 random_gen, Metric* metric);

private:
    // L distinct ranks from `ranks`, then the ids at those ranks in the set's iteration order (:76-82, :125-132)
    void drawLinks(size_t i, std::discrete_distribution<int>& ranks, const vector<Neighbor>& sorted,
                   std::mt19937& random_gen) {
        std::unordered_set<int> picked;
        while ((int)picked.size() < L) picked.insert(ranks(random_gen));
        for (int r : picked) longmatrixNN[i].push_back(sorted[r].number);
    }
};

// candidates = every other node, exact ranks (:38-85)
void KLgraph::BuildByNumber(int l, vector<float> dataset, size_t N, size_t d, std::mt19937 random_gen, Metric* metric) {
    L = l;
    for (size_t i = 0; i < N; ++i) longmatrixNN.push_back(vector<uint32_t>());
    vector<float> weight;
    for (size_t r = 0; r + 1 < N; ++r) weight.push_back(1. / (r + 1));
    std::discrete_distribution<int> ranks(weight.begin(), weight.end());
    for (size_t i = 0; i < N; ++i) {
        const float* pi = dataset.data() + i * d;
        vector<Neighbor> all;
        for (size_t j = 0; j < N; ++j)
            if (j != i) all.push_back(Neighbor{(uint32_t)j, metric->Dist(pi, dataset.data() + j * d, d)});
        std::sort(all.begin(), all.end());
        drawLinks(i, ranks, all, random_gen);
    }
}

// candidates = sqrtN nodes drawn uniformly (distinct distances: the set orders by distance alone), ranks among
// them (:88-134) -- what naive_test.cpp:80-87 builds with sqrtN = pow(n, 0.5)
void KLgraph::BuildByNumberCustom(int l, vector<float> dataset, size_t N, size_t d, size_t sqrtN, std::mt19937 random_gen,
                                  Metric* metric) {
    cout << sqrtN << ' ' << N << endl;
    L = l;
    for (size_t i = 0; i < N; ++i) longmatrixNN.push_back(vector<uint32_t>());
    vector<float> weight;
    for (size_t r = 0; r < sqrtN; ++r) weight.push_back(1. / (r + 1));
    std::discrete_distribution<int> ranks(weight.begin(), weight.end());
    std::uniform_int_distribution<int> any_node(0, (int)N - 1);
    for (size_t i = 0; i < N; ++i) {
        const float* pi = dataset.data() + i * d;
        std::set<Neighbor> sample;
        while (sample.size() < sqrtN) {
            const int num = any_node(random_gen);
            if (num != (int)i) sample.insert(Neighbor{(uint32_t)num, metric->Dist(pi, dataset.data() + (size_t)num * d, d)});
        }
        vector<Neighbor> sorted(sample.begin(), sample.end());
        std::sort(sorted.begin(), sorted.end());
        drawLinks(i, ranks, sorted, random_gen);
    }
}

// candidates = every node farther than 0.03, drawn with probability proportional to dist^-d (:137-175)
void KLgraph::BuildByDist(int l, vector<float> dataset, size_t N, size_t d, std::mt19937 random_gen, Metric* metric) {
    L = l;
    const float thr = 0.03;
    for (size_t i = 0; i < N; ++i) longmatrixNN.push_back(vector<uint32_t>());
    for (size_t i = 0; i < N; ++i) {
        const float* pi = dataset.data() + i * d;
        vector<Neighbor> far;
        for (size_t j = 0; j < N; ++j) {
            if (j == i) continue;
            const float dist = metric->Dist(pi, dataset.data() + j * d, d);
            if (dist > thr) far.push_back(Neighbor{(uint32_t)j, dist});
        }
        vector<float> weight;
        for (const Neighbor& nb : far) weight.push_back(pow(pow(nb.dist, -1), d));
        std::discrete_distribution<int> pick(weight.begin(), weight.end());
        drawLinks(i, pick, far, random_gen);
    }
}
