// dropin_units.cpp -- test driver for the host-side helpers of the C++ drop-in (gbnns_dim_red_amd/search/
// graph_utils.h, support_classes.h): reads a command and its inputs in the reference's file formats, calls the
// drop-in's function of the reference's name, writes the result as an edge file.  Built and run by
// tests/test_dropin_units.py; no GPU is involved in any of these commands.
//   dropin_units constdeg <knn.ivecs> <ds.fvecs> <n> <d> <M> <reverse> <out.ivecs>     hnswlikeGD(.., need_const_degree = true)
//   dropin_units cutk     <knn.ivecs> <ds.fvecs> <n> <d> <k> <out.ivecs>
//   dropin_units cutthr   <knn.ivecs> <ds.fvecs> <n> <d> <thr> <out.ivecs>
//   dropin_units merge    <a.ivecs> <b.ivecs> <n> <out.ivecs>
//   dropin_units fill     <a.ivecs> <b.ivecs> <n> <degree> <out.ivecs>
//   dropin_units kl       <which> <l> <ds.fvecs> <n> <d> <sqrtN> <seed> <out.ivecs>
//   dropin_units uniform  <n> <d> <seed> <out.fvecs>
//   dropin_units bvecs    <prefix> <n> <d> <out.fvecs>                                  loadVectorsAny (fvecs / bvecs / mmap)
#include "../../gbnns_dim_red_amd/search/search_function.h"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const string cmd = argv[1];
    L2Metric l2;
    if (cmd == "constdeg" && argc == 9) {
        const size_t n = atoi(argv[4]), d = atoi(argv[5]);
        vector<vector<uint32_t>> knn = loadEdges(argv[2], n, "knn");
        vector<float> ds = loadXvecs<float>(argv[3], d, n);
        writeEdges(argv[8], hnswlikeGD(knn, ds.data(), atoi(argv[6]), n, d, &l2, atoi(argv[7]) != 0, true));
    } else if (cmd == "cutk" && argc == 8) {
        const size_t n = atoi(argv[4]), d = atoi(argv[5]);
        vector<vector<uint32_t>> knn = loadEdges(argv[2], n, "knn");
        vector<float> ds = loadXvecs<float>(argv[3], d, n);
        writeEdges(argv[7], cutKNNbyK(knn, ds.data(), atoi(argv[6]), (int)n, (int)d, &l2));
    } else if (cmd == "cutthr" && argc == 8) {
        const size_t n = atoi(argv[4]), d = atoi(argv[5]);
        vector<vector<uint32_t>> knn = loadEdges(argv[2], n, "knn");
        vector<float> ds = loadXvecs<float>(argv[3], d, n);
        writeEdges(argv[7], cutKNNbyThreshold(knn, ds, (float)atof(argv[6]), (int)n, (int)d, &l2));
    } else if (cmd == "merge" && argc == 6) {
        const size_t n = atoi(argv[4]);
        vector<vector<uint32_t>> a = loadEdges(argv[2], n, "a"), b = loadEdges(argv[3], n, "b");
        writeEdges(argv[5], mergeGraph(a, b));
    } else if (cmd == "fill" && argc == 7) {
        const size_t n = atoi(argv[4]);
        vector<vector<uint32_t>> a = loadEdges(argv[2], n, "a"), b = loadEdges(argv[3], n, "b");
        writeEdges(argv[6], fillGraphToConstantDegree(a, b, atoi(argv[5])));
    } else if (cmd == "kl" && argc == 10) {
        const size_t n = atoi(argv[5]), d = atoi(argv[6]);
        vector<float> ds = loadXvecs<float>(argv[4], d, n);
        std::mt19937 gen((unsigned)strtoul(argv[8], nullptr, 10));
        KLgraph kl;
        const int which = atoi(argv[2]), l = atoi(argv[3]);
        if (which == 0) kl.BuildByNumber(l, ds, n, d, gen, &l2);
        else if (which == 1) kl.BuildByNumberCustom(l, ds, n, d, (size_t)atoi(argv[7]), gen, &l2);
        else kl.BuildByDist(l, ds, n, d, gen, &l2);
        writeEdges(argv[9], kl.longmatrixNN);
    } else if (cmd == "uniform" && argc == 6) {
        const int n = atoi(argv[2]), d = atoi(argv[3]);
        std::mt19937 gen((unsigned)strtoul(argv[4], nullptr, 10));
        vector<float> v = createUniformData(n, d, gen);
        std::ofstream out(argv[5], std::ios::binary);
        writeXvec<float>(out, v.data(), d, n);
    } else if (cmd == "bvecs" && argc == 6) {
        const size_t n = atoi(argv[3]), d = atoi(argv[4]);
        vector<float> v = loadVectorsAny(argv[2], d, n);
        std::ofstream out(argv[5], std::ios::binary);
        writeXvec<float>(out, v.data(), d, n);
    } else {
        std::cerr << "bad command line" << std::endl;
        return 2;
    }
    return 0;
}
