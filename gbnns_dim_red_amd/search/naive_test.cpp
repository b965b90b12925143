// naive_test.cpp -- the reference's second experiment driver (search/naive_test.cpp) on the MI355X path.
//
//   ./naive_test <dataset> [data_dir] [models_dir] [results_dir] [params_file]
//
// Same flow as the reference's main(): load base / query / ground-truth vectors, the low-dim ("naive")
// base and query vectors, the hnsw / knn / knn_low graphs and the two KL long-link graphs, then run the four
// sweeps of naive_test.cpp:98-105 -- "hnsw", "knn", "knn_lk" (knn + KL as auxiliary graph, llf) and
// "knn_lk_low" (the same in the low-dim space with re-rank) -- appending result lines to
// <results_dir>/naive_results_<dataset>.txt.  Paths come from argv / the environment as in final_test.cpp.
// The KL graphs are built when their files are missing (KLgraph::BuildByNumberCustom, naive_test.cpp:77-88: L =
// `kl_size` of the parameter table, candidates = sqrt(n) random nodes) and kept, exactly as the reference does.
// One difference from the reference:
//   * the low-dim query file is read with dimension d_low (the reference passes d, naive_test.cpp:59, which
//     only works when its file happens to be d wide).
// GBNNS_SEED fixes the generator of the entry points and of the KL builder (the reference seeds it from
// random_device, :22-23); GBNNS_KL_SIZE overrides kl_size;
// GBNNS_NUM_EXPER overrides the repeat count.
#include <cmath>

#include "search_function.h"

static string pickPath(int argc, char** argv, int pos, const char* env, const string& fallback) {
    if (argc > pos) return argv[pos];
    const char* e = getenv(env);
    return e ? string(e) : fallback;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        cout << " Need to specify parameters" << endl;
        return 1;
    }
    const string datasetName = argv[1];
    cout << datasetName << endl;

    const string dataDir = pickPath(argc, argv, 2, "GBNNS_DATA_DIR", "data/" + datasetName);
    const string modelsDir = pickPath(argc, argv, 3, "GBNNS_MODELS_DIR", "models/" + datasetName);
    const string resultsDir = pickPath(argc, argv, 4, "GBNNS_RESULTS_DIR", ".");
    const string paramsPath = pickPath(argc, argv, 5, "GBNNS_PARAMS", "parameters_of_databases.txt");

    std::map<string, string> params = readSearchParams(paramsPath, datasetName);
    const size_t n = atoi(params["n"].c_str());
    const size_t n_q = atoi(params["n_q"].c_str());
    const size_t n_tr = atoi(params["n_tr"].c_str());
    const size_t d = atoi(params["d"].c_str());
    const size_t d_low = atoi(params["d_low"].c_str());
    cout << n << " " << n_q << " " << n_tr << " " << d << " " << d_low << endl;
    if (n == 0 || n_q == 0 || d == 0) {
        cout << "dataset '" << datasetName << "' not found in " << paramsPath << endl;
        return 1;
    }
    vector<int> efs_hnsw_origin = getVectorFromString(params["efs_hnsw"]);
    const string hnsw_name = params["hnsw_name"];
    const string netStyle = "naive";

    const string pathData = dataDir + "/" + datasetName;
    vector<float> db = loadXvecs<float>(pathData + "_base.fvecs", d, n);
    vector<float> queries = loadXvecs<float>(pathData + "_query.fvecs", d, n_q);
    vector<uint32_t> truth = loadXvecs<uint32_t>(pathData + "_groundtruth.ivecs", n_tr, n_q);
    vector<float> db_low = loadXvecs<float>(pathData + "_base_" + netStyle + ".fvecs", d_low, n);
    vector<float> queries_low = loadXvecs<float>(pathData + "_query" + netStyle + ".fvecs", d_low, n_q);

    vector<vector<uint32_t>> hnsw = loadEdges(modelsDir + "/hnsw_" + hnsw_name + ".ivecs", n, "hnsw");
    vector<vector<uint32_t>> knn = loadEdges(modelsDir + "/" + datasetName + "knn.ivecs", n, "knn");
    vector<vector<uint32_t>> knn_low = loadEdges(modelsDir + "/" + datasetName + "knn_low.ivecs", n, "knn_low");

    const string kl_dir = modelsDir + "/" + datasetName + "_kl_sqrt_style.ivecs";
    const string kl_dir_low = modelsDir + "/" + datasetName + "_kl_llow_sqrt_style.ivecs";
    L2Metric l2 = L2Metric();
    std::mt19937 random_gen;
    if (const char* e = getenv("GBNNS_SEED")) {
        random_gen.seed((unsigned)strtoul(e, nullptr, 10));
    } else {
        std::random_device device;
        random_gen.seed(device());
    }
    // naive_test.cpp:77-88: the long-link graphs are built (and kept as files) when they are missing
    if (!checkFileExistence(kl_dir)) {
        int kl_size = atoi(params["kl_size"].c_str());  // naive_test.cpp:35 (`<dataset> kl_size <L>` in the table)
        if (const char* e = getenv("GBNNS_KL_SIZE")) kl_size = atoi(e);
        KLgraph kl_sqrt;
        kl_sqrt.BuildByNumberCustom(kl_size, db, n, d, pow(n, 0.5), random_gen, &l2);
        writeEdges(kl_dir, kl_sqrt.longmatrixNN);
        KLgraph kl_sqrt_low;
        kl_sqrt_low.BuildByNumberCustom(kl_size, db_low, n, d_low, pow(n, 0.5), random_gen, &l2);
        writeEdges(kl_dir_low, kl_sqrt_low.longmatrixNN);
    }
    vector<vector<uint32_t>> kl = loadEdges(kl_dir, n, "kl");
    vector<vector<uint32_t>> kl_low = loadEdges(kl_dir_low, n, "kl_low");

    int numberExper = 5;
    if (const char* e = getenv("GBNNS_NUM_EXPER")) numberExper = atoi(e);
    const int numberThreads = 1;

    const string output_s = resultsDir + "/naive_results_" + datasetName + ".txt";
    const char* output = output_s.c_str();
    remove(output);
    remove((string(output) + ".json").c_str());  // the machine-readable sidecar is rewritten with the result file

    performRealTests(n, d, d, n_q, n_tr, efs_hnsw_origin, random_gen, hnsw, hnsw, db, queries, db, queries, truth,
                     output, &l2, "hnsw", false, false, numberExper, numberThreads);
    performRealTests(n, d, d, n_q, n_tr, efs_hnsw_origin, random_gen, knn, knn, db, queries, db, queries, truth,
                     output, &l2, "knn", false, false, numberExper, numberThreads);
    performRealTests(n, d, d, n_q, n_tr, efs_hnsw_origin, random_gen, knn, kl, db, queries, db, queries, truth,
                     output, &l2, "knn_lk", true, true, numberExper, numberThreads);
    performRealTests(n, d, d_low, n_q, n_tr, efs_hnsw_origin, random_gen, knn_low, kl_low, db, queries, db_low,
                     queries_low, truth, output, &l2, "knn_lk_low", true, true, numberExper, numberThreads);
    return 0;
}
