// final_test.cpp -- the reference's experiment driver (search/final_test.cpp) on the MI355X path.
//
//   ./final_test <dataset> [data_dir] [models_dir] [results_dir] [params_file]
//
// Same flow as the reference's main(): read the dataset's line block from
// parameters_of_databases.txt, load base / query / ground-truth / low-dim base vectors, the two
// graphs and the three net layers, then run the plain-graph sweep (performRealTests) and the
// two-stage sweep (performRealNetTests), 5 repeats each, appending result lines to
// <results_dir>/final_results_<dataset>.txt.  The reference hard-codes absolute paths under
// /home/shekhale and /mnt/data/shekhale (final_test.cpp:26,44-45,80); here they come from argv or
// the environment (GBNNS_DATA_DIR, GBNNS_MODELS_DIR, GBNNS_RESULTS_DIR, GBNNS_PARAMS) with the
// same file naming.  GBNNS_NUM_EXPER overrides the repeat count.
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
    size_t d_hidden = atoi(params["d_hidden"].c_str());
    if (d_hidden == 0) {
        // parameters_of_databases.txt has no d_hidden row for gist and deep (the reference reads 0 there and cannot load
        // their nets): take the width from the `second_part` tag, "..._w_<width>_e_..." (:20, :31)
        const string& tag = params["second_part"];
        const size_t at = tag.find("_w_");
        if (at != string::npos) d_hidden = atoi(tag.c_str() + at + 3);
    }
    cout << n << " " << n_q << " " << n_tr << " " << d << " " << d_low << endl;
    if (n == 0 || n_q == 0 || d == 0) {
        cout << "dataset '" << datasetName << "' not found in " << paramsPath << endl;
        return 1;
    }
    vector<int> efs = getVectorFromString(params["efs"]);
    vector<int> efs_hnsw_origin = getVectorFromString(params["efs_hnsw"]);
    const string hnsw_name = params["hnsw_name"];

    const string pathData = dataDir + "/" + datasetName;
    // <name>_base.fvecs, or <name>_base.bvecs when only the byte-vector file exists (graph_utils.h: loadVectorsAny)
    vector<float> db = loadVectorsAny(pathData + "_base", d, n);
    vector<float> queries = loadVectorsAny(pathData + "_query", d, n_q);
    vector<uint32_t> truth = loadXvecs<uint32_t>(pathData + "_groundtruth.ivecs", n_tr, n_q);
    vector<float> db_ar = loadXvecs<float>(pathData + "_base_angular_optimal.fvecs", d_low, n);

    vector<vector<uint32_t>> hnsw = loadEdges(modelsDir + "/hnsw_" + hnsw_name + ".ivecs", n, "hnsw");
    vector<vector<uint32_t>> hnsw_ar =
        loadEdges(modelsDir + "/hnsw_" + hnsw_name + "_angular_optimal.ivecs", n, "hnsw_ar");

    const string pathARNets = modelsDir + "/" + datasetName + "_net_as_matrix_angular_optimal";
    Net net;
    net.layerFirst = loadXvecs<float>(pathARNets + "_1.fvecs", d + 1, d_hidden);
    net.layerSecond = loadXvecs<float>(pathARNets + "_2.fvecs", d_hidden + 1, d_hidden);
    net.layerFinal = loadXvecs<float>(pathARNets + "_3.fvecs", d_hidden + 1, d_low);

    int numberExper = 5;
    if (const char* e = getenv("GBNNS_NUM_EXPER")) numberExper = atoi(e);
    const int numberThreads = 1;

    const string output_s = resultsDir + "/final_results_" + datasetName + ".txt";
    const char* output = output_s.c_str();
    remove(output);
    remove((string(output) + ".json").c_str());  // the machine-readable sidecar is rewritten with the result file

    L2Metric l2 = L2Metric();
    std::mt19937 random_gen;
    std::random_device device;
    random_gen.seed(device());

    performRealTests(n, d, d, n_q, n_tr, efs_hnsw_origin, random_gen, hnsw, hnsw, db, queries, db, queries, truth,
                     output, &l2, "hnsw", false, false, numberExper, numberThreads);
    performRealNetTests(n, d, d_low, n_q, n_tr, efs, random_gen, hnsw_ar, hnsw_ar, db, queries, db_ar, &net,
                        d_hidden, truth, output, &l2, "hnsw_new_ar", false, false, numberExper, numberThreads);
    return 0;
}
