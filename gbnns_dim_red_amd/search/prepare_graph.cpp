// prepare_graph.cpp -- the reference's graph-preparation driver (search/prepare_graph.cpp):
// loads the low-dimensional base set and its kNN lists, prunes them with hnswlikeGD(M = 30,
// reverse edges) and writes the adjacency the search walks (prepare_graph.cpp:64-74).
//
//   ./prepare_graph <dataset> <lat_name> [data_dir] [models_dir] [params_file]
//
// Files (reference naming): <data_dir>/<dataset>_base_<lat_name>.fvecs,
// <models_dir>/<dataset>_knn_1k_<lat_name>.ivecs  ->  <models_dir>/<dataset>_gd_knn_<lat_name>.ivecs
// GBNNS_GD_M overrides M.  The builder is host code, as in the reference.  When the kNN file does not exist it
// is produced first, on the device: exact K-NN lists of the low-dim base set over itself (gbnns_exact_knn, the
// reference's distance arithmetic; K = 1000 as the file name says, GBNNS_KNN_K overrides) -- the step the
// reference leaves to dim_red/support_func.py:374-384 (faiss).
#include "search_function.h"

static string pickPath(int argc, char** argv, int pos, const char* env, const string& fallback) {
    if (argc > pos) return argv[pos];
    const char* e = getenv(env);
    return e ? string(e) : fallback;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        cout << " Need to specify parameters" << endl;
        return 1;
    }
    const string datasetName = argv[1];
    const string fileLatName = argv[2];
    const string dataDir = pickPath(argc, argv, 3, "GBNNS_DATA_DIR", "data/" + datasetName);
    const string modelsDir = pickPath(argc, argv, 4, "GBNNS_MODELS_DIR", "models/" + datasetName);
    const string paramsPath = pickPath(argc, argv, 5, "GBNNS_PARAMS", "parameters_of_databases.txt");
    cout << datasetName << endl;

    std::map<string, string> params = readSearchParams(paramsPath, datasetName);
    const size_t n = atoi(params["n"].c_str());
    const size_t d_low = atoi(params["d_low"].c_str());
    if (n == 0 || d_low == 0) {
        cout << "dataset '" << datasetName << "' not found in " << paramsPath << endl;
        return 1;
    }
    cout << n << " " << d_low << endl;

    L2Metric l2 = L2Metric();
    std::vector<float> db_low = loadXvecs<float>(dataDir + "/" + datasetName + "_base_" + fileLatName + ".fvecs", d_low, n);
    const string knnPath = modelsDir + "/" + datasetName + "_knn_1k_" + fileLatName + ".ivecs";
    if (!std::ifstream(knnPath).good()) {
        int K = 1000;
        if (const char* e = getenv("GBNNS_KNN_K")) K = atoi(e);
        if ((size_t)K > n - 1) K = (int)n - 1;
        if (K < 1) {
            cout << "kNN lists need at least two base vectors" << endl;
            return 1;
        }
        cout << "kNN file missing: building exact " << K << "-NN lists on the device" << endl;
        const char* dev = getenv("GBNNS_DEVICE");
        vector<vector<uint32_t>> knn(n);
        const size_t slice = 32768;  // queries per call: bounds the k x slice workspace and the id buffer
        vector<uint32_t> ids;
        for (size_t s0 = 0; s0 < n; s0 += slice) {
            const size_t cnt = std::min(slice, n - s0);
            ids.resize(cnt * K);
            if (gbnns_exact_knn(dev ? atoi(dev) : 0, db_low.data(), n, db_low.data() + s0 * d_low, cnt, (uint32_t)d_low, K,
                                GBNNS_METRIC_L2, (int64_t)s0, ids.data(), nullptr, GBNNS_MEM_HOST, nullptr))
                gbnnsDie("gbnns_exact_knn");
            for (size_t i = 0; i < cnt; ++i) knn[s0 + i].assign(ids.begin() + i * K, ids.begin() + (i + 1) * K);
        }
        writeEdges(knnPath, knn);
    }
    vector<vector<uint32_t>> knn_low =
        loadEdges(modelsDir + "/" + datasetName + "_knn_1k_" + fileLatName + ".ivecs", n, "knn_low");

    int M = 30;
    if (const char* e = getenv("GBNNS_GD_M")) M = atoi(e);
    // prepare_graph.cpp:70 passes need_const_degree = false; GBNNS_CONST_DEGREE=1 turns the builder's last argument on
    // (lists padded to 2M from the kNN candidates, support_func.h:466-485)
    const bool const_degree = getenv("GBNNS_CONST_DEGREE") && atoi(getenv("GBNNS_CONST_DEGREE"));
    vector<vector<uint32_t>> gd_knn_low = hnswlikeGD(knn_low, db_low.data(), M, n, d_low, &l2, true, const_degree);
    cout << "GD_knn " << findGraphAverageDegree(gd_knn_low) << endl;
    writeEdges(modelsDir + "/" + datasetName + "_gd_knn_" + fileLatName + ".ivecs", gd_knn_low);
    return 0;
}
