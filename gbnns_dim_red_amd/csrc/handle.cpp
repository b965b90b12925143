// handle.cpp -- host side of libgbnns_hip.so, the C ABI declared in include/gbnns.h: the index handle (api_internal.h lists the other units).
//
// Owns device memory (index data in HBM, growable per-index workspaces: "lanes"), converts the reference's
// host-side data structures to the device layouts, and sequences the kernels of one batch call on
// one HIP stream -- the caller's, or with GBNNS_FLAG_DEFER_JOIN a lane's own, several batches in flight:
//     [MLP layer x3 + normalise] -> walk (first pass, fused re-rank) [-> retry pass] -> walk (general kernel,
//     hand-over list) [-> re-rank]
// There is no CPU fallback anywhere in this file: if HIP is unusable every entry point fails.

#include "api_internal.h"

using namespace gbnns;
using namespace gbnns_api;

namespace gbnns_api {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

thread_local SlowLog g_slow;

// Host -> device copy of a caller's (pageable) array through the library's own page-locked staging buffer, in pieces.
// A plain hipMemcpy from pageable memory lets the runtime page-lock the caller's pages on the fly and remember the
// mapping; a caller that frees such an array and gets the same addresses back from its allocator for another one (numpy,
// std::vector) then has the next copy fault on the stale mapping now and then (round 4: "an illegal memory access" inside
// the first upload of gbnns_index_create, about one full test-suite run in three; never in a short run).  One-time
// uploads of an index do not miss the extra pass over host memory.
int h2d_staged(void* dst, const void* src, size_t bytes) {
    static std::mutex mu;
    static void* stage = nullptr;
    constexpr size_t kPiece = 8u << 20;
    std::lock_guard<std::mutex> lk(mu);
    if (!stage) HIP_TRY(hipHostMalloc(&stage, 2 * kPiece, hipHostMallocDefault));
    hipEvent_t ev[2] = {nullptr, nullptr};
    HIP_TRY(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    int rc = GBNNS_OK;
    size_t done = 0;
    for (int i = 0; done < bytes && rc == GBNNS_OK; ++i, done += kPiece) {
        const size_t nb = std::min(kPiece, bytes - done);
        char* half = static_cast<char*>(stage) + (size_t)(i & 1) * kPiece;
        hipError_t e = i >= 2 ? hipEventSynchronize(ev[i & 1]) : hipSuccess;  // the piece that used this half has left it
        if (e == hipSuccess) {
            std::memcpy(half, static_cast<const char*>(src) + done, nb);
            e = hipMemcpyAsync(static_cast<char*>(dst) + done, half, nb, hipMemcpyHostToDevice, nullptr);
        }
        if (e == hipSuccess) e = hipEventRecord(ev[i & 1], nullptr);
        if (e != hipSuccess) rc = fail(GBNNS_ERR_HIP, "staged upload: %s", hipGetErrorString(e));
    }
    if (rc == GBNNS_OK && hipStreamSynchronize(nullptr) != hipSuccess) rc = fail(GBNNS_ERR_HIP, "staged upload: %s", hipGetErrorString(hipGetLastError()));
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
    return rc;
}

int upload(DevBuf& dst, const void* src, size_t rows, size_t row_floats, size_t pad_floats,
           int mem_kind) {
    // copies a [rows x row_floats] f32 matrix into a zero-padded [rows x pad_floats] device matrix
    const size_t bytes = rows * pad_floats * sizeof(float);
    int rc = dst.ensure(bytes ? bytes : 4);
    if (rc) return rc;
    if (mem_kind == GBNNS_MEM_DEVICE) {
        if (pad_floats == row_floats) {
            HIP_TRY(hipMemcpy(dst.p, src, bytes, hipMemcpyDeviceToDevice));
        } else {
            HIP_TRY(hipMemset(dst.p, 0, bytes));
            HIP_TRY(hipMemcpy2D(dst.p, pad_floats * 4, src, row_floats * 4, row_floats * 4, rows, hipMemcpyDeviceToDevice));
        }
        return GBNNS_OK;
    }
    if (pad_floats == row_floats) return h2d_staged(dst.p, src, bytes);
    // padded rows: the rows are packed into a temporary device matrix first, then spread on the device
    DevBuf packed;
    if ((rc = packed.ensure(rows * row_floats * 4 ? rows * row_floats * 4 : 4))) return rc;
    rc = h2d_staged(packed.p, src, rows * row_floats * 4);
    hipError_t e = rc ? hipSuccess : hipMemset(dst.p, 0, bytes);
    if (!rc && e == hipSuccess) e = hipMemcpy2D(dst.p, pad_floats * 4, packed.p, row_floats * 4, row_floats * 4, rows, hipMemcpyDeviceToDevice);
    if (!rc && e == hipSuccess) e = hipDeviceSynchronize();
    packed.release();
    if (!rc && e != hipSuccess) rc = fail(GBNNS_ERR_HIP, "padded upload: %s", hipGetErrorString(e));
    return rc;
}

// [dout x (din+1)] rows = [W | b]  ->  W [dout x wstride] (zero padded) followed by bias [dout]
void repack_layer(const float* layer, uint32_t din, uint32_t dout, uint32_t wstride,
                  std::vector<float>& out) {
    const size_t base = out.size();
    out.resize(base + (size_t)dout * wstride + round_up(dout, 4), 0.f);  // keeps the next layer 16-B aligned
    float* w = out.data() + base;
    float* b = w + (size_t)dout * wstride;
    for (uint32_t o = 0; o < dout; ++o) {
        const float* row = layer + (size_t)o * (din + 1);
        std::memcpy(w + (size_t)o * wstride, row, (size_t)din * sizeof(float));
        b[o] = row[din];
    }
}

// CSR (host) -> padded adjacency.  Order inside each list is preserved.  A neighbour id that
// repeats inside one list is dropped after its first occurrence: the reference would find it
// already visited (search_function.h:25), so this changes nothing observable.
int build_ell(const uint64_t* off, const uint32_t* nbr, uint64_t n, std::vector<uint32_t>& ell,
              uint32_t& stride) {
    uint64_t maxdeg = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (off[i + 1] < off[i]) return fail(GBNNS_ERR_INVALID, "graph_offsets not monotone at %llu",
                                             (unsigned long long)i);
        maxdeg = std::max<uint64_t>(maxdeg, off[i + 1] - off[i]);
    }
    if (maxdeg > (1u << 20)) return fail(GBNNS_ERR_UNSUPPORTED, "max degree %llu too large",
                                         (unsigned long long)maxdeg);
    stride = round_up((uint32_t)std::max<uint64_t>(maxdeg, 1), 16);
    if ((double)n * stride * 4.0 > 200e9)
        return fail(GBNNS_ERR_UNSUPPORTED, "padded adjacency would need %.1f GB", n * stride * 4e-9);
    ell.assign((size_t)n * stride, kInvalidId);
    std::vector<uint32_t> tmp;
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t* src = nbr + off[i];
        const uint32_t deg = (uint32_t)(off[i + 1] - off[i]);
        uint32_t* dst = ell.data() + (size_t)i * stride;
        bool dup = false;
        for (uint32_t j = 0; j < deg; ++j) {
            if (src[j] >= n) return fail(GBNNS_ERR_INVALID, "node %llu: neighbour id %u >= n",
                                         (unsigned long long)i, src[j]);
        }
        if (deg > 1) {
            tmp.assign(src, src + deg);
            std::sort(tmp.begin(), tmp.end());
            dup = std::adjacent_find(tmp.begin(), tmp.end()) != tmp.end();
        }
        if (!dup) {
            std::memcpy(dst, src, (size_t)deg * 4);
        } else {
            uint32_t m = 0;
            for (uint32_t j = 0; j < deg; ++j) {
                bool seen = false;
                for (uint32_t l = 0; l < m && !seen; ++l) seen = dst[l] == src[j];
                if (!seen) dst[m++] = src[j];
            }
        }
    }
    return GBNNS_OK;
}

// Diagnostic knobs.  The environment gives the process-wide defaults (read once when the library loads), gbnns_debug_knob changes
// them; a handle copies the defaults when it is created and from then on has its own (gbnns_index_knob) -- round 6: until then they were
// fourteen process-global atomics and a test that flipped one changed every handle of the process.  Results never depend on them.
//   "quotient"      0 = never the quotient form of the visited set (GBNNS_QUOTIENT)
//   "vs_disp"       probe number at which a probe sequence of that form gives up, 1 .. 15 (GBNNS_DEBUG_VS_DISP; 15 = the product's)
//   "max_waves"     most first-pass wavefronts per CU the LDS shares are cut for (GBNNS_MAX_WAVES; 0 = the per-kernel defaults)
//   "spec_min_nq"   smallest batch whose ef <= 64 first pass requests the rows before the visited test (walk_hot_spec_kernel;
//                   GBNNS_SPEC_MIN_NQ; 0 = never)
//   "spec_any_form" 1: ... whatever the form of the visited set (tests; default: only tables NOT in the quotient form)
//   "mlp_small"     smallest batch IN FLIGHT whose hidden projection layers run on the small-footprint kernel (0 = never)
//   "mlp_net"       0 = never the one-launch projection (mlp_net.hip), 1 = for the shapes and batch sizes it serves (GBNNS_MLP_NET)
//   "mlp_slab"      0 = never the slab kernel for single layers (mlp_net.hip, mlp_slab_kernel), 1 = where a layer is one round of it
//   "late_rows"     generic two-list kernels over 192- / 256- / 576-byte rows -- -1 = by shape and residency (search_core.cpp), 0 = rows
//                   requested before the visited test, 1 = after it (GBNNS_LATE_ROWS)
//   "vs_fill2"      a visited-set fill (longest walk seen / capacity, per cent) for the sizing rule to aim at in the hand-laid-out kernels
//                   over two-pass adjacency rows; 0 = the one-pass rule (default).  History (second half of round 5): those kernels fell
//                   off a cliff as their tables filled (GD(M = 30) graph, ef 180: first-pass kernel 2.33 ms at a fill of 0.86, 1.63 at
//                   0.72, 11.6 at 0.95 -- tools/fill_scan.sh) and a target of 0.72 bought 17 - 30 %; the cause was that they handed a
//                   query over whenever a probe sequence ran out instead of using the stash (walk_hot.hip) -- with the stash there too the
//                   one-pass rule is the better one again (ef 140 / 160 / 180 in flight 1.03 / 1.26 / 1.43 ms against 1.10 / 1.31 / 1.56
//                   at 0.72; before either change 1.21 / 1.42 / 2.16).  Kept as an A/B knob.
//   "spec_tail"     largest partial last round of a lone launch, in percent of the device's wavefront slots, whose wavefronts request
//                   their rows before the visited test (0 = off)
//   "coop"          the two-wavefront walk for small batches (walk_coop.hip): -1 = where it serves and the batch leaves the room
//                   (default), 0 = never, 1 / 2 = wherever the shape allows, with two / three wavefronts per query (tests, A/B runs;
//                   the three-wavefront form measured slower; GBNNS_COOP)
//   "coop_pack"     1 = launch the three-wavefront form with its own LDS only (default 0: a launch of at most c workgroups per CU asks for
//                   1 / c of the CU's LDS each, so that the dispatcher spreads them evenly; GBNNS_COOP_PACK, A/B runs)
// Process-wide only (gbnns_exact_knn has no handle):
//   "knn_chunk"      most rows per filtered chunk (a multiple of 64)
//   "knn_pool_min_k" shortest list gbnns_exact_knn keeps as an unordered pool (one wavefront per query and chunk) instead of a heap
//                    (measured on 10^6 x 32: k = 48 0.327 against 0.333 s, k = 100 0.425 against 0.536 s, k = 1 000 2.4 against 7.6 s)
//   "knn_filter"     0 = gbnns_exact_knn without the matrix-core filter (GBNNS_KNN_FILTER; tests compare the two paths)
int knob_env(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
bool knob_set(Knobs& k, const char* name, int value) {
    if (!std::strcmp(name, "quotient")) k.quotient = value != 0;
    else if (!std::strcmp(name, "vs_disp")) k.vs_disp = value <= 0 ? 15 : std::min(15, value);
    else if (!std::strcmp(name, "max_waves")) k.max_waves = std::max(0, std::min(32, value));
    else if (!std::strcmp(name, "spec_any_form")) k.spec_any_form = value != 0;
    else if (!std::strcmp(name, "spec_min_nq")) k.spec_min_nq = std::max(0, value);
    else if (!std::strcmp(name, "mlp_small")) k.mlp_small = std::max(0, value);
    else if (!std::strcmp(name, "mlp_net")) k.mlp_net = value != 0;
    else if (!std::strcmp(name, "mlp_slab")) k.mlp_slab = std::max(0, std::min(2, value));
    else if (!std::strcmp(name, "late_rows")) k.late_rows = std::max(-1, std::min(1, value));
    else if (!std::strcmp(name, "vs_fill2")) k.vs_fill2 = std::max(0, std::min(95, value));
    else if (!std::strcmp(name, "spec_tail")) k.spec_tail = std::max(0, std::min(100, value));
    else if (!std::strcmp(name, "coop")) k.coop = std::max(-1, std::min(2, value));
    else if (!std::strcmp(name, "coop_pack")) k.coop_pack = value != 0;
    else return false;
    return true;
}
static std::mutex g_knob_mu;
static Knobs& knob_default_ref() {  // (function-local: initialised on first use, whatever the order of the static initialisers)
    static Knobs d = [] {
        Knobs k{};
        knob_set(k, "quotient", knob_env("GBNNS_QUOTIENT", 1));
        knob_set(k, "vs_disp", knob_env("GBNNS_DEBUG_VS_DISP", 15));
        knob_set(k, "max_waves", knob_env("GBNNS_MAX_WAVES", 0));
        knob_set(k, "spec_min_nq", knob_env("GBNNS_SPEC_MIN_NQ", 32768));
        knob_set(k, "spec_any_form", knob_env("GBNNS_SPEC_ANY_FORM", 0));
        knob_set(k, "mlp_small", knob_env("GBNNS_MLP_SMALL", 4096));
        knob_set(k, "mlp_net", knob_env("GBNNS_MLP_NET", 1));
        knob_set(k, "mlp_slab", knob_env("GBNNS_MLP_SLAB", 1));
        knob_set(k, "late_rows", knob_env("GBNNS_LATE_ROWS", -1));
        knob_set(k, "vs_fill2", knob_env("GBNNS_VS_FILL2", 0));
        knob_set(k, "spec_tail", knob_env("GBNNS_SPEC_TAIL", 50));
        knob_set(k, "coop", knob_env("GBNNS_COOP", -1));
        knob_set(k, "coop_pack", knob_env("GBNNS_COOP_PACK", 0));
        return k;
    }();
    return d;
}
Knobs knob_defaults() {
    std::lock_guard<std::mutex> g(g_knob_mu);
    return knob_default_ref();
}
std::atomic<int> g_knob_knn_chunk{std::max(64, knob_env("GBNNS_KNN_CHUNK", 1 << 15) & ~63)};  // (a multiple of 64, never 0: the chunk loops step by it)
std::atomic<int> g_knob_knn_pool_min_k{std::max(1, knob_env("GBNNS_KNN_POOL_MIN_K", 64))};
std::atomic<int> g_knob_knn_filter{knob_env("GBNNS_KNN_FILTER", 1)};

// (defined here, beside the defaults, so that gbnns_debug_knob below can reach them under the lock)
static bool knob_default_set(const char* name, int value) {
    std::lock_guard<std::mutex> g(g_knob_mu);
    return knob_set(knob_default_ref(), name, value);
}

}  // namespace gbnns_api

extern "C" {

int gbnns_version(void) { return GBNNS_VERSION; }

const char* gbnns_last_error(void) { return g_err.c_str(); }

int gbnns_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

void gbnns_free(void* p) { std::free(p); }

int gbnns_index_create(const gbnns_index_desc* desc, gbnns_index** out) {
    if (!desc || !out) return fail(GBNNS_ERR_INVALID, "null argument");
    *out = nullptr;
    if (desc->struct_size != sizeof(gbnns_index_desc))
        return fail(GBNNS_ERR_INVALID, "gbnns_index_desc.struct_size mismatch (%u != %zu)",
                    desc->struct_size, sizeof(gbnns_index_desc));
    if (desc->n == 0 || desc->n >= (1ull << 31)) return fail(GBNNS_ERR_INVALID, "n must be in [1, 2^31)");
    if (desc->d == 0 || !desc->db) return fail(GBNNS_ERR_INVALID, "db / d missing");
    if (!desc->graph_offsets || !desc->graph_nbrs) return fail(GBNNS_ERR_INVALID, "graph missing");
    if (desc->metric != GBNNS_METRIC_L2 && desc->metric != GBNNS_METRIC_NEG_DOT)
        return fail(GBNNS_ERR_INVALID, "unknown metric %d", desc->metric);
    if (desc->mem_kind != GBNNS_MEM_HOST && desc->mem_kind != GBNNS_MEM_DEVICE)
        return fail(GBNNS_ERR_INVALID, "unknown mem_kind %d", desc->mem_kind);
    if ((desc->d_low != 0) != (desc->db_low != nullptr))
        return fail(GBNNS_ERR_INVALID, "d_low and db_low must be given together");
    const bool has_net = desc->net_l1 || desc->net_l2 || desc->net_l3;
    if (has_net && !(desc->net_l1 && desc->net_l2 && desc->net_l3 && desc->d_hidden && desc->d_low))
        return fail(GBNNS_ERR_INVALID, "net needs all three layers, d_hidden and d_low");

    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(GBNNS_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (desc->device < 0 || desc->device >= count)
        return fail(GBNNS_ERR_NO_DEVICE, "device %d out of range (%d devices)", desc->device, count);
    HIP_TRY(hipSetDevice(desc->device));

    gbnns_index* ix = new (std::nothrow) gbnns_index;
    if (!ix) return fail(GBNNS_ERR_OOM, "host allocation failed");
    ix->device = desc->device;
    ix->metric = desc->metric;
    ix->n = desc->n;
    ix->d = desc->d;
    ix->d_low = desc->d_low;
    ix->d_hidden = desc->d_hidden;
    (void)hipDeviceGetAttribute(&ix->cus, hipDeviceAttributeMultiprocessorCount, ix->device);
    ix->knob = knob_defaults();
    ix->d_pad = round_up(desc->d, 4);
    ix->dl_pad = round_up(desc->d_low, 4);
    int rc = GBNNS_OK;

    auto take = [&](const float* src, uint32_t dim, uint32_t pad, DevBuf& own, const float*& dst) -> int {
        if (desc->mem_kind == GBNNS_MEM_DEVICE && pad == dim) {
            dst = src;  // borrowed: rows already 16-B aligned
            return GBNNS_OK;
        }
        int r = upload(own, src, ix->n, dim, pad, desc->mem_kind);
        dst = own.as<float>();
        return r;
    };
    rc = take(desc->db, ix->d, ix->d_pad, ix->db_own, ix->db);
    if (!rc && desc->db_low) rc = take(desc->db_low, ix->d_low, ix->dl_pad, ix->db_low_own, ix->db_low);

    if (!rc) {
        std::vector<uint32_t> ell;
        rc = build_ell(desc->graph_offsets, desc->graph_nbrs, ix->n, ell, ix->ell_stride);
        if (!rc) rc = ix->ell.ensure(ell.size() * 4);
        if (!rc) {
            rc = h2d_staged(ix->ell.p, ell.data(), ell.size() * 4);
        }
    }

    if (!rc && has_net) {
        const uint32_t d = ix->d, dh = ix->d_hidden, dl = ix->d_low;
        std::vector<float> l1, l2, l3;
        const float *p1 = desc->net_l1, *p2 = desc->net_l2, *p3 = desc->net_l3;
        if (desc->mem_kind == GBNNS_MEM_DEVICE) {
            l1.resize((size_t)dh * (d + 1));
            l2.resize((size_t)dh * (dh + 1));
            l3.resize((size_t)dl * (dh + 1));
            hipError_t e = hipMemcpy(l1.data(), p1, l1.size() * 4, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(l2.data(), p2, l2.size() * 4, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(l3.data(), p3, l3.size() * 4, hipMemcpyDeviceToHost);
            if (e != hipSuccess) rc = fail(GBNNS_ERR_HIP, "net download: %s", hipGetErrorString(e));
            p1 = l1.data();
            p2 = l2.data();
            p3 = l3.data();
        }
        if (!rc) {
            // (rows padded with zeros to 16 floats: the one-launch projection reads whole 16-input blocks)
            ix->ws1 = round_up(d, 16);
            ix->ws2 = round_up(dh, 16);
            ix->ws3 = round_up(dh, 16);

            std::vector<float> packed;
            repack_layer(p1, d, dh, ix->ws1, packed);
            const size_t o2 = packed.size();
            repack_layer(p2, dh, dh, ix->ws2, packed);
            const size_t o3 = packed.size();
            repack_layer(p3, dh, dl, ix->ws3, packed);
            rc = ix->net.ensure(packed.size() * 4);
            if (!rc) {
                hipError_t e = hipMemcpy(ix->net.p, packed.data(), packed.size() * 4, hipMemcpyHostToDevice);
                if (e != hipSuccess) rc = fail(GBNNS_ERR_HIP, "net upload: %s", hipGetErrorString(e));
            }
            float* base = ix->net.as<float>();
            ix->w1 = base;
            ix->b1 = ix->w1 + (size_t)dh * ix->ws1;
            ix->w2 = base + o2;
            ix->b2 = ix->w2 + (size_t)dh * ix->ws2;
            ix->w3 = base + o3;
            ix->b3 = ix->w3 + (size_t)dl * ix->ws3;
            ix->has_net = true;
        }
    }
    if (!rc) rc = ix->lanes[0].ctrl.ensure(512);
    if (!rc) {
        // (hipMemset on device memory may return before the fill has run: callers' streams may be non-blocking
        // ones that do not order themselves after the null stream, so wait for it here)
        hipError_t e = hipMemset(ix->lanes[0].ctrl.p, 0, 512);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) rc = fail(GBNNS_ERR_HIP, "ctrl init: %s", hipGetErrorString(e));
        ix->lanes[0].ctrl_ready = true;
    }
    if (rc) {
        gbnns_index_destroy(ix);
        return rc;
    }
    *out = ix;
    return GBNNS_OK;
}

int gbnns_index_set_aux_graph(gbnns_index* ix, const uint64_t* offsets, const uint32_t* nbrs) {
    if (!ix) return fail(GBNNS_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipDeviceSynchronize());  // no search may still be reading the old table
    ix->has_aux = false;
    if (!offsets && !nbrs) return GBNNS_OK;
    if (!offsets || !nbrs) return fail(GBNNS_ERR_INVALID, "auxiliary graph: offsets / nbrs missing");
    std::vector<uint32_t> ell;
    int rc = build_ell(offsets, nbrs, ix->n, ell, ix->aux_stride);
    if (rc) return rc;
    if ((rc = ix->aux_ell.ensure(ell.size() * 4))) return rc;
    if (int rc2 = h2d_staged(ix->aux_ell.p, ell.data(), ell.size() * 4)) return rc2;
    ix->has_aux = true;
    return GBNNS_OK;
}

uint32_t gbnns_index_d_low(const gbnns_index* ix) { return ix ? ix->d_low : 0u; }
uint64_t gbnns_index_n(const gbnns_index* ix) { return ix ? ix->n : 0u; }
uint32_t gbnns_index_d(const gbnns_index* ix) { return ix ? ix->d : 0u; }
int gbnns_index_device(const gbnns_index* ix) { return ix ? ix->device : -1; }

int gbnns_index_destroy(gbnns_index* ix) {
    if (!ix) return GBNNS_OK;
    (void)hipSetDevice(ix->device);
    for (auto& pc : ix->pending)
        for (auto& e : pc.ev) (void)hipEventDestroy(e);
    (void)hipDeviceSynchronize();  // lanes may still be running a call whose join was deferred
    if (ix->order_ev) (void)hipEventDestroy(ix->order_ev);
    if (ix->fork_ev) (void)hipEventDestroy(ix->fork_ev);
    for (Lane& L : ix->lanes) {
        if (L.stats_ev) (void)hipEventDestroy(L.stats_ev);
        if (L.done_ev) (void)hipEventDestroy(L.done_ev);
        if (L.prev_ev) (void)hipEventDestroy(L.prev_ev);
        if (L.h_stats) (void)hipHostFree(L.h_stats);
        if (L.stream) (void)hipStreamDestroy(L.stream);
        for (int i = 0; DevBuf* b = L.bufs(i); ++i) b->release();
    }
    DevBuf* bufs[] = {&ix->db_own, &ix->db_low_own, &ix->ell, &ix->aux_ell, &ix->net, &ix->net_mfma};
    for (DevBuf* b : bufs) b->release();
    delete ix;
    return GBNNS_OK;
}

}  // extern "C"

namespace gbnns_api {


// MLP over rows x [nx x xstride] (device) -> out [nx x dl_pad] (device); h1/h2 are scratch.
int run_project(gbnns_index* ix, Lane& L, const float* x, uint32_t xstride, uint32_t nx, float* out,
                hipStream_t s, bool in_flight, bool mfma) {
    // the whole net in one launch where it serves (round 5, mlp_net.hip: 0.048 against 0.075 ms on the SIFT shape; the
    // round-2 one-launch form -- csrc/project.hip, deleted in round 4 -- was slower than the three launches)
    if (!mfma && ix->knob.mlp_net) {
        NetLaunch n{};
        n.x = x; n.xstride = xstride; n.nq = nx; n.out = out; n.ostride = ix->dl_pad; n.cus = ix->cus;
        n.w[0] = ix->w1; n.w[1] = ix->w2; n.w[2] = ix->w3;
        n.wstride[0] = ix->ws1; n.wstride[1] = ix->ws2; n.wstride[2] = ix->ws3;
        n.bias[0] = ix->b1; n.bias[1] = ix->b2; n.bias[2] = ix->b3;
        n.din[0] = ix->d; n.din[1] = n.din[2] = ix->d_hidden;
        n.dout[0] = n.dout[1] = ix->d_hidden; n.dout[2] = ix->d_low;
        if (mlp_net_serves(n)) {
            HIP_TRY(launch_mlp_net(n, s));
            std::snprintf(ix->acc.project_kernel, sizeof(ix->acc.project_kernel), "mlp_net_kernel");
            return GBNNS_OK;
        }
    }
    if (mfma && !getenv("GBNNS_MFMA_LAYERS") && mlp_mfma_net_serves(ix->d, ix->d_hidden, ix->d_low)) {
        // the throughput option in one launch (round 6, mlp_mfma_net.hip); GBNNS_MFMA_LAYERS=1 keeps round 5's three launches (A/B)
        NetLaunch n{};
        n.x = x; n.xstride = xstride; n.nq = nx; n.out = out; n.ostride = ix->dl_pad; n.cus = ix->cus;
        n.w[0] = ix->w1; n.w[1] = ix->w2; n.w[2] = ix->w3;
        n.wstride[0] = ix->ws1; n.wstride[1] = ix->ws2; n.wstride[2] = ix->ws3;
        n.bias[0] = ix->b1; n.bias[1] = ix->b2; n.bias[2] = ix->b3;
        n.din[0] = ix->d; n.din[1] = n.din[2] = ix->d_hidden;
        n.dout[0] = n.dout[1] = ix->d_hidden; n.dout[2] = ix->d_low;
        if (!ix->net_mfma_ready) {
            int rc0 = ix->net_mfma.ensure(mlp_mfma_net_packed_floats(ix->d, ix->d_hidden, ix->d_low) * 4);
            if (rc0) return rc0;
            HIP_TRY(launch_mlp_mfma_pack(n, ix->net_mfma.as<float>(), s));   // (stream order: before this call's projection; later calls on
            HIP_TRY(hipStreamSynchronize(s));                                  //  other streams find it finished)
            ix->net_mfma_ready = true;
        }
        HIP_TRY(launch_mlp_mfma_net(n, ix->net_mfma.as<float>(), s));
        std::snprintf(ix->acc.project_kernel, sizeof(ix->acc.project_kernel), "mlp_mfma_net_kernel");
        return GBNNS_OK;
    }
    std::snprintf(ix->acc.project_kernel, sizeof(ix->acc.project_kernel), mfma ? "mlp_layer_mfma_kernel" : "mlp_layer_kernels");
    int rc = L.h1.ensure((size_t)nx * ix->d_hidden * 4);
    if (!rc) rc = L.h2.ensure((size_t)nx * ix->d_hidden * 4);
    if (rc) return rc;
    LayerParams p{};
    // batches in flight whose walks fill the machine: the small-footprint kernel for the hidden layers (mlp.hip; sift-like
    // +2.5 % at ef 64, +3.6 % at ef 36; a 1 000-query GIST batch -- one walk wavefront per SIMD, nothing to squeeze in
    // beside -- and every batch that runs alone are faster on the big-tile kernel)
    const int small_min = ix->knob.mlp_small;
    // (... up to 32 times that: a 1 M-query DEEP batch is twenty rounds of the machine on its own, its projection is not
    // waiting for room, and the big-tile kernel's 12 % matter again: 43.1 against 41.5 M queries/s)
    p.small_footprint = (in_flight && small_min > 0 && nx >= (uint32_t)small_min && (uint64_t)nx <= 32ull * (uint64_t)small_min) ? 1 : 0;
    // a layer that is one round of the machine for the slab kernel (small batches, the GIST shape's 1 000 x 960 -> 1 024 -> 1 024 -> 64:
    // 117 against 153 us, the 64-neuron last layer alone 15 against 43) takes that; everything else the per-layer kernels of mlp.hip
    bool slab_used = false;
    auto layer = [&](const LayerParams& lp) {
        if (mfma) return launch_mlp_layer_mfma(lp, s);
        // (batches in flight: only the narrow shape -- 35 KB, 256 threads -- finds room beside the other lanes' walk wavefronts; the
        // wide one's 103 KB workgroups wait for a CU to drain: GIST 2.27 against 2.58 M queries/s.  Knob value 2 = wide in flight too.)
        const int slab = ix->knob.mlp_slab;
        if (!lp.small_footprint && slab && (!in_flight || lp.dout <= 64u || slab >= 2) && mlp_slab_wins(lp, ix->cus)) {
            slab_used = true;
            return launch_mlp_slab(lp, ix->cus, s);
        }
        return launch_mlp_layer(lp, s);
    };
    p.x = x; p.xstride = xstride; p.w = ix->w1; p.wstride = ix->ws1; p.bias = ix->b1;
    p.out = L.h1.as<float>(); p.ostride = ix->d_hidden; p.nq = nx; p.din = ix->d;
    p.dout = ix->d_hidden; p.relu = 1;
    HIP_TRY(layer(p));
    p.x = L.h1.as<float>(); p.xstride = ix->d_hidden; p.w = ix->w2; p.wstride = ix->ws2;
    p.bias = ix->b2; p.out = L.h2.as<float>(); p.din = ix->d_hidden;
    HIP_TRY(layer(p));
    p.x = L.h2.as<float>(); p.w = ix->w3; p.wstride = ix->ws3; p.bias = ix->b3; p.out = out;
    p.ostride = ix->dl_pad; p.dout = ix->d_low; p.relu = 0; p.normalize = 1;
    HIP_TRY(layer(p));
    if (slab_used) std::snprintf(ix->acc.project_kernel, sizeof(ix->acc.project_kernel), "mlp_slab_kernel");
    return GBNNS_OK;
}

int prof_flush(gbnns_index* ix) {
    for (auto& pc : ix->pending) {
        HIP_TRY(hipEventSynchronize(pc.ev[4]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, pc.ev[0], pc.ev[1]));
        ix->acc.project_ms += ms;
        HIP_TRY(hipEventElapsedTime(&ms, pc.ev[1], pc.ev[2]));
        ix->acc.walk_ms += ms;
        HIP_TRY(hipEventElapsedTime(&ms, pc.ev[2], pc.ev[3]));
        ix->acc.walk_general_ms += ms;
        HIP_TRY(hipEventElapsedTime(&ms, pc.ev[3], pc.ev[4]));
        ix->acc.rerank_ms += ms;
        HIP_TRY(hipEventElapsedTime(&ms, pc.ev[0], pc.ev[4]));
        ix->acc.total_ms += ms;
        ix->acc.calls += 1;
        ix->acc.queries += pc.queries;
        for (int i = 0; i < 5; ++i) (void)hipEventDestroy(pc.ev[i]);
    }
    ix->pending.clear();
    return GBNNS_OK;
}

}  // namespace gbnns_api

extern "C" {

// Diagnostic (not in gbnns.h): copies the 32 stamp/histogram sums of a GBNNS_STAMPS build and clears them.
int gbnns_debug_read_stamps(gbnns_index* ix, unsigned long long* out32) {
    if (!ix || !out32) return fail(GBNNS_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out32, ix->lanes[0].ctrl.as<uint32_t>() + 8, 256, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(ix->lanes[0].ctrl.as<uint32_t>() + 8, 0, 256));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return GBNNS_OK;
}

// Diagnostic (not in gbnns.h): one batch merge of the register-list walk kernels on host-supplied keys.
// entries: sorted u64 keys [size]; surv: 64 keys, ~0 = no survivor in that lane; out: 64*regs keys;
// out_info: {new size, merged (0 = boundary tie, list untouched), new worst}.
int gbnns_debug_merge(int regs, const unsigned long long* entries, int size, const unsigned long long* surv, int ef,
                      unsigned long long* out, int* out_info) {
    if (!(regs == 1 || regs == 2 || regs == 4) || size < 1 || size > ef || ef > 64 * regs)
        return fail(GBNNS_ERR_INVALID, "bad debug_merge arguments");
    unsigned long long *d_e = nullptr, *d_s = nullptr, *d_o = nullptr;
    int* d_i = nullptr;
    HIP_TRY(hipMalloc(&d_e, 256 * 8));
    HIP_TRY(hipMalloc(&d_s, 64 * 8));
    HIP_TRY(hipMalloc(&d_o, 256 * 8));
    HIP_TRY(hipMalloc(&d_i, 16));
    HIP_TRY(hipMemcpy(d_e, entries, (size_t)size * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_s, surv, 64 * 8, hipMemcpyHostToDevice));
    HIP_TRY(launch_debug_merge(regs, reinterpret_cast<const uint64_t*>(d_e), size, reinterpret_cast<const uint64_t*>(d_s), ef,
                               reinterpret_cast<uint64_t*>(d_o), d_i, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d_o, (size_t)64 * regs * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_info, d_i, 12, hipMemcpyDeviceToHost));
    (void)hipFree(d_e); (void)hipFree(d_s); (void)hipFree(d_o); (void)hipFree(d_i);
    return GBNNS_OK;
}

int gbnns_debug_knob(const char* name, int value) {
    if (!name) return fail(GBNNS_ERR_INVALID, "gbnns_debug_knob: null name");
    if (knob_default_set(name, value)) return GBNNS_OK;  // a handle knob: the default of the handles created from now on
    if (!std::strcmp(name, "knn_chunk")) g_knob_knn_chunk.store(std::max(64, value & ~63), std::memory_order_relaxed);
    else if (!std::strcmp(name, "knn_pool_min_k")) g_knob_knn_pool_min_k.store(std::max(1, value), std::memory_order_relaxed);
    else if (!std::strcmp(name, "knn_filter")) g_knob_knn_filter.store(value, std::memory_order_relaxed);
    else return fail(GBNNS_ERR_INVALID, "gbnns_debug_knob: unknown knob '%s'", name);
    return GBNNS_OK;
}

int gbnns_index_knob_get(gbnns_index* ix, const char* name, int* out) {
    if (!ix || !name || !out) return fail(GBNNS_ERR_INVALID, "gbnns_index_knob_get: null argument");
    const Knobs& k = ix->knob;
    const struct { const char* n; int v; } all[] = {
        {"quotient", k.quotient}, {"vs_disp", k.vs_disp}, {"max_waves", k.max_waves}, {"spec_min_nq", k.spec_min_nq},
        {"spec_any_form", k.spec_any_form}, {"mlp_small", k.mlp_small}, {"mlp_net", k.mlp_net}, {"mlp_slab", k.mlp_slab},
        {"late_rows", k.late_rows}, {"vs_fill2", k.vs_fill2}, {"spec_tail", k.spec_tail}, {"coop", k.coop}, {"coop_pack", k.coop_pack}};
    for (const auto& e : all)
        if (!std::strcmp(name, e.n)) { *out = e.v; return GBNNS_OK; }
    return fail(GBNNS_ERR_INVALID, "gbnns_index_knob_get: unknown handle knob '%s'", name);
}

int gbnns_index_knob(gbnns_index* ix, const char* name, int value) {
    if (!ix || !name) return fail(GBNNS_ERR_INVALID, "gbnns_index_knob: null argument");
    if (!knob_set(ix->knob, name, value)) return fail(GBNNS_ERR_INVALID, "gbnns_index_knob: unknown handle knob '%s'", name);
    return GBNNS_OK;
}

int gbnns_profile_enable(gbnns_index* ix, int on) {
    if (!ix) return fail(GBNNS_ERR_INVALID, "null index");
    ix->profiling = on != 0;
    return GBNNS_OK;
}

int gbnns_profile_read(gbnns_index* ix, gbnns_profile* out, int reset) {
    if (!ix || !out) return fail(GBNNS_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ix->device));
    int rc = prof_flush(ix);
    if (rc) return rc;
    uint32_t total = 0;  // ctrl[5]: queries the general kernel has processed since the last reset
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&total, ix->lanes[0].ctrl.as<uint32_t>() + 5, 4, hipMemcpyDeviceToHost));
    ix->acc.general_queries = total;
    ix->acc.struct_size = sizeof(gbnns_profile);
    // out->struct_size on entry = the caller's sizeof(gbnns_profile): a caller built against an older header (the struct grew from 160 to
    // 192 bytes in round 5) gets the prefix it knows and nothing written past it; 0 (callers that never set it) = the 160-byte round-4 layout
    {
        const size_t theirs = out->struct_size ? out->struct_size : 160u;
        const size_t take = std::min(theirs, sizeof(gbnns_profile));
        if (take < 8) return fail(GBNNS_ERR_INVALID, "gbnns_profile.struct_size %zu too small", theirs);
        std::memcpy(out, &ix->acc, take);
        out->struct_size = (uint32_t)take;
    }
    if (reset) {
        ix->acc = gbnns_profile{};
        HIP_TRY(hipMemset(ix->lanes[0].ctrl.as<uint32_t>() + 5, 0, 4));
        HIP_TRY(hipStreamSynchronize(nullptr));  // callers' streams need not order themselves after the null stream
    }
    return GBNNS_OK;
}

int gbnns_project(gbnns_index* ix, const float* x, uint64_t n_x, float* out, int mem_kind,
                  void* stream) {
    if (!ix || !x || !out) return fail(GBNNS_ERR_INVALID, "null argument");
    if (!ix->has_net) return fail(GBNNS_ERR_INVALID, "index has no net");
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint64_t chunk = 1u << 16;
    int rc = enter_stream(ix, s);
    if (rc) return rc;
    Lane& L = ix->lanes[0];
    rc = L.q_low.ensure((size_t)std::min<uint64_t>(chunk, n_x) * ix->dl_pad * 4);
    if (rc) return rc;
    if (mem_kind == GBNNS_MEM_HOST) {
        rc = L.q_in.ensure((size_t)std::min<uint64_t>(chunk, n_x) * ix->d * 4);
        if (rc) return rc;
    }
    for (uint64_t b = 0; b < n_x; b += chunk) {
        const uint32_t m = (uint32_t)std::min<uint64_t>(chunk, n_x - b);
        const float* xin = x + b * ix->d;
        if (mem_kind == GBNNS_MEM_HOST) {
            HIP_TRY(hipMemcpyAsync(L.q_in.p, xin, (size_t)m * ix->d * 4, hipMemcpyHostToDevice, s));
            xin = L.q_in.as<float>();
        }
        float* dst = L.q_low.as<float>();
        const bool direct = mem_kind == GBNNS_MEM_DEVICE && ix->dl_pad == ix->d_low;
        if (direct) dst = out + b * ix->d_low;
        rc = run_project(ix, L, xin, ix->d, m, dst, s);
        if (rc) return rc;
        if (!direct) {
            const hipMemcpyKind kind = mem_kind == GBNNS_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
            HIP_TRY(hipMemcpy2DAsync(out + b * ix->d_low, (size_t)ix->d_low * 4, dst, (size_t)ix->dl_pad * 4,
                                     (size_t)ix->d_low * 4, m, kind, s));
        }
        if (mem_kind == GBNNS_MEM_HOST) HIP_TRY(hipStreamSynchronize(s));
    }
    return GBNNS_OK;
}

int gbnns_rerank(gbnns_index* ix, const float* queries, uint64_t n_q, const uint32_t* cand,
                 uint32_t cand_stride, const int32_t* count, uint32_t* out_ids, int mem_kind,
                 void* stream) {
    if (!ix || !queries || !cand || !out_ids) return fail(GBNNS_ERR_INVALID, "null argument");
    if (n_q == 0) return GBNNS_OK;
    if (cand_stride == 0 || n_q >= (1ull << 31)) return fail(GBNNS_ERR_INVALID, "bad sizes");
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t nq = (uint32_t)n_q;
    const bool host = mem_kind == GBNNS_MEM_HOST;
    int rc;
    if ((rc = enter_stream(ix, s))) return rc;
    Lane& L = ix->lanes[0];
    RerankParams r{};
    r.db = ix->db; r.dstride = ix->d_pad; r.dim = ix->d; r.qstride = ix->d; r.cand_stride = cand_stride;
    r.nq = nq; r.n = (uint32_t)ix->n;
    if ((rc = L.cnt.ensure((size_t)nq * 4))) return rc;
    int32_t* cnt_dev = L.cnt.as<int32_t>();
    if (host) {
        if ((rc = L.q_in.ensure((size_t)nq * ix->d * 4))) return rc;
        if ((rc = L.cand.ensure((size_t)nq * cand_stride * 4))) return rc;
        if ((rc = L.out.ensure((size_t)nq * 4))) return rc;
        for (uint64_t i = 0; i < n_q; ++i) {
            const uint32_t c = count ? (uint32_t)std::max(count[i], 0) : cand_stride;
            if (c > cand_stride) return fail(GBNNS_ERR_INVALID, "count[%llu] > stride", (unsigned long long)i);
            for (uint32_t j = 0; j < c; ++j)
                if (cand[i * cand_stride + j] >= ix->n)
                    return fail(GBNNS_ERR_INVALID, "candidate id %u >= n", cand[i * cand_stride + j]);
        }
        HIP_TRY(hipMemcpyAsync(L.q_in.p, queries, (size_t)nq * ix->d * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(L.cand.p, cand, (size_t)nq * cand_stride * 4, hipMemcpyHostToDevice, s));
        if (count) HIP_TRY(hipMemcpyAsync(cnt_dev, count, (size_t)nq * 4, hipMemcpyHostToDevice, s));
        r.q = L.q_in.as<float>(); r.cand = L.cand.as<uint32_t>(); r.out = L.out.as<uint32_t>();
    } else {
        r.q = queries; r.cand = cand; r.out = out_ids;
        if (count) cnt_dev = const_cast<int32_t*>(count);
    }
    if (!count) HIP_TRY(launch_fill_u32(reinterpret_cast<uint32_t*>(cnt_dev), cand_stride, nq, s));
    r.count = cnt_dev;
    HIP_TRY(launch_rerank(r, ix->metric, s));
    if (host) {
        HIP_TRY(hipMemcpyAsync(out_ids, r.out, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        ix->in_flight = false;
    }
    return GBNNS_OK;
}

}  // extern "C"
