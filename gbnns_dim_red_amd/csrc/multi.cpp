// multi.cpp -- query-sharded replicas on the GPUs of one node, below Python (include/gbnns.h, "multi-device").
//
// The path partitions by independent units: queries never interact (search_function.h:348, the body of the
// reference's `omp parallel for`, :152), the index is read-only.  So the index is replicated -- one gbnns_index
// per listed device -- and a batch is cut into contiguous blocks, one per replica.  Each
// replica has its own host thread and its own HIP stream.
//   host buffers   : every replica reads its rows of the caller's query array and writes its answers straight
//                    into the caller's output arrays -- there is nothing to exchange.
//   device buffers : every replica's answers are all-gathered so that each device ends up with the whole id
//                    vector: ONE ncclAllGather of uint32 ids per batch over RCCL / xGMI (4 bytes per query:
//                    latency-bound; sends are padded to the largest block).  librccl is loaded on first use
//                    (dlopen: the library is 0.5 GB and host-buffer users never need it).
// No CPU fallback: creation fails without HIP devices.

#include "../../include/gbnns.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

int mfail(int code, const char* fmt, ...);

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    bool load(std::string& why) {
        if (so) return true;
        // GBNNS_RCCL_LIB names the library instead of the usual places (also how the tests reach the failure path)
        const char* forced = getenv("GBNNS_RCCL_LIB");
        const char* usual[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        const char* first_err = nullptr;
        std::string kept;
        for (size_t i = 0; i < (forced ? 1u : 3u) && !so; ++i) {
            so = dlopen(forced ? forced : usual[i], RTLD_NOW | RTLD_LOCAL);
            if (!so && !first_err) {
                const char* e = dlerror();  // ONE call: dlerror() hands the message out once and clears it
                kept = e ? e : "librccl.so not found";
                first_err = kept.c_str();
            }
        }
        if (!so) {
            why = first_err ? kept : std::string("librccl.so not found");
            return false;
        }
        auto sym = [&](const char* n) { return dlsym(so, n); };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        GetVersion = reinterpret_cast<decltype(GetVersion)>(sym("ncclGetVersion"));
        if (!CommInitAll || !CommDestroy || !AllGather || !GroupStart || !GroupEnd) {
            why = "librccl.so lacks an expected symbol";
            dlclose(so);
            so = nullptr;
            return false;
        }
        return true;
    }
};

}  // namespace

struct gbnns_multi {
    std::vector<int> devices;
    std::vector<gbnns_index*> replicas;
    std::vector<hipStream_t> streams;
    // device-buffer path
    Rccl rccl;
    std::vector<ncclComm_t> comms;
    std::vector<uint32_t*> send, recv;  // per replica: [width] padded answers, [R x width] gathered
    size_t width = 0;
    uint32_t d = 0;
    bool rccl_one = false;  // a single replica goes through the communicator too (gbnns_multi_rccl_single_rank)
};

namespace {

thread_local std::string g_merr;

int mfail(int code, const char* fmt, ...) {
    char buf[600];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_merr = buf;
    return code;
}

// runs fn(r) for every replica on its own host thread; returns the first non-zero status (message kept)
template <typename F>
int for_each_replica(gbnns_multi* m, F&& fn) {
    const size_t R = m->replicas.size();
    std::vector<int> rc(R, 0);
    std::vector<std::string> msg(R);
    auto body = [&](size_t r) {
        rc[r] = fn(r);
        if (rc[r]) {
            const char* e = gbnns_last_error();  // thread-local in the worker: carry it over
            msg[r] = (e && *e) ? e : g_merr;
        }
    };
    if (R == 1) {
        body(0);
    } else {
        std::vector<std::thread> th;
        th.reserve(R);
        for (size_t r = 0; r < R; ++r) th.emplace_back(body, r);
        for (auto& t : th) t.join();
    }
    for (size_t r = 0; r < R; ++r)
        if (rc[r]) return mfail(rc[r], "replica %zu (device %d): %s", r, m->devices[r], msg[r].c_str());
    return GBNNS_OK;
}

}  // namespace

extern "C" {

const char* gbnns_multi_last_error(void) { return g_merr.c_str(); }

// Contiguous block of part `part` of `parts`: sizes differ by at most one, the first n_q % parts blocks are the
// longer ones (the arithmetic of gbnns_dim_red_amd/sharding.py::shard_bounds and of bench.py --config deep).
void gbnns_shard_bounds(uint64_t n_q, int32_t parts, int32_t part, uint64_t* lo, uint64_t* hi) {
    if (parts <= 0 || part < 0 || part >= parts) {
        if (lo) *lo = 0;
        if (hi) *hi = 0;
        return;
    }
    const uint64_t base = n_q / (uint64_t)parts, extra = n_q % (uint64_t)parts, p = (uint64_t)part;
    const uint64_t a = p * base + (p < extra ? p : extra);
    if (lo) *lo = a;
    if (hi) *hi = a + base + (p < extra ? 1 : 0);
}

int gbnns_multi_create(const gbnns_index_desc* desc, const int32_t* devices, int32_t n_devices, gbnns_multi** out) {
    if (!desc || !out) return mfail(GBNNS_ERR_INVALID, "null argument");
    *out = nullptr;
    if (desc->struct_size != sizeof(gbnns_index_desc)) return mfail(GBNNS_ERR_INVALID, "gbnns_index_desc.struct_size mismatch");
    if (desc->mem_kind != GBNNS_MEM_HOST)
        return mfail(GBNNS_ERR_INVALID, "gbnns_multi_create replicates HOST buffers (a device tensor lives on one device)");
    const int avail = gbnns_device_count();
    if (avail <= 0) return mfail(GBNNS_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    std::vector<int> devs;
    if (!devices || n_devices <= 0) {
        for (int i = 0; i < avail; ++i) devs.push_back(i);  // every visible device
    } else {
        for (int i = 0; i < n_devices; ++i) {
            if (devices[i] < 0 || devices[i] >= avail)
                return mfail(GBNNS_ERR_NO_DEVICE, "device %d out of range (%d devices)", devices[i], avail);
            devs.push_back(devices[i]);
        }
    }
    if (devs.size() > 64) return mfail(GBNNS_ERR_INVALID, "more than 64 replicas");
    gbnns_multi* m = new (std::nothrow) gbnns_multi;
    if (!m) return mfail(GBNNS_ERR_OOM, "host allocation failed");
    m->devices = devs;
    m->replicas.assign(devs.size(), nullptr);
    m->streams.assign(devs.size(), nullptr);
    m->d = desc->d;
    // one thread per replica: the uploads of the replicas run side by side
    const int rc = for_each_replica(m, [&](size_t r) -> int {
        gbnns_index_desc dd = *desc;
        dd.device = m->devices[r];
        int e = gbnns_index_create(&dd, &m->replicas[r]);
        if (e) return e;
        if (hipSetDevice(m->devices[r]) != hipSuccess ||
            hipStreamCreateWithFlags(&m->streams[r], hipStreamNonBlocking) != hipSuccess)
            return mfail(GBNNS_ERR_HIP, "stream creation failed");
        return GBNNS_OK;
    });
    if (rc) {
        const std::string keep = g_merr;
        gbnns_multi_destroy(m);
        g_merr = keep;
        return rc;
    }
    *out = m;
    return GBNNS_OK;
}

int gbnns_multi_destroy(gbnns_multi* m) {
    if (!m) return GBNNS_OK;
    for (size_t r = 0; r < m->replicas.size(); ++r) {
        (void)hipSetDevice(m->devices[r]);
        if (m->streams[r]) {
            (void)hipStreamSynchronize(m->streams[r]);
            (void)hipStreamDestroy(m->streams[r]);
        }
        if (r < m->comms.size() && m->comms[r] && m->rccl.CommDestroy) (void)m->rccl.CommDestroy(m->comms[r]);
        if (r < m->send.size() && m->send[r]) (void)hipFree(m->send[r]);
        if (r < m->recv.size() && m->recv[r]) (void)hipFree(m->recv[r]);
        if (m->replicas[r]) gbnns_index_destroy(m->replicas[r]);
    }
    delete m;
    return GBNNS_OK;
}

int gbnns_multi_size(const gbnns_multi* m) { return m ? (int)m->replicas.size() : 0; }

gbnns_index* gbnns_multi_replica(gbnns_multi* m, int32_t i) {
    return (m && i >= 0 && (size_t)i < m->replicas.size()) ? m->replicas[(size_t)i] : nullptr;
}

int gbnns_multi_device_of(const gbnns_multi* m, int32_t i) {
    return (m && i >= 0 && (size_t)i < m->devices.size()) ? m->devices[(size_t)i] : -1;
}

int gbnns_multi_set_aux_graph(gbnns_multi* m, const uint64_t* offsets, const uint32_t* nbrs) {
    if (!m) return mfail(GBNNS_ERR_INVALID, "null argument");
    return for_each_replica(m, [&](size_t r) { return gbnns_index_set_aux_graph(m->replicas[r], offsets, nbrs); });
}

int gbnns_multi_search_ex(gbnns_multi* m, const gbnns_search_args* a) {
    if (!m || !a) return mfail(GBNNS_ERR_INVALID, "null argument");
    if (a->struct_size != sizeof(gbnns_search_args)) return mfail(GBNNS_ERR_INVALID, "gbnns_search_args.struct_size mismatch");
    if (a->mem_kind != GBNNS_MEM_HOST)
        return mfail(GBNNS_ERR_INVALID, "gbnns_multi_search_ex takes HOST buffers (device blocks: gbnns_multi_search_device)");
    if (a->n_q == 0) return GBNNS_OK;
    // everything a replica would refuse is refused here, once, before any pointer is offset: a replica r > 0 handed
    // "NULL + lo" would take it for a buffer
    if (a->mode < GBNNS_MODE_NET || a->mode > GBNNS_MODE_PLAIN) return mfail(GBNNS_ERR_INVALID, "bad mode");
    if (a->ef <= 0) return mfail(GBNNS_ERR_INVALID, "ef must be >= 1");
    if (a->n_q >= (1ull << 31)) return mfail(GBNNS_ERR_INVALID, "n_q too large");
    if (!a->queries || !a->out_ids) return mfail(GBNNS_ERR_INVALID, "queries / out_ids missing");
    if (a->mode == GBNNS_MODE_LOWQ && !a->queries_low) return mfail(GBNNS_ERR_INVALID, "queries_low missing");
    if (a->n_entries > 1 && !a->entry_ids) return mfail(GBNNS_ERR_INVALID, "n_entries > 1 needs entry_ids");
    if (a->n_entries > 4096) return mfail(GBNNS_ERR_INVALID, "n_entries too large");
    const int R = (int)m->replicas.size();
    const uint32_t n_ent = a->n_entries ? a->n_entries : 1u;
    const int kk = a->mode == GBNNS_MODE_PLAIN ? (a->k > 0 ? (a->k < a->ef ? a->k : a->ef) : 1) : a->ef;  // candidate row width
    // the low dimension is only known to the index: the LOWQ block offset needs it
    return for_each_replica(m, [&](size_t r) -> int {
        uint64_t lo, hi;
        gbnns_shard_bounds(a->n_q, R, (int)r, &lo, &hi);
        if (hi == lo) return GBNNS_OK;
        gbnns_search_args b = *a;
        b.n_q = hi - lo;
        b.stream = m->streams[r];
        b.flags &= ~GBNNS_FLAG_DEFER_JOIN;  // this call is synchronous: the replicas' blocks are complete when it returns
        b.queries = a->queries + lo * m->d;
        if (a->queries_low) {
            const uint32_t dl = gbnns_index_d_low(m->replicas[r]);
            b.queries_low = a->queries_low + lo * dl;
        }
        if (a->entry_ids) b.entry_ids = a->entry_ids + lo * n_ent;
        b.out_ids = a->out_ids + lo;
        if (a->out_hops) b.out_hops = a->out_hops + lo;
        if (a->out_dist_calc) b.out_dist_calc = a->out_dist_calc + lo;
        if (a->out_edges) b.out_edges = a->out_edges + lo;
        if (a->out_cand) b.out_cand = a->out_cand + lo * (uint64_t)kk;
        if (a->out_cand_dist) b.out_cand_dist = a->out_cand_dist + lo * (uint64_t)kk;
        if (a->out_q_low) b.out_q_low = a->out_q_low + lo * gbnns_index_d_low(m->replicas[r]);
        return gbnns_search_ex(m->replicas[r], &b);
    });
}

int gbnns_multi_search_device(gbnns_multi* m, const gbnns_search_args* tmpl, uint64_t n_q, const float* const* query_blocks,
                              const uint32_t* const* entry_blocks, uint32_t* const* out_ids_all) {
    if (!m || !tmpl || !query_blocks || !out_ids_all) return mfail(GBNNS_ERR_INVALID, "null argument");
    if (tmpl->struct_size != sizeof(gbnns_search_args)) return mfail(GBNNS_ERR_INVALID, "gbnns_search_args.struct_size mismatch");
    if (tmpl->mode == GBNNS_MODE_LOWQ) return mfail(GBNNS_ERR_UNSUPPORTED, "LOWQ mode is not offered in the device-block form");
    if (n_q == 0) return GBNNS_OK;
    if (n_q >= (1ull << 31)) return mfail(GBNNS_ERR_INVALID, "n_q too large");
    const int R = (int)m->replicas.size();
    const size_t width = (size_t)((n_q + (uint64_t)R - 1) / (uint64_t)R);
    // communicators and staging buffers (first use, or a larger batch)
    const bool exchange = R > 1 || m->rccl_one;  // (one replica: a plain copy, unless the exchange leg is asked for)
    if (exchange && m->comms.empty()) {
        std::string why;
        if (!m->rccl.load(why)) return mfail(GBNNS_ERR_UNSUPPORTED, "RCCL unavailable: %s", why.c_str());
        m->comms.assign((size_t)R, nullptr);
        const ncclResult_t e = m->rccl.CommInitAll(m->comms.data(), R, m->devices.data());
        if (e != ncclSuccess) {
            m->comms.clear();
            return mfail(GBNNS_ERR_HIP, "ncclCommInitAll: %s (one communicator per DISTINCT device is required)",
                         m->rccl.GetErrorString ? m->rccl.GetErrorString(e) : "error");
        }
    }
    if (width > m->width) {
        m->send.resize((size_t)R, nullptr);
        m->recv.resize((size_t)R, nullptr);
        for (int r = 0; r < R; ++r) {
            if (hipSetDevice(m->devices[(size_t)r]) != hipSuccess) return mfail(GBNNS_ERR_HIP, "hipSetDevice failed");
            (void)hipStreamSynchronize(m->streams[(size_t)r]);
            if (m->send[(size_t)r]) (void)hipFree(m->send[(size_t)r]);
            if (m->recv[(size_t)r]) (void)hipFree(m->recv[(size_t)r]);
            m->send[(size_t)r] = m->recv[(size_t)r] = nullptr;
            if (hipMalloc(reinterpret_cast<void**>(&m->send[(size_t)r]), width * 4) != hipSuccess ||
                hipMalloc(reinterpret_cast<void**>(&m->recv[(size_t)r]), (size_t)R * width * 4) != hipSuccess)
                return mfail(GBNNS_ERR_OOM, "staging buffers for the all-gather");
        }
        m->width = width;
    }
    // searches: each replica's block on its stream, answers into its padded send buffer
    int rc = for_each_replica(m, [&](size_t r) -> int {
        uint64_t lo, hi;
        gbnns_shard_bounds(n_q, R, (int)r, &lo, &hi);
        if (hipSetDevice(m->devices[r]) != hipSuccess) return mfail(GBNNS_ERR_HIP, "hipSetDevice failed");
        if (hipMemsetAsync(m->send[r], 0xFF, m->width * 4, m->streams[r]) != hipSuccess) return mfail(GBNNS_ERR_HIP, "memset failed");
        if (hi == lo) return GBNNS_OK;
        gbnns_search_args b = *tmpl;
        b.mem_kind = GBNNS_MEM_DEVICE;
        b.flags &= ~GBNNS_FLAG_DEFER_JOIN;  // (the all-gather below reads the answers in the replica stream's order)
        b.n_q = hi - lo;
        b.stream = m->streams[r];
        b.queries = query_blocks[r];
        b.queries_low = nullptr;
        b.entry_ids = entry_blocks ? entry_blocks[r] : nullptr;
        b.out_ids = m->send[r];
        b.out_hops = nullptr; b.out_dist_calc = nullptr; b.out_cand = nullptr; b.out_cand_dist = nullptr;
        b.out_q_low = nullptr; b.out_edges = nullptr;
        return gbnns_search_ex(m->replicas[r], &b);
    });
    if (rc) return rc;
    // the path's one exchange step: all-gather of the answer ids (RCCL over xGMI; a plain copy for one replica)
    if (exchange) {
        ncclResult_t e = m->rccl.GroupStart();
        for (int r = 0; r < R && e == ncclSuccess; ++r)
            e = m->rccl.AllGather(m->send[(size_t)r], m->recv[(size_t)r], m->width, ncclUint32, m->comms[(size_t)r], m->streams[(size_t)r]);
        const ncclResult_t e2 = m->rccl.GroupEnd();
        if (e != ncclSuccess || e2 != ncclSuccess)
            return mfail(GBNNS_ERR_HIP, "ncclAllGather: %s", m->rccl.GetErrorString ? m->rccl.GetErrorString(e != ncclSuccess ? e : e2) : "error");
    }
    // unpad: block s of the gathered [R x width] array -> rows shard_bounds(n_q, R, s) of out_ids_all[r]
    for (int r = 0; r < R; ++r) {
        if (hipSetDevice(m->devices[(size_t)r]) != hipSuccess) return mfail(GBNNS_ERR_HIP, "hipSetDevice failed");
        const uint32_t* src = exchange ? m->recv[(size_t)r] : m->send[(size_t)r];
        for (int s = 0; s < R; ++s) {
            uint64_t lo, hi;
            gbnns_shard_bounds(n_q, R, s, &lo, &hi);
            if (hi == lo) continue;
            if (hipMemcpyAsync(out_ids_all[r] + lo, src + (size_t)s * m->width, (hi - lo) * 4, hipMemcpyDeviceToDevice,
                               m->streams[(size_t)r]) != hipSuccess)
                return mfail(GBNNS_ERR_HIP, "hipMemcpyAsync failed");
        }
    }
    return GBNNS_OK;
}

int gbnns_multi_synchronize(gbnns_multi* m) {
    if (!m) return mfail(GBNNS_ERR_INVALID, "null argument");
    for (size_t r = 0; r < m->replicas.size(); ++r) {
        if (hipSetDevice(m->devices[r]) != hipSuccess || hipStreamSynchronize(m->streams[r]) != hipSuccess)
            return mfail(GBNNS_ERR_HIP, "synchronising replica %zu failed", r);
    }
    return GBNNS_OK;
}

// Diagnostic (not in gbnns.h; tests/test_cabi_cpu.py): loads RCCL the way gbnns_multi_search_device does on first
// use with more than one replica.  0 = loaded, GBNNS_ERR_UNSUPPORTED + message otherwise.
int gbnns_internal_rccl_probe(void) {
    Rccl r;
    std::string why;
    if (!r.load(why)) return mfail(GBNNS_ERR_UNSUPPORTED, "RCCL unavailable: %s", why.c_str());
    dlclose(r.so);
    return GBNNS_OK;
}

int gbnns_multi_rccl_single_rank(gbnns_multi* m, int on) {
    if (!m) return mfail(GBNNS_ERR_INVALID, "null argument");
    if (m->replicas.size() != 1) return mfail(GBNNS_ERR_INVALID, "gbnns_multi_rccl_single_rank: the handle has %zu replicas", m->replicas.size());
    m->rccl_one = on != 0;
    return GBNNS_OK;
}

int gbnns_multi_rccl_version(gbnns_multi* m) {
    if (!m || !m->rccl.so || !m->rccl.GetVersion) return 0;
    int v = 0;
    return m->rccl.GetVersion(&v) == ncclSuccess ? v : 0;
}

void* gbnns_multi_stream(gbnns_multi* m, int32_t i) {
    return (m && i >= 0 && (size_t)i < m->streams.size()) ? static_cast<void*>(m->streams[(size_t)i]) : nullptr;
}

}  // extern "C"
