// pin.cpp -- gbnns_host_pin / gbnns_host_unpin: page-locking caller buffers (page-granular, shared pages counted).  Cut out of
// api.cpp in round 5.

#include "api_internal.h"

using namespace gbnns;
using namespace gbnns_api;

namespace {
constexpr uintptr_t kPage = 4096;
struct PinRange { uintptr_t hi; int users; };
std::mutex g_pin_mu;
std::map<uintptr_t, PinRange> g_pin_ranges;        // lo -> [lo, hi), page-aligned, disjoint
std::map<const void*, std::vector<uintptr_t>> g_pin_users;  // user pointer -> the ranges (by lo) it holds
}  // namespace

extern "C" {

int gbnns_host_pin(void* ptr, size_t bytes) {
    if (!ptr || bytes == 0) return fail(GBNNS_ERR_INVALID, "gbnns_host_pin: empty buffer");
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (g_pin_users.count(ptr)) return GBNNS_OK;  // pinned here already
    const uintptr_t lo = reinterpret_cast<uintptr_t>(ptr) & ~(kPage - 1);
    const uintptr_t hi = (reinterpret_cast<uintptr_t>(ptr) + bytes + kPage - 1) & ~(kPage - 1);
    // pages of [lo, hi) that an earlier call registered are shared (counted); the gaps between them are registered now
    std::vector<uintptr_t> held;
    std::vector<std::pair<uintptr_t, uintptr_t>> gaps;
    uintptr_t at = lo;
    auto it = g_pin_ranges.upper_bound(lo);
    if (it != g_pin_ranges.begin()) {
        auto prev = std::prev(it);
        if (prev->second.hi > lo) it = prev;
    }
    for (; it != g_pin_ranges.end() && it->first < hi; ++it) {
        if (it->first > at) gaps.push_back({at, it->first});
        held.push_back(it->first);
        at = std::max(at, it->second.hi);
    }
    if (at < hi) gaps.push_back({at, hi});
    if (held.empty() && pinned_alias(static_cast<char*>(ptr), bytes)) return GBNNS_OK;  // page-locked by the caller: nothing to do, nothing to undo
    const size_t shared = held.size();  // ranges of earlier calls; what follows them in `held` is registered by this call
    for (const auto& gp : gaps) {
        const hipError_t e = hipHostRegister(reinterpret_cast<void*>(gp.first), gp.second - gp.first, hipHostRegisterDefault);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            // (part of the range is page-locked by someone else, or the pages are not ours to lock: the buffer stays
            // pageable for the calls that probe it -- pinned_alias.  Nothing of this call is kept: the gaps registered so
            // far are given back and no user entry is left, so a retry tries again instead of reporting "pinned already"
            // for a buffer that is not)
            for (size_t i = shared; i < held.size(); ++i) {
                if (hipHostUnregister(reinterpret_cast<void*>(held[i])) != hipSuccess) (void)hipGetLastError();
                g_pin_ranges.erase(held[i]);
            }
            return fail(GBNNS_ERR_HIP, "hipHostRegister: %s", hipGetErrorString(e));
        }
        g_pin_ranges[gp.first] = PinRange{gp.second, 0};
        held.push_back(gp.first);
    }
    for (uintptr_t h : held) g_pin_ranges[h].users += 1;
    g_pin_users[ptr] = held;
    return GBNNS_OK;
}

int gbnns_host_unpin(void* ptr) {
    if (!ptr) return GBNNS_OK;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    auto u = g_pin_users.find(ptr);
    if (u == g_pin_users.end()) return GBNNS_OK;  // not registered by gbnns_host_pin: nothing to undo
    for (uintptr_t h : u->second) {
        auto r = g_pin_ranges.find(h);
        if (r == g_pin_ranges.end()) continue;
        if (--r->second.users <= 0) {
            if (hipHostUnregister(reinterpret_cast<void*>(h)) != hipSuccess) (void)hipGetLastError();
            g_pin_ranges.erase(r);
        }
    }
    g_pin_users.erase(u);
    return GBNNS_OK;
}

}  // extern "C"
