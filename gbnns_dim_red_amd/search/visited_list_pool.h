// visited_list_pool.h -- source-compatibility stand-in for the reference's visited set
// (search/visited_list_pool.h:8-84).  On the MI355X path the visited set lives in LDS inside the
// walk kernel (an exact open-addressing hash set per wavefront), so these host classes only keep
// the types that appear in the reference's signatures (getOneSearchResults takes a
// VisitedListPool*); nothing on the device path touches them.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

typedef uint16_t vl_type;

class VisitedList {
public:
    vl_type curV;
    vl_type* mass;
    size_t numelements;

    explicit VisitedList(size_t count) : curV((vl_type)-1), numelements(count), store_(count) {
        mass = store_.data();
    }
    // epoch bump; on wrap-around the stamps are cleared (reference :21-28)
    void reset() {
        if (++curV == 0) {
            std::memset(mass, 0, sizeof(vl_type) * numelements);
            ++curV;
        }
    }

private:
    std::vector<vl_type> store_;
};

class VisitedListPool {
public:
    VisitedListPool(size_t initial, size_t count) : count_(count) {
        for (size_t i = 0; i < initial; ++i) free_.emplace_back(new VisitedList(count_));
    }
    VisitedList* getFreeVisitedList() {
        std::unique_ptr<VisitedList> vl;
        {
            std::lock_guard<std::mutex> lock(guard_);
            if (!free_.empty()) {
                vl = std::move(free_.back());
                free_.pop_back();
            }
        }
        if (!vl) vl.reset(new VisitedList(count_));
        vl->reset();
        return vl.release();
    }
    void releaseVisitedList(VisitedList* vl) {
        std::lock_guard<std::mutex> lock(guard_);
        free_.emplace_back(vl);
    }

private:
    std::vector<std::unique_ptr<VisitedList>> free_;
    std::mutex guard_;
    size_t count_;
};
