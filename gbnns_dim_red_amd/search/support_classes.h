// support_classes.h -- StopW, the stopwatch the reference harness times its query loop with
// (search/support_classes.h:9-24).  The KLgraph long-link builders of the reference file are out
// of scope (used by naive_test.cpp only; final_test.cpp runs with use_second_graph = false).
#pragma once

#include <chrono>

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
