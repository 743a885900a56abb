// Worker threads of the host side (OBJ line parse, the two tree builds, the flattenings) and the knobs they take.
//
// Two rules, both from the cold-start crash of round 2:
//  * a worker reads no environment variable.  getenv is only safe while no other thread changes the environment,
//    and the HIP runtime's initialisation does change it (libhsa-runtime64 and librocprofiler-register import
//    setenv).  Every RT_* knob of scene creation is read once, by BuildKnobs::from_environment on the thread that
//    calls rt_scene_create, before the first worker exists, and handed down by value;
//  * a worker starts through run_async, which (with RT_SEGV_TRACE set) gives the thread an alternate signal stack
//    before its work, so that a fault on it — a stack overflow included — prints its stack instead of ending the
//    process silently.
#pragma once
#include <cstdint>
#include <future>
#include <utility>

namespace rt {

struct BuildKnobs {
    uint32_t leaf = 0;             // RT_LEAF: render-tree leaf size, 0 = the measured default
    double traversal_cost = 1.0;   // RT_CT: render-tree SAH traversal cost relative to one primitive test
    bool device_build = false;     // RT_BUILD=device
    bool debug = false;            // RT_BUILD_DEBUG: phase times on stderr
    int small = 0;                 // RT_BUILD_SMALL: device build, small-node threshold, 0 = default
    static BuildKnobs from_environment();
};

bool segv_trace_enabled();  // RT_SEGV_TRACE was set when the library was loaded
// this thread's alternate signal stack for as long as the object lives (nothing unless segv_trace_enabled())
struct AltSignalStack {
    void* mem = nullptr;
    AltSignalStack();
    ~AltSignalStack();
    AltSignalStack(const AltSignalStack&) = delete;
    AltSignalStack& operator=(const AltSignalStack&) = delete;
};

template <class F>
auto run_async(F&& f) {
    return std::async(std::launch::async, [fn = std::forward<F>(f)]() mutable {
        AltSignalStack alt;
        return fn();
    });
}

}  // namespace rt
