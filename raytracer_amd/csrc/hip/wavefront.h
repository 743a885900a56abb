// Wavefront form of the render loop: raygen -> { extend -> shade (+ wave64 ballot compaction) } per bounce.
#pragma once
#include <hip/hip_runtime.h>

#include "device_scene.h"

namespace rt {

// Per-path state in HBM, structure of arrays, one slot per in-flight path of the current batch.
struct WavefrontBuffers {
    double* org;        // 3 per path  (x[], y[], z[] planes)
    float* dir;         // 3 per path
    float* thr;         // 3 per path
    uint64_t* rng;      // 1 per path
    uint32_t* pixel;    // 1 per path
    uint32_t* depth;    // 1 per path
    float4* hit;        // 1 per path: t, slot bits, bu, bv
    uint32_t* queue[2]; // ping-pong path-index queues
    uint32_t* counts;   // queue length per bounce level (kMaxLevels entries)
    uint32_t capacity;  // paths
};

struct WavefrontState {
    WavefrontBuffers buf{};
    std::vector<void*> allocations;
    ~WavefrontState();
};

hipError_t render_wavefront(WavefrontState& st, const DevScene& sc, const DevRenderParams& rp, uint32_t tree_depth,
                            float* d_rgb_sum, DevCounters* counters, bool stats, hipStream_t stream,
                            uint32_t* n_launches);

}  // namespace rt
