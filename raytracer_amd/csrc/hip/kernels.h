// Host-callable launchers of the HIP kernels (one translation unit per kernel family).
#pragma once
#include <hip/hip_runtime.h>

#include "device_scene.h"

namespace rt {

// d_stack_overflow: persistent_overflow_bytes() of device memory for the six-wave build's deep stack levels, or null
// (the launch then runs the build that keeps the whole stack in LDS)
hipError_t launch_render_persistent(const DevScene& sc, const DevRenderParams& rp, uint32_t tree_depth,
                                    uint32_t n_cus, float* d_acc, float* d_rgb_sum, DevCounters* counters,
                                    uint32_t* work_counter, bool stats, hipStream_t stream, int* d_stack_overflow);
size_t persistent_overflow_bytes(const DevScene& sc, uint32_t tree_depth, uint32_t n_cus);
// bytes of the tile-planar accumulator the persistent kernel adds into (d_acc above)
size_t persistent_acc_bytes(uint32_t width, uint32_t height);
// rt_render_device_noalloc: the largest frame (in 8x8 tiles; 1920x1080 is 32 400) whose accumulator fits the
// library's static device scratch, and that scratch's addresses on `device`, which must be the current device
// on the first call for it (render_persistent.hip)
constexpr uint32_t kNoAllocMaxTiles = 32768;
hipError_t persistent_static_scratch(int device, float** d_acc, DevCounters** d_counters, hipEvent_t** done);
// tile-planar accumulator -> row-major interleaved framebuffer
hipError_t launch_resolve(const float* d_acc, uint32_t width, uint32_t height, float* d_rgb_sum, hipStream_t stream);
// pooled persistent kernel (render_pool.hip): same arguments; false = use launch_render_persistent instead
bool pool_kernel_supports(uint32_t width, uint32_t height, uint32_t tree_depth);
hipError_t launch_render_pool(const DevScene& sc, const DevRenderParams& rp, uint32_t tree_depth, uint32_t n_cus,
                              float* d_acc, float* d_rgb_sum, DevCounters* counters, uint32_t* work_counter,
                              bool stats, hipStream_t stream);
hipError_t launch_tonemap(const float* d_rgb_sum, uint32_t n_values, double n_samples, uint8_t* d_out,
                          hipStream_t stream);
constexpr uint32_t kPrimary64Stack = 128;  // entries of primary64_kernel's private stack; the caller checks the depth
hipError_t launch_primary64(const DevScene& sc, uint32_t width, uint32_t height, int32_t* d_prim_id, double* d_t,
                            hipStream_t stream);
hipError_t launch_primary32(const DevScene& sc, uint32_t width, uint32_t height, uint32_t tree_depth,
                            int32_t* d_prim_id, double* d_t, hipStream_t stream);

// The launch path's tuning and debugging knobs (RT_SHADE_T, RT_NODE_BIAS, RT_TAIL_K, RT_FIRST_CHUNK, RT_NO_W6,
// RT_SWAP_T, RT_WF_DEBUG, RT_STAGE_DEBUG: tools/).  They are looked up only in a process that had RT_TUNE set at its
// first launch — the tools set it —: a caller's render calls read no environment (api.cpp).
const char* tuning_env(const char* name);

}  // namespace rt
