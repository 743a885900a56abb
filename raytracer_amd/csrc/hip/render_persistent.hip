// Persistent ("megakernel") form of the per-pixel render loop main.rs:150-185: one lane owns one pixel and
// walks its strata; a lane whose path ended regenerates the next camera ray at once, so every lane of a
// wave traces one ray segment per loop trip until its pixel is done (no per-sample re-convergence).
// Radiance is summed in registers and written once: the only HBM traffic besides scene fetches is
// 12 B per pixel.
#include "device_rt.h"
#include "kernels.h"

namespace rt {

constexpr int kBlock = 256;

template <int STACK, bool STATS>
__global__ __launch_bounds__(kBlock) void render_persistent_kernel(DevScene sc, DevRenderParams rp,
                                                                   float* __restrict__ rgb_sum,
                                                                   DevCounters* __restrict__ counters) {
    __shared__ int lds_stack[STACK * kBlock];
    int* stack = lds_stack + threadIdx.x;

    // A wave covers an 8x8 pixel tile (coherent primary rays); tiles in row-major order.
    const uint32_t tiles_x = (rp.width + 7u) >> 3;
    const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t px = (wave % tiles_x) * 8u + (lane & 7u);
    const uint32_t py = (wave / tiles_x) * 8u + (lane >> 3);  // image row, 0 = top
    const bool valid = px < rp.width && py < rp.height;
    const uint32_t idx = py * rp.width + px;  // == reference's idx: pi = idx % W, pj = H - idx/W - 1
    const uint32_t pj = rp.height - py - 1u;
    const uint64_t seed = ((uint64_t)rp.seed_hi << 32) | rp.seed_lo;

    F3 lsum = {0.f, 0.f, 0.f};
    F3 thr = {1.f, 1.f, 1.f};
    Ray ray;
    ray.o = {0., 0., 0.};
    ray.d = {0.f, 0.f, 1.f};
    Rng rng;
    rng.s = 0;
    uint32_t depth = 0;
    uint32_t next = valid ? rp.stratum_begin : rp.n_strata;
    bool alive = false;
    TraceStats st;
    uint32_t n_segments = 0, n_samples = 0;

    while (true) {
        if (!alive) {
            if (next >= rp.n_strata) break;
            rng_init(rng, seed, idx, next);
            ray = camera_ray(sc, rp, px, pj, next, rng);
            next += rp.stratum_step;
            thr = {1.f, 1.f, 1.f};
            depth = 0;
            alive = true;
            if (STATS) n_samples++;
        }
        const HitRec hit = trace_closest<STATS>(sc, ray, 0.001f, stack, kBlock, st);  // scene/mod.rs:104
        if (STATS) n_segments++;
        alive = shade_bounce(sc, ray, hit, 0.001f, rng, thr, lsum, depth, rp.max_depth);
    }

    if (valid) {
        rgb_sum[3 * idx + 0] = lsum.x;
        rgb_sum[3 * idx + 1] = lsum.y;
        rgb_sum[3 * idx + 2] = lsum.z;
    }
    if (STATS) {
        atomicAdd(&counters->samples, (unsigned long long)n_samples);
        atomicAdd(&counters->segments, (unsigned long long)n_segments);
        atomicAdd(&counters->node_visits, (unsigned long long)st.nodes);
        atomicAdd(&counters->tri_tests, (unsigned long long)st.tris);
        atomicAdd(&counters->sphere_tests, (unsigned long long)st.spheres);
        atomicAdd(&counters->quad_tests, (unsigned long long)st.quads);
    }
}

template <int STACK>
static hipError_t launch_stack(const DevScene& sc, const DevRenderParams& rp, float* d_rgb_sum, DevCounters* counters,
                               bool stats, hipStream_t stream) {
    const uint32_t tiles = ((rp.width + 7u) >> 3) * ((rp.height + 7u) >> 3);
    const uint32_t blocks = (tiles * 64u + kBlock - 1) / kBlock;
    if (stats)
        hipLaunchKernelGGL((render_persistent_kernel<STACK, true>), dim3(blocks), dim3(kBlock), 0, stream, sc, rp,
                           d_rgb_sum, counters);
    else
        hipLaunchKernelGGL((render_persistent_kernel<STACK, false>), dim3(blocks), dim3(kBlock), 0, stream, sc, rp,
                           d_rgb_sum, counters);
    return hipGetLastError();
}

hipError_t launch_render_persistent(const DevScene& sc, const DevRenderParams& rp, uint32_t tree_depth,
                                    float* d_rgb_sum, DevCounters* counters, bool stats, hipStream_t stream) {
    // near-first traversal keeps at most (depth - 1) deferred children
    if (tree_depth <= 24) return launch_stack<24>(sc, rp, d_rgb_sum, counters, stats, stream);
    if (tree_depth <= 40) return launch_stack<40>(sc, rp, d_rgb_sum, counters, stats, stream);
    if (tree_depth <= 64) return launch_stack<64>(sc, rp, d_rgb_sum, counters, stats, stream);
    return hipErrorInvalidValue;  // caller reports "BVH deeper than the traversal stack"
}

// main.rs:177-180, 209-221 (average in f64 like the reference's Vec3 / f64)
__global__ void tonemap_kernel(const float* __restrict__ rgb_sum, uint32_t n_values, double n_samples,
                               uint8_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_values) return;
    const double c = sqrt((double)rgb_sum[i] / n_samples);
    // f64::clamp keeps NaN; Rust's `as u8` maps NaN to 0 and saturates
    double v = fmin(fmax(c, 0.0), 0.999);
    if (c != c) v = 0.0;
    out[i] = (uint8_t)(256.0 * v);
}

hipError_t launch_tonemap(const float* d_rgb_sum, uint32_t n_values, double n_samples, uint8_t* d_out,
                          hipStream_t stream) {
    const uint32_t blocks = (n_values + 255u) / 256u;
    hipLaunchKernelGGL(tonemap_kernel, dim3(blocks), dim3(256), 0, stream, d_rgb_sum, n_values, n_samples, d_out);
    return hipGetLastError();
}

}  // namespace rt
