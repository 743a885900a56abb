// Wavefront form of the render loop main.rs:150-185 / scene/mod.rs:86-126: per batch of camera paths
//   raygen  ->  for bounce in 0..max_depth { extend (BVH traversal)  ->  shade (scatter, emit, compact) }
// Path state lives in HBM as a structure of arrays; between bounces the surviving paths are compacted into
// a dense index queue with one wave64 ballot + one atomic per wave, so later bounces run full waves.
// Every kernel is launched with a fixed, CU-filling grid and walks its queue with a grid-stride loop; the
// queue length is read from device memory, so there is no host round trip inside a batch.
#include "wavefront.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "device_rt.h"
#include "kernels.h"

namespace rt {

namespace {

constexpr int kWBlock = 256;
constexpr uint32_t kMaxLevels = 1024;            // rp.max_depth is clamped to this many queue counters
constexpr uint32_t kTargetPaths = 16u << 20;     // paths in flight per batch (~1.4 GB of state)

struct Batch {
    uint32_t n_pixels;
    uint32_t n_paths;        // n_pixels * strata in this batch
    uint32_t first_stratum;  // index into this call's stratum sequence (begin + i*step)
};

__device__ __forceinline__ void store_ray(const WavefrontBuffers& b, uint32_t p, const Ray& r) {
    const uint32_t c = b.capacity;
    b.org[p] = r.o.x, b.org[c + p] = r.o.y, b.org[2 * c + p] = r.o.z;
    b.dir[p] = r.d.x, b.dir[c + p] = r.d.y, b.dir[2 * c + p] = r.d.z;
}
__device__ __forceinline__ Ray load_ray(const WavefrontBuffers& b, uint32_t p) {
    const uint32_t c = b.capacity;
    Ray r;
    r.o = {b.org[p], b.org[c + p], b.org[2 * c + p]};
    r.d = {b.dir[p], b.dir[c + p], b.dir[2 * c + p]};
    return r;
}

// camera.rs:76-87 + main.rs:157-174 for every (pixel, stratum) of the batch; path p -> pixel p % n_pixels.
template <bool DIRECT>
__global__ __launch_bounds__(kWBlock) void wf_raygen(DevScene sc, DevRenderParams rp, WavefrontBuffers b, Batch bt) {
    const uint64_t seed = ((uint64_t)rp.seed_hi << 32) | rp.seed_lo;
    if (blockIdx.x == 0 && threadIdx.x == 0) b.counts[0] = bt.n_paths;  // level-0 queue = every path of the batch
    for (uint32_t p = blockIdx.x * kWBlock + threadIdx.x; p < bt.n_paths; p += gridDim.x * kWBlock) {
        const uint32_t idx = p % bt.n_pixels;
        const uint32_t stratum = rp.stratum_begin + (bt.first_stratum + p / bt.n_pixels) * rp.stratum_step;
        const uint32_t pi = idx % rp.width, pj = rp.height - idx / rp.width - 1u;
        Rng rng;
        rng_init(rng, seed, idx, stratum);
        const Ray ray = camera_ray<DIRECT>(sc, rp, pi, pj, stratum, rng);
        store_ray(b, p, ray);
        const uint32_t c = b.capacity;
        b.thr[p] = 1.f, b.thr[c + p] = 1.f, b.thr[2 * c + p] = 1.f;
        b.rng[p] = rng.s;
        b.pixel[p] = idx;
        b.depth[p] = 0;
    }
}

// extend: closest hit for every queued path (bvh.rs:262-314, surfaces.rs hit functions).
template <int STACK, bool STATS>
__global__ __launch_bounds__(kWBlock) void wf_extend(DevScene sc, WavefrontBuffers b, const uint32_t* __restrict__ queue,
                                                     const uint32_t* __restrict__ count_ptr, uint32_t level,
                                                     DevCounters* __restrict__ counters) {
    __shared__ int lds_stack[STACK * kWBlock];
    const uint32_t count = count_ptr[level];
    TraceStats st;
    uint32_t n_seg = 0;
    for (uint32_t i = blockIdx.x * kWBlock + threadIdx.x; i < count; i += gridDim.x * kWBlock) {
        const uint32_t p = level == 0 ? i : queue[i];
        const Ray ray = load_ray(b, p);
        const HitRec h = trace_closest<STATS>(sc, ray, 0.001f, lds_stack + threadIdx.x, kWBlock, st);
        b.hit[p] = make_float4(h.t, __int_as_float(h.slot), h.bu, h.bv);
        if (STATS) n_seg++;
    }
    if (STATS) {
        atomicAdd(&counters->segments, (unsigned long long)n_seg);
        atomicAdd(&counters->node_visits, (unsigned long long)st.nodes);
        atomicAdd(&counters->tri_tests, (unsigned long long)st.tris);
        atomicAdd(&counters->sphere_tests, (unsigned long long)st.spheres);
        atomicAdd(&counters->quad_tests, (unsigned long long)st.quads);
    }
}

// shade: emit + scatter (material.rs:59-181, scene/mod.rs:105-123); a path that ends adds its radiance to the
// framebuffer (every path contributes exactly once: at a light, at the sky, or zero); survivors are appended
// to the next level's queue.
template <bool DIRECT>
__global__ __launch_bounds__(kWBlock) void wf_shade(DevScene sc, DevRenderParams rp, WavefrontBuffers b,
                                                    const uint32_t* __restrict__ queue_in,
                                                    uint32_t* __restrict__ queue_out, uint32_t* __restrict__ counts,
                                                    uint32_t level, float* __restrict__ rgb_sum) {
    const uint32_t count = counts[level];
    const uint32_t c = b.capacity;
    // whole waves iterate together so the ballot below sees all 64 lanes
    const uint32_t count_up = (count + 63u) & ~63u;
    for (uint32_t i = blockIdx.x * kWBlock + threadIdx.x; i < count_up; i += gridDim.x * kWBlock) {
        bool alive = false;
        uint32_t p = 0;
        if (i < count) {
            p = level == 0 ? i : queue_in[i];
            Ray ray = load_ray(b, p);
            const float4 hv = b.hit[p];
            HitRec h;
            h.t = hv.x, h.slot = __float_as_int(hv.y), h.bu = hv.z, h.bv = hv.w;
            Rng rng;
            rng.s = b.rng[p];
            F3 thr = {b.thr[p], b.thr[c + p], b.thr[2 * c + p]};
            F3 lsum = {0.f, 0.f, 0.f};
            uint32_t depth = b.depth[p];
            const uint32_t meta = h.slot >= 0 ? __float_as_uint(sc.prims[3 * h.slot].w) : 0u;
            alive = shade_bounce<DIRECT>(sc, sc.loose_rec, ray, h, meta, 0.001f, rng, thr, lsum, depth, rp.max_depth);
            if (alive) {
                store_ray(b, p, ray);
                b.thr[p] = thr.x, b.thr[c + p] = thr.y, b.thr[2 * c + p] = thr.z;
                b.rng[p] = rng.s;
                b.depth[p] = depth;
            } else if (lsum.x != 0.f || lsum.y != 0.f || lsum.z != 0.f) {
                const uint32_t idx = b.pixel[p];
                atomicAdd(&rgb_sum[3 * idx + 0], lsum.x);
                atomicAdd(&rgb_sum[3 * idx + 1], lsum.y);
                atomicAdd(&rgb_sum[3 * idx + 2], lsum.z);
            }
        }
        // wave64 compaction: one atomic per wave reserves a run of the output queue
        const unsigned long long mask = __ballot(alive);
        if (mask) {
            const uint32_t lane = threadIdx.x & 63u;
            const int leader = __ffsll((long long)mask) - 1;
            uint32_t base = 0;
            if ((int)lane == leader) base = atomicAdd(&counts[level + 1], (uint32_t)__popcll(mask));
            base = __shfl(base, leader);
            if (alive) queue_out[base + __popcll(mask & ((1ull << lane) - 1ull))] = p;
        }
    }
}

template <typename T>
hipError_t grow(WavefrontState& st, T*& ptr, size_t n) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, n * sizeof(T));
    if (e != hipSuccess) return e;
    st.allocations.push_back(p);
    ptr = (T*)p;
    return hipSuccess;
}

hipError_t ensure_capacity(WavefrontState& st, uint32_t n_paths) {
    if (st.buf.capacity >= n_paths) return hipSuccess;
    for (void* p : st.allocations) (void)hipFree(p);
    st.allocations.clear();
    st.buf = WavefrontBuffers{};
    hipError_t e;
    const size_t n = n_paths;
    if ((e = grow(st, st.buf.org, 3 * n)) != hipSuccess) return e;
    if ((e = grow(st, st.buf.dir, 3 * n)) != hipSuccess) return e;
    if ((e = grow(st, st.buf.thr, 3 * n)) != hipSuccess) return e;
    if ((e = grow(st, st.buf.rng, n)) != hipSuccess) return e;
    if ((e = grow(st, st.buf.pixel, n)) != hipSuccess) return e;
    if ((e = grow(st, st.buf.depth, n)) != hipSuccess) return e;
    if ((e = grow(st, st.buf.hit, n)) != hipSuccess) return e;
    if ((e = grow(st, st.buf.queue[0], n)) != hipSuccess) return e;
    if ((e = grow(st, st.buf.queue[1], n)) != hipSuccess) return e;
    if ((e = grow(st, st.buf.counts, (size_t)kMaxLevels + 1)) != hipSuccess) return e;
    st.buf.capacity = n_paths;
    return hipSuccess;
}

template <int STACK>
void launch_extend(bool stats, dim3 grid, hipStream_t stream, const DevScene& sc, const WavefrontBuffers& b,
                   const uint32_t* queue, uint32_t level, DevCounters* counters) {
    if (stats)
        hipLaunchKernelGGL((wf_extend<STACK, true>), grid, dim3(kWBlock), 0, stream, sc, b, queue, b.counts, level, counters);
    else
        hipLaunchKernelGGL((wf_extend<STACK, false>), grid, dim3(kWBlock), 0, stream, sc, b, queue, b.counts, level, counters);
}

}  // namespace

WavefrontState::~WavefrontState() {
    for (void* p : allocations) (void)hipFree(p);
}

hipError_t render_wavefront(WavefrontState& st, const DevScene& sc, const DevRenderParams& rp, uint32_t tree_depth,
                            float* d_rgb_sum, DevCounters* counters, bool stats, hipStream_t stream,
                            uint32_t* n_launches) {
    if (tree_depth > 64) return hipErrorInvalidValue;
    const uint32_t n_pixels = rp.width * rp.height;
    const uint32_t my_strata =
        rp.stratum_begin >= rp.n_strata ? 0 : (rp.n_strata - rp.stratum_begin + rp.stratum_step - 1) / rp.stratum_step;
    const uint32_t levels = std::min(rp.max_depth, kMaxLevels);
    hipError_t e = hipMemsetAsync(d_rgb_sum, 0, (size_t)3 * n_pixels * sizeof(float), stream);
    if (e != hipSuccess) return e;
    uint32_t launches = 0;
    if (my_strata == 0) {
        *n_launches = 0;
        return hipSuccess;
    }
    const uint32_t strata_per_batch = std::max(1u, std::min(my_strata, kTargetPaths / std::max(1u, n_pixels)));
    if ((uint64_t)strata_per_batch * n_pixels >= (1ull << 31)) return hipErrorInvalidValue;
    if ((e = ensure_capacity(st, strata_per_batch * n_pixels)) != hipSuccess) return e;

    hipDeviceProp_t prop;
    int dev = 0;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
    const uint32_t fill_blocks = (uint32_t)prop.multiProcessorCount * 8u;  // 8 blocks of 256 = 32 waves per CU

    for (uint32_t first = 0; first < my_strata; first += strata_per_batch) {
        Batch bt;
        bt.n_pixels = n_pixels;
        bt.first_stratum = first;
        bt.n_paths = std::min(strata_per_batch, my_strata - first) * n_pixels;
        const dim3 grid(std::min(fill_blocks, (bt.n_paths + kWBlock - 1) / kWBlock));
        if ((e = hipMemsetAsync(st.buf.counts, 0, ((size_t)kMaxLevels + 1) * sizeof(uint32_t), stream)) != hipSuccess) return e;
        if (rp.sampler != 0)
            hipLaunchKernelGGL(wf_raygen<true>, grid, dim3(kWBlock), 0, stream, sc, rp, st.buf, bt);
        else
            hipLaunchKernelGGL(wf_raygen<false>, grid, dim3(kWBlock), 0, stream, sc, rp, st.buf, bt);
        ++launches;
        // The metal re-sampling quirk can add more than one to `depth` per bounce, so a path dies after at most
        // max_depth levels; queues beyond the last live level are empty and their kernels fall straight through.
        for (uint32_t level = 0; level < levels; ++level) {
            const uint32_t* qin = st.buf.queue[level & 1];
            uint32_t* qout = st.buf.queue[(level + 1) & 1];
            if (tree_depth <= 24)
                launch_extend<24>(stats, grid, stream, sc, st.buf, qin, level, counters);
            else if (tree_depth <= 40)
                launch_extend<40>(stats, grid, stream, sc, st.buf, qin, level, counters);
            else
                launch_extend<64>(stats, grid, stream, sc, st.buf, qin, level, counters);
            if (rp.sampler != 0)
                hipLaunchKernelGGL(wf_shade<true>, grid, dim3(kWBlock), 0, stream, sc, rp, st.buf, qin, qout, st.buf.counts,
                                   level, d_rgb_sum);
            else
                hipLaunchKernelGGL(wf_shade<false>, grid, dim3(kWBlock), 0, stream, sc, rp, st.buf, qin, qout, st.buf.counts,
                                   level, d_rgb_sum);
            launches += 2;
        }
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if (tuning_env("RT_WF_DEBUG")) {  // queue length per bounce level of this batch
            std::vector<uint32_t> host(levels + 1);
            if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
            if ((e = hipMemcpy(host.data(), st.buf.counts, host.size() * 4, hipMemcpyDeviceToHost)) != hipSuccess) return e;
            fprintf(stderr, "[wavefront] batch @%u queue lengths:", first);
            for (uint32_t v : host) fprintf(stderr, " %u", v);
            fprintf(stderr, "\n");
        }
    }
    *n_launches = launches;
    return hipSuccess;
}

}  // namespace rt
