// Pooled form of the per-pixel render loop main.rs:150-185: more paths than lanes.
//
// The stage-scheduled persistent kernel (render_persistent.hip) binds one path to one lane, so a trip of the
// node stage only has the lanes whose path happens to be inside the tree (17 of 64 on lucy.yaml).  Here a
// wave owns 64 + RT_POOL_SLOTS paths: 64 resident in registers — the ones traversing the tree, each with its
// LDS stack column — and a pool of path records in LDS that wait either for the shade stage (trace complete)
// or for a lane (ray generated, tree traversal pending).  Four kinds of trips, chosen per wave from ballots
// and three wave-uniform queue lengths (no LDS atomics, no barriers: everything is private to the wave):
//
//   node   up to RT_NODE_STEPS inner nodes for the resident lanes standing at an inner node
//   leaf   one whole leaf for the resident lanes standing at a leaf
//   swap   resident lanes whose trace is complete trade their record for a pool record that needs a trace
//   shade  64 pool records with a complete trace are shaded IN PLACE in LDS: scatter (or end the path and
//          start the next sample of the chunk), loose spheres/quads, tree-bounds test; the record then queues
//          for a lane (tree needed) or for the next shade trip (complete already: sky, ground, walls)
//
// so shade trips run with all 64 lanes and traversal trips with most of them, whatever fraction of the
// scene's rays needs the tree.  Samples, RNG streams and arithmetic are those of the persistent kernel; only
// the order of the f32 atomic adds differs.
#include <algorithm>
#include <cstdlib>

#include "device_rt.h"
#include "kernels.h"

namespace rt {

#ifndef RT_POOL_SLOTS
#define RT_POOL_SLOTS 64
#endif
#ifndef RT_POOL_WAVES
#define RT_POOL_WAVES 3  // waves per SIMD the register budget is set for (LDS allows 12 one-wave blocks per CU)
#endif
#ifndef RT_POOL_NODE_STEPS
#define RT_POOL_NODE_STEPS 3
#endif
constexpr int kPool = RT_POOL_SLOTS;
constexpr int kShadeFull = kPool - 16 < 64 ? kPool - 16 : 64;  // shade as soon as this many records wait
constexpr uint32_t kPoolChunkStrata = 16;
constexpr uint32_t kDeadPath = 0xffffffffu;  // depth|pix word of a record without a path
constexpr int kPixBits = 26;                 // depth (<= 63) above, accumulator slot below
static_assert(kPool <= 255, "queue entries are bytes");

struct PoolPlan {
    int swap_threshold;  // run a swap trip once it would serve this many lanes (or nothing else can run)
    uint32_t tiles_x, n_tiles, my_strata, groups, n_chunks;
};

constexpr int kLaneDone = kTraversalDone;  // resident trace complete: waits for a swap
constexpr int kLaneFree = 0x7ffffffe;      // lane holds no path

__device__ __forceinline__ uint32_t lanes_before(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// A path record: 5 x 16 B.  {o.x o.y | o.z d.x d.y | d.z rng depth/pix | thr rank | t slot bu bv}
__device__ __forceinline__ void rec_store(uint4* r, const Ray& ray, const Rng& rng, const F3& thr, uint32_t dp,
                                          const HitRec& h, int rank) {
    r[0] = make_uint4((uint32_t)__double2loint(ray.o.x), (uint32_t)__double2hiint(ray.o.x),
                      (uint32_t)__double2loint(ray.o.y), (uint32_t)__double2hiint(ray.o.y));
    r[1] = make_uint4((uint32_t)__double2loint(ray.o.z), (uint32_t)__double2hiint(ray.o.z), __float_as_uint(ray.d.x),
                      __float_as_uint(ray.d.y));
    r[2] = make_uint4(__float_as_uint(ray.d.z), (uint32_t)rng.s, (uint32_t)(rng.s >> 32), dp);
    r[3] = make_uint4(__float_as_uint(thr.x), __float_as_uint(thr.y), __float_as_uint(thr.z), (uint32_t)rank);
    r[4] = make_uint4(__float_as_uint(h.t), (uint32_t)h.slot, __float_as_uint(h.bu), __float_as_uint(h.bv));
}
__device__ __forceinline__ void rec_load(const uint4* r, Ray& ray, Rng& rng, F3& thr, uint32_t& dp, HitRec& h,
                                         int& rank) {
    const uint4 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4];
    ray.o.x = __hiloint2double((int)a.y, (int)a.x);
    ray.o.y = __hiloint2double((int)a.w, (int)a.z);
    ray.o.z = __hiloint2double((int)b.y, (int)b.x);
    ray.d = {__uint_as_float(b.z), __uint_as_float(b.w), __uint_as_float(c.x)};
    rng.s = ((uint64_t)c.z << 32) | c.y;
    dp = c.w;
    thr = {__uint_as_float(d.x), __uint_as_float(d.y), __uint_as_float(d.z)};
    rank = (int)d.w;
    h.t = __uint_as_float(e.x), h.slot = (int)e.y, h.bu = __uint_as_float(e.z), h.bv = __uint_as_float(e.w);
}

template <int STACK, bool STATS, bool DIRECT>
__global__ __launch_bounds__(64, RT_POOL_WAVES) void render_pool_kernel(DevScene sc, DevRenderParams rp, PoolPlan plan,
                                                                        float* __restrict__ acc,
                                                                        DevCounters* __restrict__ counters,
                                                                        uint32_t* __restrict__ work_counter) {
    __shared__ int lds_stack[STACK * 64];
    __shared__ uint4 pool[kPool * 5];
    __shared__ unsigned char q_shade[kPool], q_trace[kPool], q_free[kPool];
    __shared__ uint4 lds_loose[kMaxLoose * kLooseRecQuads];
    const uint32_t lane = threadIdx.x;
    int* stack = lds_stack + lane;
    for (uint32_t i = lane; i < sc.n_loose * kLooseRecQuads; i += 64u) lds_loose[i] = sc.loose_rec[i];
    const uint64_t seed = ((uint64_t)rp.seed_hi << 32) | rp.seed_lo;

    // every slot starts as a record without a path, queued for the shade stage (which starts paths)
    for (uint32_t s = lane; s < (uint32_t)kPool; s += 64u) {
        pool[5 * s + 2] = make_uint4(0u, 0u, 0u, kDeadPath);
        q_shade[s] = (unsigned char)s;
    }
    int n_sq = kPool, n_tq = 0, n_fq = 0;  // wave-uniform queue lengths (the queues are stacks)

    // wave-uniform cursor over the current chunk of samples
    uint32_t chunk_tile = 0, chunk_group = 0, next_j = 0, chunk_total = 0;
    bool exhausted = false;

    // resident path (registers)
    Ray ray;
    ray.o = {0., 0., 0.};
    ray.d = {0.f, 0.f, 1.f};
    SlabRay sr = slab_ray(sc, ray);
    Closest best;
    closest_reset(best);
    Rng rng;
    rng.s = 0;
    F3 thr = {1.f, 1.f, 1.f};
    uint32_t dp = kDeadPath;
    int node = kLaneFree;
    int sp = 0;
    TraceStats st;
    uint32_t n_segments = 0;
    uint32_t trips[4] = {0, 0, 0, 0}, lanes_on[4] = {0, 0, 0, 0};
    unsigned long long cycles[4] = {0, 0, 0, 0}, t_begin = 0;
    int stage_run = 0;

    while (true) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // queue/pool writes of the last trip are visible
        const bool is_done = node == kLaneDone, is_free = node == kLaneFree;
        const unsigned long long m_done = __ballot(is_done), m_free = __ballot(is_free);
        const unsigned long long m_leaf = __ballot(node < 0);
        const unsigned long long m_node = __ballot(node >= 0 && node < kLaneFree);
        const int n_done = __popcll(m_done), n_free = __popcll(m_free);
        const int n_leaf = __popcll(m_leaf), n_node = __popcll(m_node);
        const int n_trav = n_leaf + n_node;
        // what a swap trip would do now
        const int sw_a = min(n_done, n_tq);         // complete lanes that trade for a pending ray
        const int sw_b = min(n_done - sw_a, n_fq);  // complete lanes that park their record in an empty slot
        const int sw_c = min(n_free, n_tq - sw_a);  // empty lanes that take a pending ray
        const int can_swap = sw_a + sw_b + sw_c;
        if (STATS) t_begin = __builtin_readcyclecounter();

        if (n_sq >= kShadeFull || (n_sq > 0 && n_trav == 0 && can_swap == 0)) {
            // ---------------------------------------------------------------- shade trip (pool, in place)
            const int m = min(64, n_sq);
            const bool act = (int)lane < m;
            if (STATS) trips[2]++, lanes_on[2] += m, stage_run = 2;
            int s = 0;
            Ray r2;
            r2.o = {0., 0., 0.};
            r2.d = {0.f, 0.f, 1.f};
            Rng g2;
            g2.s = 0;
            F3 thr2 = {1.f, 1.f, 1.f};
            uint32_t dp2 = kDeadPath;
            HitRec h2;
            h2.t = 0.f, h2.slot = -1, h2.bu = h2.bv = 0.f;
            int rank2 = -1;
            if (act) {
                s = q_shade[n_sq - 1 - (int)lane];
                rec_load(pool + 5 * s, r2, g2, thr2, dp2, h2, rank2);
            }
            n_sq -= m;
            bool alive2 = act && dp2 != kDeadPath;
            uint32_t depth2 = dp2 >> kPixBits, pix2 = dp2 & ((1u << kPixBits) - 1u);
            // shade the finished segment (scene/mod.rs:104-123)
            if (alive2) {
                F3 lsum = {0.f, 0.f, 0.f};
                const uint32_t meta2 = h2.slot >= 0 ? __float_as_uint(sc.prims[3 * h2.slot].w) : 0u;
                alive2 = shade_bounce<DIRECT>(sc, lds_loose, r2, h2, meta2, 0.001f, g2, thr2, lsum, depth2, rp.max_depth);
                if (!alive2 && (lsum.x != 0.f || lsum.y != 0.f || lsum.z != 0.f)) {
                    float* a = acc + (size_t)(pix2 >> 6) * 192u + (pix2 & 63u);
                    atomicAdd(a, lsum.x);
                    atomicAdd(a + 64, lsum.y);
                    atomicAdd(a + 128, lsum.z);
                }
            }
            // records without a path take the next samples of the chunk (uniform control flow)
            unsigned long long need = __ballot(act && !alive2);
            while (need != 0ull && !exhausted) {
                if (next_j >= chunk_total) {
                    uint32_t c = 0;
                    if (lane == 0) c = atomicAdd(work_counter, 1u);
                    c = __builtin_amdgcn_readfirstlane(c);
                    if (c >= plan.n_chunks) {
                        exhausted = true;
                        break;
                    }
                    chunk_group = c / plan.n_tiles;
                    chunk_tile = c - chunk_group * plan.n_tiles;
                    const uint32_t strata_here = min(kPoolChunkStrata, plan.my_strata - chunk_group * kPoolChunkStrata);
                    chunk_total = 64u * strata_here;
                    next_j = 0;
                }
                const uint32_t avail = chunk_total - next_j;
                const uint32_t before = lanes_before(need);
                if (act && !alive2 && before < avail) {
                    const uint32_t j = next_j + before;
                    const uint32_t in_tile = j & 63u;
                    const uint32_t px = (chunk_tile % plan.tiles_x) * 8u + (in_tile & 7u);
                    const uint32_t py = (chunk_tile / plan.tiles_x) * 8u + (in_tile >> 3);  // image row, 0 = top
                    if (px < rp.width && py < rp.height) {
                        const uint32_t stratum =
                            rp.stratum_begin + (chunk_group * kPoolChunkStrata + (j >> 6)) * rp.stratum_step;
                        pix2 = chunk_tile * 64u + in_tile;
                        rng_init(g2, seed, py * rp.width + px, stratum);
                        r2 = camera_ray<DIRECT>(sc, rp, px, rp.height - py - 1u, stratum, g2);
                        thr2 = {1.f, 1.f, 1.f};
                        depth2 = 0;
                        alive2 = true;
                    }
                }
                next_j += min((uint32_t)__popcll(need), avail);
                need = __ballot(act && !alive2);
            }
            // start the next segment's trace: loose spheres/quads and the tree-bounds test here, the tree itself
            // on a lane
            int dest = 3;  // 0: shade queue (trace complete), 1: trace queue, 2: empty slot, 3: not mine
            if (act) {
                Closest b2;
                closest_reset(b2);
                dest = 2;
                dp2 = kDeadPath;
                if (alive2) {
                    if (STATS) n_segments++;
                    loose_hits<STATS>(lds_loose, sc.n_loose, r2, 0.001f, b2, st);
                    dest = 0;
                    if (sc.has_tree && hits_tree_bounds(sc, slab_ray(sc, r2), 0.001f, b2.h.t)) dest = 1;
                    dp2 = (depth2 << kPixBits) | pix2;
                }
                rec_store(pool + 5 * s, r2, g2, thr2, dp2, b2.h, b2.rank);
            }
            const unsigned long long to_sq = __ballot(dest == 0), to_tq = __ballot(dest == 1), to_fq = __ballot(dest == 2);
            if (dest == 0) q_shade[n_sq + (int)lanes_before(to_sq)] = (unsigned char)s;
            if (dest == 1) q_trace[n_tq + (int)lanes_before(to_tq)] = (unsigned char)s;
            if (dest == 2) q_free[n_fq + (int)lanes_before(to_fq)] = (unsigned char)s;
            n_sq += __popcll(to_sq), n_tq += __popcll(to_tq), n_fq += __popcll(to_fq);
        } else if (can_swap >= plan.swap_threshold || (n_trav == 0 && can_swap > 0)) {
            // ---------------------------------------------------------------- swap trip
            if (STATS) trips[3]++, lanes_on[3] += can_swap, stage_run = 3;
            const int rd = (int)lanes_before(m_done), rf = (int)lanes_before(m_free);
            int src = -1, dst = -1;
            if (is_done && rd < sw_a) src = dst = q_trace[n_tq - 1 - rd];
            else if (is_done && rd < sw_a + sw_b) dst = q_free[n_fq - 1 - (rd - sw_a)];
            else if (is_free && rf < sw_c) src = q_trace[n_tq - sw_a - 1 - rf];
            // the incoming ray first (src may be dst)
            Ray r2 = ray;
            Rng g2 = rng;
            F3 thr2 = thr;
            uint32_t dp2 = dp;
            HitRec h2 = best.h;
            int rank2 = best.rank;
            if (src >= 0) rec_load(pool + 5 * src, r2, g2, thr2, dp2, h2, rank2);
            if (dst >= 0) {
                rec_store(pool + 5 * dst, ray, rng, thr, dp, best.h, best.rank);
                q_shade[n_sq + rd] = (unsigned char)dst;
            }
            // a slot an empty lane took its ray from: a new path starts there (shade stage) while samples remain
            if (src >= 0 && dst < 0) {
                if (!exhausted) {
                    pool[5 * src + 2] = make_uint4(0u, 0u, 0u, kDeadPath);
                    q_shade[n_sq + sw_a + sw_b + rf] = (unsigned char)src;
                } else {
                    q_free[n_fq - sw_b + rf] = (unsigned char)src;
                }
            }
            n_sq += sw_a + sw_b + (exhausted ? 0 : sw_c);
            n_tq -= sw_a + sw_c;
            n_fq += (exhausted ? sw_c : 0) - sw_b;
            if (src >= 0) {
                ray = r2, rng = g2, thr = thr2, dp = dp2, best.h = h2, best.rank = rank2;
                sr = slab_ray(sc, ray);
                sp = 0;
                node = 0;  // the root; the record's ray reaches the tree's bounds (tested in the shade trip)
            } else if (dst >= 0) {
                node = kLaneFree;
            }
        } else if (n_trav == 0) {
            break;  // no path anywhere: lanes empty, queues empty, samples exhausted
        } else if (n_node >= n_leaf) {
            if (STATS) trips[0]++, lanes_on[0] += n_node, stage_run = 0;
#pragma unroll 1
            for (int k = 0; k < RT_POOL_NODE_STEPS; ++k)
                if (node >= 0 && node < kLaneFree) node = node_step<STATS>(sc, sr, 0.001f, best.h.t, node, stack, 64, sp, st);
        } else {
            if (STATS) trips[1]++, lanes_on[1] += n_leaf, stage_run = 1;
            if (node < 0) {
                leaf_step<STATS>(sc, ray, sr, 0.001f, ~node, best, st);
                node = stack_pop(stack, 64, sp);
            }
        }
        if (STATS) {
            const unsigned long long dt = __builtin_readcyclecounter() - t_begin;
            for (int k = 0; k < 4; ++k) cycles[k] += stage_run == k ? dt : 0ull;
        }
    }

    if (STATS) {
        atomicAdd(&counters->segments, (unsigned long long)n_segments);
        atomicAdd(&counters->node_visits, (unsigned long long)st.nodes);
        atomicAdd(&counters->tri_tests, (unsigned long long)st.tris);
        atomicAdd(&counters->sphere_tests, (unsigned long long)st.spheres);
        atomicAdd(&counters->quad_tests, (unsigned long long)st.quads);
        if (lane == 0)
            for (int k = 0; k < 4; ++k) {
                atomicAdd(&counters->stage_trips[k], (unsigned long long)trips[k]);
                atomicAdd(&counters->stage_lanes[k], (unsigned long long)lanes_on[k]);
                atomicAdd(&counters->stage_cycles[k], cycles[k]);
            }
    }
}

template <int STACK, bool STATS, bool DIRECT>
static void launch_one(const DevScene& sc, const DevRenderParams& rp, const PoolPlan& plan, uint32_t blocks, float* d_acc,
                       DevCounters* counters, uint32_t* work_counter, hipStream_t stream) {
    hipLaunchKernelGGL((render_pool_kernel<STACK, STATS, DIRECT>), dim3(blocks), dim3(64), 0, stream, sc, rp, plan, d_acc,
                       counters, work_counter);
}

template <int STACK>
static hipError_t launch_stack(const DevScene& sc, const DevRenderParams& rp, const PoolPlan& plan, uint32_t blocks,
                               float* d_acc, DevCounters* counters, uint32_t* work_counter, bool stats,
                               hipStream_t stream) {
    const bool direct = rp.sampler != 0;
    if (stats && direct) launch_one<STACK, true, true>(sc, rp, plan, blocks, d_acc, counters, work_counter, stream);
    if (stats && !direct) launch_one<STACK, true, false>(sc, rp, plan, blocks, d_acc, counters, work_counter, stream);
    if (!stats && direct) launch_one<STACK, false, true>(sc, rp, plan, blocks, d_acc, counters, work_counter, stream);
    if (!stats && !direct) launch_one<STACK, false, false>(sc, rp, plan, blocks, d_acc, counters, work_counter, stream);
    return hipGetLastError();
}

bool pool_kernel_supports(uint32_t width, uint32_t height, uint32_t tree_depth) {
    const uint64_t n_tiles = (uint64_t)((width + 7u) >> 3) * ((height + 7u) >> 3);
    return n_tiles * 64u <= (1ull << kPixBits) && tree_depth <= 40;
}

hipError_t launch_render_pool(const DevScene& sc, const DevRenderParams& rp, uint32_t tree_depth, uint32_t n_cus,
                              float* d_acc, float* d_rgb_sum, DevCounters* counters, uint32_t* work_counter,
                              bool stats, hipStream_t stream) {
    if (!pool_kernel_supports(rp.width, rp.height, tree_depth) || rp.max_depth > 63u) return hipErrorInvalidValue;
    PoolPlan plan;
    plan.swap_threshold = tuning_env("RT_SWAP_T") ? atoi(tuning_env("RT_SWAP_T")) : 16;
    plan.tiles_x = (rp.width + 7u) >> 3;
    plan.n_tiles = plan.tiles_x * ((rp.height + 7u) >> 3);
    plan.my_strata =
        rp.stratum_begin >= rp.n_strata ? 0 : (rp.n_strata - rp.stratum_begin + rp.stratum_step - 1) / rp.stratum_step;
    plan.groups = (plan.my_strata + kPoolChunkStrata - 1) / kPoolChunkStrata;
    if ((uint64_t)plan.n_tiles * plan.groups >= (1ull << 32)) return hipErrorInvalidValue;
    plan.n_chunks = plan.n_tiles * plan.groups;

    hipError_t e = hipMemsetAsync(d_acc, 0, persistent_acc_bytes(rp.width, rp.height), stream);
    if (e != hipSuccess) return e;
    if ((e = hipMemsetAsync(work_counter, 0, sizeof(uint32_t), stream)) != hipSuccess) return e;
    if (plan.n_chunks != 0) {
        // one-wave blocks; LDS (stack + pool) sets how many fit a CU, surplus blocks find the counter exhausted
        const uint32_t blocks = std::min<uint32_t>(n_cus * 4u * RT_POOL_WAVES, plan.n_chunks);
        if (tree_depth <= 24) e = launch_stack<24>(sc, rp, plan, blocks, d_acc, counters, work_counter, stats, stream);
        else e = launch_stack<40>(sc, rp, plan, blocks, d_acc, counters, work_counter, stats, stream);
        if (e != hipSuccess) return e;
    }
    return launch_resolve(d_acc, rp.width, rp.height, d_rgb_sum, stream);
}

}  // namespace rt
