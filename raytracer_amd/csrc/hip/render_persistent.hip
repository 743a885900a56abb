// Persistent form of the per-pixel render loop main.rs:150-185.
//
// Work is the stream of all (tile, strata-group, pixel, stratum) samples, cut into chunks of one 8x8 pixel
// tile x up to 16 strata.  A fixed, CU-filling grid of waves pulls chunks from one global counter (one atomic
// per ~1024 samples) and hands the samples of the current chunk to its lanes through a wave-uniform cursor:
// a lane whose path ended takes the next sample at once (ballot + mbcnt, no LDS, no barrier), so every lane
// traces one ray segment per loop trip whatever the mix of sky, ground and mesh pixels in the tile, and a
// wave that drew cheap tiles simply pulls more chunks (no static pixel ownership, no tail of heavy waves).
// A finished path adds its radiance with three float atomics — a path contributes exactly once (at a light,
// at the sky, or nothing) — into a tile-planar accumulator acc[tile][rgb][64] (the 64 pixels of a chunk's tile
// are 256 contiguous bytes per component, so a wave's atomics fall into few cache lines); resolve_kernel then
// writes the caller's row-major interleaved framebuffer.  That is 12 B of atomic traffic per sample plus
// 24 B per pixel, and the only HBM traffic besides scene fetches.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "device_rt.h"
#include "kernels.h"

// The triangles-only lean kernel exists twice (last template argument, SIX): compiled for five waves per SIMD
// (96 VGPRs) with the whole traversal stack in LDS, and for SIX (80 VGPRs, 12 dwords spilled) with at most
// kLdsStackCap stack levels in LDS and the deeper ones in memory (SplitStack) — what holds a CU to five blocks on
// lucy.yaml is LDS, 21 KB of each block's 31 KB being the stack of a 22-level tree.  The six-wave kernel serves the
// mesh scenes whose blocks then fit six to a CU: lucy.yaml +5 %, teapot.yaml +6 %, dual.yaml +6 %, cornell_lucy.yaml
// +8 %.
namespace rt {

#ifndef RT_BLOCK
#define RT_BLOCK 256
#endif
constexpr int kBlock = RT_BLOCK;
#ifndef RT_NODE_STEPS
#define RT_NODE_STEPS 4
#endif
#ifndef RT_LEAN_WAVES
#define RT_LEAN_WAVES 5  // waves per SIMD the lean single-kind builds are compiled for (96 VGPRs)
#endif
#ifndef RT_NOTREE_WAVES
#define RT_NOTREE_WAVES 8  // the build without a render tree: 64 VGPRs with 16 dwords spilled (76 unforced, six waves: quads -7 %, distant -3 %)
#endif
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 4
#endif
#ifndef RT_EXT_WAVES
#define RT_EXT_WAVES 4
#endif
#ifndef RT_EXT_NOTREE_WAVES
#define RT_EXT_NOTREE_WAVES 6  // EXT without a render tree and without a noise texture (84 VGPRs unforced; 5: -6 %, 8: -20 %)
#endif
#ifndef RT_EXT_PLAIN_WAVES
#define RT_EXT_PLAIN_WAVES 5   // EXT with a single-kind render tree, without a noise texture (107 VGPRs unforced; 4: -9 %, 6: -32 %)
#endif
#ifndef RT_MEDIA_WAVES
#define RT_MEDIA_WAVES 3  // the build with constant media: 168 VGPRs (4 waves spill 56 dwords: -8..-12 %; 2 waves: -24 %)
#endif
// tools/dynamic_isa.py compiles this file with -DRT_REGION_MARKERS: comments in the instruction stream that say which
// stage the code after them belongs to (no instruction, no effect on the product build, which does not define it)
#ifdef RT_REGION_MARKERS
#define RT_REGION(name) asm volatile("; RT_REGION " name)
#else
#define RT_REGION(name) ((void)0)
#endif
constexpr int kPrioNode = 2, kPrioLeaf = 1;  // s_setprio levels of the node and leaf trips (everything else runs at 0)
constexpr uint32_t kChunkStrata = 16;  // strata per chunk in the first phase (a chunk is 64 pixels x this many strata)
constexpr int kPhases = 5;

struct PersistentPlan {
    int shade_threshold;     // run the shade stage once this many lanes wait in it (or nothing else can run)
    int node_bias;           // node stage runs when n_node * node_bias >= n_leaf * 4
    uint32_t tiles_x, n_tiles;
    uint32_t my_strata;      // strata this call renders (begin, begin+step, ...)
    // Guided hand-out in kPhases phases of shrinking chunks (32 or 16, then 16, 8, 4, 2 strata): phase p covers the strata
    // [phase_first[p], phase_first[p + 1]) of every tile in groups [phase_end[p - 1], phase_end[p]).  The waves
    // then run dry within a small chunk's time of each other; with 16-strata chunks throughout, the waves still
    // busy with a heavy chunk when the counter ran out cost 5 ms of a 33 ms launch on lucy.yaml, 13 of 71 on
    // teapot.yaml (1/8 of the 1024 strata, as an 8-GPU run gives each rank).
    uint32_t phase_first[kPhases + 1], phase_end[kPhases], phase_strata[kPhases];
    uint32_t n_chunks;       // n_tiles * phase_end[kPhases - 1]
    // Dynamic LDS of a block, in this order: the loose records (n_loose x kLooseRecQuads uint4), the traversal stack
    // [stack_levels][kBlock] ints, the waves' tile sums (3 x 64 floats each).  Sized per launch — tree depth and
    // loose count are the scene's — because the kernel lives on its occupancy: 160 KB of LDS per CU hold five
    // blocks only up to 32 KB each.  Last, the waves' batches of pre-drawn camera samples: [gen_words][64] dwords
    // each — direction (3), generator state (2), and where the scene needs them lens sample (2) and shutter time (1).
    uint32_t stack_levels;
    uint32_t gen_words;
    // The six-wave build keeps at most kLdsStackCap levels of the stack in LDS; a lane that defers more children than
    // that — trees deeper than kLdsStackCap + 1 only — keeps the rest here: [level - kLdsStackCap][lane of the launch].
    int* stack_overflow;
};
constexpr uint32_t kLdsStackCap = 16;
// the kernel's arguments as the kernarg segment lays them out (reload_kernarg)
struct PersistentArgs {
    DevScene sc;
    DevRenderParams rp;
    PersistentPlan plan;
    float* acc;
    DevCounters* counters;
    uint32_t* work_counter;
};
// The six-wave build's traversal stack.  What bounds a CU to five blocks on lucy.yaml is LDS, 21 KB of each block's
// 31 KB being the stack of a 22-level tree, whose lower levels a lane hardly ever reaches: the first kLdsStackCap
// levels stay in LDS ([level][lane]), deeper ones go to the launch's overflow array in memory (the address is worked
// out there and then from the kernel's arguments: nothing of it lives in registers along the hot path).
struct SplitStack {
    int* column;  // this lane's column of the LDS array, stride kBlock
    __device__ __forceinline__ int* overflow_slot(int level) const {
        int* base = reload_kernarg<int*>(offsetof(PersistentArgs, plan) + offsetof(PersistentPlan, stack_overflow));
        return base + (size_t)(level - (int)kLdsStackCap) * ((size_t)gridDim.x * kBlock) + (blockIdx.x * kBlock + threadIdx.x);
    }
    __device__ __forceinline__ void store(int level, int v) const {
        if (level < (int)kLdsStackCap) column[level * kBlock] = v;
        else *overflow_slot(level) = v;
    }
    __device__ __forceinline__ int load(int level) const {
        return level < (int)kLdsStackCap ? column[level * kBlock] : *overflow_slot(level);
    }
};
constexpr uint32_t kGenLens = 5, kGenTime = 7;  // first word of the lens sample / of the shutter time in a batch

// Lane stages, encoded in `node`: an inner node id (>= 0) = one node_step pending; a leaf reference (< 0,
// ~slot) = one prim_step pending; kStageShade = the trace is complete (or no path yet): shade / regenerate /
// start the next trace; kStageIdle = no path and no work left.
constexpr int kStageShade = kTraversalDone;
constexpr int kStageIdle = 0x7ffffffe;

// Pointers that say where they point.  The waves' tile sums and sample batches live in LDS and the accumulator in
// device memory, and the code says so: as generic pointers the two ends of the "tile sums or accumulator" choice merged
// into one flat atomic on a selected 64-bit address, the LDS aperture's high word sat in two vector registers for the
// whole kernel (the six-wave build then spilled the path's throughput), and the called functions took 64-bit pointers.
typedef __attribute__((address_space(3))) float* LdsFloats;
typedef __attribute__((address_space(1))) float* DeviceFloats;
__device__ __forceinline__ uint32_t lds_offset(LdsFloats p) { return (uint32_t)(uintptr_t)p; }
__device__ __forceinline__ LdsFloats lds_floats(uint32_t offset) { return (LdsFloats)(uintptr_t)offset; }
__device__ __forceinline__ void add_to(LdsFloats p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void add_to(DeviceFloats p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Adds the wave's LDS tile sums into the tile-planar accumulator (one 256-byte run per component) and clears
// them.  Called by all 64 lanes.
__device__ __forceinline__ void flush_tile(float* acc, LdsFloats wacc, uint32_t tile, uint32_t lane) {
    if (tile == 0xffffffffu) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    DeviceFloats g = (DeviceFloats)acc + (size_t)tile * 192u + lane;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = wacc[64 * c + lane];
        if (v != 0.f) add_to(g + 64 * c, v);
        wacc[64 * c + lane] = 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// The next chunk of the launch's hand-out: which tile, its first stratum (in this call's slice), how many samples
// (0: no work left).  The wave's tile sums go to the accumulator on the way (the chunk's tile becomes the wave's
// current one).  Out of line like draw_batch: once per chunk.
struct ChunkOfWork {
    uint32_t tile, first, total;
};
__device__ __attribute__((noinline)) ChunkOfWork next_chunk(const void* kernarg, uint32_t wacc_offset, uint32_t acc_tile) {
    const LdsFloats wacc = lds_floats(wacc_offset);
    const uintptr_t ka = uniform_address(kernarg);
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t c = 0;
    if (lane == 0) c = atomicAdd(reload_uniform<uint32_t*>(ka + offsetof(PersistentArgs, work_counter)), 1u);
    c = __builtin_amdgcn_readfirstlane(c);
    // (the plan's words one by one, by scalar loads at wave-uniform offsets)
    auto plan_word = [ka](size_t member, int i) {
        return reload_uniform<uint32_t>(ka + offsetof(PersistentArgs, plan) + member + 4u * (size_t)i);
    };
    if (c >= plan_word(offsetof(PersistentPlan, n_chunks), 0)) return ChunkOfWork{0u, 0u, 0u};
    // tile-minor order: concurrently running waves work on neighbouring tiles of one strata group
    const uint32_t n_tiles = plan_word(offsetof(PersistentPlan, n_tiles), 0);
    const uint32_t group = c / n_tiles;
    ChunkOfWork ch;
    ch.tile = c - group * n_tiles;
    flush_tile(reload_uniform<float*>(ka + offsetof(PersistentArgs, acc)), wacc, (uint32_t)__builtin_amdgcn_readfirstlane((int)acc_tile), lane);
    int p = 0;
    uint32_t prev_end = 0;
    while (p < kPhases - 1) {
        const uint32_t end = plan_word(offsetof(PersistentPlan, phase_end), p);
        if (group < end) break;
        prev_end = end, ++p;
    }
    const uint32_t strata = plan_word(offsetof(PersistentPlan, phase_strata), p);
    ch.first = plan_word(offsetof(PersistentPlan, phase_first), p) + (group - prev_end) * strata;
    ch.total = 64u * min(strata, plan_word(offsetof(PersistentPlan, phase_first), p + 1) - ch.first);
    return ch;
}

// One batch of camera samples — stratum `local_stratum` of this call's slice, for the 64 pixels of `tile`, lane l
// drawing pixel l's — into the wave's LDS batch.  Out of line: it runs once per 64 samples, and inlined its
// registers count against every stage of the kernel.
template <bool DIRECT, bool EXT>
__device__ __attribute__((noinline)) void draw_batch(const void* kernarg, uint32_t wgen_offset, uint32_t tile, uint32_t local_stratum,
                                                    bool has_lens) {
    const LdsFloats wgen = lds_floats(wgen_offset);
    // (the kernarg segment's address is the kernel's to know — in a called function the intrinsic yields null — and
    // arrives in vector registers like every argument: made scalar again for the scalar loads)
    const uintptr_t ka = uniform_address(kernarg);
    const uint32_t lane = threadIdx.x & 63u;
    const DevRenderParams gp = reload_uniform<DevRenderParams>(ka + offsetof(PersistentArgs, rp));
    const DevCamera cam = reload_uniform<DevCamera>(ka + offsetof(PersistentArgs, sc) + offsetof(DevScene, cam));
    const uint32_t tiles_x = (gp.width + 7u) >> 3;
    const uint32_t px = (tile % tiles_x) * 8u + (lane & 7u);
    const uint32_t py = (tile / tiles_x) * 8u + (lane >> 3);  // image row, 0 = top
    const uint32_t stratum = gp.stratum_begin + local_stratum * gp.stratum_step;
    Rng g;
    // the stream key is the reference's idx: pi = idx % W, pj = H - idx/W - 1
    rng_init(g, ((uint64_t)gp.seed_hi << 32) | gp.seed_lo, py * gp.width + px, stratum);
    const CameraSample cs = camera_sample<DIRECT>(cam, gp, px, gp.height - py - 1u, stratum, g);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // a pixel of the tile that lies outside the frame: NaN marks "no sample"
    wgen[lane] = (px < gp.width && py < gp.height) ? cs.d.x : __builtin_nanf("");
    wgen[64 + lane] = cs.d.y, wgen[128 + lane] = cs.d.z;
    wgen[192 + lane] = __uint_as_float((uint32_t)g.s), wgen[256 + lane] = __uint_as_float((uint32_t)(g.s >> 32));
    if (has_lens) wgen[kGenLens * 64u + lane] = cs.rx, wgen[(kGenLens + 1u) * 64u + lane] = cs.ry;
    if (EXT) wgen[kGenTime * 64u + lane] = cs.time;
}

// Waves per SIMD each build is compiled for (the register budget that follows from it): every one measured, see the
// macros above and DESIGN §4.1 "builds".
constexpr int waves_per_simd(bool ext, bool media, int tree, bool noise, bool six) {
    if (six) return 6;
    if (media) return RT_MEDIA_WAVES;
    if (!ext) return tree == 4 ? RT_NOTREE_WAVES : (tree != 0 ? RT_LEAN_WAVES : RT_MIN_WAVES);
    if (noise) return RT_EXT_WAVES;
    return tree == 4 ? RT_EXT_NOTREE_WAVES : (tree != 0 ? RT_EXT_PLAIN_WAVES : RT_EXT_WAVES);
}

// A wave-uniform number the compiler knows nothing about any more (no instruction: an empty asm on its scalar register).
__device__ __forceinline__ int forget_uniform(int v) {
    asm volatile("" : "+s"(v));
    return v;
}

template <bool STATS, bool DIRECT, bool EXT, bool MEDIA, int TREE, bool NOISE = true, bool SIX = false>
__global__ __launch_bounds__(kBlock, waves_per_simd(EXT, MEDIA, TREE, NOISE, SIX)) void render_persistent_kernel(DevScene sc, DevRenderParams rp, PersistentPlan plan,
                                                                   float* __restrict__ acc,
                                                                   DevCounters* __restrict__ counters,
                                                                   uint32_t* __restrict__ work_counter) {
    extern __shared__ uint4 lds_dyn[];
    uint4* lds_loose = lds_dyn;
    int* lds_stack = (int*)(lds_dyn + sc.n_loose * kLooseRecQuads);
    float* lds_acc = (float*)(lds_stack + plan.stack_levels * kBlock);
    // (every level in LDS; the six-wave build keeps the levels beyond kLdsStackCap in memory, see SplitStack, and
    // takes the code that knows about them only on the trips in which a lane of the wave is that deep)
    const LdsStack stack{lds_stack + threadIdx.x, kBlock};
    const SplitStack deep_stack{lds_stack + threadIdx.x};
    for (uint32_t i = threadIdx.x; i < sc.n_loose * kLooseRecQuads; i += kBlock) lds_loose[i] = sc.loose_rec[i];
    // radiance sums of the tile this wave's current chunk belongs to (3 x 64 floats per wave)
    const LdsFloats wacc = (LdsFloats)lds_acc + (threadIdx.x >> 6) * 192;
    const LdsFloats wgen = (LdsFloats)lds_acc + (kBlock / 64) * 192 + (threadIdx.x >> 6) * plan.gen_words * 64u;
    wacc[threadIdx.x & 63u] = 0.f, wacc[64 + (threadIdx.x & 63u)] = 0.f, wacc[128 + (threadIdx.x & 63u)] = 0.f;
    uint32_t acc_tile = 0xffffffffu;  // wave-uniform
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;

    // wave-uniform cursor over the current chunk
    uint32_t chunk_tile = 0, chunk_first = 0, next_j = 0, chunk_total = 0;  // chunk_first: its first stratum
    bool exhausted = false;
    // The chunk's samples are drawn 64 at a time — one stratum of the tile's 64 pixels, lane l drawing pixel l's,
    // all lanes busy, control flow uniform — into the wave's LDS batch; a lane whose path ended picks its sample up
    // from there.  (Drawn on demand by the lanes that need one, the stream seeding and the camera sample ran on a
    // third of the lanes of every shade trip.)  gen_j: first sample index of the batch the LDS holds.
    uint32_t gen_j = 0xffffffffu;
    const bool has_lens = sc.cam.lens_radius != 0.f;

    // per-lane path + traversal state
    F3 thr = {1.f, 1.f, 1.f};
    Ray ray;
    ray.o = {0., 0., 0.};
    ray.d = {0.f, 0.f, 1.f};
    SlabRay sr = slab_ray(sc, ray);
    Closest best;
    closest_reset(best);
    Rng rng;
    rng.s = 0;
    uint32_t depth = 0, pix = 0;  // pix = tile * 64 + position in tile
    bool alive = false;   // a path is in flight
    int node = kStageShade;
    int sp = 0;
    TraceStats st;
    uint32_t n_segments = 0;
    uint32_t trips[4] = {0, 0, 0, 0}, lanes_on[4] = {0, 0, 0, 0};  // counting build: stage occupancy 
    unsigned long long cycles[4] = {0, 0, 0, 0}, t_begin = 0;
    unsigned long long sub_cycles[3] = {0, 0, 0};  // shade trip split: bounce+accumulate, regenerate, loose+bounds
    int stage_run = 0;
    unsigned long long t_exhausted = 0;
    uint32_t drain_trips[3] = {0, 0, 0}, drain_lanes[3] = {0, 0, 0};
    if (STATS && lane == 0) atomicMax(&counters->t_first_inv, ~wall_clock64());

    // Every trip runs ONE stage for the lanes waiting in it — the stage most lanes wait in — so the lanes that
    // execute a trip all do the same small piece of work (one node, one primitive, or one shade+raygen),
    // whatever mix of cheap sky/ground rays and long mesh traversals the wave holds.
    while (true) {
        RT_REGION("select");
        const unsigned long long m_shade = __ballot(node == kStageShade);
        const unsigned long long m_leaf = __ballot(node < 0);
        const unsigned long long m_node = __ballot(node >= 0 && node < kStageIdle);
        const int n_shade = __popcll(m_shade), n_leaf = __popcll(m_leaf), n_node = __popcll(m_node);
        if ((n_shade | n_leaf | n_node) == 0) break;  // every lane idle: no paths in flight, no work left

        if (STATS) t_begin = __builtin_readcyclecounter();
        // Which stage runs: the shade stage (2) once shade_threshold lanes wait in it or nothing else can run, otherwise
        // the node stage (0) or the leaf stage (1), whichever has more lanes (node_bias weighs them).  TREE == 4: a scene
        // without a render tree (every primitive loose) — no node or leaf stage is compiled in.
        // The stage is a number, forgotten and compared again before each stage's code: three plain `if`s in a row.  (As
        // an if / else-if / else chain, the structuriser threads every per-lane value the stages change through the join
        // blocks it adds, and the copies it leaves there — the closest-hit record's six words back and forth, in every
        // trip, the node trips that do not touch it included — are never coalesced: 13 moves per trip.)
        int stage = 2;
        if (TREE != 4)
            asm volatile(  // on the scalar unit, and opaque to the compiler like forget_uniform's
                "s_cmp_ge_i32 %1, %2\n\ts_cselect_b32 %0, 0, 1\n\t"
                "s_cmp_ge_i32 %3, %4\n\ts_cselect_b32 %0, 2, %0\n\t"
                "s_cmp_eq_u32 %5, 0\n\ts_cselect_b32 %0, 2, %0"
                : "=&s"(stage)
                : "s"(n_node * plan.node_bias), "s"(n_leaf * 4), "s"(n_shade), "s"(plan.shade_threshold), "s"(n_node | n_leaf)
                : "scc");
        if (stage == 0) {
            if (STATS) trips[0]++, lanes_on[0] += n_node, stage_run = 0;
            if (STATS && exhausted) drain_trips[0]++, drain_lanes[0] += n_node;
            // a few node visits per trip: amortises the stage selection over the dependent fetches of one descent.
            // A node trip is a chain of dependent fetches with little arithmetic between them: the wave goes first
            // at the issue arbiter while it runs one (s_setprio), so that its next fetch is on its way before the
            // leaf and shade trips of the SIMD's other waves take their turn: lucy +3.3 %, teapot +3.8 %, cornell_lucy
            // +2.7 %, dual +4.9 %.  Leaf trips go second (another +0.5 %), shade trips last; the shade stage first
            // instead: -2.6 %; node and leaf stages at one level: +-0; only the fetch-issuing ends of the leaf and
            // shade stages raised: +-0 (profiles/r3_experiments.md).
            __builtin_amdgcn_s_setprio(kPrioNode);
            RT_REGION("node");
            if (SIX && __any(sp + RT_NODE_STEPS > (int)kLdsStackCap)) {  // a push or pop may pass the LDS levels
                RT_REGION("node.deep");
#pragma unroll 1
                for (int k = 0; k < RT_NODE_STEPS; ++k)
                    if (node >= 0 && node < kStageIdle) node = node_step<STATS>(sc, sr, 0.001f, best.h.t, node, deep_stack, sp, st);
            } else {
                RT_REGION("node");
#pragma unroll 1
                for (int k = 0; k < RT_NODE_STEPS; ++k)
                    if (node >= 0 && node < kStageIdle) node = node_step<STATS>(sc, sr, 0.001f, best.h.t, node, stack, sp, st);
            }
            __builtin_amdgcn_s_setprio(0);
            RT_REGION("select");
        }
        stage = forget_uniform(stage);
        if (stage == 1) {
            if (STATS) trips[1]++, lanes_on[1] += n_leaf, stage_run = 1;
            if (STATS && exhausted) drain_trips[1]++, drain_lanes[1] += n_leaf;
            RT_REGION("leaf");
            __builtin_amdgcn_s_setprio(kPrioLeaf);
            if (node < 0) {
                leaf_step<STATS, EXT, TREE>(sc, ray, sr, 0.001f, ~node, best, st);
                node = (SIX && sp > (int)kLdsStackCap) ? stack_pop(deep_stack, sp) : stack_pop(stack, sp);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        stage = forget_uniform(stage);
        if (stage == 2) {
            if (STATS) trips[2]++, lanes_on[2] += n_shade, stage_run = 2;
            if (STATS && exhausted) drain_trips[2]++, drain_lanes[2] += n_shade;
            RT_REGION("shade.bounce");
            const bool mine = node == kStageShade;
            // The scene's parameters as this stage sees them: read again from the kernel's arguments (scalar loads of
            // the members used below) rather than kept in SGPRs across the node and leaf stages — see reload_uniform.
            const DevScene ss = reload_kernarg<DevScene>(offsetof(PersistentArgs, sc));
            const uint32_t max_depth = reload_kernarg<uint32_t>(offsetof(PersistentArgs, rp) + offsetof(DevRenderParams, max_depth));
            float* const acc_now = reload_kernarg<float*>(offsetof(PersistentArgs, acc));
            unsigned long long t_part = 0;
            if (STATS) t_part = __builtin_readcyclecounter();
            // ---- shade the finished segment (scene/mod.rs:104-123)
            if (mine && alive) {
                F3 lsum = {0.f, 0.f, 0.f};  // a path contributes once, where it ends
                if (MEDIA) {
                    // the media are evaluated after the surface trace, and again on every re-trace of a ray whose
                    // hit did not scatter (see shade_bounce)
                    const Closest surface = best;
                    MediaSpans spans;
                    media_spans(ss, ray, spans);
                    bool again, light = false;
                    F3 emitted = {0.f, 0.f, 0.f};
                    do {
                        again = false;
                        best = surface;
                        // the same light again and again (Q1) until a medium scatters or the depth runs out
                        if (light && media_light_attempts(ss, spans, surface, 0.001f, rng, depth, max_depth)) {
                            lsum = lsum + thr * emitted;
                            alive = false;
                            break;
                        }
                        media_sample(ss, ray, 0.001f, rng, best, spans);
                        alive = shade_bounce<DIRECT, EXT, NOISE>(ss, lds_loose, ray, best.h, best.meta, 0.001f, rng, thr, lsum, depth,
                                                          max_depth, &again, &emitted);
                        light = again && best.h.slot > kMediumSlot && (emitted.x != 0.f || emitted.y != 0.f || emitted.z != 0.f);
                    } while (again && alive);
                } else {
                    alive = shade_bounce<DIRECT, EXT, NOISE, TREE == 1>(ss, lds_loose, ray, best.h, best.meta, 0.001f, rng, thr, lsum,
                                                                 depth, max_depth);
                }
                if (!alive && (lsum.x != 0.f || lsum.y != 0.f || lsum.z != 0.f)) {
                    // tile-planar accumulator: the lanes of a wave add into one 256-byte run per component
                    if ((pix >> 6) == acc_tile) {  // the wave's current tile: sum in LDS, flushed once per chunk
                        add_to(wacc + (pix & 63u), lsum.x);
                        add_to(wacc + 64 + (pix & 63u), lsum.y);
                        add_to(wacc + 128 + (pix & 63u), lsum.z);
                    } else {  // a path that outlived its chunk
                        const DeviceFloats a = (DeviceFloats)acc_now + (size_t)(pix >> 6) * 192u + (pix & 63u);
                        add_to(a, lsum.x);
                        add_to(a + 64, lsum.y);
                        add_to(a + 128, lsum.z);
                    }
                }
            }
            if (STATS) {
                const unsigned long long t = __builtin_readcyclecounter();
                sub_cycles[0] += t - t_part, t_part = t;
            }
            // ---- regenerate: lanes without a path take the next samples of the chunk (uniform control flow)
            RT_REGION("shade.regenerate");
            unsigned long long need = __ballot(mine && !alive);
            while (need != 0ull && !exhausted) {
                if (next_j >= chunk_total) {
                    const ChunkOfWork ch = next_chunk((const void*)__builtin_amdgcn_kernarg_segment_ptr(), lds_offset(wacc), acc_tile);
                    chunk_total = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch.total);
                    if (chunk_total == 0u) {
                        exhausted = true;
                        if (STATS) t_exhausted = wall_clock64();
                        break;
                    }
                    chunk_tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch.tile);
                    chunk_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch.first);
                    acc_tile = chunk_tile;
                    next_j = 0;
                    gen_j = 0xffffffffu;
                }
                if (gen_j != (next_j & ~63u)) {  // wave-uniform: draw the batch next_j starts
                    gen_j = next_j & ~63u;
                    draw_batch<DIRECT, EXT>((const void*)__builtin_amdgcn_kernarg_segment_ptr(), lds_offset(wgen), chunk_tile,
                                            chunk_first + (gen_j >> 6), has_lens);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
                const uint32_t avail = gen_j + 64u - next_j;  // what is left of the batch (chunks are whole batches)
                const uint32_t before =
                    __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
                if (mine && !alive && before < avail) {
                    const uint32_t in_tile = (next_j + before) & 63u;
                    CameraSample cs;
                    cs.d.x = wgen[in_tile];
                    if (cs.d.x == cs.d.x) {
                        cs.d.y = wgen[64 + in_tile], cs.d.z = wgen[128 + in_tile];
                        rng.s = ((uint64_t)__float_as_uint(wgen[256 + in_tile]) << 32) | __float_as_uint(wgen[192 + in_tile]);
                        cs.rx = cs.ry = cs.time = 0.f;
                        if (has_lens) cs.rx = wgen[kGenLens * 64u + in_tile], cs.ry = wgen[(kGenLens + 1u) * 64u + in_tile];
                        if (EXT) cs.time = wgen[kGenTime * 64u + in_tile];
                        pix = chunk_tile * 64u + in_tile;  // slot in the tile-planar accumulator
                        ray = camera_ray_of(ss.cam, cs);
                        thr = {1.f, 1.f, 1.f};
                        depth = 0;
                        alive = true;
                    }
                }
                next_j += min((uint32_t)__popcll(need), avail);
                need = __ballot(mine && !alive);
            }
            if (STATS) {
                const unsigned long long t = __builtin_readcyclecounter();
                sub_cycles[1] += t - t_part, t_part = t;
            }
            // ---- start the next segment's trace: loose spheres/quads now (all these lanes in lockstep), the
            // tree through the node/leaf stages
            RT_REGION("shade.loose");
            if (mine) {
                if (alive) {
                    if (STATS) n_segments++;
                    closest_reset(best);
                    loose_hits<STATS, EXT>(uniform_recs(ss.loose_rec), ss.n_loose, ray, 0.001f, best, st);
                    if (TREE != 4 && ss.has_tree) {
                        sr = slab_ray(ss, ray);
                        sp = 0;
                        // rays that never reach the tree's bounds are complete already: stay in the shade stage
                        node = hits_tree_bounds(ss, sr, 0.001f, best.h.t) ? 0 : kStageShade;
                    }
                } else {
                    node = kStageIdle;
                }
            }
            if (STATS) sub_cycles[2] += __builtin_readcyclecounter() - t_part;
            RT_REGION("select");
        }
        if (STATS) {
            const unsigned long long t_end = __builtin_readcyclecounter();
            for (int k = 0; k < 4; ++k) cycles[k] += stage_run == k ? t_end - t_begin : 0ull;
        }
    }

    flush_tile(reload_kernarg<float*>(offsetof(PersistentArgs, acc)), wacc, acc_tile, lane);

    if (STATS) {
        atomicAdd(&counters->segments, (unsigned long long)n_segments);
        atomicAdd(&counters->node_visits, (unsigned long long)st.nodes);
        atomicAdd(&counters->tri_tests, (unsigned long long)st.tris);
        atomicAdd(&counters->sphere_tests, (unsigned long long)st.spheres);
        atomicAdd(&counters->quad_tests, (unsigned long long)st.quads);
        if (lane == 0) {
            const unsigned long long t0 = ~__hip_atomic_load(&counters->t_first_inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long bucket = 50000ull;  // 0.5 ms at 100 MHz
            atomicAdd(&counters->exhaust_hist[min((unsigned long long)63, (t_exhausted - t0) / bucket)], 1u);
            const unsigned long long t_fin = wall_clock64();
            atomicAdd(&counters->finish_hist[min((unsigned long long)63, (t_fin - t0) / bucket)], 1u);
            atomicAdd(&counters->drain_hist[min((unsigned long long)63, (t_fin - t_exhausted) / 25000ull)], 1u);
            if (t_fin - t_exhausted > 200000ull) {
                const uint32_t k = atomicAdd(&counters->n_late, 1u);
                if (k < 32u) {
                    unsigned long long* r = counters->late[k];
                    r[0] = t_exhausted - t0, r[1] = t_fin - t0;
                    for (int q = 0; q < 3; ++q) r[2 + q] = drain_trips[q], r[5 + q] = drain_lanes[q];
                }
            }
        }
        if (lane == 0)
            for (int k = 0; k < 4; ++k) {
                atomicAdd(&counters->stage_trips[k], (unsigned long long)trips[k]);
                atomicAdd(&counters->stage_lanes[k], (unsigned long long)lanes_on[k]);
                atomicAdd(&counters->stage_cycles[k], cycles[k]);
                if (k < 3) atomicAdd(&counters->stage_cycles[4 + k], sub_cycles[k]);
            }
    }
}

// the six-wave kernel's launcher: the triangles-only lean kernel, by counting build and sampler
static hipError_t launch_persistent_six(const DevScene& sc, const DevRenderParams& rp, const PersistentPlan& plan, uint32_t blocks,
                                        size_t lds, float* d_acc, DevCounters* counters, uint32_t* work_counter, bool stats,
                                        hipStream_t stream) {
#define RT_LAUNCH_SIX(STATS_, DIRECT_)                                                                                    \
    hipLaunchKernelGGL((render_persistent_kernel<STATS_, DIRECT_, false, false, 1, true, true>), dim3(blocks), dim3(kBlock), \
                       lds, stream, sc, rp, plan, d_acc, counters, work_counter)
    const bool direct = rp.sampler != 0;
    if (stats && direct) RT_LAUNCH_SIX(true, true);
    else if (stats) RT_LAUNCH_SIX(true, false);
    else if (direct) RT_LAUNCH_SIX(false, true);
    else RT_LAUNCH_SIX(false, false);
#undef RT_LAUNCH_SIX
    return hipGetLastError();
}

// tile-planar accumulator -> the caller's row-major interleaved framebuffer (row 0 = top, main.rs:158)
__global__ void resolve_kernel(const float* __restrict__ acc, uint32_t width, uint32_t height, uint32_t tiles_x,
                               float* __restrict__ rgb_sum) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= width * height) return;
    const uint32_t px = i % width, py = i / width;
    const float* a = acc + (size_t)((py >> 3) * tiles_x + (px >> 3)) * 192u + ((py & 7u) * 8u + (px & 7u));
    rgb_sum[3 * i + 0] = a[0];
    rgb_sum[3 * i + 1] = a[64];
    rgb_sum[3 * i + 2] = a[128];
}

size_t persistent_lds_bytes(const DevScene& sc, const PersistentPlan& plan) {
    return (size_t)sc.n_loose * kLooseRecQuads * sizeof(uint4) + (size_t)plan.stack_levels * kBlock * sizeof(int) +
           (size_t)(kBlock / 64) * (192 + plan.gen_words * 64) * sizeof(float);
}

template <bool STATS, bool DIRECT>
static void launch_one(const DevScene& sc, const DevRenderParams& rp, const PersistentPlan& plan, uint32_t blocks,
                       float* d_acc, DevCounters* counters, uint32_t* work_counter, hipStream_t stream) {
    const size_t lds = persistent_lds_bytes(sc, plan);
    // scenes with textures or moving spheres run the EXT build, those with constant media the EXT + MEDIA one; the
    // others keep the lean one (each feature costs registers in every stage, used or not)
#define RT_LAUNCH(EXT_, MEDIA_, TREE_)                                                                                  \
    hipLaunchKernelGGL((render_persistent_kernel<STATS, DIRECT, EXT_, MEDIA_, TREE_>), dim3(blocks), dim3(kBlock), lds, \
                       stream, sc, rp, plan, d_acc, counters, work_counter)
#define RT_LAUNCH_PLAIN(TREE_)                                                                                          \
    hipLaunchKernelGGL((render_persistent_kernel<STATS, DIRECT, true, false, TREE_, false>), dim3(blocks), dim3(kBlock), \
                       lds, stream, sc, rp, plan, d_acc, counters, work_counter)
    const bool noise = (sc.ext & 2u) != 0u;
    if (sc.ext && sc.n_media) RT_LAUNCH(true, true, 0);
    else if (sc.ext && !noise && !sc.has_tree) RT_LAUNCH_PLAIN(4);
    else if (sc.ext && !noise && sc.tree_kind == 2u) RT_LAUNCH_PLAIN(2);
    else if (sc.ext && !noise) RT_LAUNCH_PLAIN(0);
    else if (sc.ext && !sc.has_tree) RT_LAUNCH(true, false, 4);
    else if (sc.ext && sc.tree_kind == 2u) RT_LAUNCH(true, false, 2);
    else if (sc.ext) RT_LAUNCH(true, false, 0);
    else if (!sc.has_tree) RT_LAUNCH(false, false, 4);
    else if (sc.tree_kind == 1u) RT_LAUNCH(false, false, 1);
    else if (sc.tree_kind == 2u) RT_LAUNCH(false, false, 2);
    else if (sc.tree_kind == 3u) RT_LAUNCH(false, false, 3);
    else RT_LAUNCH(false, false, 0);
#undef RT_LAUNCH
#undef RT_LAUNCH_PLAIN
}

// Six 256-thread blocks share a CU's 160 KB of LDS when each needs at most this much (allocated in 512-byte steps)
constexpr size_t kLdsForSixBlocks = (160u * 1024u / 6u) / 512u * 512u;

static hipError_t launch_any(const DevScene& sc, const DevRenderParams& rp, const PersistentPlan& plan, uint32_t blocks,
                             float* d_acc, DevCounters* counters, uint32_t* work_counter, bool stats, hipStream_t stream) {
    const bool direct = rp.sampler != 0;
    if (stats && direct) launch_one<true, true>(sc, rp, plan, blocks, d_acc, counters, work_counter, stream);
    if (stats && !direct) launch_one<true, false>(sc, rp, plan, blocks, d_acc, counters, work_counter, stream);
    if (!stats && direct) launch_one<false, true>(sc, rp, plan, blocks, d_acc, counters, work_counter, stream);
    if (!stats && !direct) launch_one<false, false>(sc, rp, plan, blocks, d_acc, counters, work_counter, stream);
    return hipGetLastError();
}

static uint32_t persistent_grid_limit(uint32_t n_cus) { return n_cus * 8u * (256u / kBlock); }

size_t persistent_overflow_bytes(const DevScene& sc, uint32_t tree_depth, uint32_t n_cus) {
    const uint32_t levels = std::max(tree_depth, 2u) - 1u;
    if (sc.ext || !sc.has_tree || sc.tree_kind != 1u || levels <= kLdsStackCap) return 0;
    return (size_t)(levels - kLdsStackCap) * persistent_grid_limit(n_cus) * kBlock * sizeof(int);
}

hipError_t launch_render_persistent(const DevScene& sc, const DevRenderParams& rp, uint32_t tree_depth,
                                    uint32_t n_cus, float* d_acc, float* d_rgb_sum, DevCounters* counters,
                                    uint32_t* work_counter, bool stats, hipStream_t stream, int* d_stack_overflow) {
    PersistentPlan plan;
    plan.stack_overflow = nullptr;
    // (36 for the large triangle meshes, 44 for the small ones — trees of up to 18 levels, which stay in the caches: a
    // node trip is then cheap beside a shade trip, and the shade stage had better wait for more lanes: teapot.yaml
    // +3.3 %, dual.yaml +1.6 %, where lucy.yaml would lose 2.7 %; trees of spheres and quads, whose rays nearly all
    // enter the tree, take 48: one-weekend +3.5 %, random-spheres +6 %, final-scene +3 %)
    plan.shade_threshold = tuning_env("RT_SHADE_T") ? atoi(tuning_env("RT_SHADE_T"))
                           : (sc.tree_kind != 1u ? 48 : (tree_depth <= 18u ? 44 : 36));
    plan.node_bias = tuning_env("RT_NODE_BIAS") ? atoi(tuning_env("RT_NODE_BIAS")) : 2;
    plan.tiles_x = (rp.width + 7u) >> 3;
    plan.n_tiles = plan.tiles_x * ((rp.height + 7u) >> 3);
    plan.my_strata =
        rp.stratum_begin >= rp.n_strata ? 0 : (rp.n_strata - rp.stratum_begin + rp.stratum_step - 1) / rp.stratum_step;
    // A phase leaves kTailChunks of its own chunks' worth of strata to the finer phases after it (a heavy chunk
    // takes several times the mean), but never more than half of what is left; the last phase takes the rest.
    // Without a tree the chunks cost much the same and the samples are cheap (40-50 Gsamples/s): there the small
    // chunks' own overhead shows, and one chunk's worth per phase is enough.
    // With a tree: four chunks' worth, eight for the small trees (up to 18 levels) whose scenes run at six waves per
    // SIMD with short node trips — there the finer tail pays more (measured on the six-wave kernel, RT_TAIL_K 4 / 8:
    // teapot.yaml 6450 / 6605, dual.yaml 6605 / 6815 Msamples/s; lucy.yaml 13580 / 13500, cornell_lucy.yaml 4401 / 4396).
    const uint32_t tail_chunks = tuning_env("RT_TAIL_K") ? (uint32_t)std::max(0, atoi(tuning_env("RT_TAIL_K")))
                                                     : (sc.has_tree ? (tree_depth <= 18u ? 8u : 4u) : 1u);
    // Frames of 512 strata and more start with chunks twice as large (+1 %: fewer flushes, longer stays on a tile).
    // The large meshes (trees deeper than 18 levels) take larger ones still, and take them for the strata slices of
    // several GPUs too — 64 strata from 256 on, 32 from 128 (tools/overlap_probe.py with RT_FIRST_CHUNK, awaited /
    // three in flight, ms: lucy.yaml whole frame 140.6 / 139.8 -> 139.2 / 139.1, 1/2 72.0 -> 70.9, 1/4 38.0 / 37.5 ->
    // 37.2 / 36.3, 1/8 20.7 / 19.9 -> 20.2 / 19.5; cornell_lucy.yaml +0.1...0.3 % throughout).  Not the small trees:
    // dual.yaml's 1/8 slice at 32 takes 55 ms awaited instead of 40 (a heavy chunk late in the frame), 1/4 at 64: 107 for 76.
    const bool large_mesh = sc.has_tree && tree_depth > 18u;
    const uint32_t first_chunk =
        tuning_env("RT_FIRST_CHUNK") ? (uint32_t)std::max(2, atoi(tuning_env("RT_FIRST_CHUNK")))
        : large_mesh             ? (plan.my_strata >= 256u ? 4u * kChunkStrata : (plan.my_strata >= 128u ? 2u * kChunkStrata : kChunkStrata))
                                 : (plan.my_strata >= 512u ? 2u * kChunkStrata : kChunkStrata);
    uint32_t done = 0, groups = 0;
    for (int p = 0; p < kPhases; ++p) {
        const uint32_t per = std::max(2u, first_chunk >> p), left = plan.my_strata - done;
        const uint32_t take = p == kPhases - 1 ? left : (left - std::min(left / 2u, tail_chunks * per)) / per * per;
        plan.phase_first[p] = done, plan.phase_strata[p] = per;
        groups += (take + per - 1) / per;
        plan.phase_end[p] = groups;
        done += take;
    }
    plan.phase_first[kPhases] = plan.my_strata;
    if ((uint64_t)plan.n_tiles * groups >= (1ull << 32)) return hipErrorInvalidValue;
    plan.n_chunks = plan.n_tiles * groups;

    hipError_t e = hipMemsetAsync(d_acc, 0, persistent_acc_bytes(rp.width, rp.height), stream);
    if (e != hipSuccess) return e;
    if ((e = hipMemsetAsync(work_counter, 0, sizeof(uint32_t), stream)) != hipSuccess) return e;
    if (plan.n_chunks == 0) return launch_resolve(d_acc, rp.width, rp.height, d_rgb_sum, stream);
    // Enough blocks to fill every CU at the occupancy the kernel's registers and LDS allow; surplus blocks are
    // harmless (they start when a slot frees up and find the counter exhausted).
    const uint32_t blocks = std::min<uint32_t>(persistent_grid_limit(n_cus), (plan.n_chunks + 3u) / 4u * (256u / kBlock));
    // near-first traversal keeps at most (depth - 1) deferred children
    // (LDS per block at ~1 KB per level: up to 27 levels -> 5 blocks per CU, 35 -> 4, 48 -> 3, 56 -> 2)
    if (tree_depth > 56) return hipErrorInvalidValue;  // caller reports "BVH deeper than the traversal stack"
    plan.stack_levels = std::max(tree_depth, 2u) - 1u;
    plan.gen_words = sc.ext ? kGenTime + 1u : (sc.cam.lens_radius != 0.f ? kGenLens + 2u : kGenLens);
    // Mesh scenes whose blocks leave room for a sixth per CU run the six-wave build (see the top of this file), with
    // the stack's levels beyond kLdsStackCap in the overflow array when the tree is deeper than that (a caller
    // without one — rt_render_device_noalloc — then stays with the default build).
    static const bool no_w6 = tuning_env("RT_NO_W6") != nullptr;  // A/B aid: the five-wave kernel throughout
    PersistentPlan w6 = plan;
    w6.stack_levels = std::min(plan.stack_levels, kLdsStackCap);
    w6.stack_overflow = d_stack_overflow;
    const bool w6_fits = !sc.ext && sc.has_tree && sc.tree_kind == 1u && persistent_lds_bytes(sc, w6) <= kLdsForSixBlocks &&
                         (plan.stack_levels <= kLdsStackCap || d_stack_overflow != nullptr);
    if (w6_fits && !no_w6)
        e = launch_persistent_six(sc, rp, w6, blocks, persistent_lds_bytes(sc, w6), d_acc, counters, work_counter, stats, stream);
    else
        e = launch_any(sc, rp, plan, blocks, d_acc, counters, work_counter, stats, stream);
    if (e != hipSuccess) return e;
    return launch_resolve(d_acc, rp.width, rp.height, d_rgb_sum, stream);
}

hipError_t launch_resolve(const float* d_acc, uint32_t width, uint32_t height, float* d_rgb_sum, hipStream_t stream) {
    hipLaunchKernelGGL(resolve_kernel, dim3((width * height + 255u) / 256u), dim3(256), 0, stream, d_acc, width, height,
                       (width + 7u) >> 3, d_rgb_sum);
    return hipGetLastError();
}

size_t persistent_acc_bytes(uint32_t width, uint32_t height) {
    return (size_t)((width + 7u) >> 3) * ((height + 7u) >> 3) * 192u * sizeof(float);
}

// ---- scratch of rt_render_device_noalloc (include/rt.h): the tile-planar accumulator of a frame of up to
// kNoAllocMaxTiles 8x8 tiles, then one DevCounters block (the chunk cursor).  One copy per GPU, held from module
// load until the process ends.  In host code `scratch` is only a handle for hipGetSymbolAddress: kernels,
// memsets and copies take the device address that call returns, looked up once per device below.
constexpr size_t kStaticAccBytes = (size_t)kNoAllocMaxTiles * 192u * sizeof(float);
constexpr size_t kStaticScratchBytes = kStaticAccBytes + 4096u;
static_assert(sizeof(DevCounters) <= 4096u, "the counters block must fit the scratch's tail");
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mutex;
static void* g_scratch_ptr[kMaxDevices] = {};
static hipEvent_t g_scratch_done[kMaxDevices] = {};  // end of the last rt_render_device_noalloc call on the device

// Once per device, with that device current (rt_scene_create): the device address of `scratch` and the event that
// orders the calls which share it.  *done points into a table that lives as long as the library.
hipError_t persistent_static_scratch(int device, float** d_acc, DevCounters** d_counters, hipEvent_t** done) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        int current = -1;
        hipError_t e = hipGetDevice(&current);
        if (e != hipSuccess) return e;
        if (current != device) return hipErrorInvalidDevice;  // each GPU has its own copy at its own address
        void* ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!ptr) return hipErrorInvalidSymbol;
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = ptr, g_scratch_done[device] = ev;
    }
    *d_acc = (float*)g_scratch_ptr[device];
    *d_counters = (DevCounters*)((unsigned char*)g_scratch_ptr[device] + kStaticAccBytes);
    *done = &g_scratch_done[device];
    return hipSuccess;
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
