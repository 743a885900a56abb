// Device functions shared by the persistent and the wavefront render kernels (gfx950, wave64).
//   sampling     main.rs:165-173, camera.rs:76-87, util.rs:15-35
//   traversal    bvh.rs:262-314 re-ordered near-child-first with interval narrowing; equal-t ties go to the
//                higher in-order leaf slot, which is what the reference's recursion yields
//   intersection surfaces.rs:73-106 (sphere), 246-279 (triangle), 330-354 (quad)
//   shading      scene/mod.rs:86-135, material.rs:99-181
// Triangles and boxes are f32; the few spheres/quads of a scene are intersected in f64 from an f64 ray
// origin (radius-1000 ground spheres and 555-unit Cornell walls are where f32 self-intersection fails).
#pragma once
#include "device_scene.h"

namespace rt {

// ---------------------------------------------------------------- RNG
// Counter-seeded PCG-XSH-RR 64/32: one stream per (seed, pixel, stratum), so an image does not depend on
// how strata are sharded over GPUs or how lanes are scheduled.  24-bit uniforms are exact in f32 AND f64,
// so the f64 oracle draws the very same variates (oracle/rt_ref.cpp restates this).
struct Rng {
    uint64_t s;
};

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ void rng_init(Rng& r, uint64_t seed, uint32_t pixel, uint32_t stratum) {
    r.s = splitmix64(seed ^ splitmix64(((uint64_t)pixel << 32) | stratum));
}
__device__ __forceinline__ uint32_t rng_next(Rng& r) {
    uint64_t old = r.s;
    // (the increment made where it is used, in scalar registers: as a plain literal it was hoisted out of the render
    // loop into a vector-register pair and, in the tightest builds, spilled to scratch and reloaded at every draw)
    uint64_t inc = 1442695040888963407ull;
    asm volatile("" : "+s"(inc));
    r.s = old * 6364136223846793005ull + inc;
    uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t)(old >> 59);
    return (xs >> rot) | (xs << ((32u - rot) & 31u));
}
__device__ __forceinline__ float rng_u01(Rng& r) { return (float)(rng_next(r) >> 8) * 0x1p-24f; }  // [0,1)
__device__ __forceinline__ float rng_pm1(Rng& r) { return rng_u01(r) * 2.0f - 1.0f; }              // [-1,1), exact

struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 f3(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ F3 f3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ F3 f3(float4 v) { return {v.x, v.y, v.z}; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 operator*(float c, F3 v) { return {c * v.x, c * v.y, c * v.z}; }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross(F3 a, F3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ F3 neg(F3 a) { return {-a.x, -a.y, -a.z}; }
// The same products with every fused multiply-add spelled out and the compiler forbidden to fuse or split anything
// else: wherever the triangle test is inlined — first slot of a leaf, second slot, the loop over a long leaf, either
// build of the persistent kernel, the other kernels — a triangle gets the same bits.  (Left to the compiler's
// contraction, two copies of the test in one kernel rounded differently: a triangle's t depended on its place in its
// leaf, so host-built and device-built trees gave different images, and a ray slipped between two coplanar triangles.)
__device__ __forceinline__ float dot_exact(F3 a, F3 b) {
#pragma clang fp contract(off)
    return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z));
}
__device__ __forceinline__ F3 cross_exact(F3 a, F3 b) {
#pragma clang fp contract(off)
    return {fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}

struct D3 {
    double x, y, z;
};
__device__ __forceinline__ D3 d3(const double* p) { return {p[0], p[1], p[2]}; }
// Wave-uniform read-only data addressed through the constant address space: the compiler fetches it with
// scalar loads into SGPRs (one fetch per wave through the scalar cache) instead of 64 identical lane reads.
typedef const double __attribute__((address_space(4)))* UniformF64;
typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));  // uint4 is a class: no copy out of address space 4
typedef const U32x4 __attribute__((address_space(4)))* UniformRecs;
__device__ __forceinline__ D3 d3(UniformF64 p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ D3 d3(F3 v) { return {(double)v.x, (double)v.y, (double)v.z}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 operator*(double c, D3 v) { return {c * v.x, c * v.y, c * v.z}; }
__device__ __forceinline__ double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ D3 cross(D3 a, D3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ F3 tof(D3 v) { return {(float)v.x, (float)v.y, (float)v.z}; }

// Hardware approximations (1 ulp) where the reference's f64 sqrt / division is not what bounds the f32 result:
// the IEEE-exact f32 expansions cost 10-14 instructions each.
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }

// Loads that say where they read: device memory.  The shade stage takes the scene's arrays from pointers it has read
// back from the kernel's arguments (reload_kernarg), and those are generic pointers to the compiler: it fetched through
// flat_load, which counts on the LDS/scalar counter as well as on the vector-memory one — so the wait for a scalar load
// (the next array's address) also waited for every vector load in flight, and the material fetch and the normal fetch
// of a hit, meant to overlap, ran one after the other.  Typed as address space 1 they are global_load again.
typedef float F32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_global(const float4* p) {
    const F32x4 v = *(const F32x4 __attribute__((address_space(1)))*)(uintptr_t)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int load_global(const int* p) { return *(const int __attribute__((address_space(1)))*)(uintptr_t)p; }
__device__ __forceinline__ double load_global(const double* p) {
    return *(const double __attribute__((address_space(1)))*)(uintptr_t)p;
}

// util.rs:15-24 / 26-35.  Two samplers with the same distributions:
//   DIRECT == false  the reference's rejection loops, same draw order (parity mode: the oracle's default);
//   DIRECT == true   closed-form maps with a fixed number of draws (2 for a direction or a disk point, 3 for
//                    a ball point): no data-dependent loop, so no lane waits for the unluckiest lane of the
//                    wave.  oracle/rt_ref.cpp restates both; v_sin/v_cos take their argument in turns.
template <bool DIRECT>
__device__ __forceinline__ F3 rng_unit_vector(Rng& r) {  // uniform on the unit sphere: unit(in_unit_sphere)
    if (DIRECT) {
        const float z = 1.0f - 2.0f * rng_u01(r);
        const float turn = rng_u01(r);
        const float rad = fast_sqrt(fmaxf(0.0f, 1.0f - z * z));
        return {rad * __builtin_amdgcn_cosf(turn), rad * __builtin_amdgcn_sinf(turn), z};
    }
    while (true) {
        float x = rng_pm1(r), y = rng_pm1(r), z = rng_pm1(r);
        const float l2 = x * x + y * y + z * z;
        if (l2 < 1.0f) {
            const float inv_len = fast_rsq(l2);
            return {inv_len * x, inv_len * y, inv_len * z};
        }
    }
}
template <bool DIRECT>
__device__ __forceinline__ F3 rng_in_unit_sphere(Rng& r) {  // uniform in the unit ball
    if (DIRECT) {
        const F3 u = rng_unit_vector<true>(r);
        const float rad = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(rng_u01(r)) * (1.0f / 3.0f));  // cbrt
        return rad * u;
    }
    while (true) {
        float x = rng_pm1(r), y = rng_pm1(r), z = rng_pm1(r);
        if (x * x + y * y + z * z < 1.0f) return {x, y, z};
    }
}
template <bool DIRECT>
__device__ __forceinline__ void rng_in_unit_disk(Rng& r, float& x, float& y) {
    if (DIRECT) {
        const float rad = fast_sqrt(rng_u01(r));
        const float turn = rng_u01(r);
        x = rad * __builtin_amdgcn_cosf(turn), y = rad * __builtin_amdgcn_sinf(turn);
        return;
    }
    while (true) {
        x = rng_pm1(r), y = rng_pm1(r);
        if (x * x + y * y < 1.0f) return;
    }
}

// ---------------------------------------------------------------- ray state
struct Ray {
    D3 o;  // f64 origin: exact input of the sphere/quad tests and of o + t*d
    F3 d;  // direction, NOT normalised (camera.rs:84, material.rs:110)
    float time;  // shutter time in [0, 1) (camera.rs:80), kept along the path; read by the EXT builds only
};

struct HitRec {
    float t;
    int slot;      // leaf slot of the closest hit, -1 = miss
    float bu, bv;  // triangle barycentrics
};

// Sphere/quad hits keep their f64 root in the (otherwise unused) barycentric slots, so the shade step can
// place the hit point on the surface without repeating the intersection.
__device__ __forceinline__ void hit_set_t64(HitRec& h, double t) {
    h.bu = __int_as_float(__double2loint(t));
    h.bv = __int_as_float(__double2hiint(t));
}
__device__ __forceinline__ double hit_get_t64(const HitRec& h) {
    return __hiloint2double(__float_as_int(h.bv), __float_as_int(h.bu));
}

struct TraceStats {
    uint32_t nodes = 0, tris = 0, spheres = 0, quads = 0;
};

// Camera ray for (pixel, stratum): main.rs:157-174 + camera.rs:76-87, in two halves so that the persistent kernel
// can draw a whole batch of them at once: the draws (direction for a zero lens offset, lens sample scaled by the
// lens radius, shutter time), and the ray they make.
struct CameraSample {
    F3 d;          // ll + u*hor + v*ver - origin (before the lens offset is added back)
    float rx, ry;  // lens_radius * unit-disk sample
    float time;
};
template <bool DIRECT>
__device__ __forceinline__ CameraSample camera_sample(const DevCamera& cam, const DevRenderParams& rp, uint32_t pi, uint32_t pj,
                                                      uint32_t stratum, Rng& rng) {
    const uint32_t si = rp.sn_magic ? __umulhi(stratum, rp.sn_magic) : stratum / rp.sn, sj = stratum - si * rp.sn;
    float ju = -0.5f + rp.recip_sn * ((float)si + rng_u01(rng));
    float jv = -0.5f + rp.recip_sn * ((float)sj + rng_u01(rng));
    float s = (float)(pi + 1) + ju;  // in pixel units; *hor_px == (centeru + jitteru*pixel_du)*horizontal
    float t = (float)(pj + 1) + jv;
    CameraSample cs;
    rng_in_unit_disk<DIRECT>(rng, cs.rx, cs.ry);  // always drawn, even for a pinhole (camera.rs:77)
    cs.rx *= cam.lens_radius, cs.ry *= cam.lens_radius;
    cs.time = rng_u01(rng);  // shutter time (camera.rs:80): without effect on stationary scenes
    cs.d = f3(cam.ll_rel) + s * f3(cam.hor_px) + t * f3(cam.ver_px);
    return cs;
}
__device__ __forceinline__ Ray camera_ray_of(const DevCamera& cam, const CameraSample& cs) {
    const F3 off = cs.rx * f3(cam.u) + cs.ry * f3(cam.v);
    Ray ray;
    ray.o = d3(cam.origin) - d3(off);
    ray.d = cs.d + off;
    ray.time = cs.time;
    return ray;
}
// Wave-uniform parameters that only rare code needs (the camera where samples are drawn, the hand-out plan where a
// chunk is fetched) are read again from memory there, through the constant address space (scalar loads), instead of
// living in SGPRs across every stage: the kernel holds more scalars than the 102 registers there are, and what
// does not fit is parked in vector-register lanes and moved back with v_readlane on the hot path.  The address
// passes through an empty asm so that the compiler cannot hoist the loads back to the kernel's entry.
template <class T>
__device__ __forceinline__ T reload_uniform(uintptr_t a) {
    asm volatile("" : "+s"(a));
    typedef const uint32_t __attribute__((address_space(4)))* UniformWords;
    const UniformWords w = (UniformWords)a;
    static_assert(sizeof(T) % 4 == 0, "copied word by word");
    uint32_t words[sizeof(T) / 4];
#pragma unroll
    for (uint32_t i = 0; i < sizeof(T) / 4; ++i) words[i] = w[i];
    T out;
    __builtin_memcpy(&out, words, sizeof out);
    return out;
}
// a by-value kernel argument, by its offset in the kernarg segment.  KERNELS ONLY: in a called (noinline) function
// the intrinsic lowers to a null pointer — pass the address down and use reload_uniform there.
template <class T>
__device__ __forceinline__ T reload_kernarg(size_t offset) {
    return reload_uniform<T>((uintptr_t)__builtin_amdgcn_kernarg_segment_ptr() + offset);
}
template <bool DIRECT>
__device__ __forceinline__ Ray camera_ray(const DevScene& sc, const DevRenderParams& rp, uint32_t pi, uint32_t pj,
                                          uint32_t stratum, Rng& rng) {
    return camera_ray_of(sc.cam, camera_sample<DIRECT>(sc.cam, rp, pi, pj, stratum, rng));
}

// ---------------------------------------------------------------- primitive tests
// f64 reciprocal / square root from the f32 hardware approximation plus one Newton step in f64: relative
// error ~1e-14, a third of the instructions of the IEEE-exact expansions.  (The exact-parity kernel in
// primary64.hip uses the IEEE operations; these are for the render path, whose arguments are well inside
// the f32 exponent range.)
__device__ __forceinline__ double rcp64(double b) {
    const double r = (double)__builtin_amdgcn_rcpf((float)b);
    return fma(fma(-b, r, 1.0), r, r);
}
__device__ __forceinline__ double sqrt64(double x) {  // x > 0
    const double y = (double)__builtin_amdgcn_rsqf((float)x);  // ~ 1/sqrt(x)
    const double s = x * y;                                    // ~ sqrt(x)
    return fma(fma(-s, s, x), 0.5 * y, s);
}

// Sphere, f64 (surfaces.rs:73-106).  Returns the first root inside [tmin, tmax], or -1.
// Sphere::center, surfaces.rs:52-54; the stationary builds read c0 only.
template <bool EXT, class P0, class P1>
__device__ __forceinline__ D3 sphere_centre(P0 c0, P1 c1, float time) {
    if (!EXT) return d3(c0);
    const double t = (double)time;
    return (1.0 - t) * d3(c0) + t * d3(c1);
}
__device__ __forceinline__ double sphere_root(D3 centre, double r, D3 o, D3 d, double tmin, double tmax) {
    D3 oc = o - centre;
    double a = dot(d, d);
    double half_b = dot(oc, d);
    double c = dot(oc, oc) - r * r;
    double disc = half_b * half_b - a * c;
    if (disc > 0.0) {
        double root = sqrt64(disc);
        double ia = rcp64(a);
        double t0 = (-half_b - root) * ia;
        if (tmin <= t0 && t0 <= tmax) return t0;
        double t1 = (-half_b + root) * ia;
        if (tmin <= t1 && t1 <= tmax) return t1;
    }
    return -1.0;
}

// Aabb::hit (trace.rs:131-149) on the box of the reference tree's leaf that holds a surface whose own
// reference bounding box does not cover it: a negative-radius sphere (Sphere::bounding_box inverts) or a
// parallelogram whose sides share an axis (Quad::bounding_box spans only q and q+u+v).  Such a box adds too
// little to its leaf's box, and the reference tests the surface only for rays that hit the leaf's box
// (BVHLeafNode::hit, bvh.rs:300-303) — never, for a negative sphere alone in a leaf.  gate = lo.xyz hi.xyz.
__device__ __attribute__((noinline)) bool leaf_gate_hit(const double* gate, D3 o, D3 d, double t_min) {
    const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    double t_max = __builtin_inf();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // 1/d, +-inf for d = 0 like the reference (trace.rs:18)
        const double inv_d = dd[a] == 0.0 ? __builtin_copysign(__builtin_inf(), dd[a]) : rcp64(dd[a]);
        double t0 = (gate[a] - oo[a]) * inv_d, t1 = (gate[3 + a] - oo[a]) * inv_d;
        if (inv_d < 0.0) {
            const double tmp = t0;
            t0 = t1, t1 = tmp;
        }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max < t_min) return false;
    }
    return true;
}
template <bool EXT, class P>
__device__ __forceinline__ double sphere_root_gated(P sp, P gate, P c1, float time, D3 o, D3 d, double tmin, double tmax) {
    double t = sphere_root(sphere_centre<EXT>(sp, c1, time), sp[3], o, d, tmin, tmax);
    // gate[0] == -inf: the host found the leaf's box to enclose the sphere, so every hit passes
    if (t >= 0.0 && sp[3] < 0.0 && gate[0] != -__builtin_inf() && !leaf_gate_hit((const double*)gate, o, d, tmin)) t = -1.0;
    return t;
}

// Quad, f64 (surfaces.rs:330-354).  alpha = w·(p×v) = p·(v×w), beta = w·(u×p) = p·(w×u).
template <class P>
__device__ __forceinline__ double quad_root(P q, D3 o, D3 d, double tmin, double tmax) {
    D3 n = d3(q);
    double denom = dot(n, d);
    if (fabs(denom) < 1e-8) return -1.0;
    double t = (q[3] - dot(n, o)) * rcp64(denom);
    if (t < 1e-4 || !(tmin <= t && t <= tmax)) return -1.0;
    D3 p = (o + t * d) - d3(q + 4);
    double alpha = dot(p, d3(q + 7));
    double beta = dot(p, d3(q + 10));
    if (!(0.0 <= alpha && alpha < 1.0 && 0.0 <= beta && beta < 1.0)) return -1.0;
    return t;
}

template <class P>
__device__ __forceinline__ double quad_root_gated(P q, P gate, D3 o, D3 d, double tmin, double tmax) {
    double t = quad_root(q, o, d, tmin, tmax);
    if (t >= 0.0 && gate[0] != -__builtin_inf() && !leaf_gate_hit((const double*)gate, o, d, tmin)) t = -1.0;
    return t;
}

// ---------------------------------------------------------------- traversal
// Hit encoding: slot >= 0 is a leaf slot of the render tree, slot <= -2 is loose primitive (-2 - slot),
// -1 is a miss.  Closest wins; an exact tie goes to the surface that comes later in the reference tree's
// in-order leaf sequence (its "rank"), which is what the reference's recursion yields (bvh.rs:272-273, 308).
//
// The traversal is written as three steps — loose_hits (once per ray), node_step (one inner node),
// prim_step (one leaf slot) — so that the wavefront `extend` kernel can run them as the classic
// while-while loop (trace_closest) and the persistent kernel can schedule them stage by stage.
//
// `stack` points at this lane's column of an LDS array laid out [level][lane] (stride = block size), so
// a push/pop is one conflict-free ds_write/ds_read.  The host checks tree depth <= stack levels.
constexpr int kTraversalDone = 0x7fffffff;  // "node" value once the stack has run empty

struct Closest {
    HitRec h;
    int rank;       // rank of h.slot's surface (ties)
    uint32_t meta;  // meta word of h.slot (tree hits): type and material / index for the shade step
};
// t of the closest hit as exactly as it is known: the f64 root of a sphere or quad (loose, or a tree hit that is not a
// triangle), the f32 t of a triangle; +inf before the first hit.
__device__ __forceinline__ double closest_t64(const Closest& c) {
    const bool has_root = c.h.slot <= -2 || (c.h.slot >= 0 && (c.meta & kMetaTypeMask) != 1u);
    return has_root ? hit_get_t64(c.h) : (double)c.h.t;
}
__device__ __forceinline__ void closest_reset(Closest& c) {
    c.h.t = __builtin_inff();
    c.h.slot = -1;
    c.h.bu = c.h.bv = 0.f;
    c.rank = -1;
    c.meta = 0u;
}

// Loose spheres/quads: every lane tests the same primitive at the same time (no divergence); f64 because
// radius-1000 spheres and 555-unit walls need it.  `recs` are the self-contained records of
// DevScene::loose_rec — in LDS for the persistent kernels (one broadcast read per 16 B, no dependent
// global round trips inside the loop), in global memory for the others.
__device__ __forceinline__ const double* loose_data(const uint4* recs, uint32_t i) {
    return (const double*)(recs + kLooseRecQuads * i + 1);
}
__device__ __forceinline__ UniformF64 loose_data(UniformRecs recs, uint32_t i) {
    return (UniformF64)(recs + kLooseRecQuads * i + 1);
}
__device__ __forceinline__ UniformRecs uniform_recs(const uint4* p) { return (UniformRecs)(uintptr_t)p; }
// `recs`: const uint4* (LDS or global, read lane by lane) or UniformRecs (global, read once per wave).
template <bool STATS, bool EXT = false, class Recs>
__device__ __forceinline__ void loose_hits(Recs recs, uint32_t n_loose, const Ray& ray, float tmin,
                                           Closest& best, TraceStats& st) {
    const D3 dd = d3(ray.d);
    // The closest root so far in f64: these tests are the reference's own arithmetic, and an exact equal-t tie between
    // two of them (coplanar quads) is decided on the f64 roots, as the reference decides it — on best.h.t, their f32
    // rounding, a root that rounded DOWN shut the tie's other party out (t64 > (double)best.h.t) whatever its rank.
    double best64 = closest_t64(best);
    for (uint32_t i = 0; i < n_loose; ++i) {
        const auto m = recs[kLooseRecQuads * i];
        double t;
        if (m.x == 0u) {
            if (STATS) st.spheres++;
            t = sphere_root_gated<EXT>(loose_data(recs, i), loose_data(recs, i) + 4, loose_data(recs, i) + 10, ray.time, ray.o,
                                       dd, (double)tmin, best64);
        } else {
            if (STATS) st.quads++;
            t = quad_root_gated(loose_data(recs, i), loose_data(recs, i) + 13, ray.o, dd, (double)tmin, best64);
        }
        // (a root lies in [tmin, best64] or is -1)
        if (t >= 0.0 && (t < best64 || (int)m.z > best.rank)) {
            best64 = t;
            best.h.t = (float)t, best.h.slot = -2 - (int)i, best.rank = (int)m.z;
            hit_set_t64(best.h, t);
        }
    }
}

struct SlabRay {  // f32 ray in the form the fma slab test wants, in grid units: the ray reaches plane q of axis a
                  // at t = q * a_[a] + b_[a], with a_ = grid_step / d and b_ = (grid_lo - o) / d
    float ax, ay, az, bx, by, bz;
};
struct DevGrid {  // DevScene::grid_lo, grid_step
    float lo[3], step[3];
};
__device__ __forceinline__ SlabRay slab_ray(const DevGrid& grid, const Ray& ray) {
    SlabRay r;
    const F3 o = tof(ray.o), d = ray.d;
    // A zero direction component would make q*inf + (-o)*inf a NaN or a wrongly signed infinity in the fma slab
    // form; clamp |d| away from zero (2^-80) so every slab bound stays finite and correctly ordered.
    const float ooeps = 0x1p-80f;
    const float idx = __builtin_amdgcn_rcpf(fabsf(d.x) > ooeps ? d.x : copysignf(ooeps, d.x));
    const float idy = __builtin_amdgcn_rcpf(fabsf(d.y) > ooeps ? d.y : copysignf(ooeps, d.y));
    const float idz = __builtin_amdgcn_rcpf(fabsf(d.z) > ooeps ? d.z : copysignf(ooeps, d.z));
    r.ax = grid.step[0] * idx, r.ay = grid.step[1] * idy, r.az = grid.step[2] * idz;
    r.bx = (grid.lo[0] - o.x) * idx, r.by = (grid.lo[1] - o.y) * idy, r.bz = (grid.lo[2] - o.z) * idz;
    return r;
}
__device__ __forceinline__ SlabRay slab_ray(const DevScene& sc, const Ray& ray) {
    static_assert(offsetof(DevScene, grid_step) == offsetof(DevScene, grid_lo) + 12, "DevGrid mirrors the two arrays");
    return slab_ray(DevGrid{{sc.grid_lo[0], sc.grid_lo[1], sc.grid_lo[2]}, {sc.grid_step[0], sc.grid_step[1], sc.grid_step[2]}}, ray);
}

// Does the ray reach the render tree's bounds (grid planes 0 and 65535) at all within [tmin, best_t]?  Most
// sky/ground rays do not, and then skip the traversal stages entirely.
__device__ __forceinline__ bool hits_tree_bounds(const DevScene& sc, const SlabRay& r, float tmin, float best_t) {
    const float ax = r.bx, bx = fmaf(65535.0f, r.ax, r.bx);
    const float ay = r.by, by = fmaf(65535.0f, r.ay, r.by);
    const float az = r.bz, bz = fmaf(65535.0f, r.az, r.bz);
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), best_t * 1.0000004f));
    return tn <= tf;
}

// A lane's traversal stack.  LdsStack: every level in LDS (this lane's column of the [level][lane] array).  A kernel
// may bring its own type with the same two members (render_persistent.hip's SplitStack keeps only the first levels
// in LDS).
struct LdsStack {
    int* column;
    int stride;
    __device__ __forceinline__ void store(int level, int v) const { column[level * stride] = v; }
    __device__ __forceinline__ int load(int level) const { return column[level * stride]; }
};
template <class Stack>
__device__ __forceinline__ int stack_pop(const Stack& stack, int& sp) {
    if (sp == 0) return kTraversalDone;
    --sp;
    return stack.load(sp);
}
__device__ __forceinline__ int stack_pop(int* stack, int stride, int& sp) { return stack_pop(LdsStack{stack, stride}, sp); }

// One inner node: test both children's boxes against [tmin, best_t], go to the nearer hit child, defer the
// other.  Returns the next node (>= 0 inner, < 0 leaf ~first_slot, kTraversalDone).
// A child's box: per axis one word, centre (u16) | half-extent (binary16, rounded up) << 16, in grid units.  One fma
// reaches the centre plane, one each the two faces (m -/+ e*|a|: whichever the ray's direction, the minus side is the
// near one), so no per-axis min/max is needed — those issue at half the fma's rate
// (profiles/r1_valu_issue_rates.txt) — and the half-extent enters v_fma_mix_f32 as it is, without a conversion.
__device__ __forceinline__ void child_slab(const uint4 c, const SlabRay& r, float tmin, float tmax_pad, float& tn, float& tf) {
    typedef _Float16 Half2 __attribute__((ext_vector_type(2)));
    const float mx = fmaf((float)(c.x & 0xffffu), r.ax, r.bx), ex = (float)__builtin_bit_cast(Half2, c.x).y;
    const float my = fmaf((float)(c.y & 0xffffu), r.ay, r.by), ey = (float)__builtin_bit_cast(Half2, c.y).y;
    const float mz = fmaf((float)(c.z & 0xffffu), r.az, r.bz), ez = (float)__builtin_bit_cast(Half2, c.z).y;
    const float nx = fmaf(-ex, fabsf(r.ax), mx), fx = fmaf(ex, fabsf(r.ax), mx);
    const float ny = fmaf(-ey, fabsf(r.ay), my), fy = fmaf(ey, fabsf(r.ay), my);
    const float nz = fmaf(-ez, fabsf(r.az), mz), fz = fmaf(ez, fabsf(r.az), mz);
    tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, tmin));
    tf = fminf(fminf(fx, fy), fminf(fz, tmax_pad));
}
template <bool STATS, class Stack>
__device__ __forceinline__ int node_step(const DevScene& sc, const SlabRay& r, float tmin, float best_t, int node,
                                         const Stack& stack, int& sp, TraceStats& st) {
    if (STATS) st.nodes++;
    const uint4 n0 = sc.nodes[2 * node + 0];
    const uint4 n1 = sc.nodes[2 * node + 1];
    const float tmax_pad = best_t * 1.0000004f;
    float t0n, t0f, t1n, t1f;
    child_slab(n0, r, tmin, tmax_pad, t0n, t0f);
    child_slab(n1, r, tmin, tmax_pad, t1n, t1f);
    const bool h0 = t0n <= t0f, h1 = t1n <= t1f;
    const int c0 = (int)n0.w, c1 = (int)n1.w;
    if (h0 && h1) {
        const bool swap = t1n < t0n;
        stack.store(sp, swap ? c0 : c1);
        ++sp;
        return swap ? c1 : c0;
    }
    if (h0) return c0;
    if (h1) return c1;
    return stack_pop(stack, sp);
}
template <bool STATS>
__device__ __forceinline__ int node_step(const DevScene& sc, const SlabRay& r, float tmin, float best_t, int node,
                                         int* stack, int stride, int& sp, TraceStats& st) {
    return node_step<STATS>(sc, r, tmin, best_t, node, LdsStack{stack, stride}, sp, st);
}

// One leaf slot whose record (p0 p1 p2) is in registers.  Returns true when it was the last slot of its leaf.
template <bool STATS, bool EXT = false, int TREE = 0>
__device__ __forceinline__ bool prim_test(const DevScene& sc, const Ray& ray, const SlabRay& r, float tmin, int slot,
                                          float4 p0, float4 p1, float4 p2, Closest& best, TraceStats& st) {
    const uint32_t meta = __float_as_uint(p0.w);
    const uint32_t type = meta & kMetaTypeMask;
    float t = -1.f, bu = 0.f, bv = 0.f;
    double t64 = -1.0, best64 = 0.0;  // best64: the closest hit's t as closest_t64 gives it (sphere / quad tests only)
    // TREE: what the host found in the render tree — 1: nothing but triangles (every mesh scene), 2: nothing but
    // spheres (one-weekend), 3: nothing but quads (the cornell example), 0: a mix.  The tests that cannot occur are
    // not compiled in: each costs registers.
    if (TREE == 1 || (TREE == 0 && type == 1u)) {  // triangle, Möller–Trumbore (surfaces.rs:246-279) with host-precomputed edges
        if (STATS) st.tris++;
        const F3 e1 = f3(p1), e2 = f3(p2);
        const F3 pvec = cross_exact(ray.d, e2);
        const float det = dot_exact(e1, pvec);
        if (!(-1e-8f < det && det < 1e-8f)) {
            const float inv_det = __builtin_amdgcn_rcpf(det);
            const F3 tvec = tof(ray.o) - f3(p0);  // the f32 origin is re-derived here rather than kept in registers
            // The reference's edges are inclusive (u in [0, 1], v >= 0, u + v <= 1, surfaces.rs:262-270) and its f64
            // barycentrics leave no ray between two triangles that share an edge; f32 ones are off by ~1e-7 either
            // way, each triangle with its own rounding, and one sample in 4e6 slipped through a shared edge of a flat
            // mesh (tests/test_gpu_analytic.py::test_furnace_flat_triangle_mesh).  The edges therefore take in
            // kEdgeSlack more: a ray that close to a shared edge hits both triangles at very nearly the same t, the
            // nearer one counts (an exact tie goes by rank, like an exact edge hit in the reference).
            constexpr float kEdgeSlack = 0x1p-20f;
            const float u = dot_exact(tvec, pvec) * inv_det;
            if (u >= -kEdgeSlack && u <= 1.f + kEdgeSlack) {
                const F3 qvec = cross_exact(tvec, e1);
                const float v = dot_exact(ray.d, qvec) * inv_det;
                if (!(v < -kEdgeSlack || u + v > 1.f + kEdgeSlack)) {
                    const float tt = dot_exact(e2, qvec) * inv_det;
                    if (!(tt < 1e-4f) && tmin <= tt) t = tt, bu = u, bv = v;
                }
            }
        }
    } else if (TREE == 2 || (TREE == 0 && type == 0u)) {  // only scenes with more than kMaxLoose spheres+quads keep them in the tree
        if (STATS) st.spheres++;
        const uint32_t si = meta >> kMetaIndexShift;
        best64 = closest_t64(best);
        t64 = sphere_root_gated<EXT>(sc.spheres + 4 * si, sc.sphere_gate + 6 * si, sc.sphere_end + 3 * si, ray.time, ray.o,
                                     d3(ray.d), (double)tmin, best64);
        t = (float)t64;
    } else {
        if (STATS) st.quads++;
        best64 = closest_t64(best);
        t64 = quad_root_gated(sc.quads + 16 * (meta >> kMetaIndexShift), sc.quad_gate + 6 * (meta >> kMetaIndexShift), ray.o,
                              d3(ray.d), (double)tmin, best64);
        t = (float)t64;
    }
    if (t >= tmin && t <= best.h.t) {
        const int rank = __float_as_int(p1.w);
        // a sphere's or quad's root is compared as the f64 number it is (see loose_hits); a triangle's t is f32
        const bool nearer = (TREE != 1 && type != 1u) ? t64 < best64 : t < best.h.t;
        if (nearer || rank > best.rank) {
            best.h.t = t, best.h.slot = slot, best.h.bu = bu, best.h.bv = bv, best.rank = rank, best.meta = meta;
            if (TREE != 1 && type != 1u) hit_set_t64(best.h, t64);
        }
    }
    return (meta & kMetaLast) != 0u;
}
template <bool STATS, bool EXT = false, int TREE = 0>
__device__ __forceinline__ bool prim_step(const DevScene& sc, const Ray& ray, const SlabRay& r, float tmin, int slot,
                                          Closest& best, TraceStats& st) {
    return prim_test<STATS, EXT, TREE>(sc, ray, r, tmin, slot, sc.prims[3 * slot], sc.prims[3 * slot + 1], sc.prims[3 * slot + 2], best,
                            st);
}
// A whole leaf.  The records of its first two slots are fetched together (render-tree leaves hold at most two
// in all but degenerate cases; the array is padded by one slot), so a two-triangle leaf costs one memory
// round trip, not two.
template <bool STATS, bool EXT = false, int TREE = 0>
__device__ __forceinline__ void leaf_step(const DevScene& sc, const Ray& ray, const SlabRay& r, float tmin, int slot,
                                          Closest& best, TraceStats& st) {
    if (TREE >= 2) {  // spheres and quads fetch their own data by index: one copy of the test, slot by slot
        while (!prim_step<STATS, EXT, TREE>(sc, ray, r, tmin, slot, best, st)) ++slot;
        return;
    }
    const float4 a0 = sc.prims[3 * slot + 0], a1 = sc.prims[3 * slot + 1], a2 = sc.prims[3 * slot + 2];
    const float4 b0 = sc.prims[3 * slot + 3], b1 = sc.prims[3 * slot + 4], b2 = sc.prims[3 * slot + 5];
    // the first triangle's v0 is needed only once its determinant has passed, and the compiler moves its fetch down
    // there: a second memory round trip inside the leaf trip for nearly every lane.  A use right here keeps the fetch
    // among the others.
    asm volatile("" ::"v"(a0.x), "v"(a0.y), "v"(a0.z));
    if (prim_test<STATS, EXT, TREE>(sc, ray, r, tmin, slot, a0, a1, a2, best, st)) return;
    if (prim_test<STATS, EXT, TREE>(sc, ray, r, tmin, slot + 1, b0, b1, b2, best, st)) return;
    slot += 2;
    while (!prim_step<STATS, EXT, TREE>(sc, ray, r, tmin, slot, best, st)) ++slot;
}

// The classic while-while loop over the steps (used by wf_extend and the f32 AOV kernel).
template <bool STATS>
__device__ __forceinline__ HitRec trace_closest(const DevScene& sc, const Ray& ray, float tmin, int* stack,
                                                int stack_stride, TraceStats& st) {
    Closest best;
    closest_reset(best);
    loose_hits<STATS>(sc.loose_rec, sc.n_loose, ray, tmin, best, st);
    if (!sc.has_tree) return best.h;
    const SlabRay r = slab_ray(sc, ray);
    int sp = 0;
    int node = hits_tree_bounds(sc, r, tmin, best.h.t) ? 0 : kTraversalDone;  // the root is always an inner node
    while (node != kTraversalDone) {
        while (node >= 0 && node != kTraversalDone) node = node_step<STATS>(sc, r, tmin, best.h.t, node, stack, stack_stride, sp, st);
        if (node == kTraversalDone) break;
        int slot = ~node;
        while (!prim_step<STATS>(sc, ray, r, tmin, slot, best, st)) ++slot;
        node = stack_pop(stack, stack_stride, sp);
    }
    return best.h;
}

// ---------------------------------------------------------------- shading
__device__ __forceinline__ F3 reflect(F3 v, F3 n) { return v - (2.0f * dot(v, n)) * n; }  // material.rs:171-173

// One bounce (one iteration of the loop at scene/mod.rs:103-125).  `lsum` gathers radiance in the forward
// form  L += T*emitted; T *= albedo  (equal to resolve_color's back-to-front fold).  Returns whether the
// path continues; `depth` follows the reference's counter including the metal re-sampling quirk (Q1).
// ---- constant media (EXT builds): ConstantMedium::hit, surfaces.rs:611-646.
// The reference evaluates a medium inside the BVH traversal and draws its free path from thread_rng() — not
// even from the seeded generator.  Here the media are evaluated after the surface trace, in index order: a
// medium whose boundary the ray crosses in front of the closest hit so far draws one uniform from the path's
// own stream (the oracle does the same, so both follow the same paths).  The boundary tests keep the
// reference's semantics: a sphere is entered at its first root whatever its sign (Interval::UNIVERSE), a box
// of quads only at parameters >= 1e-4 (Quad::hit's own cut-off, surfaces.rs:337), so a ray that starts inside
// a box-shaped medium sees its far wall as the entry and finds no exit: such a medium is invisible from inside.
// The boundary crossings are a property of the ray, the free paths are drawn per attempt (a ray whose hit does
// not scatter is traced again, see shade_bounce): media_spans does the geometry once, media_sample the draws.
struct MediaSpans {
    double t_in[kMaxMedia], t_out[kMaxMedia];  // t_out < t_in: the ray does not cross this medium's boundary
    double len, inv_len;                       // |d| and its reciprocal
};
// (Not inlined, like leaf_gate_hit: the kernel's register allocation then does not have to hold this function's
// working set — a box's six roots — next to the path state; its operands travel as plain pointers and values.)
// The arguments of a function that is not inlined arrive in vector registers; for data that is the same for every
// lane, make the address a scalar again so that the constant-address-space reads below become scalar loads.
__device__ __forceinline__ uintptr_t uniform_address(const void* p) {
    const uintptr_t a = (uintptr_t)p;
    // (the builtin returns int: widen through uint32_t, or a low half with bit 31 set smears into the high half)
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    return ((uintptr_t)hi << 32) | (uintptr_t)lo;
}
// The media and their boundaries are the same for every lane: scalar loads, not two dozen dependent vector round
// trips per ray (media_spans took 46 000 cycles of a cornell-smoke shade trip with those, 9 500 without).
typedef const int32_t __attribute__((address_space(4)))* UniformI32;
// -(1 / density) of medium k through a scalar load; `media` must be a wave-uniform address
__device__ __forceinline__ double medium_neg_inv_density(uintptr_t media, uint32_t k) {
    return ((UniformF64)(media + 32u * k))[2];
}
__device__ __attribute__((noinline)) void media_spans_impl(const DevMedium* media_g, uint32_t n_media, const double* spheres_g,
                                                           const double* quads_g, D3 o, D3 dd, MediaSpans* out) {
    const UniformI32 media = (UniformI32)uniform_address(media_g);  // DevMedium: boundary first count material | f64 f64
    const UniformF64 spheres = (UniformF64)uniform_address(spheres_g), quads = (UniformF64)uniform_address(quads_g);
    MediaSpans sp;
    const double inf = (double)__builtin_inff();
    sp.len = sqrt64(dot(dd, dd)), sp.inv_len = rcp64(sp.len);
#pragma unroll
    for (uint32_t k = 0; k < kMaxMedia; ++k) {
        sp.t_in[k] = 0.0, sp.t_out[k] = -1.0;
        if (k >= n_media) continue;
        struct {
            int32_t boundary, first, count;
        } m = {media[8 * k], media[8 * k + 1], media[8 * k + 2]};
        if (m.boundary == 0) {
            const UniformF64 s = spheres + 4 * m.first;
            const D3 oc = o - d3(s);
            const double a = dot(dd, dd), half_b = dot(oc, dd), c = dot(oc, oc) - s[3] * s[3];
            const double disc = half_b * half_b - a * c;
            if (disc > 0.0) {
                const double root = sqrt64(disc), ia = rcp64(a);
                const double t0 = (-half_b - root) * ia, t1 = (-half_b + root) * ia;
                if (t1 >= t0 + 0.0001) sp.t_in[k] = t0, sp.t_out[k] = t1;  // the second Sphere::hit, on (t0 + 0.0001, inf)
            }
        } else if (m.count > 6) {
            // a boundary of any number of quads (the reference takes any Hittable, surfaces.rs:601-608): the closest
            // root, then the closest root beyond it — the two Quad::hit passes as they stand, nothing kept between them
            double first = inf, second = inf;
            for (int32_t q = 0; q < m.count; ++q) {
                const double t = quad_root(quads + 16 * (m.first + q), o, dd, -inf, inf);
                if (t >= 0.0 && t <= first) first = t;
            }
            if (first < inf) {
                for (int32_t q = 0; q < m.count; ++q) {
                    const double t = quad_root(quads + 16 * (m.first + q), o, dd, -inf, inf);
                    if (t >= first + 0.0001 && t <= second) second = t;
                }
                if (second < inf) sp.t_in[k] = first, sp.t_out[k] = second;
            }
        } else {
            // the six sides once: the second Quad::hit pass, on (first + 0.0001, inf), sees the same roots
            double tq[6], first = inf, second = inf;
#pragma unroll
            for (int32_t q = 0; q < 6; ++q) {
                tq[q] = q < m.count ? quad_root(quads + 16 * (m.first + q), o, dd, -inf, inf) : -1.0;
                if (tq[q] >= 0.0 && tq[q] <= first) first = tq[q];
            }
            if (first < inf) {
#pragma unroll
                for (int32_t q = 0; q < 6; ++q)
                    if (tq[q] >= first + 0.0001 && tq[q] <= second) second = tq[q];
                if (second < inf) sp.t_in[k] = first, sp.t_out[k] = second;
            }
        }
    }
    *out = sp;
}
__device__ __forceinline__ void media_spans(const DevScene& sc, const Ray& ray, MediaSpans& sp) {
    media_spans_impl(sc.media, sc.n_media, sc.spheres, sc.quads, ray.o, d3(ray.d), &sp);
}
// The free path -ln(U)/density takes the hardware's f32 logarithm of the 24-bit uniform (the oracle's is f64:
// the two decide differently only when the free path equals the span to ~1e-7).
__device__ __forceinline__ void media_sample(const DevScene& sc, const Ray& ray, float tmin, Rng& rng, Closest& best,
                                             const MediaSpans& sp) {
    const double inf = (double)__builtin_inff();
#pragma unroll
    for (uint32_t k = 0; k < kMaxMedia; ++k) {
        if (k >= sc.n_media || !(sp.t_out[k] >= sp.t_in[k])) continue;
        const double t_best = best.h.slot == -1 ? inf
                              : (best.h.slot >= 0 && (best.meta & kMetaTypeMask) == 1u ? (double)best.h.t : hit_get_t64(best.h));
        double t_in = fmax(sp.t_in[k], (double)tmin);
        const double t_out = fmin(sp.t_out[k], t_best);
        if (!(t_in < t_out)) continue;
        t_in = fmax(t_in, 0.0);
        const double inside = (t_out - t_in) * sp.len;
        const double hit_distance = medium_neg_inv_density((uintptr_t)sc.media, k) * (double)__logf(rng_u01(rng));
        if (hit_distance > inside) continue;
        const double t = t_in + hit_distance * sp.inv_len;
        best.h.t = (float)t, best.h.slot = kMediumSlot - (int)k, best.rank = -1, best.meta = 0u;
        hit_set_t64(best.h, t);
    }
}

// The further attempts of a ray whose surface hit is a light (Q1, see shade_bounce): every attempt draws the free
// paths of the media the ray crosses in front of the light, and as long as none of them scatters the attempt is
// those draws and nothing else.  Returns true when the depth ran out that way (the light then emits); false when
// an attempt scatters: `rng` is back at that attempt's first draw and `depth` counts the attempts before it, so
// that media_sample repeats the attempt and finds the scatter.  Same draws and same comparisons as media_sample.
__device__ __attribute__((noinline)) bool media_light_attempts_impl(const DevMedium* media, uint32_t n_media,
                                                                   const MediaSpans* spans, double t_best, float tmin,
                                                                   uint64_t* rng_state, uint32_t* depth_io,
                                                                   uint32_t max_depth) {
    const MediaSpans sp = *spans;
    Rng rng;
    rng.s = *rng_state;
    uint32_t depth = *depth_io;
    double inside[kMaxMedia], neg_inv[kMaxMedia];
    bool any = false;
#pragma unroll
    for (uint32_t k = 0; k < kMaxMedia; ++k) {
        inside[k] = -1.0, neg_inv[k] = 0.0;
        if (k >= n_media || !(sp.t_out[k] >= sp.t_in[k])) continue;
        double t_in = fmax(sp.t_in[k], (double)tmin);
        const double t_out = fmin(sp.t_out[k], t_best);
        if (!(t_in < t_out)) continue;
        t_in = fmax(t_in, 0.0);
        inside[k] = (t_out - t_in) * sp.len;
        neg_inv[k] = medium_neg_inv_density(uniform_address(media), k);
        any = true;
    }
    bool emits = true;
    if (!any) {  // no medium in front of the light: the attempts are identical and draw nothing
        depth = max_depth;
    } else {
        while (true) {
            const Rng attempt = rng;
            bool scatter = false;
#pragma unroll
            for (uint32_t k = 0; k < kMaxMedia; ++k) {
                if (!(inside[k] >= 0.0)) continue;
                const double hit_distance = neg_inv[k] * (double)__logf(rng_u01(rng));
                scatter |= !(hit_distance > inside[k]);
            }
            if (scatter) {
                rng = attempt;
                emits = false;
                break;
            }
            if (++depth >= max_depth) break;
        }
    }
    *rng_state = rng.s, *depth_io = depth;
    return emits;
}
__device__ __forceinline__ bool media_light_attempts(const DevScene& sc, const MediaSpans& sp, const Closest& surface,
                                                     float tmin, Rng& rng, uint32_t& depth, uint32_t max_depth) {
    const double t_best = surface.h.slot >= 0 && (surface.meta & kMetaTypeMask) == 1u ? (double)surface.h.t
                                                                                      : hit_get_t64(surface.h);
    uint64_t state = rng.s;
    uint32_t d = depth;
    const bool emits = media_light_attempts_impl(sc.media, sc.n_media, &sp, t_best, tmin, &state, &d, max_depth);
    rng.s = state, depth = d;
    return emits;
}

#ifndef RT_PERLIN_INLINE
#define RT_PERLIN_INLINE __attribute__((noinline))
#endif
// ---- textures (EXT builds): Texture::color_at, material.rs:212-221
// PerlinNoise::noise + trilinear_interp, material.rs:325-368
__device__ __forceinline__ float perlin_noise(const float4* vec, const int32_t* perm, F3 p) {
    const float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    const int i = (int)fx, j = (int)fy, k = (int)fz;
    const float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    const float uu = u * u * (3.f - 2.f * u), vv = v * v * (3.f - 2.f * v), ww = w * w * (3.f - 2.f * w);
    float acc = 0.f;
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) {
        const int di = ii & 1, dj = (ii & 2) >> 1, dk = (ii & 4) >> 2;
        const float4 g = vec[perm[(i + di) & 255] ^ perm[256 + ((j + dj) & 255)] ^ perm[512 + ((k + dk) & 255)]];
        const float fi = (float)di, fj = (float)dj, fk = (float)dk;
        acc += (fi * uu + (1.f - fi) * (1.f - uu)) * (fj * vv + (1.f - fj) * (1.f - vv)) *
               (fk * ww + (1.f - fk) * (1.f - ww)) * (g.x * (u - fi) + g.y * (v - fj) + g.z * (w - fk));
    }
    return acc;
}
__device__ RT_PERLIN_INLINE float perlin_turb(const float4* vec, const int32_t* perm, F3 p) {  // material.rs:311-323
    float acc = 0.f, weight = 1.f;
    for (int d = 0; d < 7; ++d) {
        acc += weight * perlin_noise(vec, perm, p);
        weight *= 0.5f;
        p = 2.0f * p;
    }
    return fabsf(acc);
}
// `n_out` is the sphere's outward unit normal (get_sphere_uv, surfaces.rs:108-118); u = v = 0 for triangles and
// quads, which never set them (trace.rs:59-84).
template <bool NOISE = true>
__device__ __forceinline__ F3 texture_color(const DevScene& sc, int mat, F3 albedo, bool is_sphere, F3 n_out, D3 p) {
    const float4 t0 = sc.mat_tex[2 * mat];
    const int tex = __float_as_int(t0.x), index = __float_as_int(t0.y);
    if (tex == 0) return albedo;
    const F3 odd = f3(sc.mat_tex[2 * mat + 1]);
    float u = 0.f, v = 0.f;
    if (is_sphere && (tex == 2 || tex == 4)) {
        const float pi = 3.14159265358979f;
        u = (atan2f(-n_out.z, n_out.x) + pi) * (0.5f / pi);
        v = acosf(fminf(fmaxf(-n_out.y, -1.f), 1.f)) * (1.0f / pi);
    }
    if (tex == 1) {  // Checker, material.rs:236-247 (the point in f64, like the reference)
        const double inv = sc.mat_scale[2 * mat + 1];
        const long long s = (long long)floor(inv * p.x) + (long long)floor(inv * p.y) + (long long)floor(inv * p.z);
        return (s & 1) == 0 ? albedo : odd;
    }
    if (tex == 2) {  // UVChecker, material.rs:257-267
        const float sc_ = (float)sc.mat_scale[2 * mat];
        const int s = (int)floorf(sc_ * u) + (int)floorf(sc_ * v);
        return (s & 1) == 0 ? albedo : odd;
    }
    if (NOISE && tex == 3) {  // NoiseTexture::marbled, material.rs:300-303
        const F3 pf = tof(p);
        const float n = 0.5f * (1.0f + sinf((float)sc.mat_scale[2 * mat] * pf.z +
                                            10.0f * perlin_turb(sc.perlin_vec + 256 * index, sc.perlin_perm + 768 * index, pf)));
        return {n, n, n};
    }
    // Image::color_at, material.rs:386-424
    const uint4 dim = sc.image_dims[index];
    if (dim.y == 0u || dim.z == 0u) return {0.f, 1.f, 1.f};
    u = fminf(fmaxf(u, 0.f), 1.f);
    v = 1.0f - fminf(fmaxf(v, 0.f), 1.f);
    const uint32_t i = (uint32_t)(u * (float)(dim.y - 1u)), j = (uint32_t)(v * (float)(dim.z - 1u));
    const uint8_t* px = sc.image_rgb + dim.x + 3u * (j * dim.y + i);
    return {(float)px[0] * (1.0f / 255.0f), (float)px[1] * (1.0f / 255.0f), (float)px[2] * (1.0f / 255.0f)};
}

// `meta` is the hit slot's meta word (Closest::meta; tree hits only), `recs` the loose records as in loose_hits.
// TRI: the render tree holds nothing but triangles (a tree hit is a triangle; spheres and quads are loose).
template <bool DIRECT, bool EXT = false, bool NOISE = true, bool TRI = false>
__device__ __forceinline__ bool shade_bounce(const DevScene& sc, const uint4* recs, Ray& ray, const HitRec& hit,
                                             uint32_t meta, float tmin, Rng& rng, F3& thr, F3& lsum, uint32_t& depth,
                                             uint32_t max_depth, bool* retrace = nullptr, F3* emitted = nullptr) {
    // `retrace` (scenes with constant media): a hit that does not scatter (a light, an absorbed metal reflection)
    // makes the reference trace the SAME ray again, one depth unit per attempt (Q1) — and every attempt draws the
    // media's free paths anew, so the ray may scatter in a medium instead.  With `retrace` given, such a hit
    // returns with *retrace set and the ray unchanged, and the caller evaluates the media again; without media
    // the attempts are identical, so the light ends the path at once and the metal re-samples its fuzz in place.
    // A light also reports what it emits (*emitted): its further attempts need no second look at the surface.
    if (hit.slot == -1) {  // escaped: (0, background), scene/mod.rs:122
        lsum = lsum + thr * f3(sc.background);
        return false;
    }
    if (EXT && hit.slot <= kMediumSlot) {  // scattered inside a constant medium: Isotropic, material.rs:120-123
        const int mat = sc.media[kMediumSlot - hit.slot].material;
        const D3 p = ray.o + hit_get_t64(hit) * d3(ray.d);
        thr = thr * f3(sc.materials[mat]);
        ray.o = p;
        ray.d = rng_in_unit_sphere<DIRECT>(rng);
        return ++depth < max_depth;
    }
    // type and material first: the material fetch then overlaps the normal fetch
    uint32_t type, index;
    int mat;
    // (The vertex normals are fetched under `type == 1u` below, after the material's fetch has come back: fetched here,
    // beside it — one memory round trip instead of two — they cost ten more live registers, 16 spilled dwords instead
    // of 6 in the six-wave build, and 12 % of the frame rate: profiles/r3_experiments.md.)
    if (hit.slot >= 0) {
        type = TRI ? 1u : (meta & kMetaTypeMask), index = meta >> kMetaIndexShift;
        mat = type == 1u ? (int)index : (type == 0u ? load_global(sc.sphere_mat + index) : load_global(sc.quad_mat + index));
    } else {
        const uint4 m = recs[kLooseRecQuads * (uint32_t)(-2 - hit.slot)];
        type = m.x, index = m.y, mat = (int)m.w;
    }
    const float4 m = load_global(sc.materials + mat);
    const int kind = load_global(sc.mat_kind + mat);
    const D3 dd = d3(ray.d);
    F3 n;
    D3 p;
    if (type == 1u) {
        const float4 s0 = load_global(sc.tri_shade + 3 * hit.slot + 0);
        const float4 s1 = load_global(sc.tri_shade + 3 * hit.slot + 1);
        const float4 s2 = load_global(sc.tri_shade + 3 * hit.slot + 2);
        // un-normalised barycentric blend (Q3), surfaces.rs:277
        n = hit.bu * f3(s1) + hit.bv * f3(s2) + (1.0f - hit.bu - hit.bv) * f3(s0);
        p = ray.o + (double)hit.t * dd;
    } else {
        p = ray.o + hit_get_t64(hit) * dd;  // the f64 root (hit.t is its f32 rounding)
        double c[4];                        // sphere: c.xyz r; quad: normal.xyz d
        if (hit.slot >= 0) {
            const double* g = type == 0u ? sc.spheres + 4 * index : sc.quads + 16 * index;
            c[0] = load_global(g), c[1] = load_global(g + 1), c[2] = load_global(g + 2), c[3] = load_global(g + 3);
        } else {
            const double* l = loose_data(recs, (uint32_t)(-2 - hit.slot));
            c[0] = l[0], c[1] = l[1], c[2] = l[2], c[3] = l[3];
        }
        if (type == 0u) {
            D3 centre = d3(c);
            if (EXT) {  // the centre at this ray's time (surfaces.rs:52-54)
                const double* c1 = hit.slot >= 0 ? sc.sphere_end + 3 * index : loose_data(recs, (uint32_t)(-2 - hit.slot)) + 10;
                centre = sphere_centre<true>(c, c1, ray.time);
            }
            D3 nn = rcp64(c[3]) * (p - centre);  // signed radius: a negative one flips the normal
            n = tof(nn);
        } else {
            n = tof(d3(c));
        }
    }
    const F3 n_out = n;  // outward normal, before the face flip: what the sphere's (u, v) is taken from
    // Hit::new, trace.rs:59-76: front face from the (shading) normal (Q4)
    const bool front = dot(ray.d, n) < 0.0f;
    if (!front) n = neg(n);

    const F3 albedo = EXT ? texture_color<NOISE>(sc, mat, f3(m), type == 0u, n_out, p) : f3(m);
    if (kind == 3) {  // DiffuseLight: emits on both faces, never scatters (material.rs:66-72, 94)
        if (EXT && retrace && ++depth < max_depth) {
            *retrace = true;
            if (emitted) *emitted = albedo;
            return true;
        }
        lsum = lsum + thr * albedo;
        return false;  // the reference re-traces this ray up to depth 40 to the same end (Q1)
    }
    F3 dir;
    if (kind == 0) {  // Lambertian, material.rs:100-118
        dir = n + rng_unit_vector<DIRECT>(rng);
        if (fabsf(dir.x) < 1e-8f && fabsf(dir.y) < 1e-8f && fabsf(dir.z) < 1e-8f) dir = n;
    } else if (kind == 1) {  // Metal, material.rs:126-138 + the re-sampling loop of Q1
        const F3 refl = reflect(ray.d, n);  // incoming direction not normalised (Q2)
        while (true) {
            const F3 s = rng_in_unit_sphere<DIRECT>(rng);
            dir = refl + m.w * s;
            if (dot(dir, n) > 1e-8f) break;
            // absorbed: nothing pushed, depth += 1, same ray re-traced to the same hit, fuzz re-drawn
            if (++depth >= max_depth) return false;
            if (EXT && retrace) {
                *retrace = true;
                return true;
            }
        }
    } else {  // Dielectric, material.rs:141-181
        const float ratio = front ? fast_rcp(m.w) : m.w;
        const float inv_len = fast_rsq(dot(ray.d, ray.d));
        const F3 ud = inv_len * ray.d;
        const float cos_theta = fminf(dot(neg(ud), n), 1.0f);
        const float sin_theta = fast_sqrt(1.0f - cos_theta * cos_theta);
        bool refl = ratio * sin_theta > 1.0f;
        if (!refl) {  // short-circuit: the uniform is drawn only when refraction is possible
            float r0 = (1.0f - ratio) * fast_rcp(1.0f + ratio);
            r0 = r0 * r0;
            const float c1 = 1.0f - cos_theta;
            const float c2 = c1 * c1;
            refl = r0 + (1.0f - r0) * (c2 * c2 * c1) > rng_u01(rng);
        }
        if (refl) {
            dir = reflect(ud, n);
        } else {
            const F3 perp = ratio * (ud + cos_theta * n);
            const float par = -fast_sqrt(fabsf(1.0f - dot(perp, perp)));
            dir = perp + par * n;
        }
    }
    thr = thr * albedo;
    ray.o = p;
    ray.d = dir;
    return ++depth < max_depth;
}

}  // namespace rt
