// Scene layout in HBM (uploaded once per scene, read-only afterwards) and the kernel parameter blocks.
// See DESIGN.md "Data layout".  All f32 arrays are float4-granular so every fetch is one dwordx4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

// prim meta word (prims[3*slot].w bits): type in bits 0-1, last-in-leaf in bit 2; above them the material id
// (triangles) or the type-local index (spheres, quads).
constexpr uint32_t kMetaTypeMask = 3u;
constexpr uint32_t kMetaLast = 4u;
constexpr int kMetaIndexShift = 3;
constexpr int kNoChild = 0x7fffffff;  // "empty" child of the synthetic root when the tree is one leaf
#ifndef RT_MAX_LOOSE
#define RT_MAX_LOOSE 32
#endif
constexpr uint32_t kMaxLoose = RT_MAX_LOOSE;    // at most this many spheres+quads are tested by every ray ("loose") ...
#ifndef RT_MAX_LOOSE_MESH
#define RT_MAX_LOOSE_MESH 16
#endif
constexpr uint32_t kMaxLooseBesideMesh = RT_MAX_LOOSE_MESH;    // ... and this many when the scene has triangles (api.cpp, loose_spheres_and_quads)
constexpr uint32_t kLooseRecQuads = 11;  // uint4 per loose record: {type, index, rank, material} + 20 doubles

// f64 records for the exact-parity kernel (array of structs; speed is irrelevant there).
struct Node64 {
    double lo[2][3], hi[2][3];  // child boxes, un-rounded
    int32_t child[2];           // >= 0 inner node, < 0 leaf: ~first_slot, kNoChild: none
    int32_t pad[2];
};
struct Prim64 {
    int32_t type, last, surface, pad;
    // tri: v0 v1 v2 (9)          sphere: c (3), r          quad: normal(3) d q(3) u(3) v(3) w(3)
    double a[16];
};

// ConstantMedium (surfaces.rs:594-653): boundary = spheres[first] or quads[first .. first+count)
struct DevMedium {
    int32_t boundary, first, count, material;
    double neg_inv_density;  // -(1 / d), surfaces.rs:605
    double pad;
};
static_assert(sizeof(DevMedium) == 32, "media_spans_impl reads DevMedium as eight 32-bit words");
constexpr uint32_t kMaxMedia = 4;
constexpr int kMediumSlot = -100;  // hit slot of medium k is kMediumSlot - k (loose primitives use -2 .. -17)

struct DevCamera {
    double origin[3];  // ray origins are carried in f64 (spheres/quads are intersected in f64)
    float ll_rel[3];   // lower_left - origin            (f64 on host, then rounded)
    float hor_px[3];   // horizontal / width
    float ver_px[3];   // vertical / height
    float u[3], v[3];  // lens basis
    float lens_radius;
};

struct DevScene {
    // render tree (binned SAH, leaves <= 2): triangles, plus spheres/quads when a scene has too many of
    // them to test "loose"
    const uint4* nodes;         // 2 per inner node (32 B), one per child: {x, y, z, ref} with each axis word holding
                                //   the child box's centre plane (low half, u16) and half-extent (high half, binary16) on a 16-bit grid over
                                //   the tree's bounds (rounded outward by at least a step), ref = child id: >= 0 inner
                                //   node, < 0 leaf ~first_slot.  A node visit is two 16-byte fetches per lane
                                //   instead of four: the kernels are bound by the vector L1's request rate.
    const float4* prims;        // 3 per leaf slot: tri {v0|meta, e1|rank, e2}; rank = position in the
                                //   reference tree's in-order leaf sequence (decides equal-t ties)
    const float4* tri_shade;    // 3 per leaf slot: vertex normals n0 n1 n2; [0].w = material id bits
    const int4* loose;          // spheres/quads tested by every ray before traversal: type, index, rank, surface id
    const uint4* loose_rec;     // the same, self-contained (kLooseRecQuads uint4 each): {type, index, rank, material},
                                //   then the sphere's c.xyz r + gate lo hi + centre at time 1, or the quad's normal d q A B + gate lo hi, as doubles; the
                                //   persistent kernels stage these in LDS
    uint32_t n_loose;
    uint32_t has_tree;          // 0: no surface lives in the render tree (all are loose)
    float grid_lo[3], grid_step[3];  // the grid: plane q of axis a lies at grid_lo[a] + q * grid_step[a], q in
                                     //   0 .. 65535; planes 0 and 65535 are the (inflated) bounds of the render tree
    const double* spheres;      // 4 per sphere: c.xyz r
    const double* sphere_gate;  // 6 per sphere: lo.xyz hi.xyz of the reference tree's leaf that holds it; read for
                                //   negative radii only (their inverted bounding box adds nothing to the leaf's
                                //   box, so the reference reaches them only through the leaf's other members)
    const double* quads;        // 16 per quad: normal(3) d q(3) A=v×w (3) B=w×u (3) pad(3)
    const double* quad_gate;    // 6 per quad, like sphere_gate: Quad::bounding_box (surfaces.rs:356-359) spans only
                                //   q and q+u+v, which is too small for a parallelogram whose sides share an axis
    const int32_t* sphere_mat;  // material id per sphere
    const int32_t* quad_mat;    // material id per quad
    // textures and moving spheres (the EXT builds of the kernels; `ext` == 0 means none are present; bit 1: a noise
    // texture is among them — the builds without one leave the Perlin code, and its registers, out)
    uint32_t ext;
    uint32_t tree_kind;         // what the render tree holds: 1 triangles only, 2 spheres only, 3 quads only, 0 a mix
    uint32_t n_media;
    const DevMedium* media;
    const double* sphere_end;   // 3 per sphere: centre at ray time 1 (Sphere::moving, surfaces.rs:43-54)
    const float4* mat_tex;      // 2 per material: {texture kind, table/image index, -, -} (as bits), {albedo2.xyz, -}
    const double* mat_scale;    // 2 per material: tex_scale, 1 / tex_scale
    const float4* perlin_vec;   // 256 per noise table
    const int32_t* perlin_perm; // 3 x 256 per noise table
    const uint8_t* image_rgb;   // all images' pixels, back to back
    const uint4* image_dims;    // per image: byte offset, width, height, -
    const float4* materials;    // 1 per material: albedo.xyz, w = fuzz | refractive index
    const int32_t* mat_kind;    // RT_MAT_* per material
    const int32_t* slot_surface;  // surface id per leaf slot
    const Node64* nodes64;
    const Prim64* prims64;
    float background[3];
    DevCamera cam;
    double cam64[4][3];  // origin, lower_left, horizontal, vertical (parity kernel)
    uint32_t n_slots;
};

struct DevRenderParams {
    uint32_t width, height, sn, max_depth;
    uint32_t stratum_begin, stratum_step, n_strata;  // n_strata = sn*sn
    uint32_t seed_lo, seed_hi;
    float recip_sn;
    uint32_t sn_magic;  // division of a stratum index by sn as one multiply-high (0: divide)
    uint32_t sampler;  // RT_SAMPLER_*
};

struct DevCounters {  // one instance in HBM, bumped with one atomic per wave at kernel end
    unsigned long long samples, segments, node_visits, tri_tests, sphere_tests, quad_tests;
    unsigned int work;  // chunk cursor of the persistent kernel
    unsigned int pad;
    unsigned long long stage_trips[8], stage_lanes[8], stage_cycles[8];  // persistent kernels, counting build: per stage
    // counting build: when the waves ran out of work and when they finished (0.5 ms buckets from the first wave's
    // start, 100 MHz wall clock) — the launch's tail
    unsigned long long t_first_inv;  // ~(earliest start)
    unsigned int exhaust_hist[64], finish_hist[64];
    unsigned int drain_hist[64];          // finish - out-of-work, 0.25 ms buckets
    unsigned int n_late, pad2;
    unsigned long long late[32][8];       // waves that drained for > 2 ms: times, trips and lanes per stage while draining
};

}  // namespace rt
