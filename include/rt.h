/*
 * rt.h — C ABI of the MI355X-native render path (librt_amd.so).
 *
 * The reference (cwbriones/raytracer, Rust) has no FFI; its per-pixel render
 * path sits behind three crate-internal calls made from src/main.rs:
 *     scene::load_scene(path, aspect, use_bvh)      main.rs:100  load.rs:39
 *     camera.get_ray(rng, u, v)                     main.rs:174  camera.rs:76
 *     scene.ray_color(ray, rng, max_depth)          main.rs:175  scene/mod.rs:79
 * wrapped by the rayon pixel loop main.rs:150-185 and the tonemap main.rs:178-180.
 * The cut is below the loader and above the pixel loop: the host hands over a
 * flat immutable scene, the backend returns an un-normalised f32 radiance-sum
 * framebuffer.  Every entry point is extern "C", plain pointers and sizes.
 *
 * Conventions: every call returning int returns 0 on success, non-zero on
 * error; rt_last_error() then returns a thread-local message.  The caller owns
 * all host buffers; handles are opaque and released by the matching destroy.
 * A handle is used from one host thread at a time.
 */
#ifndef RT_AMD_RT_H
#define RT_AMD_RT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- flat scene description (host memory, f64 like the reference: geom.rs:19,114) ---- */

/* RT_MAT_ISOTROPIC (material.rs:52-57, 120-123) is the phase function of a constant medium */
enum { RT_MAT_LAMBERTIAN = 0, RT_MAT_METAL = 1, RT_MAT_DIELECTRIC = 2, RT_MAT_DIFFUSE_LIGHT = 3, RT_MAT_ISOTROPIC = 4 };
/* RT_SURF_MEDIUM: index into rt_scene_desc.media */
enum { RT_SURF_SPHERE = 0, RT_SURF_TRIANGLE = 1, RT_SURF_QUAD = 2, RT_SURF_MEDIUM = 3 };

/* Texture, material.rs:183-222.  Only Lambertian materials carry one other than SOLID (material.rs:23-54). */
enum { RT_TEX_SOLID = 0, RT_TEX_CHECKER = 1, RT_TEX_UV_CHECKER = 2, RT_TEX_MARBLE = 3, RT_TEX_IMAGE = 4 };

/* material.rs:16-56 (kind + texture; param = fuzz | refractive index) */
typedef struct rt_material {
    int32_t kind;
    int32_t texture;    /* RT_TEX_* */
    double albedo[3];   /* SolidColor; the "even" colour of the two checkers */
    double param;
    double albedo2[3];  /* the "odd" colour of the two checkers (material.rs:229-268) */
    double tex_scale;   /* Checker: scale (its reciprocal multiplies the point, material.rs:193-199);
                           UVChecker: scale of (u, v); NoiseTexture: scale of z in `marbled` (material.rs:300-303) */
    int32_t tex_index;  /* MARBLE: index into rt_scene_desc.perlin; IMAGE: index into rt_scene_desc.images */
    int32_t _pad;
} rt_material;

/* surfaces.rs:26-54: the centre moves linearly from `center` (ray time 0) to `center1` (time 1) when `moving`
 * is set; radius may be negative (distant.yaml:31) */
typedef struct rt_sphere {
    double center[3];
    double radius;
    int32_t material;
    int32_t moving;
    double center1[3];
} rt_sphere;

/* ConstantMedium, surfaces.rs:594-653: a boundary filled with an isotropic scatterer of constant density.
 * The boundary is one sphere (spheres[first]) or a closed surface of quads (quads[first .. first+count): the six of
 * one make_box, surfaces.rs:542-590, or any number); boundary primitives are NOT listed in `surfaces` (a scene that also shows the boundary
 * as a surface lists a copy).  The medium itself is listed in `surfaces` as {RT_SURF_MEDIUM, index}. */
typedef struct rt_medium {
    int32_t boundary;  /* RT_SURF_SPHERE | RT_SURF_QUAD */
    int32_t first, count; /* one sphere (count 1), or `count` quads that enclose a volume (a make_box's six, or more) */
    int32_t material;  /* an RT_MAT_ISOTROPIC material */
    double density;
} rt_medium;

/* PerlinNoise, material.rs:281-309: 256 unit vectors and three permutations of 0..255 */
typedef struct rt_perlin {
    double ranvec[256][3];
    int32_t perm[3][256];
} rt_perlin;

/* Image texture, material.rs:379-424: tightly packed RGB8 rows, top row first; width == 0 is the reference's
 * "image could not be loaded" case, which shades (0, 1, 1) */
typedef struct rt_image {
    const uint8_t* rgb;
    uint32_t width, height;
} rt_image;

/* surfaces.rs:195-243: the three vertices and vertex normals a Triangle resolves through its Mesh */
typedef struct rt_triangle {
    double v[3][3];
    double n[3][3];
    int32_t material;
    int32_t _pad;
} rt_triangle;

/* surfaces.rs:296-327: q,u,v as given; normal,d,w are the caches Quad::new computes */
typedef struct rt_quad {
    double q[3], u[3], v[3];
    double normal[3];
    double d;
    double w[3];
    int32_t material;
    int32_t _pad;
} rt_quad;

/* One entry per surface in the order the loader added it to the SceneBuilder
 * (load.rs:57-158).  Order matters: the BVH build uses a stable sort (bvh.rs:218).
 * A primitive id reported by this library is an index into this array. */
typedef struct rt_surface_ref {
    int32_t type;  /* RT_SURF_* */
    int32_t index; /* into spheres / triangles / quads */
} rt_surface_ref;

/* A tree inside the tree: Surface::Bvh (surfaces.rs:374), a member of its parent's BVH that is a BVH itself — the
 * small spheres of one-weekend beside the ground sphere (example.rs:96-139), make_box's one leaf of six quads
 * (surfaces.rs:542-591), final-scene's tree of boxes and its turned cluster of spheres (example.rs:487-498, 553-570).
 * Hits do not depend on the nesting; what does is the in-order leaf sequence, i.e. which of two surfaces wins an exact
 * equal-t tie (bvh.rs:272-273, 308): the parent's build sees the group as ONE item with ONE box, and the group's own
 * tree is built over its members alone.  Instancing wrappers around a group (Rotated / Translated, surfaces.rs:461-540)
 * are baked into the members' coordinates; what they leave behind is the group's box in its parent's build (`box`)
 * and, since the reference builds a wrapped group's tree in the group's own space, the members' boxes there
 * (rt_scene_desc.surface_build_box). */
typedef struct rt_bvh_group {
    int32_t parent;  /* enclosing group (listed before this one), -1 = the scene's own list (scene/mod.rs:31) */
    int32_t leaf;    /* non-zero: Bvh::new_leaf (bvh.rs:63-68) — one leaf, the members in their listing order */
    double box[6];   /* lo.xyz hi.xyz of the group as its parent's build sees it (Bvh::bounding_box, through
                        Aabb::rotate / Aabb::translate where the reference wraps the group, trace.rs:185-211);
                        box[0] > box[3]: none given — the union of the members' boxes */
} rt_bvh_group;

/* camera.rs:11-18 after Camera::new (camera.rs:43-74) */
typedef struct rt_camera {
    double origin[3], lower_left[3], horizontal[3], vertical[3];
    double u[3], v[3], w[3];
    double lens_radius;
} rt_camera;

typedef struct rt_scene_desc {
    const rt_material* materials;
    size_t n_materials;
    const rt_sphere* spheres;
    size_t n_spheres;
    const rt_triangle* triangles;
    size_t n_triangles;
    const rt_quad* quads;
    size_t n_quads;
    const rt_surface_ref* surfaces;
    size_t n_surfaces;
    double background[3];
    rt_camera camera;
    int32_t use_bvh; /* 0 = one leaf holding every surface (--no-bvh, main.rs:60-65, scene/mod.rs:61-65); 1 = trees
                      * built on the host; 2 = as 1, but the tree the render kernels traverse is built on the GPU
                      * (same binned-SAH criteria, level by level; the environment's RT_BUILD=device asks for the same).
                      * The reference-topology tree (bvh.rs:131-234: tie ranks, parity kernel) is always a host build */
    int32_t _pad;
    const rt_perlin* perlin; /* noise tables the MARBLE textures refer to (may be null when n_perlin == 0) */
    size_t n_perlin;
    const rt_image* images;  /* images the IMAGE textures refer to */
    size_t n_images;
    const rt_medium* media;  /* constant media (RT_SURF_MEDIUM surfaces index this) */
    size_t n_media;
    /* nesting of the reference-topology tree (null / 0: none — every scene file; the example scenes nest) */
    const rt_bvh_group* groups;
    size_t n_groups;
    const int32_t* surface_group;    /* n_surfaces entries: the group a surface is a member of, -1 = the scene's own
                                        list; the surfaces of a group, its sub-groups' included, are contiguous */
    const double* surface_build_box; /* 6 per surface or null: lo.xyz hi.xyz of the surface in its group's build where
                                        that is not its own bounding box (a member of a rotated group: the box in the
                                        group's own space); lo.x > hi.x: its own bounding box */
} rt_scene_desc;

/* ---- host-side loader: replaces scene::load_scene (load.rs:39-160) ---- */

/* Parses the scene YAML, loads OBJ(.gz) meshes, bakes transforms, builds the camera for
 * `aspect_ratio`.  *out is owned by the library; release with rt_scene_desc_free. */
/* use_bvh: 0, 1 or 2 as rt_scene_desc.use_bvh says. */
int rt_load_scene(const char* yaml_path, double aspect_ratio, int use_bvh, rt_scene_desc** out);

/* Procedural example scenes: replaces Example::from_str + Example::scene (scene/example.rs:53-79) for all
 * eight names the reference knows (example.rs:53-66): "one-weekend" (the reference's default when no --scene is
 * given, main.rs:102-106), "random-spheres", "two-spheres", "two-perlin", "earth", "cornell", "cornell-smoke",
 * "final-scene"; any other name is an error.  The reference draws the random ones from thread_rng(); here `seed`
 * fixes the draws.  Release *out with rt_scene_desc_free. */
int rt_example_scene(const char* name, double aspect_ratio, uint64_t seed, int use_bvh, rt_scene_desc** out);
void rt_scene_desc_free(rt_scene_desc* desc);

/* ---- device scene: BVH build (bvh.rs:131-234) + SoA flattening + upload to HBM ---- */

typedef struct rt_scene rt_scene;

/* device >= 0: that HIP device; RT_DEVICE_CURRENT keeps the current one; RT_DEVICE_NONE builds and
 * flattens on the host only (inspection and CPU tests) — every render call on such a scene fails. */
enum { RT_DEVICE_CURRENT = -1, RT_DEVICE_NONE = -2 };
int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);

typedef struct rt_scene_info {
    /* the reference-topology tree (bvh.rs:131-234): parity kernel and equal-t tie ranks */
    uint64_t n_surfaces, n_nodes, n_leaves;
    uint32_t max_depth;     /* depth of the binary tree (root = 1) */
    uint32_t max_leaf_size; /* largest leaf (can exceed 8: bvh.rs:219,233) */
    uint64_t device_bytes;  /* bytes resident in HBM for this scene */
    double build_ms;        /* host BVH builds + flatten */
    /* the tree the f32 render kernels traverse (binned SAH; spheres/quads kept out of it when few) */
    uint64_t render_nodes;
    uint32_t render_depth, render_max_leaf;
    uint32_t n_loose;       /* spheres + quads tested by every ray before traversal */
    uint32_t device_build;  /* 1 if the render tree was built on the GPU */
    double render_build_ms; /* the render tree's build alone (it runs beside the reference-topology build) */
} rt_scene_info;
int rt_scene_get_info(const rt_scene* scene, rt_scene_info* info);

/* In-order (left-to-right) leaf sequence of the reference-topology tree as surface ids
 * (n_surfaces entries) — the sequence that decides equal-t ties (bvh.rs:272-273,308). */
int rt_scene_leaf_order(const rt_scene* scene, int32_t* out_surface_ids);

/* ---- the hot path: replaces the pixel loop main.rs:150-185 ---- */

/* Three organisations of the same loop (same samples, same arithmetic):
 *   PERSISTENT: one path per lane, stage-scheduled waves;
 *   WAVEFRONT:  one kernel per stage, path state in HBM queues;
 *   POOL:       stage-scheduled waves that own more paths than lanes (LDS pool); falls back to PERSISTENT
 *               where it does not apply (no tree in the scene, max_depth > 63, frames beyond 2^26 pixels). */
enum { RT_KERNEL_PERSISTENT = 0, RT_KERNEL_WAVEFRONT = 1, RT_KERNEL_POOL = 2 };
/* How unit-sphere / unit-disk samples are drawn (util.rs:15-35).  Same distributions:
 *   REFERENCE: the reference's rejection loops, in its draw order;
 *   DIRECT:    closed-form maps with a fixed number of draws (no data-dependent loop on the GPU). */
enum { RT_SAMPLER_REFERENCE = 0, RT_SAMPLER_DIRECT = 1 };

typedef struct rt_render_params {
    uint32_t width, height;
    uint32_t sn;             /* strata per axis: floor(sqrt(spp)), main.rs:142 */
    uint32_t max_depth;      /* 40 in the reference, main.rs:118 */
    uint64_t seed;
    uint32_t stratum_begin;  /* this call renders strata k = begin, begin+step, ... < sn*sn   */
    uint32_t stratum_step;   /* (k = si*sn + sj, main.rs:165-166); 0 is treated as 1          */
    int32_t kernel;          /* RT_KERNEL_* */
    int32_t collect_stats;   /* non-zero: count node visits / primitive tests (slower build of the kernel) */
    int32_t sampler;         /* RT_SAMPLER_* */
    int32_t _pad;
} rt_render_params;

typedef struct rt_stats {
    uint64_t samples;        /* camera paths traced by this call */
    uint64_t segments;       /* ray segments traced incl. bounces ("rays" in the bounce sense) */
    uint64_t node_visits;    /* BVH nodes whose child boxes were tested   (collect_stats only) */
    uint64_t tri_tests, sphere_tests, quad_tests; /* primitive tests      (collect_stats only) */
    double kernel_ms;        /* device time of the render kernels, HIP events on the launch stream */
    uint32_t n_launches;     /* kernels launched */
    uint32_t _pad;
} rt_stats;

/* Renders into DEVICE memory: d_rgb_sum = 3*width*height floats, row 0 = top row of the image
 * (pj = height - idx/width - 1, main.rs:158), un-normalised radiance SUM over the strata of this
 * call (overwritten, not accumulated).  `stream` is a hipStream_t (NULL = default stream); the call
 * returns after enqueueing unless stats != NULL, in which case it synchronises the stream.
 * Scratch (tile accumulator, chunk cursor, counters, events) is kept with the scene in THREE sets used in turn, so
 * one host thread may keep up to three renders of one scene in flight on as many streams (frames in turn); a
 * fourth call re-uses the first set and is ordered after that set's previous use by an event wait on its stream.
 * The persistent and pool organisations take the sets in turn; the wavefront one has one set of queues and
 * serialises.  Growing the accumulator for a larger frame (hipFree + hipMalloc) waits for the set's last use. */
int rt_render_device(rt_scene* scene, const rt_render_params* params, float* d_rgb_sum, void* stream,
                     rt_stats* stats);

/* rt_render_device for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Calls no hipMalloc / hipFree or other
 * allocating routine: the kernels' scratch (tile accumulator + chunk cursor) is a static device array of the
 * library, 24 MiB + 4 KiB per GPU, held until the process ends; its address is looked up once per device at
 * rt_scene_create.  Limits: RT_KERNEL_PERSISTENT only, collect_stats == 0, frames of at most 32 768 8x8 tiles
 * (ceil(width/8) * ceil(height/8); 1920x1080 is 32 400) - anything larger is rejected with an error, as is a
 * scene created with RT_DEVICE_NONE.  Enqueues on `stream` and returns; same output as rt_render_device.  Every
 * call on a GPU shares that GPU's one scratch array.  Calls made by ONE host thread are ordered by the library,
 * whatever their streams and scenes: an event per GPU, created at rt_scene_create, is waited on by each call's
 * stream and recorded at the call's end (no allocation; skipped on a stream under graph capture, where the graph's
 * own order is the caller's).  Calls made by several host threads at once still race: order those yourself. */
int rt_render_device_noalloc(rt_scene* scene, const rt_render_params* params, float* d_rgb_sum, void* stream);

/* Same, into HOST memory (synchronous; allocates and frees a device framebuffer). */
int rt_render(rt_scene* scene, const rt_render_params* params, float* rgb_sum, rt_stats* stats);

/* Parity AOV: closest hit of the pixel-centre ray (u=(pi+1)/W, v=(pj+1)/H, main.rs:161-162; no jitter,
 * no lens offset), interval (0.001, inf) (scene/mod.rs:104).  prim_id = surface id or -1.
 * precision: 64 = f64 arithmetic, no contraction, reference visiting rule (exact vs the oracle);
 *            32 = the f32 arithmetic and pruned traversal the render kernels use. */
int rt_primary_hits(rt_scene* scene, uint32_t width, uint32_t height, int precision, int32_t* prim_id,
                    double* t);

/* average + gamma-2 + quantise, main.rs:177-180,209-221: u8 = (256*clamp(sqrt(sum/n),0,0.999)) as u8,
 * NaN -> 0.  Host buffers; runs on the current device. */
int rt_tonemap_rgb8(const float* rgb_sum, uint32_t width, uint32_t height, uint64_t n_samples, uint8_t* out);
/* Device-buffer form on a stream. */
int rt_tonemap_rgb8_device(const float* d_rgb_sum, uint32_t width, uint32_t height, uint64_t n_samples,
                           uint8_t* d_out, void* stream);

/* ---- several GPUs of one process: the path's one exchange step ----
 * No counterpart in the reference (its pixels are shared memory under rayon, main.rs:150): with the strata of a
 * frame sharded over GPUs (rt_render_params.stratum_begin / stratum_step), the per-GPU radiance sums are added
 * onto one GPU by ONE collective, an RCCL sum-reduce over xGMI.  One process per GPU does the same through its
 * own communicator (bench.py: torch.distributed); these calls serve one process that drives several GPUs (the CLI's
 * --gpus N).  librccl is loaded at the first rt_group_create, not with the library. */
typedef struct rt_group rt_group;
/* A communicator clique over the `n` HIP devices `devices` of this process (ncclCommInitAll). */
int rt_group_create(const int* devices, int n, rt_group** out);
/* d_bufs[i] (count floats in device i's memory) summed over i into d_bufs[root], in place; enqueued on streams[i]
 * (streams == NULL: every device's default stream) and not awaited: synchronise the root's stream before reading. */
int rt_group_reduce_sum(rt_group* group, float* const* d_bufs, size_t count, int root, void* const* streams);
void rt_group_destroy(rt_group* group);

/* ---- image output: replaces the png crate use at main.rs:192-200 (RGB8, row 0 = top) ---- */
int rt_write_png(const char* path, const uint8_t* rgb, uint32_t width, uint32_t height);

/* The reference's loader reports mesh statistics on stderr (load.rs:231-244); off by default here. */
void rt_set_verbose(int verbose);

const char* rt_last_error(void);
const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_RT_H */
