// C ABI (include/rt.h): scene flattening + upload, kernel launches, error plumbing.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <future>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/rt.h"
#include "../host/bvh_build.h"
#include "../host/scene_host.h"
#include "../host/worker.h"
#include "device_scene.h"
#include "kernels.h"
#include "wavefront.h"

namespace rt {
extern int g_verbose;
void write_png_rgb8(const std::string& path, const uint8_t* rgb, uint32_t width, uint32_t height);
}  // namespace rt

namespace rt {
const char* tuning_env(const char* name) {
    static const bool tuning = getenv("RT_TUNE") != nullptr;
    return tuning ? getenv(name) : nullptr;
}
}  // namespace rt

using namespace rt;

namespace {

thread_local std::string g_last_error;

int set_error(const std::string& m) {
    g_last_error = m;
    return 1;
}

}  // namespace
int rt_set_error_message(const std::string& m) { return set_error(m); }  // for the library's other translation units
namespace {

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
#define HIP_CHECK(expr)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            throw HipError(std::string(#expr) + " failed: " + hipGetErrorString(e_) + " (" __FILE__ ":" + \
                           std::to_string(__LINE__) + ")");                                               \
    } while (0)

float round_down(double x) {
    float f = (float)x;
    if ((double)f > x) f = std::nextafterf(f, -INFINITY);
    return f;
}
float round_up(double x) {
    float f = (float)x;
    if ((double)f < x) f = std::nextafterf(f, INFINITY);
    return f;
}

float int_bits(int32_t v) {
    float f;
    std::memcpy(&f, &v, 4);
    return f;
}
float uint_bits(uint32_t v) {
    float f;
    std::memcpy(&f, &v, 4);
    return f;
}

}  // namespace

struct rt_scene {
    int device = -1;  // RT_DEVICE_NONE: host-only
    Bvh bvh;         // reference topology (bvh.rs:131-234): parity kernel + tie ranks
    Bvh render_bvh;  // binned-SAH tree the f32 render kernels traverse
    bool has_tree = false;
    float grid_lo[3] = {0, 0, 0}, grid_step[3] = {1, 1, 1};  // 16-bit plane grid of the render tree's nodes
    rt_scene_info info{};
    uint32_t stack_depth = 0;  // depth of the flattened render tree incl. a synthetic root
    uint32_t n_cus = 256;
    DevScene dev{};
    std::vector<void*> allocations;
    // Scratch of a render call: the persistent kernels' tile-planar accumulator, the counters block (chunk
    // cursor + statistics) and the timing events.  Three sets, used in turn, so that up to three renders of one scene
    // can be in flight on as many streams (frame i+1 starts on the CUs that frame i's draining waves leave idle;
    // measured per 1/8-slice frame of lucy.yaml: 24.7 ms awaited, 24.2 with two in flight, 23.8 with three, 24.2 with
    // four); a call
    // whose slot was last used on another stream makes its stream wait for that use to end (`done`).
    struct Slot {
        DevCounters* d_counters = nullptr;
        float* d_acc = nullptr;
        size_t acc_bytes = 0;
        int* d_overflow = nullptr;  // the six-wave build's stack levels beyond LDS (deep mesh trees only)
        size_t overflow_bytes = 0;
        hipEvent_t ev0 = nullptr, ev1 = nullptr, done = nullptr;
        hipStream_t last_stream = nullptr;
        bool used = false;
    };
    static constexpr int kSlots = 3;
    Slot slots[kSlots];
    int next_slot = 0;
    // rt_render_device_noalloc: this device's copy of the library's static scratch (looked up at scene creation)
    float* static_acc = nullptr;
    DevCounters* static_counters = nullptr;
    std::unique_ptr<WavefrontState> wf;  // its queues are one set: wavefront renders all use slot 0

    // flattened host copies (kept for host-only inspection and tests)
    std::vector<uint4> nodes;
    std::vector<float4> prims, tri_shade, materials;
    std::vector<int4> loose;
    std::vector<char> is_loose;  // per surface: tested by every ray instead of living in the render tree
    std::vector<uint4> loose_rec;
    std::vector<int32_t> render_ids;  // surfaces that live in the render tree
    std::vector<double> sphere_gate, quad_gate, sphere_end, mat_scale;
    std::vector<float4> mat_tex, perlin_vec;
    std::vector<DevMedium> media;
    std::vector<int32_t> perlin_perm;
    std::vector<uint8_t> image_rgb;
    std::vector<uint4> image_dims;
    bool ext = false;  // any texture or moving sphere
    bool noise = false;  // a Perlin-noise texture among them (the kernels' NOISE builds)
    std::vector<double> spheres, quads;
    std::vector<int32_t> sphere_mat, quad_mat, mat_kind, slot_surface;
    // The exact-parity kernel's f64 arrays (the un-rounded reference-topology tree) are parity-only data: flattened
    // and uploaded at the first rt_primary_hits(precision = 64), not at scene creation.  Until then the scene keeps
    // what that flattening reads beside the tree itself — the caller's rt_scene_desc need not outlive the call:
    struct Reference64Source {  // (made on a worker thread, kept with the scene: OsBlocks, host/worker.h)
        std::vector<rt_surface_ref, OsBlocks<rt_surface_ref>> surfaces;
        std::vector<double, OsBlocks<double>> tri_v;  // 9 per triangle
        std::vector<rt_sphere> spheres;
        std::vector<rt_quad> quads;
    };
    std::unique_ptr<Reference64Source> ref64_source;  // released once the arrays are on the device
    std::vector<Node64> nodes64;
    std::vector<Prim64> prims64;
    bool ref64_ready = false;
    // orders rt_render_device_noalloc calls on this scene's device (they share one static scratch array per GPU)
    hipEvent_t* noalloc_done = nullptr;

    ~rt_scene() {
        if (device >= 0) {
            int prev = 0;
            if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(device) == hipSuccess) {
                wf.reset();
                for (void* p : allocations) (void)hipFree(p);
                for (Slot& sl : slots) {
                    if (sl.d_acc) (void)hipFree(sl.d_acc);
                    if (sl.d_overflow) (void)hipFree(sl.d_overflow);
                    if (sl.ev0) (void)hipEventDestroy(sl.ev0);
                    if (sl.ev1) (void)hipEventDestroy(sl.ev1);
                    if (sl.done) (void)hipEventDestroy(sl.done);
                }
                (void)hipSetDevice(prev);
            }
        }
    }
};

namespace {

template <typename T>
const T* upload(rt_scene& s, const std::vector<T>& v) {
    void* p = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    HIP_CHECK(hipMalloc(&p, bytes));
    s.allocations.push_back(p);
    if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    s.info.device_bytes += bytes;
    return (const T*)p;
}

void validate(const rt_scene_desc& d) {
    if (d.n_surfaces == 0) throw std::runtime_error("scene is empty");
    if (d.n_surfaces >= (1u << 28)) throw std::runtime_error("too many surfaces");
    auto check_mat = [&](int32_t m) {
        if (m < 0 || (size_t)m >= d.n_materials) throw std::runtime_error("material index out of range");
    };
    for (size_t i = 0; i < d.n_materials; ++i) {
        const rt_material& m = d.materials[i];
        if (m.kind < 0 || m.kind > RT_MAT_ISOTROPIC) throw std::runtime_error("unknown material kind");
        if (m.texture < 0 || m.texture > RT_TEX_IMAGE) throw std::runtime_error("unknown texture kind");
        if (m.texture == RT_TEX_MARBLE && (m.tex_index < 0 || (size_t)m.tex_index >= d.n_perlin))
            throw std::runtime_error("noise table index out of range");
        if (m.texture == RT_TEX_IMAGE && (m.tex_index < 0 || (size_t)m.tex_index >= d.n_images))
            throw std::runtime_error("image index out of range");
        if ((m.texture == RT_TEX_CHECKER) && !(m.tex_scale != 0.0)) throw std::runtime_error("checker scale must not be 0");
    }
    for (size_t i = 0; i < d.n_spheres; ++i) check_mat(d.spheres[i].material);
    for (size_t i = 0; i < d.n_triangles; ++i) check_mat(d.triangles[i].material);
    for (size_t i = 0; i < d.n_quads; ++i) check_mat(d.quads[i].material);
    for (size_t i = 0; i < d.n_media; ++i) {
        const rt_medium& m = d.media[i];
        check_mat(m.material);
        if (d.materials[m.material].kind != RT_MAT_ISOTROPIC) throw std::runtime_error("a medium needs an isotropic material");
        if (!(m.density > 0.0)) throw std::runtime_error("medium density must be positive");
        const size_t n = m.boundary == RT_SURF_SPHERE ? d.n_spheres : m.boundary == RT_SURF_QUAD ? d.n_quads : 0;
        if (m.first < 0 || m.count < 1 || (size_t)m.first + (size_t)m.count > n || (m.boundary == RT_SURF_SPHERE && m.count != 1))
            throw std::runtime_error("medium boundary out of range");
    }
    if (d.n_media > kMaxMedia) throw std::runtime_error("too many media (at most " + std::to_string(kMaxMedia) + ")");
    if (d.n_groups > 0 && (!d.groups || !d.surface_group)) throw std::runtime_error("groups given without surface_group");
    for (size_t i = 0; i < d.n_surfaces; ++i) {
        const rt_surface_ref& s = d.surfaces[i];
        size_t n = s.type == RT_SURF_SPHERE ? d.n_spheres
                   : s.type == RT_SURF_TRIANGLE ? d.n_triangles
                   : s.type == RT_SURF_QUAD ? d.n_quads
                   : s.type == RT_SURF_MEDIUM ? d.n_media
                                              : 0;
        if (s.index < 0 || (size_t)s.index >= n) throw std::runtime_error("surface reference out of range");
    }
}

struct FlatTree {
    std::vector<int32_t> inner_id;  // tree node -> device node index (inner nodes only)
    bool root_is_leaf = false;
    size_t n_dev_nodes = 0;
    explicit FlatTree(const Bvh& bvh) : inner_id(bvh.nodes.size(), -1) {
        root_is_leaf = bvh.nodes.empty() || bvh.nodes[0].is_leaf();
        int32_t n = 0;
        for (size_t i = 0; i < bvh.nodes.size(); ++i)
            if (!bvh.nodes[i].is_leaf()) inner_id[i] = n++;
        n_dev_nodes = root_is_leaf ? 1 : (size_t)n;
    }
    int32_t child_ref(const Bvh& bvh, int32_t node) const {
        const BvhNode& n = bvh.nodes[node];
        return n.is_leaf() ? ~n.first : inner_id[node];
    }
};

// What flatten_reference64 reads of the caller's scene description (kept with the scene until it has run).
void capture_reference64_source(const rt_scene_desc& d, rt_scene& s) {
    auto src = std::make_unique<rt_scene::Reference64Source>();
    src->surfaces.assign(d.surfaces, d.surfaces + d.n_surfaces);
    src->tri_v.resize(9 * d.n_triangles);
    for (size_t i = 0; i < d.n_triangles; ++i) std::memcpy(&src->tri_v[9 * i], d.triangles[i].v, 9 * sizeof(double));
    src->spheres.assign(d.spheres, d.spheres + d.n_spheres);
    src->quads.assign(d.quads, d.quads + d.n_quads);
    s.ref64_source = std::move(src);
}

// Reference-topology tree -> f64 arrays of the exact-parity kernel.  Inner nodes keep their children's boxes;
// leaves become runs of slots in the in-order sequence, closed by a last-in-leaf flag.
void flatten_reference64(rt_scene& s) {
    const Bvh& bvh = s.bvh;
    const rt_scene::Reference64Source& d = *s.ref64_source;
    const size_t n_surf = d.surfaces.size();
    const FlatTree ft(bvh);
    s.nodes64.assign(ft.n_dev_nodes, Node64{});
    auto put = [&](size_t node, int c, const Aabb& b, int32_t ref) {
        for (int a = 0; a < 3; ++a) s.nodes64[node].lo[c][a] = b.lo.get(a), s.nodes64[node].hi[c][a] = b.hi.get(a);
        s.nodes64[node].child[c] = ref;
    };
    if (ft.root_is_leaf) {
        put(0, 0, bvh.nodes[0].box, ~0);
        put(0, 1, bvh.nodes[0].box, kNoChild);
    } else {
        for (size_t i = 0; i < bvh.nodes.size(); ++i) {
            const BvhNode& n = bvh.nodes[i];
            if (n.is_leaf()) continue;
            put(ft.inner_id[i], 0, bvh.nodes[n.left].box, ft.child_ref(bvh, n.left));
            put(ft.inner_id[i], 1, bvh.nodes[n.right].box, ft.child_ref(bvh, n.right));
        }
    }
    std::vector<uint8_t> last(n_surf, 0);
    for (const BvhNode& n : bvh.nodes)
        if (n.is_leaf() && n.count > 0) last[n.first + n.count - 1] = 1;
    s.prims64.assign(n_surf, Prim64{});
    for (size_t slot = 0; slot < n_surf; ++slot) {
        const int32_t sid = bvh.order[slot];
        const rt_surface_ref& ref = d.surfaces[sid];
        Prim64& p64 = s.prims64[slot];
        p64.type = ref.type, p64.last = last[slot], p64.surface = sid;
        if (ref.type == RT_SURF_TRIANGLE) {
            std::memcpy(p64.a, &d.tri_v[9 * (size_t)ref.index], 9 * sizeof(double));
        } else if (ref.type == RT_SURF_MEDIUM) {
            // no deterministic hit: the parity AOV sees surfaces only (prim_hit knows no type 3)
        } else if (ref.type == RT_SURF_SPHERE) {
            const rt_sphere& sp = d.spheres[ref.index];
            for (int a = 0; a < 3; ++a) p64.a[a] = sp.center[a];
            p64.a[3] = sp.radius;
        } else {
            const rt_quad& q = d.quads[ref.index];
            for (int a = 0; a < 3; ++a) {
                p64.a[a] = q.normal[a], p64.a[4 + a] = q.q[a], p64.a[7 + a] = q.u[a];
                p64.a[10 + a] = q.v[a], p64.a[13 + a] = q.w[a];
            }
            p64.a[3] = q.d;
        }
    }
}

// the reference-topology tree's figures (rt_scene_info)
void reference_tree_info(rt_scene& s, size_t n_surfaces) {
    s.info.n_surfaces = n_surfaces;
    s.info.n_nodes = s.bvh.ref_nodes ? s.bvh.ref_nodes : s.bvh.nodes.size();
    s.info.n_leaves = s.bvh.n_leaves;
    s.info.max_depth = s.bvh.max_depth;
    s.info.max_leaf_size = s.bvh.max_leaf_size;
}

// A tree that came back from the GPU is checked before anything traverses it: every primitive in exactly one leaf,
// every leaf's box around its primitives, every inner box around its children, ranges that tile the parent's.
void check_device_tree(const std::vector<float>& bounds, const std::vector<int32_t>& ids, const Bvh& t) {
    auto bad = [](const char* what) { throw std::runtime_error(std::string("device BVH build: invalid tree: ") + what); };
    const size_t n = ids.size();
    if (t.order.size() != n) bad("order length");
    if (n == 0) return;
    if (t.nodes.empty() || t.nodes.size() > 2 * n) bad("node count");
    std::vector<int32_t> prim_of(n);  // slot -> primitive index, through the surface ids
    {
        int32_t max_id = 0;
        for (int32_t id : ids) max_id = std::max(max_id, id);
        std::vector<int32_t> prim_of_id((size_t)max_id + 1, -1);
        for (size_t i = 0; i < n; ++i) prim_of_id[(size_t)ids[i]] = (int32_t)i;
        for (size_t slot = 0; slot < n; ++slot) {
            const int32_t id = t.order[slot];
            if (id < 0 || id > max_id || prim_of_id[(size_t)id] < 0) bad("order is not a permutation");
            prim_of[slot] = prim_of_id[(size_t)id], prim_of_id[(size_t)id] = -1;  // a second use fails
        }
    }
    if (t.nodes[0].first != 0 || (size_t)t.nodes[0].count != n) bad("root range");
    std::vector<uint8_t> visited(t.nodes.size(), 0);
    std::vector<int32_t> stack{0};
    size_t leaves = 0, covered = 0;
    while (!stack.empty()) {
        const int32_t i = stack.back();
        stack.pop_back();
        if (i < 0 || (size_t)i >= t.nodes.size() || visited[(size_t)i]) bad("child index");
        visited[(size_t)i] = 1;
        const BvhNode& nd = t.nodes[(size_t)i];
        if (nd.count <= 0 || nd.first < 0 || (size_t)nd.first + (size_t)nd.count > n) bad("node range");
        if (nd.is_leaf()) {
            ++leaves, covered += (size_t)nd.count;
            for (int32_t k = nd.first; k < nd.first + nd.count; ++k) {
                const float* b = &bounds[6 * (size_t)prim_of[(size_t)k]];
                if (!(nd.box.lo.x <= b[0] && nd.box.lo.y <= b[1] && nd.box.lo.z <= b[2] && nd.box.hi.x >= b[3] &&
                      nd.box.hi.y >= b[4] && nd.box.hi.z >= b[5]))
                    bad("leaf box");
            }
            continue;
        }
        if (nd.left < 0 || nd.right < 0 || (size_t)nd.left >= t.nodes.size() || (size_t)nd.right >= t.nodes.size()) bad("child index");
        const BvhNode &l = t.nodes[(size_t)nd.left], &r = t.nodes[(size_t)nd.right];
        if (l.first != nd.first || r.first != l.first + l.count || l.count + r.count != nd.count) bad("children do not tile the parent's range");
        for (const BvhNode* c : {&l, &r})
            if (!(nd.box.lo.x <= c->box.lo.x && nd.box.lo.y <= c->box.lo.y && nd.box.lo.z <= c->box.lo.z &&
                  nd.box.hi.x >= c->box.hi.x && nd.box.hi.y >= c->box.hi.y && nd.box.hi.z >= c->box.hi.z))
                bad("inner box");
        stack.push_back(nd.right), stack.push_back(nd.left);
    }
    if (covered != n || leaves != t.n_leaves) bad("leaves do not cover the primitives");
}

// Which of the scene's spheres and quads are tested by every ray ("loose") rather than put into the render tree.
// Counted are the surfaces of their own (a medium's boundary is not one: it is met through the medium).  Measured
// (tools/loose_crossover.py, the example scenes): without triangles — no tree at all then, the shade-only build with
// every lane busy — loose wins up to 32 of them (20: +19 %, 32: +5 %, 40: -6 %; the cornell example's 18 quads in
// their closed box, where a tree culls nothing: 2.2x); beside a mesh the tree is there anyway, but the scene files'
// ground spheres and room walls span it: teapot.yaml and cornell_lucy.yaml lose 55 % and 42 % with their six in the
// tree, which also takes the general kernel build and its fourth wave.  A scene with more than that keeps its giants
// loose all the same — a sphere or quad whose box reaches over at least half the extent, on two axes or more, of the
// box of everything smaller than it: the radius-1000 ground of one-weekend (inside the tree it would stretch the
// root over the whole sky, and no ray would be spared the traversal by the bounds test), the walls of a room
// (cornell_lucy.yaml with its six in the tree: -46 %).
static double box_area(const Aabb& b) {
    const V3 l = b.hi - b.lo;
    return 2.0 * (l.x * l.y + l.y * l.z + l.z * l.x);
}
// does the box reach over at least half of `rest`'s extent on two axes or more?
static bool spans_two_axes(const Aabb& b, const Aabb& rest) {
    int n = 0;
    for (int a = 0; a < 3; ++a) {
        const double mine = b.hi.get(a) - b.lo.get(a), theirs = rest.hi.get(a) - rest.lo.get(a);
        n += theirs > 0.0 && mine >= 0.5 * theirs;
    }
    return n >= 2;
}
static std::vector<char> loose_surfaces(const rt_scene_desc& d) {
    std::vector<char> loose(d.n_surfaces, 0);
    std::vector<std::pair<double, size_t>> cand;  // (box area, surface)
    Aabb triangles_box = Aabb::empty();
    size_t triangles = 0;
    for (size_t i = 0; i < d.n_surfaces; ++i) {
        const int32_t type = d.surfaces[i].type;
        if (type == RT_SURF_SPHERE || type == RT_SURF_QUAD) cand.push_back({box_area(true_bounds(d, d.surfaces[i])), i});
        if (type == RT_SURF_TRIANGLE) triangles_box = triangles_box.merge(true_bounds(d, d.surfaces[i])), ++triangles;
    }
    if (cand.size() <= (triangles ? kMaxLooseBesideMesh : kMaxLoose)) {
        for (const auto& c : cand) loose[c.second] = 1;
        return loose;
    }
    std::stable_sort(cand.begin(), cand.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
    std::vector<Aabb> rest(cand.size() + 1);  // rest[k]: the box of candidates k.. and of the triangles
    rest[cand.size()] = triangles_box;
    for (size_t k = cand.size(); k-- > 0;) rest[k] = rest[k + 1].merge(true_bounds(d, d.surfaces[cand[k].second]));
    for (size_t k = 0; k < std::min<size_t>(cand.size(), kMaxLooseBesideMesh); ++k) {
        if (!spans_two_axes(true_bounds(d, d.surfaces[cand[k].second]), rest[k + 1])) break;
        loose[cand[k].second] = 1;
    }
    return loose;
}

constexpr uint32_t kRenderLeaf = 2;   // leaf size of the render tree (measured: 2 beats 4 and 8 by 4-5 %) ...
constexpr uint32_t kRenderLeafF64 = 1;  // ... and 1 when it holds spheres or quads, whose f64 tests cost three box tests
                                        //     (final-scene +8.5 %, random-spheres +5 %, one-weekend +2 % over 2)

// The tree the render kernels traverse: triangles, plus spheres/quads when the scene has too many of them to
// test "loose".  Independent of the reference-topology tree, so the two builds run side by side.
void build_render_tree(const rt_scene_desc& d, rt_scene& s, int build_device, const BuildKnobs& knobs) {
    s.is_loose = loose_surfaces(d);
    std::vector<int32_t>& tree_ids = s.render_ids;
    tree_ids.clear();
    for (size_t i = 0; i < d.n_surfaces; ++i) {
        if (d.surfaces[i].type == RT_SURF_MEDIUM) continue;  // media are evaluated after the surface trace
        if (!s.is_loose[i]) tree_ids.push_back((int32_t)i);
    }
    Bvh& rb = s.render_bvh;
    bool only_triangles = true;
    for (int32_t id : tree_ids) only_triangles = only_triangles && d.surfaces[id].type == RT_SURF_TRIANGLE;
    const uint32_t max_leaf = knobs.leaf ? knobs.leaf : (only_triangles ? kRenderLeaf : kRenderLeafF64);
    if (d.use_bvh && build_device >= 0) {
        // opt-in (use_bvh == 2 or RT_BUILD=device): the same builder as build_render_bvh, level by level on the GPU.
        // Its input is each primitive's f32 box, rounded outward from the bounds the host builder uses.
        std::vector<float> bounds(6 * tree_ids.size());
        for (size_t i = 0; i < tree_ids.size(); ++i) {
            const rt_surface_ref& ref = d.surfaces[tree_ids[i]];
            const Aabb b = true_bounds(d, ref).pad(ref.type == RT_SURF_QUAD ? 0.0001 : 0.0);
            for (int a = 0; a < 3; ++a) {
                float lo = (float)b.lo.get(a), hi = (float)b.hi.get(a);
                if ((double)lo > b.lo.get(a)) lo = std::nextafterf(lo, -INFINITY);
                if ((double)hi < b.hi.get(a)) hi = std::nextafterf(hi, INFINITY);
                bounds[6 * i + a] = lo, bounds[6 * i + 3 + a] = hi;
            }
        }
        if (hipSetDevice(build_device) != hipSuccess) throw std::runtime_error("device BVH build: cannot select the device");
        build_render_bvh_device(bounds, tree_ids, max_leaf, knobs, rb);
        check_device_tree(bounds, tree_ids, rb);
    } else if (d.use_bvh) {
        build_render_bvh(d, tree_ids, max_leaf, knobs.traversal_cost, rb);
    } else {  // --no-bvh: one leaf, every surface tested by every ray (main.rs:60-65)
        rb = Bvh{};
        rb.order.assign(tree_ids.begin(), tree_ids.end());
        if (!tree_ids.empty()) {
            BvhNode leaf;
            leaf.box = Aabb::empty();
            for (int32_t id : tree_ids) {
                leaf.box = leaf.box.merge(true_bounds(d, d.surfaces[id]));
            }
            leaf.first = 0, leaf.count = (int32_t)tree_ids.size();
            rb.nodes.push_back(leaf);
            rb.max_depth = 1, rb.n_leaves = 1, rb.max_leaf_size = (uint32_t)tree_ids.size();
        }
    }
}

// An integer 0..32768 as IEEE binary16 bits, rounded up to the next representable value (11 significant bits):
// the half-extent of a child box, which the kernels feed to v_fma_mix_f32 without a conversion instruction.
uint32_t half_bits_up(uint32_t e) {
    if (e == 0) return 0;
    int top = 31 - __builtin_clz(e);  // e = 1.m * 2^top
    if (top > 10) {
        const int drop = top - 10;
        e = ((e + (1u << drop) - 1u) >> drop) << drop;
        top = 31 - __builtin_clz(e);  // the round-up may have carried into the next power of two
    }
    const uint32_t mant = top > 10 ? (e >> (top - 10)) & 0x3ffu : (e << (10 - top)) & 0x3ffu;
    return ((uint32_t)(top + 15) << 10) | mant;
}

// Render tree + loose list -> f32 arrays of the render kernels.
void flatten_render(const rt_scene_desc& d, rt_scene& s) {
    const size_t n_surf = d.n_surfaces;
    // rank = position in the reference tree's in-order leaf sequence: decides equal-t ties (bvh.rs:272-273,308)
    std::vector<int32_t> rank(n_surf);
    for (size_t i = 0; i < n_surf; ++i) rank[s.bvh.order[i]] = (int32_t)i;

    const std::vector<int32_t>& tree_ids = s.render_ids;
    s.loose.clear();
    for (size_t i = 0; i < n_surf; ++i) {
        const rt_surface_ref& ref = d.surfaces[i];
        if (s.is_loose[i]) s.loose.push_back(int4{ref.type, ref.index, rank[i], (int32_t)i});
    }
    Bvh& rb = s.render_bvh;
    const size_t n_tree = tree_ids.size();
    s.has_tree = n_tree > 0;
    bool only_tris = true, only_spheres = true, only_quads = true;
    for (int32_t id : tree_ids) {
        only_tris = only_tris && d.surfaces[id].type == RT_SURF_TRIANGLE;
        only_spheres = only_spheres && d.surfaces[id].type == RT_SURF_SPHERE;
        only_quads = only_quads && d.surfaces[id].type == RT_SURF_QUAD;
    }
    s.dev.tree_kind = only_tris ? 1u : (only_spheres ? 2u : (only_quads ? 3u : 0u));
    const FlatTree ft(rb);
    const size_t n_slots = n_tree + ((s.has_tree && ft.root_is_leaf) ? 1 : 0);

    // f32 boxes must stay conservative under the fma slab form lo*inv_d - o*inv_d, whose rounding error in
    // position space is about |o| * 2^-24: inflate every box by the scene's largest coordinate * 2^-22.
    double maxabs = 0.0;
    const Aabb& all = s.bvh.nodes[0].box;
    for (int a = 0; a < 3; ++a) {
        maxabs = std::fmax(maxabs, std::fabs(all.lo.get(a)));
        maxabs = std::fmax(maxabs, std::fabs(all.hi.get(a)));
        maxabs = std::fmax(maxabs, std::fabs(d.camera.origin[a]));
    }
    const double inflate = maxabs * 0x1p-22;

    // The node boxes live on a 16-bit grid over the tree's (inflated) bounds: plane q of axis a = grid_lo[a] +
    // q * grid_step[a].  lo planes round down and hi planes up, each by one further step, which covers the f32
    // rounding of the kernels' q * (step / d) + (grid_lo - o) / d many times over (a step is 2^-16 of the extent).
    double glo[3], gstep[3];
    for (int a = 0; a < 3; ++a) {
        const double lo = s.has_tree ? rb.nodes[0].box.lo.get(a) - inflate : 0.0;
        const double hi = s.has_tree ? rb.nodes[0].box.hi.get(a) + inflate : 1.0;
        // two steps of margin at either end, so that the outward steps below stay on the grid
        const double step = std::fmax((hi - lo) / 65531.0, std::fmax(std::fabs(lo), std::fabs(hi)) * 0x1p-20 + 1e-30);
        gstep[a] = (double)(float)step;
        if (gstep[a] < step) gstep[a] = (double)std::nextafterf((float)step, INFINITY);
        glo[a] = (double)round_down(lo - 2.0 * gstep[a]);
        s.grid_lo[a] = (float)glo[a], s.grid_step[a] = (float)gstep[a];
        if (glo[a] + 65535.0 * gstep[a] < hi + gstep[a]) throw std::runtime_error("render tree does not fit its plane grid");
    }
    s.nodes.assign(2 * std::max<size_t>(ft.n_dev_nodes, 1), uint4{0, 0, 0, 0});
    auto put_child = [&](size_t node, int c, const Aabb& b, int32_t ref) {
        uint32_t w[3];
        for (int a = 0; a < 3; ++a) {
            const double qlo = std::floor((b.lo.get(a) - inflate - glo[a]) / gstep[a]) - 1.0;
            const double qhi = std::ceil((b.hi.get(a) + inflate - glo[a]) / gstep[a]) + 1.0;
            const uint32_t lo = (uint32_t)std::fmin(std::fmax(qlo, 0.0), 65535.0), hi = (uint32_t)std::fmin(std::fmax(qhi, 0.0), 65535.0);
            // stored as centre and half-extent (child_slab: one fma to the centre plane, one each to the faces; no
            // per-axis min/max): centre = floor of the mid-plane, half-extent = what still covers both faces
            const uint32_t centre = (lo + hi) >> 1, half = std::max(centre - lo, hi - centre);
            w[a] = centre | (half_bits_up(half) << 16);
        }
        s.nodes[2 * node + c] = uint4{w[0], w[1], w[2], (uint32_t)ref};
    };
    s.stack_depth = 2;
    if (s.has_tree && ft.root_is_leaf) {
        // synthetic root: child 0 = the leaf; child 1 = the grid's far corner over one dummy slot that can never
        // be hit
        put_child(0, 0, rb.nodes[0].box, ~0);
        s.nodes[1] = uint4{0x0000ffffu, 0x0000ffffu, 0x0000ffffu, (uint32_t)~(int32_t)n_tree};  // centre 65535, half-extent +0.0
    } else if (s.has_tree) {
        parallel_chunks(rb.nodes.size(), 8192, 8, [&](size_t begin, size_t end) {
            for (size_t i = begin; i < end; ++i) {
                const BvhNode& n = rb.nodes[i];
                if (n.is_leaf()) continue;
                put_child(ft.inner_id[i], 0, rb.nodes[n.left].box, ft.child_ref(rb, n.left));
                put_child(ft.inner_id[i], 1, rb.nodes[n.right].box, ft.child_ref(rb, n.right));
            }
        });
        s.stack_depth = rb.max_depth;
    }

    std::vector<uint8_t> last(std::max<size_t>(n_slots, 1), 0);
    for (const BvhNode& n : rb.nodes)
        if (n.is_leaf() && n.count > 0) last[n.first + n.count - 1] = 1;
    s.prims.assign(3 * (std::max<size_t>(n_slots, 1) + 1), float4{0, 0, 0, 0});  // + one slot of padding (leaf_step)
    s.tri_shade.assign(3 * std::max<size_t>(n_slots, 1), float4{0, 0, 0, 0});
    s.slot_surface.assign(std::max<size_t>(n_slots, 1), -1);
    parallel_chunks(n_tree, 8192, 8, [&](size_t slot_begin, size_t slot_end) {
    for (size_t slot = slot_begin; slot < slot_end; ++slot) {
        const int32_t sid = rb.order[slot];
        const rt_surface_ref& ref = d.surfaces[sid];
        s.slot_surface[slot] = sid;
        uint32_t meta = (uint32_t)ref.type | (last[slot] ? kMetaLast : 0u);
        float4* p = &s.prims[3 * slot];
        if (ref.type == RT_SURF_TRIANGLE) {
            const rt_triangle& t = d.triangles[ref.index];
            p[0] = {(float)t.v[0][0], (float)t.v[0][1], (float)t.v[0][2], uint_bits(meta)};
            p[1] = {(float)(t.v[1][0] - t.v[0][0]), (float)(t.v[1][1] - t.v[0][1]), (float)(t.v[1][2] - t.v[0][2]), 0};
            p[2] = {(float)(t.v[2][0] - t.v[0][0]), (float)(t.v[2][1] - t.v[0][1]), (float)(t.v[2][2] - t.v[0][2]), 0};
            float4* sh = &s.tri_shade[3 * slot];
            for (int k = 0; k < 3; ++k) sh[k] = {(float)t.n[k][0], (float)t.n[k][1], (float)t.n[k][2], 0};
            sh[0].w = int_bits(t.material);
            if ((uint32_t)t.material >= (1u << (32 - kMetaIndexShift))) throw std::runtime_error("material id out of range");
            p[0].w = uint_bits(meta | ((uint32_t)t.material << kMetaIndexShift));
        } else {
            meta |= (uint32_t)ref.index << kMetaIndexShift;
            p[0].w = uint_bits(meta);
        }
        p[1].w = int_bits(rank[sid]);
    }
    });
    // The dummy slot is read by whatever test the kernel build compiles in (the spheres-only and quads-only builds
    // ignore the type bits): make it a record that test can never hit — a triangle with zero edges (det = 0), the
    // extra sphere with a NaN radius (disc > 0 fails), the extra quad with a zero normal (|denom| < 1e-8) below.
    if (s.has_tree && ft.root_is_leaf) {
        const uint32_t kind = s.dev.tree_kind;
        s.prims[3 * n_tree].w = kind == 2u   ? uint_bits(0u | kMetaLast | ((uint32_t)d.n_spheres << kMetaIndexShift))
                                : kind == 3u ? uint_bits(2u | kMetaLast | ((uint32_t)d.n_quads << kMetaIndexShift))
                                             : uint_bits(1u | kMetaLast);
    }

    s.spheres.assign(4 * (d.n_spheres + 1), 0.0);  // + the never-hit dummy
    s.spheres[4 * d.n_spheres + 3] = std::nan("");
    s.sphere_end.assign(3 * (d.n_spheres + 1), 0.0);
    s.ext = false, s.noise = false;
    s.sphere_mat.assign(d.n_spheres + 1, 0);
    for (size_t i = 0; i < d.n_spheres; ++i) {
        for (int a = 0; a < 3; ++a) s.spheres[4 * i + a] = d.spheres[i].center[a];
        s.spheres[4 * i + 3] = d.spheres[i].radius;
        for (int a = 0; a < 3; ++a)
            s.sphere_end[3 * i + a] = d.spheres[i].moving ? d.spheres[i].center1[a] : d.spheres[i].center[a];
        if (d.spheres[i].moving) s.ext = true;
        s.sphere_mat[i] = d.spheres[i].material;
    }
    // gate of a sphere / quad = box of the reference tree's leaf that holds it (see leaf_gate_hit in device_rt.h)
    s.sphere_gate.assign(6 * (d.n_spheres + 1), 0.0);
    s.quad_gate.assign(6 * (d.n_quads + 1), 0.0);
    for (const BvhNode& n : s.bvh.nodes) {
        if (!n.is_leaf()) continue;
        for (int32_t k = n.first; k < n.first + n.count; ++k) {
            const rt_surface_ref& ref = d.surfaces[s.bvh.order[k]];
            if (ref.type == RT_SURF_TRIANGLE || ref.type == RT_SURF_MEDIUM) continue;  // Triangle::bounding_box is exact
            double* gate = ref.type == RT_SURF_SPHERE ? &s.sphere_gate[6 * ref.index] : &s.quad_gate[6 * ref.index];
            // (a run of surfaces inside a reference leaf that also holds nested trees: the whole leaf's box)
            const Aabb& leaf_box = n.gate >= 0 ? s.bvh.nodes[n.gate].box : n.box;
            for (int a = 0; a < 3; ++a) gate[a] = leaf_box.lo.get(a), gate[3 + a] = leaf_box.hi.get(a);
            // a leaf box that encloses the surface is crossed by every ray that hits the surface: no test needed
            // (up to the rounding of a baked corner: the six quads of a turned box meet in corners that each quad reaches
            // by its own sums, q + u + v here, q there — an ulp apart, which is no reason to gate a face by its own box)
            const Aabb tb = true_bounds(d, ref);
            bool encloses = true;
            for (int a = 0; a < 3; ++a) {
                const double slack = 1e-9 * (tb.hi.get(a) - tb.lo.get(a));
                encloses = encloses && leaf_box.lo.get(a) <= tb.lo.get(a) + slack && tb.hi.get(a) - slack <= leaf_box.hi.get(a);
            }
            if (encloses) gate[0] = -INFINITY;
        }
    }
    s.quads.assign(16 * (d.n_quads + 1), 0.0);  // + the never-hit dummy (all zeros)
    s.quad_mat.assign(d.n_quads + 1, 0);
    for (size_t i = 0; i < d.n_quads; ++i) {
        const rt_quad& q = d.quads[i];
        V3 u{q.u[0], q.u[1], q.u[2]}, v{q.v[0], q.v[1], q.v[2]}, w{q.w[0], q.w[1], q.w[2]};
        V3 A = cross(v, w), B = cross(w, u);  // w·(p×v) = p·(v×w), w·(u×p) = p·(w×u)
        double* o = &s.quads[16 * i];
        for (int a = 0; a < 3; ++a) o[a] = q.normal[a], o[4 + a] = q.q[a];
        o[3] = q.d;
        o[7] = A.x, o[8] = A.y, o[9] = A.z, o[10] = B.x, o[11] = B.y, o[12] = B.z;
        s.quad_mat[i] = q.material;
    }
    // loose records: header + the sphere's / quad's doubles, self-contained (staged in LDS by the kernels)
    s.loose_rec.assign(kLooseRecQuads * std::max<size_t>(s.loose.size(), 1), uint4{0, 0, 0, 0});
    for (size_t i = 0; i < s.loose.size(); ++i) {
        const int4& l = s.loose[i];
        uint4* rec = &s.loose_rec[kLooseRecQuads * i];
        const int32_t mat = l.x == RT_SURF_SPHERE ? s.sphere_mat[l.y] : s.quad_mat[l.y];
        rec[0] = uint4{(uint32_t)l.x, (uint32_t)l.y, (uint32_t)l.z, (uint32_t)mat};
        if (l.x == RT_SURF_SPHERE) {
            std::memcpy(rec + 1, &s.spheres[4 * l.y], 4 * sizeof(double));
            std::memcpy(rec + 3, &s.sphere_gate[6 * l.y], 6 * sizeof(double));
            std::memcpy(rec + 6, &s.sphere_end[3 * l.y], 3 * sizeof(double));
        } else {
            double* data = reinterpret_cast<double*>(rec + 1);
            std::memcpy(data, &s.quads[16 * l.y], 13 * sizeof(double));
            std::memcpy(data + 13, &s.quad_gate[6 * l.y], 6 * sizeof(double));
        }
    }
    s.materials.resize(d.n_materials);
    s.mat_kind.resize(d.n_materials);
    s.mat_tex.assign(2 * std::max<size_t>(d.n_materials, 1), float4{0, 0, 0, 0});
    s.mat_scale.assign(2 * std::max<size_t>(d.n_materials, 1), 1.0);
    for (size_t i = 0; i < d.n_materials; ++i) {
        const rt_material& m = d.materials[i];
        s.materials[i] = {(float)m.albedo[0], (float)m.albedo[1], (float)m.albedo[2], (float)m.param};
        s.mat_kind[i] = m.kind;
        s.mat_tex[2 * i] = {int_bits(m.texture), int_bits(m.tex_index), 0, 0};
        s.mat_tex[2 * i + 1] = {(float)m.albedo2[0], (float)m.albedo2[1], (float)m.albedo2[2], 0};
        s.mat_scale[2 * i] = m.tex_scale, s.mat_scale[2 * i + 1] = 1.0 / m.tex_scale;  // f64::recip, material.rs:195
        if (m.texture != RT_TEX_SOLID) s.ext = true;
        if (m.texture == RT_TEX_MARBLE) s.noise = true;
    }
    s.media.assign(std::max<size_t>(d.n_media, 1), DevMedium{});
    for (size_t k = 0; k < d.n_media; ++k) {
        const rt_medium& m = d.media[k];
        s.media[k] = DevMedium{m.boundary, m.first, m.count, m.material, -(1.0 / m.density), 0.0};
        s.ext = true;
    }
    s.perlin_vec.assign(256 * std::max<size_t>(d.n_perlin, 1), float4{0, 0, 0, 0});
    s.perlin_perm.assign(768 * std::max<size_t>(d.n_perlin, 1), 0);
    for (size_t k = 0; k < d.n_perlin; ++k)
        for (int i = 0; i < 256; ++i) {
            const double* v = d.perlin[k].ranvec[i];
            s.perlin_vec[256 * k + i] = {(float)v[0], (float)v[1], (float)v[2], 0};
            for (int a = 0; a < 3; ++a) s.perlin_perm[768 * k + 256 * a + i] = d.perlin[k].perm[a][i];
        }
    s.image_rgb.assign(1, 0);
    s.image_dims.assign(std::max<size_t>(d.n_images, 1), uint4{0, 0, 0, 0});
    for (size_t k = 0; k < d.n_images; ++k) {
        const rt_image& im = d.images[k];
        const size_t n = (size_t)3 * im.width * im.height;
        s.image_dims[k] = uint4{(uint32_t)s.image_rgb.size(), n ? im.width : 0u, n ? im.height : 0u, 0u};
        if (n) s.image_rgb.insert(s.image_rgb.end(), im.rgb, im.rgb + n);
    }
    s.info.render_nodes = rb.nodes.size();
    s.info.render_depth = s.stack_depth;
    s.info.render_max_leaf = rb.max_leaf_size;
    s.info.n_loose = (uint32_t)s.loose.size();
}

void fill_camera(const rt_scene_desc& d, DevScene& dev) {
    const rt_camera& c = d.camera;
    for (int a = 0; a < 3; ++a) {
        dev.cam.origin[a] = c.origin[a];
        dev.cam.ll_rel[a] = (float)(c.lower_left[a] - c.origin[a]);
        dev.cam.u[a] = (float)c.u[a];
        dev.cam.v[a] = (float)c.v[a];
        dev.cam64[0][a] = c.origin[a], dev.cam64[1][a] = c.lower_left[a];
        dev.cam64[2][a] = c.horizontal[a], dev.cam64[3][a] = c.vertical[a];
        dev.background[a] = (float)d.background[a];
    }
    dev.cam.lens_radius = (float)c.lens_radius;
}

struct DeviceGuard {
    int prev = -1;
    bool active = false;
    explicit DeviceGuard(int device) {
        if (device < 0) return;
        HIP_CHECK(hipGetDevice(&prev));
        if (prev != device) {
            HIP_CHECK(hipSetDevice(device));
            active = true;
        }
    }
    ~DeviceGuard() {
        if (active) (void)hipSetDevice(prev);
    }
};

DevRenderParams make_params(const rt_scene& s, const rt_render_params& p) {
    if (p.width == 0 || p.height == 0) throw std::runtime_error("width and height must be positive");
    if ((uint64_t)p.width * p.height >= (1ull << 31)) throw std::runtime_error("image too large");
    if (p.sn == 0 || p.sn > 65535) throw std::runtime_error("sn must be in [1, 65535]");
    if (p.max_depth == 0) throw std::runtime_error("max_depth must be positive");
    if (s.device < 0) throw std::runtime_error("scene was created host-only (no device): cannot render");
    DevRenderParams rp{};
    rp.width = p.width, rp.height = p.height, rp.sn = p.sn, rp.max_depth = p.max_depth;
    rp.n_strata = p.sn * p.sn;
    rp.stratum_begin = p.stratum_begin;
    rp.stratum_step = p.stratum_step ? p.stratum_step : 1;
    rp.seed_lo = (uint32_t)p.seed, rp.seed_hi = (uint32_t)(p.seed >> 32);
    rp.recip_sn = (float)(1.0 / (double)p.sn);
    // floor(k / sn) == umulhi(k, magic) for every stratum k < sn*sn as long as sn^3 < 2^32
    rp.sn_magic = p.sn < 1600u ? (uint32_t)((1ull << 32) / p.sn + 1ull) : 0u;
    if (p.sampler != RT_SAMPLER_REFERENCE && p.sampler != RT_SAMPLER_DIRECT) throw std::runtime_error("unknown sampler");
    rp.sampler = (uint32_t)p.sampler;
    return rp;
}

}  // namespace

// hor_px / ver_px depend on the image size, so they are filled per call from the f64 camera.
static void camera_for_size(const rt_scene& s, uint32_t width, uint32_t height, DevScene& dev) {
    dev = s.dev;
    for (int a = 0; a < 3; ++a) {
        dev.cam.hor_px[a] = (float)(s.dev.cam64[2][a] / (double)width);
        dev.cam.ver_px[a] = (float)(s.dev.cam64[3][a] / (double)height);
    }
}

// The f64 parity arrays, flattened and uploaded at their first use (the scene's device must be current).
static void ensure_reference64(rt_scene& s) {
    if (s.ref64_ready) return;
    if (!s.ref64_source) throw std::runtime_error("the scene lost the source of its f64 parity arrays");
    flatten_reference64(s);
    s.dev.nodes64 = upload(s, s.nodes64);
    s.dev.prims64 = upload(s, s.prims64);
    s.ref64_source.reset();
    s.nodes64 = {}, s.prims64 = {};
    s.ref64_ready = true;
}

extern "C" {

const char* rt_last_error(void) { return g_last_error.c_str(); }
const char* rt_version(void) { return "raytracer_amd 0.1 (gfx950)"; }
void rt_set_verbose(int v) { rt::g_verbose = v; }

int rt_load_scene(const char* yaml_path, double aspect_ratio, int use_bvh, rt_scene_desc** out) {
    if (!yaml_path || !out) return set_error("rt_load_scene: null argument");
    try {
        auto hs = std::make_unique<HostScene>();
        set_workers_disabled(BuildKnobs::from_environment().one_thread);
        load_scene_file(yaml_path, aspect_ratio, use_bvh != 0, *hs);
        if (use_bvh == 2) hs->desc.use_bvh = 2;
        *out = &hs.release()->desc;
        return 0;
    } catch (const std::exception& e) {
        return set_error(std::string("load scene file '") + yaml_path + "': " + e.what());
    }
}

int rt_example_scene(const char* name, double aspect_ratio, uint64_t seed, int use_bvh, rt_scene_desc** out) {
    if (!name || !out) return set_error("rt_example_scene: null argument");
    try {
        auto hs = std::make_unique<HostScene>();
        example_scene(name, aspect_ratio, seed, use_bvh != 0, *hs);
        if (use_bvh == 2) hs->desc.use_bvh = 2;
        *out = &hs.release()->desc;
        return 0;
    } catch (const std::exception& e) {
        return set_error(e.what());
    }
}

void rt_scene_desc_free(rt_scene_desc* desc) {
    if (desc) delete reinterpret_cast<HostScene*>(desc);
}

int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** out) {
    if (!desc || !out) return set_error("rt_scene_create: null argument");
    try {
        validate(*desc);
        auto s = std::make_unique<rt_scene>();
        auto t0 = std::chrono::steady_clock::now();
        // Every knob of scene creation comes from the environment HERE, on the calling thread, before the first
        // worker starts: the workers read no environment (host/worker.h).
        const BuildKnobs knobs = BuildKnobs::from_environment();
        set_workers_disabled(knobs.one_thread);
        // The device context (0.1-0.2 s in a cold process: HIP runtime, code objects) is created on the calling
        // thread, by the first HIP call below, and never beside this library's workers (with the runtime
        // initialising on a short-lived thread while the builds started and joined some thirty threads, one cold
        // start in 50-100 died of a SIGSEGV that no handler saw — the runtime's own threads block every signal —
        // and 0 of 760 have without the overlap).
        {
            int build_device = -1;  // -1: the render tree is built on the host
            if ((desc->use_bvh == 2 || (desc->use_bvh && knobs.device_build)) && device != RT_DEVICE_NONE) {
                if (device >= 0) build_device = device;
                else HIP_CHECK(hipGetDevice(&build_device));
            }
            s->info.device_build = build_device >= 0 ? 1u : 0u;
            if (build_device >= 0) {
                // the device build runs as a task of the pool beside the reference-topology build: the context is
                // created here first, on this thread, before any other thread exists (see above)
                DeviceGuard first_use(build_device);
                HIP_CHECK(hipFree(nullptr));
            }
            // (tasks of the pool, host/worker.h: beside this thread's work, or — with RT_BUILD_THREADS=1 — right here)
            ScopedTask render_job([&, build_device]() {
                const auto r0 = std::chrono::steady_clock::now();
                build_render_tree(*desc, *s, build_device, knobs);
                s->info.render_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r0).count();
            });
            build_bvh(*desc, s->bvh);
            reference_tree_info(*s, desc->n_surfaces);
            const auto t1 = std::chrono::steady_clock::now();
            render_job.get();
            const auto t2 = std::chrono::steady_clock::now();
            // beside the render flattening: the copy of what the f64 parity arrays will be made of, if ever asked for
            ScopedTask source64([&]() { capture_reference64_source(*desc, *s); });
            flatten_render(*desc, *s);
            const auto t3 = std::chrono::steady_clock::now();
            source64.get();
            if (knobs.debug) {
                auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
                fprintf(stderr, "scene create: reference-topology tree %.1f ms (render tree beside it: %.1f ms, waited %.1f ms more), "
                                "flatten render %.1f ms (f64 source copy beside it, waited %.1f ms more)\n",
                        ms(t0, t1), s->info.render_build_ms, ms(t1, t2), ms(t2, t3), ms(t3, std::chrono::steady_clock::now()));
            }
        }
        s->info.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (s->stack_depth > 56)
            throw std::runtime_error("BVH depth " + std::to_string(s->stack_depth) +
                                     " exceeds the traversal stack (56 levels)");
        fill_camera(*desc, s->dev);
        s->dev.n_slots = (uint32_t)s->slot_surface.size();
        s->dev.n_loose = (uint32_t)s->loose.size();
        s->dev.has_tree = s->has_tree ? 1u : 0u;
        for (int a = 0; a < 3; ++a) s->dev.grid_lo[a] = s->grid_lo[a], s->dev.grid_step[a] = s->grid_step[a];
        if (device == RT_DEVICE_NONE) {
            s->device = -1;
            *out = s.release();
            return 0;
        }
        if (device < 0) HIP_CHECK(hipGetDevice(&device));
        s->device = device;
        DeviceGuard guard(device);
        s->dev.nodes = upload(*s, s->nodes);
        s->dev.prims = upload(*s, s->prims);
        s->dev.tri_shade = upload(*s, s->tri_shade);
        s->dev.loose = upload(*s, s->loose);
        s->dev.loose_rec = upload(*s, s->loose_rec);
        s->dev.spheres = upload(*s, s->spheres);
        s->dev.sphere_gate = upload(*s, s->sphere_gate);
        s->dev.quad_gate = upload(*s, s->quad_gate);
        s->dev.ext = s->ext ? (s->noise ? 3u : 1u) : 0u;
        s->dev.sphere_end = upload(*s, s->sphere_end);
        s->dev.n_media = (uint32_t)desc->n_media;
        s->dev.media = upload(*s, s->media);
        s->dev.mat_tex = upload(*s, s->mat_tex);
        s->dev.mat_scale = upload(*s, s->mat_scale);
        s->dev.perlin_vec = upload(*s, s->perlin_vec);
        s->dev.perlin_perm = upload(*s, s->perlin_perm);
        s->dev.image_rgb = upload(*s, s->image_rgb);
        s->dev.image_dims = upload(*s, s->image_dims);
        s->dev.quads = upload(*s, s->quads);
        s->dev.sphere_mat = upload(*s, s->sphere_mat);
        s->dev.quad_mat = upload(*s, s->quad_mat);
        s->dev.materials = upload(*s, s->materials);
        s->dev.mat_kind = upload(*s, s->mat_kind);
        s->dev.slot_surface = upload(*s, s->slot_surface);
        s->dev.nodes64 = nullptr, s->dev.prims64 = nullptr;  // on demand: ensure_reference64
        std::vector<DevCounters> zero(1);
        for (rt_scene::Slot& sl : s->slots) {
            sl.d_counters = const_cast<DevCounters*>(upload(*s, zero));
            HIP_CHECK(hipEventCreate(&sl.ev0));
            HIP_CHECK(hipEventCreate(&sl.ev1));
            HIP_CHECK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        }
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device));
        s->n_cus = (uint32_t)prop.multiProcessorCount;
        // set-up of rt_render_device_noalloc: the static scratch's address on this device (this device is current)
        // and the event that orders that entry point's calls on it.  A failed lookup (a device index beyond the
        // table, say) only disables that entry point — it reports the missing scratch itself.
        if (persistent_static_scratch(device, &s->static_acc, &s->static_counters, &s->noalloc_done) != hipSuccess) {
            (void)hipGetLastError();
            s->static_acc = nullptr, s->static_counters = nullptr, s->noalloc_done = nullptr;
        }
        *out = s.release();
        return 0;
    } catch (const std::exception& e) {
        return set_error(std::string("rt_scene_create: ") + e.what());
    }
}

void rt_scene_destroy(rt_scene* scene) { delete scene; }

int rt_scene_get_info(const rt_scene* scene, rt_scene_info* info) {
    if (!scene || !info) return set_error("rt_scene_get_info: null argument");
    *info = scene->info;
    return 0;
}

int rt_scene_leaf_order(const rt_scene* scene, int32_t* out_surface_ids) {
    if (!scene || !out_surface_ids) return set_error("rt_scene_leaf_order: null argument");
    std::memcpy(out_surface_ids, scene->bvh.order.data(), scene->bvh.order.size() * sizeof(int32_t));
    return 0;
}

// Which kernel organisation a call runs (the other organisations defer to the persistent kernel for textures,
// moving spheres and media; the pooled one only pays when rays traverse a tree, and packs depth and pixel into
// one word).
enum class Organisation { Persistent, Pool, Wavefront };
static Organisation organisation(const rt_scene& scene, const DevScene& dev, const DevRenderParams& rp, int32_t kernel) {
    if (kernel == RT_KERNEL_WAVEFRONT) return dev.ext ? Organisation::Persistent : Organisation::Wavefront;
    if (kernel == RT_KERNEL_POOL)
        return dev.has_tree && !dev.ext && rp.max_depth <= 63u && pool_kernel_supports(rp.width, rp.height, scene.stack_depth)
                   ? Organisation::Pool
                   : Organisation::Persistent;
    if (kernel == RT_KERNEL_PERSISTENT) return Organisation::Persistent;
    throw std::runtime_error("unknown kernel selector " + std::to_string(kernel));
}

static void print_stage_debug(const DevCounters& c) {
    for (int k = 0; k < 4; ++k)
        fprintf(stderr, "[stages] %d: trips %llu lanes/trip %.1f cycles/trip %.0f total Gcycles %.2f\n", k, c.stage_trips[k],
                c.stage_trips[k] ? (double)c.stage_lanes[k] / (double)c.stage_trips[k] : 0.0,
                c.stage_trips[k] ? (double)c.stage_cycles[k] / (double)c.stage_trips[k] : 0.0,
                (double)c.stage_cycles[k] * 1e-9);
    for (int pass = 0; pass < 2; ++pass) {
        fprintf(stderr, pass ? "[waves finished, per 0.5 ms]" : "[waves out of work, per 0.5 ms]");
        int last = 63;
        while (last > 0 && (pass ? c.finish_hist : c.exhaust_hist)[last] == 0) --last;
        for (int k = 0; k <= last; ++k) fprintf(stderr, " %u", (pass ? c.finish_hist : c.exhaust_hist)[k]);
        fprintf(stderr, "\n");
    }
    fprintf(stderr, "[drain time, per 0.25 ms]");
    int last = 63;
    while (last > 0 && c.drain_hist[last] == 0) --last;
    for (int k = 0; k <= last; ++k) fprintf(stderr, " %u", c.drain_hist[k]);
    fprintf(stderr, "\n%u waves drained for more than 2 ms\n", c.n_late);
    for (unsigned k = 0; k < std::min(c.n_late, 32u); ++k)
        fprintf(stderr, "  out of work %.2f ms, done %.2f ms; node/leaf/shade trips %llu %llu %llu lanes %llu %llu %llu\n",
                c.late[k][0] * 1e-5, c.late[k][1] * 1e-5, c.late[k][2], c.late[k][3], c.late[k][4], c.late[k][5],
                c.late[k][6], c.late[k][7]);
    if (c.stage_trips[2])
        fprintf(stderr, "[shade trip] bounce+accumulate %.0f, regenerate %.0f, loose+bounds %.0f cycles/trip\n",
                (double)c.stage_cycles[4] / (double)c.stage_trips[2], (double)c.stage_cycles[5] / (double)c.stage_trips[2],
                (double)c.stage_cycles[6] / (double)c.stage_trips[2]);
}

int rt_render_device(rt_scene* scene, const rt_render_params* params, float* d_rgb_sum, void* stream_v,
                     rt_stats* stats) {
    if (!scene || !params || !d_rgb_sum) return set_error("rt_render_device: null argument");
    try {
        DevRenderParams rp = make_params(*scene, *params);
        DeviceGuard guard(scene->device);
        hipStream_t stream = (hipStream_t)stream_v;
        DevScene dev;
        camera_for_size(*scene, rp.width, rp.height, dev);
        const Organisation org = organisation(*scene, dev, rp, params->kernel);
        // this call's scratch: the slots in turn (the wavefront form's queues exist once: always slot 0)
        rt_scene::Slot& sl = scene->slots[org == Organisation::Wavefront ? 0 : scene->next_slot];
        if (org != Organisation::Wavefront) scene->next_slot = (scene->next_slot + 1) % rt_scene::kSlots;
        // the slot's previous use, if it was enqueued on another stream, must have ended before this one starts
        if (sl.used && sl.last_stream != stream) HIP_CHECK(hipStreamWaitEvent(stream, sl.done, 0));
        const bool count = stats && params->collect_stats;
        if (count) HIP_CHECK(hipMemsetAsync(sl.d_counters, 0, sizeof(DevCounters), stream));
        if (stats) HIP_CHECK(hipEventRecord(sl.ev0, stream));
        uint32_t n_launches = 1;
        if (org != Organisation::Wavefront) {
            // the tile-planar accumulator is kept with the slot and grown on demand
            const size_t need = persistent_acc_bytes(rp.width, rp.height);
            if (sl.acc_bytes < need) {
                if (sl.d_acc) {
                    if (sl.used) HIP_CHECK(hipEventSynchronize(sl.done));  // whatever stream used it last
                    HIP_CHECK(hipFree(sl.d_acc));
                    sl.d_acc = nullptr, sl.acc_bytes = 0;
                }
                HIP_CHECK(hipMalloc((void**)&sl.d_acc, need));
                sl.acc_bytes = need;
            }
            if (org == Organisation::Pool) {
                HIP_CHECK(launch_render_pool(dev, rp, scene->stack_depth, scene->n_cus, sl.d_acc, d_rgb_sum, sl.d_counters,
                                             &sl.d_counters->work, count, stream));
            } else {
                const size_t overflow = persistent_overflow_bytes(dev, scene->stack_depth, scene->n_cus);
                if (sl.overflow_bytes < overflow) {  // once per slot: the size depends on the scene alone
                    HIP_CHECK(hipMalloc((void**)&sl.d_overflow, overflow));
                    sl.overflow_bytes = overflow;
                    scene->info.device_bytes += overflow;
                }
                HIP_CHECK(launch_render_persistent(dev, rp, scene->stack_depth, scene->n_cus, sl.d_acc, d_rgb_sum, sl.d_counters,
                                                   &sl.d_counters->work, count, stream, sl.d_overflow));
            }
        } else {
            if (!scene->wf) scene->wf = std::make_unique<WavefrontState>();
            HIP_CHECK(render_wavefront(*scene->wf, dev, rp, scene->stack_depth, d_rgb_sum, sl.d_counters, count, stream,
                                       &n_launches));
        }
        if (stats) HIP_CHECK(hipEventRecord(sl.ev1, stream));
        HIP_CHECK(hipEventRecord(sl.done, stream));
        sl.used = true, sl.last_stream = stream;
        if (stats) {
            HIP_CHECK(hipStreamSynchronize(stream));
            std::memset(stats, 0, sizeof *stats);
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, sl.ev0, sl.ev1));
            stats->kernel_ms = ms;
            stats->n_launches = n_launches;
            const uint64_t mine =
                rp.stratum_begin >= rp.n_strata ? 0 : (rp.n_strata - rp.stratum_begin + rp.stratum_step - 1) / rp.stratum_step;
            stats->samples = (uint64_t)rp.width * rp.height * mine;
            if (count) {
                DevCounters c;
                HIP_CHECK(hipMemcpy(&c, sl.d_counters, sizeof c, hipMemcpyDeviceToHost));
                stats->segments = c.segments, stats->node_visits = c.node_visits, stats->tri_tests = c.tri_tests;
                stats->sphere_tests = c.sphere_tests, stats->quad_tests = c.quad_tests;
                if (tuning_env("RT_STAGE_DEBUG")) print_stage_debug(c);
            }
        }
        return 0;
    } catch (const std::exception& e) {
        return set_error(std::string("rt_render_device: ") + e.what());
    }
}

int rt_render_device_noalloc(rt_scene* scene, const rt_render_params* params, float* d_rgb_sum, void* stream_v) {
    if (!scene || !params || !d_rgb_sum) return set_error("rt_render_device_noalloc: null argument");
    try {
        DevRenderParams rp = make_params(*scene, *params);
        if (params->kernel != RT_KERNEL_PERSISTENT)
            throw std::runtime_error("only RT_KERNEL_PERSISTENT renders without allocating");
        if (params->collect_stats) throw std::runtime_error("collect_stats needs rt_render_device");
        const uint64_t tiles = (uint64_t)((rp.width + 7u) >> 3) * ((rp.height + 7u) >> 3);
        if (tiles > kNoAllocMaxTiles)
            throw std::runtime_error("frame of " + std::to_string(rp.width) + "x" + std::to_string(rp.height) + " = " +
                                     std::to_string(tiles) + " 8x8 tiles exceeds the static scratch (" +
                                     std::to_string(kNoAllocMaxTiles) + " tiles, e.g. 1920x1080): use rt_render_device");
        if (!scene->static_acc) throw std::runtime_error("the static scratch was not set up for this scene's device");
        DeviceGuard guard(scene->device);
        DevScene dev;
        camera_for_size(*scene, rp.width, rp.height, dev);
        // launch_render_persistent clears the accumulator it needs and the chunk cursor on `stream` before the
        // kernel: the scratch keeps the previous call's contents, nothing else of it is read
        // Every call on this GPU shares the one scratch array: a call first makes its stream wait for the previous
        // call's end (an event made at scene creation, one per GPU, so nothing is allocated here) and records its own
        // end on it.  Calls from ONE host thread — across streams and scenes — therefore serialise instead of racing;
        // calls from several host threads must still be ordered by the caller.
        // (Not on a stream under graph capture, where an event from outside the capture may not be waited on: there
        // the graph's own order is the caller's.)
        hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream_v, &capture) != hipSuccess) (void)hipGetLastError(), capture = hipStreamCaptureStatusNone;
        const bool ordered = capture == hipStreamCaptureStatusNone;
        if (ordered) HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream_v, *scene->noalloc_done, 0));
        HIP_CHECK(launch_render_persistent(dev, rp, scene->stack_depth, scene->n_cus, scene->static_acc, d_rgb_sum,
                                           scene->static_counters, &scene->static_counters->work, false,
                                           (hipStream_t)stream_v, nullptr));
        if (ordered) HIP_CHECK(hipEventRecord(*scene->noalloc_done, (hipStream_t)stream_v));
        return 0;
    } catch (const std::exception& e) {
        return set_error(std::string("rt_render_device_noalloc: ") + e.what());
    }
}

int rt_render(rt_scene* scene, const rt_render_params* params, float* rgb_sum, rt_stats* stats) {
    if (!scene || !params || !rgb_sum) return set_error("rt_render: null argument");
    float* d = nullptr;
    try {
        if (scene->device < 0) throw std::runtime_error("scene was created host-only (no device): cannot render");
        DeviceGuard guard(scene->device);
        const size_t bytes = (size_t)3 * params->width * params->height * sizeof(float);
        HIP_CHECK(hipMalloc((void**)&d, bytes));
        rt_stats local;
        int rc = rt_render_device(scene, params, d, nullptr, stats ? stats : &local);
        if (rc == 0) HIP_CHECK(hipMemcpy(rgb_sum, d, bytes, hipMemcpyDeviceToHost));
        (void)hipFree(d);
        return rc;
    } catch (const std::exception& e) {
        if (d) (void)hipFree(d);
        return set_error(std::string("rt_render: ") + e.what());
    }
}

int rt_primary_hits(rt_scene* scene, uint32_t width, uint32_t height, int precision, int32_t* prim_id, double* t) {
    if (!scene || !prim_id || !t) return set_error("rt_primary_hits: null argument");
    int32_t* d_id = nullptr;
    double* d_t = nullptr;
    try {
        if (scene->device < 0) throw std::runtime_error("scene was created host-only (no device)");
        if (width == 0 || height == 0 || (uint64_t)width * height >= (1ull << 31))
            throw std::runtime_error("bad image size");
        if (precision != 32 && precision != 64) throw std::runtime_error("precision must be 32 or 64");
        // the f64 kernel's depth-first walk keeps at most one entry per level plus one on its private stack
        if (precision == 64 && std::max(scene->info.max_depth, scene->bvh.flat_depth) + 1 > kPrimary64Stack)
            throw std::runtime_error("reference-topology BVH of depth " + std::to_string(scene->info.max_depth) +
                                     " exceeds the parity kernel's stack (" + std::to_string(kPrimary64Stack) + " entries)");
        DeviceGuard guard(scene->device);
        if (precision == 64) ensure_reference64(*scene);
        const size_t n = (size_t)width * height;
        HIP_CHECK(hipMalloc((void**)&d_id, n * sizeof(int32_t)));
        HIP_CHECK(hipMalloc((void**)&d_t, n * sizeof(double)));
        DevScene dev;
        camera_for_size(*scene, width, height, dev);
        if (precision == 64)
            HIP_CHECK(launch_primary64(dev, width, height, d_id, d_t, nullptr));
        else
            HIP_CHECK(launch_primary32(dev, width, height, scene->stack_depth, d_id, d_t, nullptr));
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(prim_id, d_id, n * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(t, d_t, n * sizeof(double), hipMemcpyDeviceToHost));
        (void)hipFree(d_id), (void)hipFree(d_t);
        return 0;
    } catch (const std::exception& e) {
        if (d_id) (void)hipFree(d_id);
        if (d_t) (void)hipFree(d_t);
        return set_error(std::string("rt_primary_hits: ") + e.what());
    }
}

int rt_tonemap_rgb8_device(const float* d_rgb_sum, uint32_t width, uint32_t height, uint64_t n_samples,
                           uint8_t* d_out, void* stream) {
    if (!d_rgb_sum || !d_out) return set_error("rt_tonemap_rgb8_device: null argument");
    if (n_samples == 0) return set_error("rt_tonemap_rgb8_device: n_samples must be positive");
    try {
        HIP_CHECK(launch_tonemap(d_rgb_sum, 3u * width * height, (double)n_samples, d_out, (hipStream_t)stream));
        return 0;
    } catch (const std::exception& e) {
        return set_error(std::string("rt_tonemap_rgb8_device: ") + e.what());
    }
}

int rt_tonemap_rgb8(const float* rgb_sum, uint32_t width, uint32_t height, uint64_t n_samples, uint8_t* out) {
    if (!rgb_sum || !out) return set_error("rt_tonemap_rgb8: null argument");
    float* d_in = nullptr;
    uint8_t* d_out = nullptr;
    try {
        const size_t n = (size_t)3 * width * height;
        HIP_CHECK(hipMalloc((void**)&d_in, n * sizeof(float)));
        HIP_CHECK(hipMalloc((void**)&d_out, n));
        HIP_CHECK(hipMemcpy(d_in, rgb_sum, n * sizeof(float), hipMemcpyHostToDevice));
        int rc = rt_tonemap_rgb8_device(d_in, width, height, n_samples, d_out, nullptr);
        if (rc == 0) {
            HIP_CHECK(hipDeviceSynchronize());
            HIP_CHECK(hipMemcpy(out, d_out, n, hipMemcpyDeviceToHost));
        }
        (void)hipFree(d_in), (void)hipFree(d_out);
        return rc;
    } catch (const std::exception& e) {
        if (d_in) (void)hipFree(d_in);
        if (d_out) (void)hipFree(d_out);
        return set_error(std::string("rt_tonemap_rgb8: ") + e.what());
    }
}

int rt_write_png(const char* path, const uint8_t* rgb, uint32_t width, uint32_t height) {
    if (!path || !rgb) return set_error("rt_write_png: null argument");
    try {
        write_png_rgb8(path, rgb, width, height);
        return 0;
    } catch (const std::exception& e) {
        return set_error(e.what());
    }
}

}  // extern "C"
