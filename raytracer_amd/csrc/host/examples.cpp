// Procedural example scenes: Example::scene (scene/example.rs:69-79) for the examples whose surfaces and
// materials the render path covers — `one-weekend` (example.rs:84-142, the CLI's default scene, main.rs:102-106),
// `random-spheres`, `two-spheres`, `two-perlin`, `cornell`, `cornell-smoke`, `earth` and `final-scene`.
//
// The reference draws one-weekend from rand::thread_rng(), so no two of its runs show the same scene; here
// the draws come from a splitmix64 stream seeded by the caller, in the reference's order of draws.
// Its instancing wrappers (Rotated about the bounding box's centroid, then Translated, surfaces.rs:461-534)
// are baked into the quads, as the YAML loader does for `box` (load.rs:90-156).
#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <utility>
#include <vector>

#include "scene_host.h"

namespace rt {

namespace {

struct Draws {
    uint64_t s;
    double next() {  // 53-bit uniform in [0, 1), like rand's Standard for f64
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return (double)(z >> 11) * 0x1p-53;
    }
    double range(double lo, double hi) { return lo + (hi - lo) * next(); }
};

int add_material(HostScene& hs, int kind, V3 albedo, double param) {
    rt_material m{};
    m.kind = kind;
    m.albedo[0] = albedo.x, m.albedo[1] = albedo.y, m.albedo[2] = albedo.z;
    m.param = param;
    hs.materials.push_back(m);
    return (int)hs.materials.size() - 1;
}

// ---- nesting (rt_bvh_group): which Bvh of the reference a surface is added to.  `g_current` is the group the
// next surfaces join (-1: the SceneBuilder's own list); every add_* below notes it.
thread_local int g_current = -1;
const double kNoBox[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};

void note_surface(HostScene& hs, const double* build_box = nullptr) {
    hs.surface_group.push_back(g_current);
    const double* b = build_box ? build_box : kNoBox;
    hs.surface_build_box.insert(hs.surface_build_box.end(), b, b + 6);
}
// Opens a group inside the current one; returns its index.  `box`: what the parent's build sees (null: the union
// of the members' boxes, Bvh::bounding_box of an unwrapped tree).
int open_group(HostScene& hs, bool leaf, const double* box = nullptr) {
    rt_bvh_group g{};
    g.parent = g_current, g.leaf = leaf ? 1 : 0;
    for (int a = 0; a < 6; ++a) g.box[a] = box ? box[a] : kNoBox[a];
    hs.groups.push_back(g);
    return g_current = (int)hs.groups.size() - 1;
}
void close_group(HostScene& hs) { g_current = hs.groups[(size_t)g_current].parent; }

// Aabb::rotate about +y by `degrees` (trace.rs:191-211: the eight corners turned about the box's midpoint), then
// Aabb::translate (trace.rs:185-189): the box of Translated::new(Rotated::new(x, jhat, angle), offset).
void wrapped_box(V3 lo, V3 hi, double degrees, V3 offset, double* out6) {
    const V3 delta = hi - lo, mid = lo + 0.5 * delta;
    const Quat rot = Quat::rotation(V3{0, 1, 0}, degrees * 0.017453292519943295);
    V3 mn{INFINITY, INFINITY, INFINITY}, mx{-INFINITY, -INFINITY, -INFINITY};
    for (int ii = 0; ii < 8; ++ii) {
        const V3 corner = lo + V3{(double)(ii & 1) * delta.x, (double)((ii >> 1) & 1) * delta.y, (double)((ii >> 2) & 1) * delta.z};
        const V3 r = rotate_point(rot, mid, corner);
        mn = vmin(mn, r), mx = vmax(mx, r);
    }
    mn = mn + offset, mx = mx + offset;
    out6[0] = mn.x, out6[1] = mn.y, out6[2] = mn.z, out6[3] = mx.x, out6[4] = mx.y, out6[5] = mx.z;
}

void add_sphere(HostScene& hs, V3 c, double r, int material) {
    rt_sphere s{};
    s.center[0] = s.center1[0] = c.x, s.center[1] = s.center1[1] = c.y, s.center[2] = s.center1[2] = c.z;
    s.radius = r;
    s.material = material;
    hs.surfaces.push_back(rt_surface_ref{RT_SURF_SPHERE, (int32_t)hs.spheres.size()});
    hs.spheres.push_back(s);
    note_surface(hs);
}

void add_quad(HostScene& hs, V3 q, V3 u, V3 v, int material) {
    hs.surfaces.push_back(rt_surface_ref{RT_SURF_QUAD, (int32_t)hs.quads.size()});
    hs.quads.push_back(make_quad(q, u, v, material));
    note_surface(hs);
}

// make_box (surfaces.rs:542-590), Rotated::new about the box's centroid (surfaces.rs:507-534) by
// `degrees` around +y, then Translated (surfaces.rs:468-491).  `as_surfaces` == false adds the quads as the
// boundary of a medium only (not hittable themselves).
void add_box(HostScene& hs, V3 a, V3 b, int material, double degrees, V3 offset, bool as_surfaces = true) {
    const V3 lo = vmin(a, b), hi = vmax(a, b);
    const V3 dx{hi.x - lo.x, 0, 0}, dy{0, hi.y - lo.y, 0}, dz{0, 0, hi.z - lo.z};
    // Rotated::new turns about Aabb::centroid (min + 0.5 * (max - min), trace.rs:169-174) of the box's bounding
    // box: the merge of its quads' boxes, each padded by 0.0001 where it is flat (Quad::bounding_box,
    // surfaces.rs:356-359; Aabb::pad, trace.rs:116-129) — the box's own extent widened by 0.00005 on every side
    const V3 pad{0.00005, 0.00005, 0.00005};
    const V3 blo = lo - pad, bhi = hi + pad;
    const V3 centre = blo + 0.5 * (bhi - blo);
    const Quat rot = Quat::rotation(V3{0, 1, 0}, degrees * 0.017453292519943295);
    // As surfaces the six quads are one Bvh::new_leaf (make_box) inside whatever the box is added to; where the
    // example wraps it (cornell: Rotated, then Translated) the parent's build sees the wrapped box.
    if (as_surfaces) {
        double box[6];
        const bool wrapped = degrees != 0.0 || offset.x != 0.0 || offset.y != 0.0 || offset.z != 0.0;
        if (wrapped) wrapped_box(blo, bhi, degrees, offset, box);
        open_group(hs, true, wrapped ? box : nullptr);
    }
    auto put = [&](V3 q, V3 u, V3 v) {
        const V3 p0 = rotate_point(rot, centre, q) + offset;
        const V3 p1 = rotate_point(rot, centre, q + u) + offset;
        const V3 p2 = rotate_point(rot, centre, q + v) + offset;
        if (as_surfaces) add_quad(hs, p0, p1 - p0, p2 - p0, material);
        else hs.quads.push_back(make_quad(p0, p1 - p0, p2 - p0, material));
    };
    put(V3{lo.x, lo.y, hi.z}, dx, dy);              // front
    put(V3{hi.x, lo.y, hi.z}, -1.0 * dz, dy);       // right
    put(V3{hi.x, lo.y, lo.z}, -1.0 * dx, dy);       // back
    put(V3{lo.x, lo.y, lo.z}, dz, dy);              // left
    put(V3{lo.x, hi.y, hi.z}, dx, -1.0 * dz);       // top
    put(V3{lo.x, lo.y, lo.z}, dx, dz);              // bottom
    if (as_surfaces) close_group(hs);
}

// ConstantMedium::new(boundary, d, albedo), surfaces.rs:601-608, over boundary primitives added just before
void add_medium(HostScene& hs, int boundary, int first, int count, double density, V3 albedo) {
    rt_medium m{};
    m.boundary = boundary, m.first = first, m.count = count;
    m.material = add_material(hs, RT_MAT_ISOTROPIC, albedo, 0);
    m.density = density;
    hs.surfaces.push_back(rt_surface_ref{RT_SURF_MEDIUM, (int32_t)hs.media.size()});
    hs.media.push_back(m);
    note_surface(hs);
}

void one_weekend(double aspect_ratio, uint64_t seed, HostScene& hs) {
    hs.desc.camera = build_camera(20.0, aspect_ratio, V3{13, 2, 3}, V3{0, 0, 0}, V3{0, 1, 0}, 0.1, true, 10.0);
    hs.desc.background[0] = 0.7, hs.desc.background[1] = 0.8, hs.desc.background[2] = 1.0;
    Draws rng{seed};
    add_sphere(hs, V3{0, -1000, 0}, 1000.0, add_material(hs, RT_MAT_LAMBERTIAN, V3{0.5, 0.5, 0.5}, 0));
    open_group(hs, false);  // `surfaces = Bvh::builder()` (example.rs:96), added to the scene at :139
    for (int a = -11; a < 11; ++a)
        for (int b = -11; b < 11; ++b) {
            const double choose = rng.next();
            const double cx = a + 0.9 * rng.next();
            const double cz = b + 0.9 * rng.next();
            const V3 centre{cx, 0.2, cz};
            if (!(length(centre - V3{4.0, 0.2, 0.0}) > 0.9)) continue;
            int m;
            if (choose < 0.8) {  // diffuse: Vec3 * Vec3 pointwise
                const V3 p{rng.next(), rng.next(), rng.next()}, q{rng.next(), rng.next(), rng.next()};
                m = add_material(hs, RT_MAT_LAMBERTIAN, V3{p.x * q.x, p.y * q.y, p.z * q.z}, 0);
            } else if (choose < 0.95) {  // metal
                const V3 albedo{rng.range(0.5, 1.0), rng.range(0.5, 1.0), rng.range(0.5, 1.0)};
                m = add_material(hs, RT_MAT_METAL, albedo, rng.range(0.0, 0.5));
            } else {  // glass
                m = add_material(hs, RT_MAT_DIELECTRIC, V3{1, 1, 1}, 1.5);
            }
            add_sphere(hs, centre, 0.2, m);
        }
    add_sphere(hs, V3{-4, 1, 0}, 1.0, add_material(hs, RT_MAT_LAMBERTIAN, V3{0.05, 0.2, 0.6}, 0));
    add_sphere(hs, V3{0, 1, 0}, 1.0, add_material(hs, RT_MAT_DIELECTRIC, V3{1, 1, 1}, 1.5));
    add_sphere(hs, V3{4, 1, 0}, 1.0, add_material(hs, RT_MAT_METAL, V3{0.7, 0.6, 0.5}, 0.0));
    close_group(hs);
}

int add_textured(HostScene& hs, int texture, V3 even, V3 odd, double scale, int index) {
    const int m = add_material(hs, RT_MAT_LAMBERTIAN, even, 0);
    rt_material& mm = hs.materials[(size_t)m];
    mm.texture = texture;
    mm.albedo2[0] = odd.x, mm.albedo2[1] = odd.y, mm.albedo2[2] = odd.z;
    mm.tex_scale = scale;
    mm.tex_index = index;
    return m;
}

// PerlinNoise::new, material.rs:289-309: 256 unit vectors from U[-1,1)^3, three shuffles of 0..255
int add_perlin(HostScene& hs, Draws& rng) {
    rt_perlin p{};
    for (int i = 0; i < 256; ++i) {
        const V3 v = unit(V3{rng.range(-1.0, 1.0), rng.range(-1.0, 1.0), rng.range(-1.0, 1.0)});
        p.ranvec[i][0] = v.x, p.ranvec[i][1] = v.y, p.ranvec[i][2] = v.z;
    }
    for (int a = 0; a < 3; ++a) {
        for (int i = 0; i < 256; ++i) p.perm[a][i] = i;
        for (int i = 255; i > 0; --i) {  // Fisher-Yates, as SliceRandom::shuffle does
            const int j = (int)(rng.next() * (double)(i + 1));
            std::swap(p.perm[a][i], p.perm[a][j]);
        }
    }
    hs.perlin.push_back(p);
    return (int)hs.perlin.size() - 1;
}

// example.rs:145-206: one-weekend with a checkered ground and bouncing (moving) diffuse spheres
void random_spheres(double aspect_ratio, uint64_t seed, HostScene& hs) {
    hs.desc.camera = build_camera(20.0, aspect_ratio, V3{13, 2, 3}, V3{0, 0, 0}, V3{0, 1, 0}, 0.1, true, 10.0);
    hs.desc.background[0] = 0.7, hs.desc.background[1] = 0.8, hs.desc.background[2] = 1.0;
    Draws rng{seed};
    add_sphere(hs, V3{0, -1000, 0}, 1000.0, add_textured(hs, RT_TEX_CHECKER, V3{0.2, 0.3, 0.1}, V3{0.9, 0.9, 0.9}, 0.32, 0));
    for (int a = -11; a < 11; ++a)
        for (int b = -11; b < 11; ++b) {
            const double choose = rng.next();
            const double cx = a + 0.9 * rng.next();
            const double cz = b + 0.9 * rng.next();
            const V3 centre{cx, 0.2, cz};
            if (!(length(centre - V3{4.0, 0.2, 0.0}) > 0.9)) continue;
            if (choose < 0.8) {
                const V3 p{rng.next(), rng.next(), rng.next()}, q{rng.next(), rng.next(), rng.next()};
                const int m = add_material(hs, RT_MAT_LAMBERTIAN, V3{p.x * q.x, p.y * q.y, p.z * q.z}, 0);
                const V3 centre2 = centre + V3{0, rng.range(0.0, 0.5), 0};
                add_sphere(hs, centre, 0.2, m);
                rt_sphere& sp = hs.spheres.back();  // Sphere::moving(center, center2, ...)
                sp.moving = 1;
                sp.center1[0] = centre2.x, sp.center1[1] = centre2.y, sp.center1[2] = centre2.z;
            } else if (choose < 0.95) {
                const V3 albedo{rng.range(0.5, 1.0), rng.range(0.5, 1.0), rng.range(0.5, 1.0)};
                add_sphere(hs, centre, 0.2, add_material(hs, RT_MAT_METAL, albedo, rng.range(0.0, 0.5)));
            } else {
                add_sphere(hs, centre, 0.2, add_material(hs, RT_MAT_DIELECTRIC, V3{1, 1, 1}, 1.5));
            }
        }
    add_sphere(hs, V3{-4, 1, 0}, 1.0, add_material(hs, RT_MAT_LAMBERTIAN, V3{0.05, 0.2, 0.6}, 0));
    add_sphere(hs, V3{0, 1, 0}, 1.0, add_material(hs, RT_MAT_DIELECTRIC, V3{1, 1, 1}, 1.5));
    add_sphere(hs, V3{4, 1, 0}, 1.0, add_material(hs, RT_MAT_METAL, V3{0.7, 0.6, 0.5}, 0.0));
}

void two_spheres(double aspect_ratio, HostScene& hs) {  // example.rs:208-234
    hs.desc.camera = build_camera(20.0, aspect_ratio, V3{13, 2, 3}, V3{0, 0, 0}, V3{0, 1, 0}, 0.0, false, 0.0);
    hs.desc.background[0] = 0.7, hs.desc.background[1] = 0.8, hs.desc.background[2] = 1.0;
    const int checker = add_textured(hs, RT_TEX_UV_CHECKER, V3{0.2, 0.3, 0.1}, V3{0.9, 0.9, 0.9}, 30.0, 0);
    add_sphere(hs, V3{0, -10, 0}, 10.0, checker);
    add_sphere(hs, V3{0, 10, 0}, 10.0, checker);
}

void two_perlin_spheres(double aspect_ratio, uint64_t seed, HostScene& hs) {  // example.rs:236-260
    hs.desc.camera = build_camera(20.0, aspect_ratio, V3{13, 2, 3}, V3{0, 0, 0}, V3{0, 1, 0}, 0.0, false, 0.0);
    hs.desc.background[0] = 0.7, hs.desc.background[1] = 0.8, hs.desc.background[2] = 1.0;
    Draws rng{seed};
    const int noise = add_perlin(hs, rng);
    const int perlin = add_textured(hs, RT_TEX_MARBLE, V3{1, 1, 1}, V3{0, 0, 0}, 4.0, noise);
    add_sphere(hs, V3{0, -1000, 0}, 1000.0, perlin);
    add_sphere(hs, V3{0, 2, 0}, 2.0, perlin);
}

// example.rs:405-428.  The reference reads ./scenes/res/earth.png, which its repository does not ship: the
// load fails, earth_texture() falls back to an empty image (example.rs:430-438) and Image::color_at shades
// (0, 1, 1) (material.rs:390-392).  That is what is reproduced: an empty image.  The globe's two Rotated
// wrappers only turn the texture, so they have no effect on it.
void earth(double aspect_ratio, HostScene& hs) {
    hs.desc.camera = build_camera(20.0, aspect_ratio, V3{0, 0, 12}, V3{0, 0, 0}, V3{0, 1, 0}, 0.0, false, 0.0);
    hs.desc.background[0] = 0.7, hs.desc.background[1] = 0.8, hs.desc.background[2] = 1.0;
    hs.images.push_back(rt_image{nullptr, 0, 0});
    hs.image_data.emplace_back();
    add_sphere(hs, V3{0, 0, 0}, 2.0, add_textured(hs, RT_TEX_IMAGE, V3{0, 1, 1}, V3{0, 0, 0}, 1.0, 0));
}

void cornell(double aspect_ratio, HostScene& hs) {
    // Camera::builder defaults: no aperture, focus distance = |from - towards| (camera.rs:43-74)
    hs.desc.camera = build_camera(40.0, aspect_ratio, V3{278, 278, -800}, V3{278, 278, 0}, V3{0, 1, 0}, 0.0, false, 0.0);
    hs.desc.background[0] = hs.desc.background[1] = hs.desc.background[2] = 0.0;  // SceneBuilder's default
    const int red = add_material(hs, RT_MAT_LAMBERTIAN, V3{0.65, 0.05, 0.05}, 0);
    const int white = add_material(hs, RT_MAT_LAMBERTIAN, V3{0.73, 0.73, 0.73}, 0);
    const int green = add_material(hs, RT_MAT_LAMBERTIAN, V3{0.12, 0.45, 0.15}, 0);
    const int light = add_material(hs, RT_MAT_DIFFUSE_LIGHT, V3{15, 15, 15}, 0);
    add_quad(hs, V3{555, 0, 0}, V3{0, 555, 0}, V3{0, 0, 555}, green);
    add_quad(hs, V3{0, 0, 0}, V3{0, 555, 0}, V3{0, 0, 555}, red);
    add_quad(hs, V3{343, 554, 332}, V3{-130, 0, 0}, V3{0, 0, -105}, light);
    add_quad(hs, V3{0, 555, 0}, V3{555, 0, 0}, V3{0, 0, 555}, white);
    add_quad(hs, V3{0, 0, 0}, V3{555, 0, 0}, V3{0, 0, 555}, white);
    add_quad(hs, V3{0, 0, 555}, V3{555, 0, 0}, V3{0, 555, 0}, white);
    add_box(hs, V3{0, 0.001, 0}, V3{165, 333, 165}, white, 15.0, V3{265, 0, 295});
    add_box(hs, V3{0, 0.001, 0}, V3{165, 165, 165}, white, -18.0, V3{130, 0, 65});
}

// example.rs:331-403: the Cornell box with its two boxes turned into smoke (black) and fog (white)
void cornell_smoke(double aspect_ratio, HostScene& hs) {
    hs.desc.camera = build_camera(40.0, aspect_ratio, V3{278, 278, -800}, V3{278, 278, 0}, V3{0, 1, 0}, 0.0, false, 0.0);
    hs.desc.background[0] = hs.desc.background[1] = hs.desc.background[2] = 0.0;
    const int red = add_material(hs, RT_MAT_LAMBERTIAN, V3{0.65, 0.05, 0.05}, 0);
    const int white = add_material(hs, RT_MAT_LAMBERTIAN, V3{0.73, 0.73, 0.73}, 0);
    const int green = add_material(hs, RT_MAT_LAMBERTIAN, V3{0.12, 0.45, 0.15}, 0);
    const int light = add_material(hs, RT_MAT_DIFFUSE_LIGHT, V3{7, 7, 7}, 0);
    add_quad(hs, V3{555, 0, 0}, V3{0, 555, 0}, V3{0, 0, 555}, green);
    add_quad(hs, V3{0, 0, 0}, V3{0, 555, 0}, V3{0, 0, 555}, red);
    add_quad(hs, V3{113, 554, 127}, V3{330, 0, 0}, V3{0, 0, 305}, light);
    add_quad(hs, V3{0, 555, 0}, V3{555, 0, 0}, V3{0, 0, 555}, white);
    add_quad(hs, V3{0, 0, 0}, V3{555, 0, 0}, V3{0, 0, 555}, white);
    add_quad(hs, V3{0, 0, 555}, V3{555, 0, 0}, V3{0, 555, 0}, white);
    int first = (int)hs.quads.size();
    add_box(hs, V3{0, 0.001, 0}, V3{165, 333, 165}, white, 15.0, V3{265, 0, 295}, false);
    add_medium(hs, RT_SURF_QUAD, first, 6, 0.01, V3{0, 0, 0});
    first = (int)hs.quads.size();
    add_box(hs, V3{0, 0.001, 0}, V3{165, 165, 165}, white, -18.0, V3{130, 0, 65}, false);
    add_medium(hs, RT_SURF_QUAD, first, 6, 0.01, V3{1, 1, 1});
}

// example.rs:466-572: every feature at once.  SceneBuilder's default background is black.
void final_scene(double aspect_ratio, uint64_t seed, HostScene& hs) {
    hs.desc.camera = build_camera(40.0, aspect_ratio, V3{478, 278, -600}, V3{278, 278, 0}, V3{0, 1, 0}, 0.0, false, 0.0);
    hs.desc.background[0] = hs.desc.background[1] = hs.desc.background[2] = 0.0;
    Draws rng{seed};
    const int ground = add_material(hs, RT_MAT_LAMBERTIAN, V3{0.48, 0.83, 0.53}, 0);
    open_group(hs, false);  // Bvh::new(boxes1), example.rs:498: a tree whose 400 members are make_box leaves
    for (int i = 0; i < 20; ++i)
        for (int j = 0; j < 20; ++j) {
            const double w = 100.0, x0 = -1000.0 + i * w, z0 = -1000.0 + j * w, y1 = rng.range(1.0, 101.0);
            add_box(hs, V3{x0, 0.0, z0}, V3{x0 + w, y1, z0 + w}, ground, 0.0, V3{0, 0, 0});
        }
    close_group(hs);
    add_quad(hs, V3{123, 554, 147}, V3{300, 0, 0}, V3{0, 0, 265}, add_material(hs, RT_MAT_DIFFUSE_LIGHT, V3{7, 7, 7}, 0));
    add_sphere(hs, V3{400, 400, 200}, 50.0, add_material(hs, RT_MAT_LAMBERTIAN, V3{0.7, 0.3, 0.1}, 0));
    hs.spheres.back().moving = 1;  // Sphere::moving(center1, center1 + (30, 0, 0), ...)
    hs.spheres.back().center1[0] = 430.0;
    add_sphere(hs, V3{260, 150, 45}, 50.0, add_material(hs, RT_MAT_DIELECTRIC, V3{1, 1, 1}, 1.5));
    add_sphere(hs, V3{0, 150, 145}, 50.0, add_material(hs, RT_MAT_METAL, V3{0.8, 0.8, 0.9}, 1.0));
    // a glass ball filled with blue smoke: the boundary is a surface AND (a copy) the medium's boundary
    const int glass = add_material(hs, RT_MAT_DIELECTRIC, V3{1, 1, 1}, 1.5);
    add_sphere(hs, V3{360, 150, 145}, 70.0, glass);
    auto boundary_sphere = [&](V3 c, double r) {
        rt_sphere sp{};
        sp.center[0] = sp.center1[0] = c.x, sp.center[1] = sp.center1[1] = c.y, sp.center[2] = sp.center1[2] = c.z;
        sp.radius = r, sp.material = glass;
        hs.spheres.push_back(sp);
        return (int)hs.spheres.size() - 1;
    };
    add_medium(hs, RT_SURF_SPHERE, boundary_sphere(V3{360, 150, 145}, 70.0), 1, 0.2, V3{0.2, 0.4, 0.9});
    add_medium(hs, RT_SURF_SPHERE, boundary_sphere(V3{0, 0, 0}, 5000.0), 1, 0.0001, V3{1, 1, 1});  // thin mist everywhere
    hs.images.push_back(rt_image{nullptr, 0, 0});  // earth_texture(): see earth() above
    hs.image_data.emplace_back();
    add_sphere(hs, V3{400, 200, 400}, 100.0, add_textured(hs, RT_TEX_IMAGE, V3{0, 1, 1}, V3{0, 0, 0}, 1.0, 0));
    const int noise = add_perlin(hs, rng);
    add_sphere(hs, V3{220, 280, 300}, 80.0, add_textured(hs, RT_TEX_MARBLE, V3{1, 1, 1}, V3{0, 0, 0}, 0.1, noise));
    // 1000 small spheres in a 165-cube, the cluster turned 15 degrees about its bounding box's centre, then moved
    const int white = add_material(hs, RT_MAT_LAMBERTIAN, V3{0.73, 0.73, 0.73}, 0);
    std::vector<V3> pts(1000);
    V3 lo{INFINITY, INFINITY, INFINITY}, hi{-INFINITY, -INFINITY, -INFINITY};
    for (V3& p : pts) {
        p = V3{rng.range(0.0, 165.0), rng.range(0.0, 165.0), rng.range(0.0, 165.0)};
        lo = vmin(lo, p - V3{10, 10, 10}), hi = vmax(hi, p + V3{10, 10, 10});
    }
    const V3 centre = lo + 0.5 * (hi - lo);  // Aabb::centroid of the cluster's box (Rotated::new, surfaces.rs:507-517)
    const Quat rot = Quat::rotation(V3{0, 1, 0}, 15.0 * 0.017453292519943295);
    // Translated::new(Rotated::new(Bvh::new(boxes2), ..)): the cluster's tree is built over the spheres where they
    // were drawn (their boxes in the cluster's own space: surface_build_box); the scene's build sees the turned box
    double cluster_box[6];
    wrapped_box(lo, hi, 15.0, V3{-100, 270, 395}, cluster_box);
    open_group(hs, false, cluster_box);
    for (const V3& p : pts) {
        add_sphere(hs, rotate_point(rot, centre, p) + V3{-100, 270, 395}, 10.0, white);
        const double own[6] = {p.x - 10, p.y - 10, p.z - 10, p.x + 10, p.y + 10, p.z + 10};
        std::copy(own, own + 6, hs.surface_build_box.end() - 6);
    }
    close_group(hs);
}

}  // namespace

void example_scene(const std::string& name, double aspect_ratio, uint64_t seed, bool use_bvh, HostScene& out) {
    g_current = -1;
    if (name == "one-weekend") one_weekend(aspect_ratio, seed, out);
    else if (name == "cornell") cornell(aspect_ratio, out);
    else if (name == "random-spheres") random_spheres(aspect_ratio, seed, out);
    else if (name == "two-spheres") two_spheres(aspect_ratio, out);
    else if (name == "two-perlin") two_perlin_spheres(aspect_ratio, seed, out);
    else if (name == "earth") earth(aspect_ratio, out);
    else if (name == "cornell-smoke") cornell_smoke(aspect_ratio, out);
    else if (name == "final-scene") final_scene(aspect_ratio, seed, out);
    else
        throw std::runtime_error("invalid value '" + name + "' for '--example <EXAMPLE>'");  // Example::from_str
    g_current = -1;
    out.desc.use_bvh = use_bvh ? 1 : 0;
    out.sync_desc();
}

}  // namespace rt
