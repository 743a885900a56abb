// Procedural example scenes: Example::scene (scene/example.rs:69-79) for the examples whose surfaces and
// materials the render path covers — `one-weekend` (example.rs:84-142, the CLI's default scene, main.rs:102-106)
// and `cornell` (example.rs:262-329).  The others need textures, moving spheres or volumes and
// are refused by name.
//
// The reference draws one-weekend from rand::thread_rng(), so no two of its runs show the same scene; here
// the draws come from a splitmix64 stream seeded by the caller, in the reference's order of draws.
// Its instancing wrappers (Rotated about the bounding box's centroid, then Translated, surfaces.rs:461-534)
// are baked into the quads, as the YAML loader does for `box` (load.rs:90-156).
#include <cmath>
#include <stdexcept>

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

void add_sphere(HostScene& hs, V3 c, double r, int material) {
    rt_sphere s{};
    s.center[0] = s.center1[0] = c.x, s.center[1] = s.center1[1] = c.y, s.center[2] = s.center1[2] = c.z;
    s.radius = r;
    s.material = material;
    hs.surfaces.push_back(rt_surface_ref{RT_SURF_SPHERE, (int32_t)hs.spheres.size()});
    hs.spheres.push_back(s);
}

void add_quad(HostScene& hs, V3 q, V3 u, V3 v, int material) {
    hs.surfaces.push_back(rt_surface_ref{RT_SURF_QUAD, (int32_t)hs.quads.size()});
    hs.quads.push_back(make_quad(q, u, v, material));
}

// make_box (surfaces.rs:542-590), Rotated::new about the box's centroid (surfaces.rs:507-534) by
// `degrees` around +y, then Translated (surfaces.rs:468-491).
void add_box(HostScene& hs, V3 a, V3 b, int material, double degrees, V3 offset) {
    const V3 lo = vmin(a, b), hi = vmax(a, b);
    const V3 dx{hi.x - lo.x, 0, 0}, dy{0, hi.y - lo.y, 0}, dz{0, 0, hi.z - lo.z};
    const V3 centre = 0.5 * (lo + hi);  // the quads' boxes are padded symmetrically: same centroid
    const Quat rot = Quat::rotation(V3{0, 1, 0}, degrees * 0.017453292519943295);
    auto put = [&](V3 q, V3 u, V3 v) {
        const V3 p0 = rotate_point(rot, centre, q) + offset;
        const V3 p1 = rotate_point(rot, centre, q + u) + offset;
        const V3 p2 = rotate_point(rot, centre, q + v) + offset;
        add_quad(hs, p0, p1 - p0, p2 - p0, material);
    };
    put(V3{lo.x, lo.y, hi.z}, dx, dy);              // front
    put(V3{hi.x, lo.y, hi.z}, -1.0 * dz, dy);       // right
    put(V3{hi.x, lo.y, lo.z}, -1.0 * dx, dy);       // back
    put(V3{lo.x, lo.y, lo.z}, dz, dy);              // left
    put(V3{lo.x, hi.y, hi.z}, dx, -1.0 * dz);       // top
    put(V3{lo.x, lo.y, lo.z}, dx, dz);              // bottom
}

void one_weekend(double aspect_ratio, uint64_t seed, HostScene& hs) {
    hs.desc.camera = build_camera(20.0, aspect_ratio, V3{13, 2, 3}, V3{0, 0, 0}, V3{0, 1, 0}, 0.1, true, 10.0);
    hs.desc.background[0] = 0.7, hs.desc.background[1] = 0.8, hs.desc.background[2] = 1.0;
    Draws rng{seed};
    add_sphere(hs, V3{0, -1000, 0}, 1000.0, add_material(hs, RT_MAT_LAMBERTIAN, V3{0.5, 0.5, 0.5}, 0));
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

}  // namespace

void example_scene(const std::string& name, double aspect_ratio, uint64_t seed, bool use_bvh, HostScene& out) {
    if (name == "one-weekend") one_weekend(aspect_ratio, seed, out);
    else if (name == "cornell") cornell(aspect_ratio, out);
    else if (name == "random-spheres" || name == "two-spheres" || name == "two-perlin" || name == "cornell-smoke" ||
             name == "earth" || name == "final-scene")
        throw std::runtime_error("example `" + name + "` needs textures, moving spheres or volumes, which this build's "
                                 "render path does not cover (supported: one-weekend, cornell)");
    else
        throw std::runtime_error("invalid value '" + name + "' for '--example <EXAMPLE>'");  // Example::from_str
    out.desc.use_bvh = use_bvh ? 1 : 0;
    out.sync_desc();
}

}  // namespace rt
