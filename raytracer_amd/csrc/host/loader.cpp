// Scene loader: the host-side replacement of scene::load_scene (load.rs:39-160) producing the flat
// rt_scene_desc the device path consumes.  Arithmetic that moves geometry (mesh re-centring, transform
// baking, vertex normals, quad caches, camera frame) keeps the reference's operation order in f64.
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "scene_host.h"
#include "worker.h"
#include "yaml_lite.h"

namespace rt {

int g_verbose = 0;  // rt_set_verbose: the CLI turns the reference's stderr chatter (load.rs:231-244) on

namespace {

using yaml::Node;

[[noreturn]] void fail(const std::string& m) { throw std::runtime_error(m); }

const Node& field(const Node& map, const char* key, const char* where) {
    if (map.kind != Node::Map) fail(std::string(where) + ": expected a mapping at line " + std::to_string(map.line));
    const Node* n = map.find(key);
    if (!n) fail(std::string(where) + ": missing field `" + key + "` at line " + std::to_string(map.line));
    return *n;
}

double num(const Node& n, const char* what) {
    if (!n.is_number()) fail(std::string(what) + ": expected a number at line " + std::to_string(n.line));
    return n.as_double();
}

V3 vec3(const Node& n, const char* what) {
    if (n.kind != Node::Seq || n.seq.size() != 3)
        fail(std::string(what) + ": expected a sequence of 3 numbers at line " + std::to_string(n.line));
    return {num(*n.seq[0], what), num(*n.seq[1], what), num(*n.seq[2], what)};
}

std::string tag(const Node& n, const char* where) {
    const Node& t = field(n, "type", where);
    if (t.kind != Node::Scalar) fail(std::string(where) + ": `type` must be a string at line " + std::to_string(t.line));
    return t.scalar;
}

// load.rs:471-522 — 6-hex-digit string (bytes / 256, not 255) or a float triple.
V3 albedo(const Node& n) {
    if (n.kind == Node::Seq) return vec3(n, "albedo");
    if (n.kind != Node::Scalar || n.is_number())
        fail("albedo: invalid type at line " + std::to_string(n.line) + ", expected string or array of floats");
    const std::string& s = n.scalar;
    if (s.size() != 6) fail("expected hex color format aabbcc, got " + s);
    char* end = nullptr;
    errno = 0;
    unsigned long parsed = std::strtoul(s.c_str(), &end, 16);
    if (*end || s[0] == '+' || s[0] == '-') fail("could not parse hex color: invalid digit found in string");
    return {(double)((parsed >> 16) & 0xff) / 256.0, (double)((parsed >> 8) & 0xff) / 256.0,
            (double)(parsed & 0xff) / 256.0};
}

// load.rs:445-466
rt_material material(const Node& n) {
    rt_material m{};
    std::string t = tag(n, "material");
    V3 a{1.0, 1.0, 1.0};
    if (t == "lambertian") {
        m.kind = RT_MAT_LAMBERTIAN;
        a = albedo(field(n, "albedo", "material"));
    } else if (t == "dielectric") {
        m.kind = RT_MAT_DIELECTRIC;  // albedo (1,1,1): material.rs:37-42
        m.param = num(field(n, "index", "material"), "index");
    } else if (t == "metal") {
        m.kind = RT_MAT_METAL;
        a = albedo(field(n, "albedo", "material"));
        m.param = num(field(n, "fuzz", "material"), "fuzz");
    } else if (t == "diffuseLight") {
        m.kind = RT_MAT_DIFFUSE_LIGHT;
        a = albedo(field(n, "albedo", "material"));
    } else {
        fail("unknown variant `" + t + "`, expected one of `lambertian`, `dielectric`, `metal`, `diffuseLight`");
    }
    m.albedo[0] = a.x, m.albedo[1] = a.y, m.albedo[2] = a.z;
    return m;
}

struct Transform {
    enum Kind { Scale, Rotate, Translate, TranslateTo } kind;
    double factor = 0, angle = 0;
    V3 v;  // axis | dir | dest
};

// load.rs:409-442
std::vector<Transform> transforms(const Node* n) {
    std::vector<Transform> out;
    if (!n || n->kind == Node::Null) return out;  // #[serde(default)]
    if (n->kind != Node::Seq) fail("transform: expected a sequence at line " + std::to_string(n->line));
    for (auto& item : n->seq) {
        Transform t;
        std::string k = tag(*item, "transform");
        if (k == "scale") {
            t.kind = Transform::Scale;
            t.factor = num(field(*item, "factor", "transform"), "factor");
        } else if (k == "rotate") {
            t.kind = Transform::Rotate;
            const Node& ax = field(*item, "axis", "transform");
            if (ax.kind == Node::Scalar && ax.scalar == "x") t.v = {1, 0, 0};
            else if (ax.kind == Node::Scalar && ax.scalar == "y") t.v = {0, 1, 0};
            else if (ax.kind == Node::Scalar && ax.scalar == "z") t.v = {0, 0, 1};
            else if (ax.kind == Node::Seq) t.v = vec3(ax, "axis");
            else fail("data did not match any variant of untagged enum RotationAxis at line " + std::to_string(ax.line));
            t.angle = num(field(*item, "angle", "transform"), "angle");
        } else if (k == "translate") {
            t.kind = Transform::Translate;
            t.v = vec3(field(*item, "dir", "transform"), "dir");
        } else if (k == "translateTo") {
            t.kind = Transform::TranslateTo;
            t.v = vec3(field(*item, "dest", "transform"), "dest");
        } else {
            fail("unknown variant `" + k + "`, expected one of `scale`, `rotate`, `translate`, `translateTo`");
        }
        out.push_back(t);
    }
    return out;
}

const double kDegToRad = 0.017453292519943295;  // f64::to_radians multiplies by PI/180

// load.rs:332-344 — bbox centre; max deliberately starts at (0,0,0) (quirk Q6).
V3 compute_mesh_center(const std::vector<V3>& pts) {
    V3 lo{INFINITY, INFINITY, INFINITY}, hi{0, 0, 0};
    for (const V3& v : pts) {
        hi = vmax(hi, v);
        lo = vmin(lo, v);
    }
    return lo + 0.5 * (hi - lo);
}

void v3out(double* dst, V3 v) { dst[0] = v.x, dst[1] = v.y, dst[2] = v.z; }

// load.rs:162-200
rt_quad transform_quad(const rt_quad& quad, const std::vector<Transform>& ts, V3 origin) {
    V3 q{quad.q[0], quad.q[1], quad.q[2]}, u{quad.u[0], quad.u[1], quad.u[2]}, v{quad.v[0], quad.v[1], quad.v[2]};
    V3 p[3] = {q, q + u, q + v};
    for (const Transform& t : ts) {
        switch (t.kind) {
            case Transform::Scale:
                for (V3& x : p) x = origin + t.factor * (x - origin);
                break;
            case Transform::Rotate: {
                Quat r = Quat::rotation(t.v, t.angle * kDegToRad);
                for (V3& x : p) x = rotate_point(r, origin, x);
                break;
            }
            case Transform::Translate:
                for (V3& x : p) x = x + t.v;
                origin = origin + t.v;
                break;
            case Transform::TranslateTo: {
                V3 dir = t.v - origin;
                for (V3& x : p) x = x + dir;
                origin = origin + dir;
                break;
            }
        }
    }
    return make_quad(p[0], p[1] - p[0], p[2] - p[0], quad.material);
}

// load.rs:202-313 + surfaces.rs:146-190
void load_mesh(const std::string& path, const std::vector<Transform>& ts, int mat, HostScene& out) {
    ObjMesh obj;
    load_obj_first_model(path, obj);
    if (g_verbose) {
        std::fprintf(stderr, "loaded mesh: %s\nvertices: %zu\nindices: %zu (%zu triangles)\nnormals: %zu\n",
                     obj.name.c_str(), obj.positions.size(), obj.indices.size(), obj.indices.size() / 3,
                     obj.normals.size());
    }
    if (obj.normals.size() % 3 != 0) fail("number of normals not divisible by 3: " + std::to_string(obj.normals.size()));
    if (obj.indices.size() % 3 != 0) fail("number of indices not divisible by 3: " + std::to_string(obj.indices.size()));

    std::vector<V3> verts(obj.positions.size() / 3);
    for (size_t i = 0; i < verts.size(); ++i)
        verts[i] = {(double)obj.positions[3 * i], (double)obj.positions[3 * i + 1], (double)obj.positions[3 * i + 2]};

    V3 mesh_origin = compute_mesh_center(verts);
    V3 origin_vec = mesh_origin;
    for (V3& v : verts) v = v - origin_vec;
    for (const Transform& t : ts) {
        switch (t.kind) {
            case Transform::Scale:
                for (V3& v : verts) v = t.factor * v;
                mesh_origin = compute_mesh_center(verts);
                break;
            case Transform::Rotate: {
                // pivots about mesh_origin, which still is the PRE-centring centre (quirk Q6)
                Quat r = Quat::rotation(t.v, t.angle * kDegToRad);
                for (V3& v : verts) v = rotate_point(r, mesh_origin, v);
                break;
            }
            case Transform::Translate:
                for (V3& v : verts) v = v + t.v;
                mesh_origin = mesh_origin + t.v;
                break;
            case Transform::TranslateTo: {
                V3 dir = t.v - mesh_origin;
                for (V3& v : verts) v = v + dir;
                mesh_origin = mesh_origin + dir;
                break;
            }
        }
    }

    std::vector<V3> normals;
    if (obj.normals.empty()) {
        // Mesh::new, surfaces.rs:146-169: area-weighted, un-normalised face normals summed per vertex.
        normals.assign(verts.size(), V3{});
        for (size_t f = 0; f + 2 < obj.indices.size(); f += 3) {
            V3 v0 = verts[obj.indices[f]], v1 = verts[obj.indices[f + 1]], v2 = verts[obj.indices[f + 2]];
            V3 e1 = v1 - v0, e2 = v0 - v2;
            V3 n = cross(e2, e1);
            for (int k = 0; k < 3; ++k) normals[obj.indices[f + k]] = normals[obj.indices[f + k]] + n;
        }
        for (V3& n : normals) n = 1.0 * unit(n);
    } else {
        normals.resize(obj.normals.size() / 3);
        for (size_t i = 0; i < normals.size(); ++i)
            normals[i] = {(double)obj.normals[3 * i], (double)obj.normals[3 * i + 1], (double)obj.normals[3 * i + 2]};
        if (normals.size() < verts.size())
            fail("mesh has fewer normals than vertices (" + std::to_string(normals.size()) + " < " +
                 std::to_string(verts.size()) + "); the reference indexes normals by vertex index (surfaces.rs:231-243)");
    }

    const size_t ntri = obj.indices.size() / 3, tri_base = out.triangles.size(), surf_base = out.surfaces.size();
    out.triangles.resize(tri_base + ntri), out.surfaces.resize(surf_base + ntri);  // (not filled: DefaultInit)
    parallel_chunks(ntri, 8192, 8, [&](size_t begin, size_t end) {
        for (size_t t = begin; t < end; ++t) {
            rt_triangle tri{};
            for (int k = 0; k < 3; ++k) {
                uint32_t idx = obj.indices[3 * t + k];
                v3out(tri.v[k], verts[idx]);
                v3out(tri.n[k], normals[idx]);
            }
            tri.material = mat;
            out.surfaces[surf_base + t] = {RT_SURF_TRIANGLE, (int32_t)(tri_base + t)};
            out.triangles[tri_base + t] = tri;
        }
    });
}

std::string dirname_of(const std::string& p) {
    size_t s = p.find_last_of('/');
    if (s == std::string::npos) return "";
    if (s == 0) return "/";
    return p.substr(0, s);
}

}  // namespace

void HostScene::sync_desc() {
    desc.materials = materials.data(), desc.n_materials = materials.size();
    desc.spheres = spheres.data(), desc.n_spheres = spheres.size();
    desc.triangles = triangles.data(), desc.n_triangles = triangles.size();
    desc.quads = quads.data(), desc.n_quads = quads.size();
    desc.surfaces = surfaces.data(), desc.n_surfaces = surfaces.size();
    desc.media = media.data(), desc.n_media = media.size();
    desc.perlin = perlin.data(), desc.n_perlin = perlin.size();
    for (size_t i = 0; i < images.size() && i < image_data.size(); ++i) images[i].rgb = image_data[i].data();
    desc.images = images.data(), desc.n_images = images.size();
    // nesting: only when a group exists, and then one group entry (and one build box, if any) per surface
    const bool nested = !groups.empty() && surface_group.size() == surfaces.size();
    desc.groups = nested ? groups.data() : nullptr, desc.n_groups = nested ? groups.size() : 0;
    desc.surface_group = nested ? surface_group.data() : nullptr;
    desc.surface_build_box = nested && surface_build_box.size() == 6 * surfaces.size() ? surface_build_box.data() : nullptr;
}

rt_quad make_quad(V3 q, V3 u, V3 v, int material) {
    rt_quad out{};
    V3 n = cross(u, v);
    V3 normal = unit(n);
    double d = dot(normal, q);
    V3 w = n / dot(n, n);
    v3out(out.q, q), v3out(out.u, u), v3out(out.v, v), v3out(out.normal, normal), v3out(out.w, w);
    out.d = d;
    out.material = material;
    return out;
}

rt_camera build_camera(double vfov_degrees, double aspect_ratio, V3 from, V3 towards, V3 vup, double aperture,
                       bool has_focus, double focus_dist_in) {
    double vfov = vfov_degrees * kDegToRad;
    double h = std::tan(vfov / 2.0);
    double viewport_height = 2.0 * h;
    double viewport_width = aspect_ratio * viewport_height;
    double focus_dist = has_focus ? focus_dist_in : length(from - towards);
    V3 w = unit(from - towards);
    V3 u = unit(cross(vup, w));
    V3 v = cross(w, u);
    V3 origin = from;
    V3 horizontal = focus_dist * viewport_width * u;  // (f*vw)*u: f64*f64 first, then Mul<Vec3> for f64
    V3 vertical = focus_dist * viewport_height * v;
    V3 lower_left = origin - horizontal / 2.0 - vertical / 2.0 - focus_dist * w;
    rt_camera c{};
    v3out(c.origin, origin), v3out(c.lower_left, lower_left), v3out(c.horizontal, horizontal);
    v3out(c.vertical, vertical), v3out(c.u, u), v3out(c.v, v), v3out(c.w, w);
    c.lens_radius = aperture / 2.0;
    return c;
}

void load_scene_file(const std::string& yaml_path, double aspect_ratio, bool use_bvh, HostScene& out) {
    std::ifstream f(yaml_path, std::ios::binary);
    if (!f) fail(std::string(std::strerror(errno)) + " (os error " + std::to_string(errno) + ")");
    std::stringstream ss;
    ss << f.rdbuf();
    yaml::NodePtr root = yaml::parse(ss.str());

    const Node& cam = field(*root, "camera", "config");
    const Node& scn = field(*root, "scene", "config");
    const Node& surfs = field(scn, "surfaces", "scene");
    const Node& bg = field(scn, "background", "scene");  // required: load.rs:379 has no #[serde(default)]
    if (surfs.kind != Node::Seq && surfs.kind != Node::Null)
        fail("scene.surfaces: expected a sequence at line " + std::to_string(surfs.line));
    V3 background = vec3(bg, "background");
    if (surfs.seq.empty()) fail("scene is empty");

    out = HostScene{};
    auto add_material = [&](const Node& s) {
        out.materials.push_back(material(field(s, "material", "surface")));
        return (int)out.materials.size() - 1;
    };
    for (auto& sp : surfs.seq) {
        const Node& s = *sp;
        std::string t = tag(s, "surface");
        if (t == "mesh") {
            const Node& rel = field(s, "path", "mesh");
            int mat = add_material(s);
            std::vector<Transform> ts = transforms(s.find("transform"));
            std::string dir = dirname_of(yaml_path);
            std::string mesh_path = dir.empty() ? rel.scalar : dir + "/" + rel.scalar;
            try {
                load_mesh(mesh_path, ts, mat, out);
            } catch (const std::exception& e) {
                // the reference's context message names the scene path, not the mesh path (load.rs:69)
                fail("could not load mesh at '" + yaml_path + "': " + e.what());
            }
        } else if (t == "sphere") {
            rt_sphere sph{};
            sph.radius = num(field(s, "radius", "sphere"), "radius");
            v3out(sph.center, vec3(field(s, "position", "sphere"), "position"));
            v3out(sph.center1, vec3(field(s, "position", "sphere"), "position"));  // Sphere::stationary
            sph.material = add_material(s);
            out.surfaces.push_back({RT_SURF_SPHERE, (int32_t)out.spheres.size()});
            out.spheres.push_back(sph);
        } else if (t == "quad") {
            V3 q = vec3(field(s, "position", "quad"), "position");
            const Node& sides = field(s, "sides", "quad");
            if (sides.kind != Node::Seq || sides.seq.size() != 2)
                fail("quad.sides: expected two vectors at line " + std::to_string(sides.line));
            int mat = add_material(s);
            out.surfaces.push_back({RT_SURF_QUAD, (int32_t)out.quads.size()});
            out.quads.push_back(make_quad(q, vec3(*sides.seq[0], "sides"), vec3(*sides.seq[1], "sides"), mat));
        } else if (t == "box") {
            // load.rs:90-156: six quads, transformed about the box's bbox centre
            const Node& cs = field(s, "corners", "box");
            if (cs.kind != Node::Seq || cs.seq.size() != 2)
                fail("box.corners: expected two points at line " + std::to_string(cs.line));
            V3 a = vec3(*cs.seq[0], "corners"), b = vec3(*cs.seq[1], "corners");
            int mat = add_material(s);
            std::vector<Transform> ts = transforms(s.find("transform"));
            V3 lo = vmin(a, b), hi = vmax(a, b);
            V3 dx{hi.x - lo.x, 0, 0}, dy{0, hi.y - lo.y, 0}, dz{0, 0, hi.z - lo.z};
            auto neg = [](V3 v) { return V3{-v.x, -v.y, -v.z}; };
            rt_quad qs[6] = {
                make_quad({lo.x, lo.y, hi.z}, dx, dy, mat),       // front
                make_quad({hi.x, lo.y, hi.z}, neg(dz), dy, mat),  // right
                make_quad({hi.x, lo.y, lo.z}, neg(dx), dy, mat),  // back
                make_quad({lo.x, lo.y, lo.z}, dz, dy, mat),       // left
                make_quad({lo.x, hi.y, hi.z}, dx, neg(dz), mat),  // top
                make_quad({lo.x, lo.y, lo.z}, dx, dz, mat),       // bottom
            };
            std::vector<V3> pts;
            for (auto& q : qs) {
                V3 qq{q.q[0], q.q[1], q.q[2]};
                pts.push_back(qq);
                pts.push_back(qq + V3{q.u[0], q.u[1], q.u[2]} + V3{q.v[0], q.v[1], q.v[2]});
            }
            V3 origin = compute_mesh_center(pts);
            for (auto& q : qs) {
                out.surfaces.push_back({RT_SURF_QUAD, (int32_t)out.quads.size()});
                out.quads.push_back(transform_quad(q, ts, origin));
            }
        } else {
            fail("unknown variant `" + t + "`, expected one of `sphere`, `quad`, `box`, `mesh`");
        }
    }

    // load.rs:352-374
    double fov = num(field(cam, "fov", "camera"), "fov");
    V3 from = vec3(field(cam, "from", "camera"), "from");
    V3 towards = vec3(field(cam, "towards", "camera"), "towards");
    const Node* fd = cam.find("focus_distance");
    const Node* ap = cam.find("aperture");
    bool has_fd = fd && fd->kind != Node::Null;
    double aperture = (ap && ap->kind != Node::Null) ? num(*ap, "aperture") : 0.0;
    out.desc.camera = build_camera(fov, aspect_ratio, from, towards, V3{0, 1, 0}, aperture, has_fd,
                                   has_fd ? num(*fd, "focus_distance") : 0.0);
    v3out(out.desc.background, background);
    out.desc.use_bvh = use_bvh ? 1 : 0;
    out.sync_desc();
}

}  // namespace rt
