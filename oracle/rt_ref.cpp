// rt_ref — CPU oracle for the per-pixel render path of cwbriones/raytracer.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load it; the product (raytracer_amd/) never does.
//
// It is an f64 restatement of the reference's Rust (which cannot be compiled here: no cargo/rustc), file by
// file, same recursion, same operation order, same quirks; build with -ffp-contract=off.  Each function
// cites the reference lines it follows.  Pinning: the reference's own tests hold known answers only for
// Aabb::{centroid,surface_area}, Sphere::bounding_box and the quaternion algebra (trace.rs:247-263,
// surfaces.rs:665-681, geom.rs:493-562); tests/test_oracle_kat.py checks those against this file and
// oracle/loader_ref.py.  Intersection, traversal, scatter and the integrator are pinned by NO reference
// test or golden image, so for them the status is "parity unpinned" (see DESIGN.md): what holds them is
// hand-derived — single-ray analytic cases (tests/test_oracle_kat.py) and closed-form image truths derived on paper
// from the reference's sources (tests/analytic.py: furnaces, a sphere light's cosine law, the fuzzy-metal quirk, the
// thin lens, Schlick's split and Snell's direction through a glass pane, a medium's free path, a moving sphere's
// shutter share, the checker and uv textures), which this file and the GPU path must both meet.
//
// One deliberate difference: the random stream.  The reference draws from rand 0.8.2 SmallRng seeded per
// rayon split (main.rs:154), which is not reproducible even with --seed.  The oracle keeps the
// reference's draw ORDER (jitter u, jitter v, lens disk pairs, shutter time, then per-bounce draws) but
// takes the variates from the counter-seeded PCG stream the GPU kernels use, as 24-bit uniforms that are
// exact in both f32 and f64, so GPU and oracle follow the same paths.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

namespace {

// ------------------------------------------------------------------ geom.rs
struct Vec3 {
    double x, y, z;
    double get(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(double c, Vec3 v) { return {v.x * c, v.y * c, v.z * c}; }  // geom.rs:306-313
inline Vec3 operator*(Vec3 v, double c) { return {v.x * c, v.y * c, v.z * c}; }  // geom.rs:315-322
inline Vec3 operator/(Vec3 v, double c) { return {v.x / c, v.y / c, v.z / c}; }  // geom.rs:332-339
inline Vec3 negate(Vec3 v) { return {-v.x, -v.y, -v.z}; }
inline double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // geom.rs:177-179
inline Vec3 cross(Vec3 a, Vec3 b) {                                              // geom.rs:182-188
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double square_length(Vec3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }  // geom.rs:167-169
inline double length(Vec3 v) { return std::sqrt(square_length(v)); }
inline Vec3 unit(Vec3 v) { return v / length(v); }                                 // geom.rs:191-193
inline Vec3 mul_pointwise(Vec3 a, Vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline bool near_zero(Vec3 v) {  // geom.rs:200-203
    const double s = 1e-8;
    return std::fabs(v.x) < s && std::fabs(v.y) < s && std::fabs(v.z) < s;
}
inline Vec3 min_pointwise(Vec3 a, Vec3 b) { return {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
inline Vec3 max_pointwise(Vec3 a, Vec3 b) { return {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }

const double INF = INFINITY;

// ------------------------------------------------------------------ RNG (shared spec with device_rt.h)
struct Rng {
    uint64_t s;
};
inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline Rng rng_init(uint64_t seed, uint32_t pixel, uint32_t stratum) {
    return {splitmix64(seed ^ splitmix64(((uint64_t)pixel << 32) | stratum))};
}
inline uint32_t rng_next(Rng& r) {
    uint64_t old = r.s;
    r.s = old * 6364136223846793005ull + 1442695040888963407ull;
    uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t)(old >> 59);
    return (xs >> rot) | (xs << ((32u - rot) & 31u));
}
inline double gen_f64(Rng& r) { return (double)(rng_next(r) >> 8) * 0x1p-24; }  // stands in for rng.gen::<f64>()
inline double gen_pm1(Rng& r) { return gen_f64(r) * 2.0 - 1.0; }                // stands in for Uniform::new(-1,1)

// Sampler 1 ("direct"): closed-form maps with the same distributions as the rejection loops of util.rs:15-35
// and a fixed number of draws.  NOT in the reference — it is the GPU's loop-free sampler
// (RT_SAMPLER_DIRECT), restated here so that mode can be compared path by path as well;
// tests/test_oracle_kat.py checks that both samplers give the same distribution and images.
const double TWO_PI = 6.283185307179586476925286766559;
inline Vec3 direct_unit_vector(Rng& r) {
    double z = 1.0 - 2.0 * gen_f64(r);
    double turn = gen_f64(r);
    double rad = std::sqrt(std::fmax(0.0, 1.0 - z * z));
    return {rad * std::cos(TWO_PI * turn), rad * std::sin(TWO_PI * turn), z};
}

// util.rs:15-24
inline Vec3 gen_in_unit_sphere(Rng& r, int sampler = 0) {
    if (sampler == 1) {
        Vec3 u = direct_unit_vector(r);
        return std::cbrt(gen_f64(r)) * u;
    }
    while (true) {
        double x = gen_pm1(r), y = gen_pm1(r), z = gen_pm1(r);  // Vec3::rand_within, geom.rs:196-198
        Vec3 v{x, y, z};
        if (square_length(v) < 1.0) return v;
    }
}
// util.rs:26-35
inline Vec3 gen_in_unit_disk(Rng& r, int sampler = 0) {
    if (sampler == 1) {
        double rad = std::sqrt(gen_f64(r));
        double turn = gen_f64(r);
        return {rad * std::cos(TWO_PI * turn), rad * std::sin(TWO_PI * turn), 0.0};
    }
    while (true) {
        double x = gen_pm1(r), y = gen_pm1(r);
        Vec3 v{x, y, 0.0};
        if (square_length(v) < 1.0) return v;
    }
}

// ------------------------------------------------------------------ trace.rs
struct Ray {  // trace.rs:9-25
    Vec3 origin, dir, inv_dir;
    double time;
    Ray() = default;
    Ray(Vec3 o, Vec3 d, double t) : origin(o), dir(d), inv_dir{1.0 / d.x, 1.0 / d.y, 1.0 / d.z}, time(t) {}
    Vec3 at(double t) const { return origin + dir * t; }  // trace.rs:39-41
};

struct Interval {  // trace.rs:214-225
    double lo, hi;
    bool contains(double t) const { return lo <= t && t <= hi; }
};

struct Material;
struct Hit {  // trace.rs:48-76
    Vec3 point{0, 0, 0}, normal{0, 0, 0};
    double t = 0.0;
    bool front_face = false;
    const Material* material = nullptr;
    int surface = -1;  // oracle-only: id of the surface that produced the hit
    double u = 0.0, v = 0.0;  // Hit::with_uv, trace.rs:78-84: set by spheres only
};
inline Hit make_hit(const Ray& ray, double t, Vec3 outward_normal, const Material* m, int surface) {
    Hit h;
    h.point = ray.at(t);
    h.front_face = dot(ray.dir, outward_normal) < 0.0;
    h.normal = h.front_face ? outward_normal : negate(outward_normal);
    h.t = t;
    h.material = m;
    h.surface = surface;
    return h;
}

struct Aabb {  // trace.rs:93-212
    Vec3 min, max;
    static Aabb empty() { return {{INF, INF, INF}, {-INF, -INF, -INF}}; }
    Aabb pad(double delta) const {  // trace.rs:116-129
        double x = std::fabs(max.x - min.x), y = std::fabs(max.y - min.y), z = std::fabs(max.z - min.z);
        double dx = x >= delta ? 0.0 : delta / 2.0;
        double dy = y >= delta ? 0.0 : delta / 2.0;
        double dz = z >= delta ? 0.0 : delta / 2.0;
        return {min - Vec3{dx, dy, dz}, max + Vec3{dx, dy, dz}};
    }
    bool hit(const Ray& ray, Interval iv) const {  // trace.rs:131-149
        double t_min = iv.lo, t_max = iv.hi;
        for (int a = 0; a < 3; ++a) {
            double inv_d = ray.inv_dir.get(a);
            double t0 = (min.get(a) - ray.origin.get(a)) * inv_d;
            double t1 = (max.get(a) - ray.origin.get(a)) * inv_d;
            if (inv_d < 0.0) std::swap(t0, t1);
            t_min = t0 > t_min ? t0 : t_min;
            t_max = t1 < t_max ? t1 : t_max;
            if (t_max < t_min) return false;
        }
        return true;
    }
    Aabb merge(const Aabb& o) const { return {min_pointwise(min, o.min), max_pointwise(max, o.max)}; }
    double surface_area() const {  // trace.rs:157-167
        if (min.x >= max.x || min.y >= max.y || min.z >= max.z) return 0.0;
        Vec3 l = max - min;
        return 2.0 * (l.x * l.y + l.y * l.z + l.z * l.x);
    }
    Vec3 centroid() const {  // trace.rs:169-174
        Vec3 offset = 0.5 * (max - min);
        return min + offset;
    }
};

// ------------------------------------------------------------------ material.rs
enum Kind { Lambertian = 0, Metal = 1, Dielectric = 2, DiffuseLight = 3, Isotropic = 4 };
enum Tex { SolidColor = 0, Checker = 1, UVChecker = 2, Marble = 3, Image = 4 };

// PerlinNoise, material.rs:281-377
struct PerlinNoise {
    Vec3 ranvec[256];
    int perm[3][256];
    double noise(Vec3 p) const {  // material.rs:325-344
        const long i = (long)std::floor(p.x), j = (long)std::floor(p.y), k = (long)std::floor(p.z);
        Vec3 buf[8];
        for (int ii = 0; ii < 8; ++ii) {
            const int di = ii & 1, dj = (ii & 2) >> 1, dk = (ii & 4) >> 2;
            buf[ii] = ranvec[perm[0][(i + di) & 255] ^ perm[1][(j + dj) & 255] ^ perm[2][(k + dk) & 255]];
        }
        const double u = p.x - std::floor(p.x), v = p.y - std::floor(p.y), w = p.z - std::floor(p.z);
        // trilinear_interp, material.rs:346-368
        const double uu = u * u * (3. - 2. * u), vv = v * v * (3. - 2. * v), ww = w * w * (3. - 2. * w);
        double acc = 0.0;
        for (int ii = 0; ii < 8; ++ii) {
            const double fi = (double)(ii & 1), fj = (double)((ii & 2) >> 1), fk = (double)((ii & 4) >> 2);
            const Vec3 weight{u - fi, v - fj, w - fk};
            acc += (fi * uu + (1.0 - fi) * (1.0 - uu)) * (fj * vv + (1.0 - fj) * (1.0 - vv)) *
                   (fk * ww + (1.0 - fk) * (1.0 - ww)) * dot(buf[ii], weight);
        }
        return acc;
    }
    double turb(Vec3 p) const {  // material.rs:311-323
        double acc = 0.0, weight = 1.0;
        Vec3 temp = p;
        for (int d = 0; d < 7; ++d) {
            acc += weight * noise(temp);
            weight *= 0.5;
            temp = temp * 2.0;
        }
        return std::fabs(acc);
    }
};

struct ImageTex {  // material.rs:379-424
    std::vector<uint8_t> buf;
    size_t width = 0, height = 0;
    Vec3 color_at(double u, double v) const {
        u = std::fmin(std::fmax(u, 0.0), 1.0);
        v = 1.0 - std::fmin(std::fmax(v, 0.0), 1.0);
        if (buf.empty()) return {0.0, 1.0, 1.0};
        u = std::fmin(std::fmax(u, 0.0), 1.0);
        v = std::fmin(std::fmax(v, 0.0), 1.0);
        const size_t i = (size_t)(u * (double)(width - 1)), j = (size_t)(v * (double)(height - 1));
        const uint8_t* px = &buf[3 * (j * width + i)];
        const double color_scale = 1.0 / 255.0;
        return {color_scale * (double)px[0], color_scale * (double)px[1], color_scale * (double)px[2]};
    }
};

struct Material {
    int kind;
    Vec3 albedo;   // Texture::SolidColor; the checkers' "even"
    double param;  // fuzz | refractive index
    int tex = SolidColor;
    Vec3 albedo2{0, 0, 0};  // the checkers' "odd"
    double tex_scale = 1.0;
    const PerlinNoise* perlin = nullptr;
    const ImageTex* image = nullptr;

    // Texture::color_at, material.rs:212-221
    Vec3 color_at(double u, double v, Vec3 p) const {
        switch (tex) {
            case Checker: {  // material.rs:236-247; inv_scale = scale.recip()
                const double inv_scale = 1.0 / tex_scale;
                const long x = (long)std::floor(inv_scale * p.x), y = (long)std::floor(inv_scale * p.y),
                           z = (long)std::floor(inv_scale * p.z);
                return (x + y + z) % 2 == 0 ? albedo : albedo2;
            }
            case UVChecker: {  // material.rs:257-267
                const long a = (long)std::floor(tex_scale * u), b = (long)std::floor(tex_scale * v);
                return (a + b) % 2 == 0 ? albedo : albedo2;
            }
            case Marble: {  // NoiseTexture::marbled, material.rs:300-303
                const double n = 0.5 * (1.0 + std::sin(tex_scale * p.z + 10.0 * perlin->turb(p)));
                return {n, n, n};
            }
            case Image:
                return image->color_at(u, v);
            default:
                return albedo;
        }
    }
};

inline Vec3 reflect(Vec3 v, Vec3 n) { return v - (2.0 * dot(v, n)) * n; }  // material.rs:171-173
inline Vec3 refract(Vec3 uv, Vec3 n, double etai_over_etat) {            // material.rs:175-181
    double cos_theta = std::fmin(dot(negate(uv), n), 1.0);
    Vec3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
    Vec3 r_out_parallel = (-1.0 * std::sqrt(std::fabs(1.0 - square_length(r_out_perp)))) * n;
    return r_out_perp + r_out_parallel;
}
// f64::powi(5) lowers to compiler-rt __powidf2 / LLVM's powi expansion: square-and-multiply, x * ((x*x)*(x*x))
inline double powi5(double x) {
    double x2 = x * x;
    double x4 = x2 * x2;
    return x * x4;
}
inline double reflectance(double cosine, double refractive_index) {  // material.rs:161-168
    double r0 = (1.0 - refractive_index) / (1.0 + refractive_index);
    r0 = r0 * r0;
    return r0 + (1.0 - r0) * powi5(1.0 - cosine);
}

// Material::scatter, material.rs:59-64 / 86-159.  Returns false for "None".
inline bool scatter(const Material& m, const Ray& ray, const Hit& hit, Rng& rng, Vec3& attenuation, Ray& out,
                    int sampler = 0) {
    switch (m.kind) {
        case Lambertian: {  // material.rs:100-118
            Vec3 dir = hit.normal + (sampler == 1 ? direct_unit_vector(rng) : unit(gen_in_unit_sphere(rng)));
            if (near_zero(dir)) dir = hit.normal;
            out = Ray(hit.point, dir, ray.time);
            break;
        }
        case Metal: {  // material.rs:126-138
            Vec3 reflected = reflect(ray.dir, hit.normal);
            Ray scattered(hit.point, reflected + m.param * gen_in_unit_sphere(rng, sampler), ray.time);
            if (!(dot(scattered.dir, hit.normal) > 1e-8)) return false;
            out = scattered;
            break;
        }
        case Dielectric: {  // material.rs:141-159
            double ratio = hit.front_face ? 1.0 / m.param : m.param;
            Vec3 unit_dir = unit(ray.dir);
            double cos_theta = std::fmin(dot(negate(unit_dir), hit.normal), 1.0);
            double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
            bool cannot_refract = ratio * sin_theta > 1.0;
            Vec3 dir = (cannot_refract || reflectance(cos_theta, ratio) > gen_f64(rng))
                           ? reflect(unit_dir, hit.normal)
                           : refract(unit_dir, hit.normal, ratio);
            out = Ray(hit.point, dir, ray.time);
            break;
        }
        case Isotropic:  // material.rs:120-123: the direction is NOT normalised
            out = Ray(hit.point, gen_in_unit_sphere(rng, sampler), ray.time);
            break;
        default:  // DiffuseLight, material.rs:94
            return false;
    }
    attenuation = m.color_at(hit.u, hit.v, hit.point);  // Material::scatter, material.rs:59-64
    return true;
}
inline Vec3 emit(const Material& m, const Hit& hit) {  // material.rs:66-72
    return m.kind == DiffuseLight ? m.color_at(hit.u, hit.v, hit.point) : Vec3{0, 0, 0};
}

// ------------------------------------------------------------------ surfaces.rs
struct Counters {
    uint64_t box_tests = 0, tri_tests = 0, sphere_tests = 0, quad_tests = 0, segments = 0;
};

struct BVHNode;
struct Surface {
    // BvhT: Surface::Bvh (surfaces.rs:374), a tree that is a member of another tree's leaf (example.rs:96-139, 498, 568;
    // make_box, surfaces.rs:542-591)
    enum Type { SphereT = 0, TriangleT = 1, QuadT = 2, MediumT = 3, BvhT = 4 } type;
    std::shared_ptr<BVHNode> sub;  // BvhT
    // Where the reference wraps a tree in Rotated / Translated it builds that tree in the tree's own space and turns
    // the ray at run time; the scenes arrive here with the wrappers baked into the members' coordinates, so a member
    // has two boxes: `box`, what the reference's BUILD sees (its own-space box; for a wrapped tree the turned box,
    // trace.rs:185-211), and `wbox`, the world-space bound the traversal tests (equal to `box` wherever nothing is wrapped).
    Aabb wbox;
    int id;  // index in the scene's surface list
    const Material* material;
    // sphere
    Vec3 center_start, center_end;
    double radius;
    // triangle
    Vec3 v0, v1, v2, n0, n1, n2;
    // quad (surfaces.rs:296-327)
    Vec3 q, u, v, normal, w;
    double d;
    Aabb box;  // bounding_box() evaluated once (the reference recomputes it on every use; same values)

    Vec3 center(double t) const { return (1. - t) * center_start + t * center_end; }  // surfaces.rs:52-54

    // get_sphere_uv, surfaces.rs:108-118 (the reference computes it at every sphere hit; only the uv textures read it)
    static void get_sphere_uv(Vec3 n, double& u, double& v) {
        const double PI = 3.14159265358979323846;
        const double theta = std::acos(-n.y);
        const double phi = std::atan2(-n.z, n.x) + PI;
        u = phi / (2.0 * PI), v = theta / PI;
    }
    void sphere_uv(Vec3 n, Hit& h) const {
        if (material->tex != UVChecker && material->tex != Image) return;
        get_sphere_uv(n, h.u, h.v);
    }

    Aabb bounding_box() const {
        switch (type) {
            case MediumT:  // ConstantMedium::bounding_box = the boundary's (surfaces.rs:648-653), set by the caller
                return box;
            case SphereT: {  // surfaces.rs:57-69
                Vec3 r{radius, radius, radius};
                Aabb b1{center(0.0) - r, center(0.0) + r};
                Aabb b2{center(1.0) - r, center(1.0) + r};
                return b1.merge(b2);
            }
            case TriangleT:  // surfaces.rs:281-291
                return {min_pointwise(min_pointwise(v0, v1), v2), max_pointwise(max_pointwise(v0, v1), v2)};
            default: {  // surfaces.rs:356-359
                Vec3 b = q + u + v;
                return Aabb{min_pointwise(q, b), max_pointwise(q, b)}.pad(0.0001);
            }
        }
    }

    bool sub_hit(const Ray& ray, Interval iv, Hit& out, Counters& c) const;  // Bvh::hit -> its root (bvh.rs:98-102)

    bool hit(const Ray& ray, Interval iv, Hit& out, Counters& c) const {
        switch (type) {
            case BvhT:
                return sub_hit(ray, iv, out, c);
            case MediumT:  // evaluated after the surface trace (Scene::media_hit), where the path's RNG is at hand
                return false;
            case SphereT: {  // surfaces.rs:73-106; the sphere uv (:108-118) only where a texture reads it
                c.sphere_tests++;
                Vec3 cen = center(ray.time);
                Vec3 oc = ray.origin - cen;
                double a = square_length(ray.dir);
                double half_b = dot(oc, ray.dir);
                double cc = square_length(oc) - radius * radius;
                double discriminant = half_b * half_b - a * cc;
                if (discriminant > 0.0) {
                    double root = std::sqrt(discriminant);
                    double temp = (-half_b - root) / a;
                    if (iv.contains(temp)) {
                        Vec3 outward = (ray.at(temp) - cen) / radius;
                        out = make_hit(ray, temp, outward, material, id);
                        sphere_uv(outward, out);
                        return true;
                    }
                    temp = (-half_b + root) / a;
                    if (iv.contains(temp)) {
                        Vec3 outward = (ray.at(temp) - cen) / radius;
                        out = make_hit(ray, temp, outward, material, id);
                        sphere_uv(outward, out);
                        return true;
                    }
                }
                return false;
            }
            case TriangleT: {  // surfaces.rs:246-279
                c.tri_tests++;
                Vec3 e1 = v1 - v0, e2 = v2 - v0;
                Vec3 dir = ray.dir;
                Vec3 pvec = cross(dir, e2);
                double det = dot(e1, pvec);
                if (-1E-8 < det && det < 1E-8) return false;
                double inv_det = 1.0 / det;
                Vec3 tvec = ray.origin - v0;
                double uu = dot(tvec, pvec) * inv_det;
                if (!(0.0 <= uu && uu <= 1.0)) return false;
                Vec3 qvec = cross(tvec, e1);
                double vv = dot(dir, qvec) * inv_det;
                if (vv < 0.0 || uu + vv > 1.0) return false;
                double t = dot(e2, qvec) * inv_det;
                if (t < 1E-4 || !iv.contains(t)) return false;
                Vec3 normal = uu * n1 + vv * n2 + (1.0 - uu - vv) * n0;  // not normalised (Q3)
                out = make_hit(ray, t, normal, material, id);
                return true;
            }
            default: {  // surfaces.rs:330-354
                c.quad_tests++;
                double denom = dot(normal, ray.dir);
                if (std::fabs(denom) < 1E-8) return false;
                double t = (d - dot(normal, ray.origin)) / denom;
                if (t < 1E-4 || !iv.contains(t)) return false;
                Vec3 intersection = ray.at(t);
                Vec3 planar = intersection - q;
                double alpha = dot(w, cross(planar, v));
                double beta = dot(w, cross(u, planar));
                if (0.0 <= alpha && alpha < 1.0 && 0.0 <= beta && beta < 1.0) {  // is_interior, surfaces.rs:362-364
                    out = make_hit(ray, t, normal, material, id);
                    return true;
                }
                return false;
            }
        }
    }
};

// ------------------------------------------------------------------ surfaces/bvh.rs
const double INTERSECT_COST = 1.0, TRAVERSAL_COST = 2.0;
const size_t MAX_LEAF_SIZE = 8;

struct BVHNode {
    bool leaf = false;
    Aabb bound;
    std::unique_ptr<BVHNode> left, right;  // Inner
    std::vector<Surface> surfaces;         // Leaf

    bool hit(const Ray& ray, Interval iv, Hit& out, Counters& c) const {
        c.box_tests++;
        if (!bound.hit(ray, iv)) return false;
        if (leaf) {  // bvh.rs:300-314
            double closest = iv.hi;
            bool any = false;
            Hit h;
            for (const Surface& o : surfaces) {
                if (o.hit(ray, Interval{iv.lo, closest}, h, c)) {
                    closest = h.t;
                    out = h;
                    any = true;
                }
            }
            return any;
        }
        // bvh.rs:262-275: both children, same interval; strictly smaller t keeps left, otherwise right
        Hit a, b;
        bool ha = left->hit(ray, iv, a, c);
        bool hb = right->hit(ray, iv, b, c);
        if (!ha && !hb) return false;
        if (ha && !hb) out = a;
        else if (!ha && hb) out = b;
        else if (a.t < b.t) out = a;
        else out = b;
        return true;
    }
};

bool Surface::sub_hit(const Ray& ray, Interval iv, Hit& out, Counters& c) const { return sub->hit(ray, iv, out, c); }

// After a build over own-space boxes: every node's bound as the world-space bound of what lies below it (a no-op
// where nothing is wrapped: merging is exact).
void rebound(BVHNode* n) {
    if (n->leaf) {
        n->bound = Aabb::empty();
        for (Surface& s : n->surfaces) {
            if (s.type == Surface::BvhT) {
                rebound(s.sub.get());
                s.wbox = s.sub->bound;
            }
            n->bound = n->bound.merge(s.wbox);
        }
        return;
    }
    rebound(n->left.get());
    rebound(n->right.get());
    n->bound = n->left->bound.merge(n->right->bound);
}

std::unique_ptr<BVHNode> new_leaf(std::vector<Surface> surfaces) {  // bvh.rs:289-298
    auto n = std::make_unique<BVHNode>();
    n->leaf = true;
    n->bound = Aabb::empty();
    for (const Surface& s : surfaces) n->bound = n->bound.merge(s.box);
    n->surfaces = std::move(surfaces);
    return n;
}

// SAHSplitStrategy::split_at, bvh.rs:131-174; returns -1 for None
long sah_split_at(const std::vector<Surface>& surfaces) {
    size_t n = surfaces.size();
    if (n <= 1) return -1;
    std::vector<Aabb> merge_left(n), merge_right(n);
    for (size_t i = 0; i < n; ++i) merge_left[i] = surfaces[i].box;
    merge_right = merge_left;
    std::reverse(merge_right.begin(), merge_right.end());
    for (size_t i = 1; i < n; ++i) {
        merge_left[i] = merge_left[i].merge(merge_left[i - 1]);
        merge_right[i] = merge_right[i].merge(merge_right[i - 1]);
    }
    const Aabb& total_bb = merge_left.back();
    double no_split_cost = total_bb.surface_area() * (double)n * INTERSECT_COST;
    long best = -1;
    double best_cost = 0;
    for (size_t split_idx = 1; split_idx < n - 1; ++split_idx) {
        const Aabb& l = merge_left[split_idx - 1];
        const Aabb& r = merge_right[n - split_idx - 1];
        double split_cost = TRAVERSAL_COST + l.surface_area() * (double)split_idx * INTERSECT_COST +
                            r.surface_area() * (double)(n - split_idx) * INTERSECT_COST;
        if (best < 0 || split_cost < best_cost) {  // Iterator::min_by returns the first minimum
            best = (long)split_idx;
            best_cost = split_cost;
        }
    }
    if (best >= 0 && best_cost < no_split_cost) return best;
    return -1;
}

// BVHNode::new_with_strategy, bvh.rs:183-234
std::unique_ptr<BVHNode> build_node(std::vector<Surface> surfaces) {
    if (surfaces.size() <= MAX_LEAF_SIZE) return new_leaf(std::move(surfaces));
    Vec3 mins{INF, INF, INF}, maxs{-INF, -INF, -INF};
    for (const Surface& s : surfaces) {
        Vec3 c = s.box.centroid();
        mins = min_pointwise(mins, c);
        maxs = max_pointwise(maxs, c);
    }
    Vec3 span = maxs - mins;
    int axis = 0;
    if (span.get(axis) < span.get(1)) axis = 1;
    if (span.get(axis) < span.get(2)) axis = 2;
    // slice::sort_by is a stable sort
    std::stable_sort(surfaces.begin(), surfaces.end(), [axis](const Surface& a, const Surface& b) {
        return a.box.min.get(axis) < b.box.min.get(axis);
    });
    long idx = sah_split_at(surfaces);
    if (idx >= 0) {
        std::vector<Surface> right(surfaces.begin() + idx, surfaces.end());  // split_off(idx)
        surfaces.resize((size_t)idx);
        auto n = std::make_unique<BVHNode>();
        n->left = build_node(std::move(surfaces));
        n->right = build_node(std::move(right));
        n->bound = n->left->bound.merge(n->right->bound);
        return n;
    }
    return new_leaf(std::move(surfaces));
}

// ------------------------------------------------------------------ camera.rs
struct Camera {
    Vec3 lower_left, horizontal, vertical, origin;
    double lens_radius;
    Vec3 bu, bv, bw;

    // camera.rs:43-74
    static Camera make(double vfov_deg, double aspect_ratio, Vec3 from, Vec3 towards, double aperture, bool has_focus,
                       double focus_in) {
        double vfov_radians = vfov_deg * (3.14159265358979323846 / 180.0);  // f64::to_radians
        double h = std::tan(vfov_radians / 2.0);
        double viewport_height = 2.0 * h;
        double viewport_width = aspect_ratio * viewport_height;
        double focus_dist = has_focus ? focus_in : length(from - towards);
        Vec3 vup{0., 1., 0.};
        Vec3 w = unit(from - towards);
        Vec3 u = unit(cross(vup, w));
        Vec3 v = cross(w, u);
        Camera c;
        c.origin = from;
        c.horizontal = (focus_dist * viewport_width) * u;
        c.vertical = (focus_dist * viewport_height) * v;
        c.lower_left = c.origin - c.horizontal / 2.0 - c.vertical / 2.0 - focus_dist * w;
        c.lens_radius = aperture / 2.0;
        c.bu = u, c.bv = v, c.bw = w;
        return c;
    }
    // camera.rs:76-87
    Ray get_ray(Rng& rng, double u, double v, int sampler = 0) const {
        Vec3 rd = lens_radius * gen_in_unit_disk(rng, sampler);
        Vec3 offset = bu * rd.x + bv * rd.y;
        Vec3 offset_origin = origin - offset;
        double ray_time = gen_f64(rng);  // gen_range(0.0..1.0)
        return Ray(offset_origin, (lower_left + u * horizontal + v * vertical) - offset_origin, ray_time);
    }
};

// ------------------------------------------------------------------ scene/mod.rs
// ConstantMedium, surfaces.rs:594-653
struct Medium {
    std::vector<Surface> boundary;  // one sphere, or the quads of a make_box (a Bvh of <= 8: one leaf)
    double neg_inv_density;
    const Material* phase;
    int surface;  // id in the scene's surface list

    // the boundary's Hittable::hit: a leaf scan with the interval narrowed to the closest so far (bvh.rs:300-314)
    bool boundary_hit(const Ray& ray, Interval iv, Hit& out) const {
        Counters dummy;
        bool any = false;
        for (const Surface& s : boundary) {
            Hit h;
            if (s.hit(ray, iv, h, dummy)) {
                any = true;
                iv.hi = h.t;
                out = h;
            }
        }
        return any;
    }
};

struct Scene {
    std::vector<Material> materials;
    std::vector<Medium> media;

    // ConstantMedium::hit, surfaces.rs:611-646, for every medium in index order, after the surface trace.  The
    // reference draws the free path from thread_rng() inside the traversal; here (and on the GPU) a medium
    // whose boundary the ray crosses in front of the closest hit so far takes one uniform of the path's stream.
    void media_hit(const Ray& ray, Interval interval, Rng& rng, bool& found, Hit& hit) const {
        for (const Medium& m : media) {
            Hit hit_in, hit_out;
            if (!m.boundary_hit(ray, Interval{-INF, INF}, hit_in)) continue;
            if (!m.boundary_hit(ray, Interval{hit_in.t + 0.0001, INF}, hit_out)) continue;
            double t_in = std::fmax(hit_in.t, interval.lo);
            double t_out = std::fmin(hit_out.t, found ? hit.t : interval.hi);
            if (t_in >= t_out) continue;
            t_in = std::fmax(t_in, 0.0);
            const double ray_length = length(ray.dir);
            const double distance_inside_boundary = (t_out - t_in) * ray_length;
            const double hit_distance = m.neg_inv_density * std::log(gen_f64(rng));
            if (hit_distance > distance_inside_boundary) continue;
            const double t = t_in + hit_distance / ray_length;
            hit = make_hit(ray, t, Vec3{1.0, 0.0, 0.0}, m.phase, m.surface);
            hit.front_face = true;
            found = true;
        }
    }

    std::vector<PerlinNoise> perlin;
    std::vector<ImageTex> images;
    std::unique_ptr<BVHNode> root;
    Vec3 background;
    Camera camera;
    size_t n_surfaces = 0;

    // ray_color_impl + resolve_color, scene/mod.rs:86-135 (the no-scatter case re-traces the same ray: Q1)
    Vec3 ray_color(Ray ray, Rng& rng, size_t max_depth, std::vector<std::pair<Vec3, Vec3>>& stack, Counters& c,
                   int sampler = 0) const {
        stack.clear();
        size_t depth = 0;
        while (true) {
            Interval interval{0.001, INF};
            Hit hit;
            c.segments++;
            bool found = root->hit(ray, interval, hit, c);
            if (!media.empty()) media_hit(ray, interval, rng, found, hit);
            if (found) {
                Vec3 emitted = emit(*hit.material, hit);
                Vec3 att;
                Ray scattered;
                if (scatter(*hit.material, ray, hit, rng, att, scattered, sampler)) {
                    stack.push_back({emitted, att});
                    ray = scattered;
                }
                depth += 1;
                if (depth < max_depth) continue;
                stack.push_back({emitted, Vec3{0, 0, 0}});
            } else {
                stack.push_back({Vec3{0, 0, 0}, background});
            }
            Vec3 acc{1.0, 1.0, 1.0};
            for (size_t i = stack.size(); i-- > 0;) acc = stack[i].first + mul_pointwise(stack[i].second, acc);
            return acc;
        }
    }
};

void collect_order(const BVHNode* n, std::vector<int32_t>& out, uint32_t depth, uint32_t* stats) {
    stats[0]++;                                  // nodes
    stats[2] = std::max(stats[2], depth);        // depth
    if (n->leaf) {
        stats[1]++;                              // leaves
        stats[3] = std::max<uint32_t>(stats[3], (uint32_t)n->surfaces.size());
        for (const Surface& s : n->surfaces) {
            if (s.type == Surface::BvhT) collect_order(s.sub.get(), out, depth + 1, stats);  // a nested tree, in place
            else out.push_back(s.id);
        }
        return;
    }
    collect_order(n->left.get(), out, depth + 1, stats);
    collect_order(n->right.get(), out, depth + 1, stats);
}

}  // namespace

extern "C" {

// Flat inputs (all row-major doubles unless noted):
//   materials  n x 5 : kind r g b param
//   spheres    n x 5 : cx cy cz radius material
//   triangles  n x 19: v0 v1 v2 n0 n1 n2 material
//   quads      n x 10: q u v material          (normal, d, w derived here per Quad::new, surfaces.rs:313-327)
//   surfaces   n x 2 int32: type index         (load order, load.rs:57-158)
//   camera     11    : fov aspect from(3) towards(3) aperture has_focus focus_distance
// Optional extras (null = none), for the example scenes' textures and moving spheres:
//   mat_ext     n_mat x 6 : texture r2 g2 b2 tex_scale tex_index
//   sphere_end  n_sph x 3 : centre at ray time 1 (Sphere::moving, surfaces.rs:43-50)
//   perlin_vec  n_perlin x 256 x 3 doubles, perlin_perm n_perlin x 3 x 256 int32
//   image_dims  n_images x 3 int64: byte offset into image_bytes, width, height
//   media       n_media x 5   : boundary type (0 sphere | 2 quads), first, count, material, density; a surface
//                               row {3, k} places medium k in the surface list
// Nesting (null / 0 = none), rt_bvh_group in include/rt.h:
//   groups             n_groups x 8 : parent (-1 = the scene's own list; listed before its children), leaf (1 =
//                                     Bvh::new_leaf), box lo.xyz hi.xyz in the parent's build (lo.x > hi.x: the members' union)
//   surface_group      n_surf int32 : the group a surface is a member of, -1 = the scene's own list
//   surface_build_box  n_surf x 6   : the surface's box in its group's build where that is not its own (lo.x > hi.x: its own)
void* ref_scene_create_nested(const double* materials, size_t n_mat, const double* spheres, size_t n_sph,
                              const double* triangles, size_t n_tri, const double* quads, size_t n_quad,
                              const int32_t* surfaces, size_t n_surf, const double* background, const double* camera,
                              int use_bvh, const double* mat_ext, const double* sphere_end, const double* perlin_vec,
                              const int32_t* perlin_perm, size_t n_perlin, const uint8_t* image_bytes,
                              const int64_t* image_dims, size_t n_images, const double* media, size_t n_media,
                              const double* groups, size_t n_groups, const int32_t* surface_group,
                              const double* surface_build_box) {
    auto sc = std::make_unique<Scene>();
    sc->perlin.resize(n_perlin);
    for (size_t k = 0; k < n_perlin; ++k) {
        for (int i = 0; i < 256; ++i) sc->perlin[k].ranvec[i] = {perlin_vec[(k * 256 + i) * 3], perlin_vec[(k * 256 + i) * 3 + 1], perlin_vec[(k * 256 + i) * 3 + 2]};
        for (int a = 0; a < 3; ++a)
            for (int i = 0; i < 256; ++i) sc->perlin[k].perm[a][i] = perlin_perm[(k * 3 + a) * 256 + i];
    }
    sc->images.resize(n_images);
    for (size_t k = 0; k < n_images; ++k) {
        sc->images[k].width = (size_t)image_dims[3 * k + 1], sc->images[k].height = (size_t)image_dims[3 * k + 2];
        const size_t n = 3 * sc->images[k].width * sc->images[k].height;
        sc->images[k].buf.assign(image_bytes + image_dims[3 * k], image_bytes + image_dims[3 * k] + n);
    }
    sc->materials.resize(n_mat);
    for (size_t i = 0; i < n_mat; ++i) {
        const double* m = materials + 5 * i;
        sc->materials[i] = {(int)m[0], {m[1], m[2], m[3]}, m[4]};
        if (mat_ext) {
            const double* e = mat_ext + 6 * i;
            Material& mm = sc->materials[i];
            mm.tex = (int)e[0], mm.albedo2 = {e[1], e[2], e[3]}, mm.tex_scale = e[4];
            if (mm.tex == Marble) {
                if ((size_t)e[5] >= n_perlin) return nullptr;
                mm.perlin = &sc->perlin[(size_t)e[5]];
            }
            if (mm.tex == Image) {
                if ((size_t)e[5] >= n_images) return nullptr;
                mm.image = &sc->images[(size_t)e[5]];
            }
        }
    }
    auto V = [](const double* p) { return Vec3{p[0], p[1], p[2]}; };
    auto make_sphere = [&](int index, Surface& s) {
        const double* p = spheres + 5 * index;
        s.type = Surface::SphereT;
        s.center_start = s.center_end = V(p);  // Sphere::stationary, surfaces.rs:34-41
        if (sphere_end) s.center_end = V(sphere_end + 3 * index);  // Sphere::moving
        s.radius = p[3];
        s.material = &sc->materials[(size_t)p[4]];
    };
    auto make_quad = [&](int index, Surface& s) {
        const double* p = quads + 10 * index;
        s.type = Surface::QuadT;
        s.q = V(p), s.u = V(p + 3), s.v = V(p + 6);
        Vec3 n = cross(s.u, s.v);  // Quad::new
        s.normal = unit(n);
        s.d = dot(s.normal, s.q);
        s.w = n / dot(n, n);
        s.material = &sc->materials[(size_t)p[9]];
    };
    sc->media.resize(n_media);
    std::vector<Surface> list;
    list.reserve(n_surf);
    for (size_t i = 0; i < n_surf; ++i) {
        int type = surfaces[2 * i], index = surfaces[2 * i + 1];
        Surface s{};
        s.id = (int)i;
        if (type == 3) {
            if ((size_t)index >= n_media) return nullptr;
            const double* m = media + 5 * index;
            Medium& med = sc->media[(size_t)index];
            const int btype = (int)m[0], first = (int)m[1], count = (int)m[2];
            s.type = Surface::MediumT;
            s.box = Aabb::empty();
            for (int k = 0; k < count; ++k) {
                Surface b{};
                b.id = (int)i;
                if (btype == 0) {
                    if ((size_t)(first + k) >= n_sph) return nullptr;
                    make_sphere(first + k, b);
                } else {
                    if ((size_t)(first + k) >= n_quad) return nullptr;
                    make_quad(first + k, b);
                }
                b.box = b.bounding_box();
                s.box = s.box.merge(b.box);
                med.boundary.push_back(b);
            }
            med.neg_inv_density = -(1.0 / m[4]);
            med.phase = &sc->materials[(size_t)m[3]];
            med.surface = (int)i;
            s.material = med.phase;
            list.push_back(s);
            continue;
        }
        if (type == 0) {
            if ((size_t)index >= n_sph) return nullptr;
            const double* p = spheres + 5 * index;
            s.type = Surface::SphereT;
            s.center_start = s.center_end = V(p);  // Sphere::stationary, surfaces.rs:34-41
            if (sphere_end) s.center_end = V(sphere_end + 3 * index);  // Sphere::moving
            s.radius = p[3];
            s.material = &sc->materials[(size_t)p[4]];
        } else if (type == 1) {
            if ((size_t)index >= n_tri) return nullptr;
            const double* p = triangles + 19 * index;
            s.type = Surface::TriangleT;
            s.v0 = V(p), s.v1 = V(p + 3), s.v2 = V(p + 6), s.n0 = V(p + 9), s.n1 = V(p + 12), s.n2 = V(p + 15);
            s.material = &sc->materials[(size_t)p[18]];
        } else {
            if ((size_t)index >= n_quad) return nullptr;
            const double* p = quads + 10 * index;
            s.type = Surface::QuadT;
            s.q = V(p), s.u = V(p + 3), s.v = V(p + 6);
            Vec3 n = cross(s.u, s.v);  // Quad::new
            s.normal = unit(n);
            s.d = dot(s.normal, s.q);
            s.w = n / dot(n, n);
            s.material = &sc->materials[(size_t)p[9]];
        }
        s.box = s.bounding_box();
        list.push_back(s);
    }
    sc->n_surfaces = n_surf;
    for (Surface& s : list) s.wbox = s.box;
    if (groups && n_groups && surface_group) {
        // members of every list in the order the reference adds them: a group joins its parent's list where its first
        // surface stands
        std::vector<std::vector<int32_t>> members(n_groups + 1);
        std::vector<char> listed(n_groups, 0);
        auto parent_of = [&](int32_t g) { return (int32_t)groups[8 * (size_t)g]; };
        for (size_t g = 0; g < n_groups; ++g)
            if (parent_of((int32_t)g) < -1 || parent_of((int32_t)g) >= (int32_t)g) return nullptr;
        for (size_t i = 0; i < n_surf; ++i) {
            const int32_t g = surface_group[i];
            if (g < -1 || g >= (int32_t)n_groups) return nullptr;
            std::vector<int32_t> chain;
            for (int32_t k = g; k >= 0 && !listed[(size_t)k]; k = parent_of(k)) chain.push_back(k);
            for (size_t c = chain.size(); c-- > 0;) {
                listed[(size_t)chain[c]] = 1;
                members[(size_t)(parent_of(chain[c]) + 1)].push_back(~chain[c]);
            }
            members[(size_t)(g + 1)].push_back((int32_t)i);
        }
        std::function<std::vector<Surface>(int32_t)> items_of;
        auto tree_of = [&](int32_t g) {  // Surface::Bvh for group g
            Surface b{};
            b.type = Surface::BvhT;
            b.id = -1;
            std::vector<Surface> items = items_of(g);
            b.sub = groups[8 * (size_t)g + 1] != 0.0 ? std::shared_ptr<BVHNode>(new_leaf(std::move(items)))
                                                      : std::shared_ptr<BVHNode>(build_node(std::move(items)));
            const double* gb = groups + 8 * (size_t)g + 2;
            b.box = gb[0] > gb[3] ? b.sub->bound : Aabb{V(gb), V(gb + 3)};  // Bvh::bounding_box = the root's
            b.wbox = b.box;  // set by rebound
            return b;
        };
        items_of = [&](int32_t g) {
            std::vector<Surface> items;
            for (int32_t code : members[(size_t)(g + 1)]) {
                if (code < 0) {
                    items.push_back(tree_of(~code));
                    continue;
                }
                Surface s = list[(size_t)code];
                if (surface_build_box) {
                    const double* bb = surface_build_box + 6 * (size_t)code;
                    if (!(bb[0] > bb[3])) s.box = Aabb{V(bb), V(bb + 3)};
                }
                items.push_back(s);
            }
            return items;
        };
        std::vector<Surface> top = items_of(-1);
        sc->root = use_bvh ? build_node(std::move(top)) : new_leaf(std::move(top));
        rebound(sc->root.get());
    } else
    sc->root = use_bvh ? build_node(std::move(list)) : new_leaf(std::move(list));  // scene/mod.rs:60-71
    sc->background = V(background);
    sc->camera = Camera::make(camera[0], camera[1], V(camera + 2), V(camera + 5), camera[8], camera[9] != 0.0, camera[10]);
    return sc.release();
}

void* ref_scene_create_ext(const double* materials, size_t n_mat, const double* spheres, size_t n_sph,
                           const double* triangles, size_t n_tri, const double* quads, size_t n_quad,
                           const int32_t* surfaces, size_t n_surf, const double* background, const double* camera,
                           int use_bvh, const double* mat_ext, const double* sphere_end, const double* perlin_vec,
                           const int32_t* perlin_perm, size_t n_perlin, const uint8_t* image_bytes,
                           const int64_t* image_dims, size_t n_images, const double* media, size_t n_media) {
    return ref_scene_create_nested(materials, n_mat, spheres, n_sph, triangles, n_tri, quads, n_quad, surfaces, n_surf,
                                   background, camera, use_bvh, mat_ext, sphere_end, perlin_vec, perlin_perm, n_perlin,
                                   image_bytes, image_dims, n_images, media, n_media, nullptr, 0, nullptr, nullptr);
}

void* ref_scene_create(const double* materials, size_t n_mat, const double* spheres, size_t n_sph,
                       const double* triangles, size_t n_tri, const double* quads, size_t n_quad,
                       const int32_t* surfaces, size_t n_surf, const double* background, const double* camera,
                       int use_bvh) {
    return ref_scene_create_ext(materials, n_mat, spheres, n_sph, triangles, n_tri, quads, n_quad, surfaces, n_surf,
                                background, camera, use_bvh, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0,
                                nullptr, 0);
}

void ref_scene_destroy(void* h) { delete (Scene*)h; }

// in-order leaf sequence + {nodes, leaves, depth, max leaf size}
void ref_leaf_order(void* h, int32_t* out, uint32_t* stats4) {
    Scene* sc = (Scene*)h;
    std::vector<int32_t> order;
    uint32_t st[4] = {0, 0, 0, 0};
    collect_order(sc->root.get(), order, 1, st);
    std::memcpy(out, order.data(), order.size() * sizeof(int32_t));
    std::memcpy(stats4, st, sizeof st);
}

// camera vectors as built by Camera::new: origin lower_left horizontal vertical u v w (21) + lens_radius
void ref_camera(void* h, double* out22) {
    const Camera& c = ((Scene*)h)->camera;
    const Vec3 v[7] = {c.origin, c.lower_left, c.horizontal, c.vertical, c.bu, c.bv, c.bw};
    for (int i = 0; i < 7; ++i) out22[3 * i] = v[i].x, out22[3 * i + 1] = v[i].y, out22[3 * i + 2] = v[i].z;
    out22[21] = c.lens_radius;
}

// Closest hit of the pixel-centre rays (u = (pi+1)/W, v = (pj+1)/H; no jitter, zero lens offset, time 0)
// through the reference's recursive traversal with interval (0.001, inf).
void ref_primary_hits(void* h, uint32_t width, uint32_t height, int32_t* prim_id, double* t_out, int threads) {
    const Scene* sc = (const Scene*)h;
    std::atomic<uint32_t> next_row{0};
    auto work = [&]() {
        Counters c;
        for (uint32_t row; (row = next_row.fetch_add(1)) < height;) {
            for (uint32_t col = 0; col < width; ++col) {
                uint32_t idx = row * width + col;
                uint32_t pi = idx % width, pj = height - idx / width - 1;
                double u = (double)(pi + 1) / (double)width;
                double v = (double)(pj + 1) / (double)height;
                const Camera& cam = sc->camera;
                Ray ray(cam.origin, (cam.lower_left + u * cam.horizontal + v * cam.vertical) - cam.origin, 0.0);
                Hit hit;
                if (sc->root->hit(ray, Interval{0.001, INF}, hit, c)) {
                    prim_id[idx] = hit.surface;
                    t_out[idx] = hit.t;
                } else {
                    prim_id[idx] = -1;
                    t_out[idx] = -1.0;
                }
            }
        }
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < threads; ++i) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
}

// The pixel loop main.rs:150-185 for strata k = begin, begin+step, ... < sn*sn.  rgb_sum: 3*W*H doubles,
// un-normalised sums, row 0 = top.  counters5: box tests, tri, sphere, quad tests, segments.
void ref_render(void* h, uint32_t width, uint32_t height, uint32_t sn, uint32_t max_depth, uint64_t seed,
                uint32_t stratum_begin, uint32_t stratum_step, int threads, double* rgb_sum, uint64_t* counters5,
                int sampler) {
    const Scene* sc = (const Scene*)h;
    if (stratum_step == 0) stratum_step = 1;
    const double recip_sn = 1.0 / (double)sn;
    const double pixel_du = 1.0 / (double)width, pixel_dv = 1.0 / (double)height;
    std::atomic<uint32_t> next_row{0};
    std::vector<Counters> totals((size_t)std::max(threads, 1));
    auto work = [&](int tid) {
        Counters c;
        std::vector<std::pair<Vec3, Vec3>> stack;
        for (uint32_t row; (row = next_row.fetch_add(1)) < height;) {
            for (uint32_t col = 0; col < width; ++col) {
                uint32_t idx = row * width + col;
                uint32_t pi = idx % width, pj = height - idx / width - 1;  // main.rs:157-158
                double centeru = (double)(pi + 1) / (double)width;         // main.rs:161
                double centerv = (double)(pj + 1) / (double)height;        // main.rs:162
                Vec3 acc{0, 0, 0};
                for (uint32_t k = stratum_begin; k < sn * sn; k += stratum_step) {
                    uint32_t si = k / sn, sj = k % sn;  // main.rs:165-166: si outer, sj inner
                    Rng rng = rng_init(seed, idx, k);
                    double jitteru = -0.5 + recip_sn * ((double)si + gen_f64(rng));
                    double jitterv = -0.5 + recip_sn * ((double)sj + gen_f64(rng));
                    double u = centeru + jitteru * pixel_du;
                    double v = centerv + jitterv * pixel_dv;
                    Ray ray = sc->camera.get_ray(rng, u, v, sampler);
                    Vec3 col3 = sc->ray_color(ray, rng, max_depth, stack, c, sampler);
                    acc = acc + col3;
                }
                rgb_sum[3 * idx] = acc.x, rgb_sum[3 * idx + 1] = acc.y, rgb_sum[3 * idx + 2] = acc.z;
            }
        }
        totals[(size_t)tid] = c;
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < threads; ++i) pool.emplace_back(work, i);
    work(0);
    for (auto& th : pool) th.join();
    if (counters5) {
        Counters s;
        for (const Counters& c : totals) {
            s.box_tests += c.box_tests, s.tri_tests += c.tri_tests, s.sphere_tests += c.sphere_tests;
            s.quad_tests += c.quad_tests, s.segments += c.segments;
        }
        counters5[0] = s.box_tests, counters5[1] = s.tri_tests, counters5[2] = s.sphere_tests;
        counters5[3] = s.quad_tests, counters5[4] = s.segments;
    }
}

// main.rs:177-180: u8 = (256 * clamp(sqrt(sum / n), 0, 0.999)) as u8 ; NaN -> 0
void ref_tonemap(const double* rgb_sum, size_t n_values, double n_samples, uint8_t* out) {
    for (size_t i = 0; i < n_values; ++i) {
        double c = std::sqrt(rgb_sum[i] / n_samples);
        double v = c != c ? 0.0 : std::fmin(std::fmax(c, 0.0), 0.999);
        out[i] = (uint8_t)(256. * v);
    }
}

// ---- single-function hooks for the known-answer tests
double ref_aabb_surface_area(const double* min3, const double* max3) {
    return Aabb{{min3[0], min3[1], min3[2]}, {max3[0], max3[1], max3[2]}}.surface_area();
}
void ref_aabb_centroid(const double* min3, const double* max3, double* out3) {
    Vec3 c = Aabb{{min3[0], min3[1], min3[2]}, {max3[0], max3[1], max3[2]}}.centroid();
    out3[0] = c.x, out3[1] = c.y, out3[2] = c.z;
}
void ref_sphere_bbox(const double* center3, double radius, double* out6) {
    Surface s{};
    s.type = Surface::SphereT;
    s.center_start = s.center_end = {center3[0], center3[1], center3[2]};
    s.radius = radius;
    Aabb b = s.bounding_box();
    out6[0] = b.min.x, out6[1] = b.min.y, out6[2] = b.min.z, out6[3] = b.max.x, out6[4] = b.max.y, out6[5] = b.max.z;
}
int ref_aabb_hit(const double* min3, const double* max3, const double* o3, const double* d3, double tmin, double tmax) {
    Ray r({o3[0], o3[1], o3[2]}, {d3[0], d3[1], d3[2]}, 0.0);
    return Aabb{{min3[0], min3[1], min3[2]}, {max3[0], max3[1], max3[2]}}.hit(r, Interval{tmin, tmax}) ? 1 : 0;
}
// One ray against the whole scene: returns surface id or -1; out: t, point(3), normal(3), front_face
int ref_trace(void* h, const double* o3, const double* d3, double tmin, double tmax, double* out8) {
    const Scene* sc = (const Scene*)h;
    Ray r({o3[0], o3[1], o3[2]}, {d3[0], d3[1], d3[2]}, 0.0);
    Hit hit;
    Counters c;
    if (!sc->root->hit(r, Interval{tmin, tmax}, hit, c)) return -1;
    out8[0] = hit.t;
    out8[1] = hit.point.x, out8[2] = hit.point.y, out8[3] = hit.point.z;
    out8[4] = hit.normal.x, out8[5] = hit.normal.y, out8[6] = hit.normal.z;
    out8[7] = hit.front_face ? 1.0 : 0.0;
    return hit.surface;
}
// Material::scatter on a synthetic hit; returns 1 if scattered.  out: attenuation(3) dir(3)
int ref_scatter(int kind, const double* albedo3, double param, const double* dir3, const double* normal3, int front_face,
                uint64_t rng_state, double* out6) {
    Material m{kind, {albedo3[0], albedo3[1], albedo3[2]}, param};
    Ray ray({0, 0, 0}, {dir3[0], dir3[1], dir3[2]}, 0.0);
    Hit hit{};
    hit.point = {0, 0, 0};
    hit.normal = {normal3[0], normal3[1], normal3[2]};
    hit.front_face = front_face != 0;
    Rng rng{rng_state};
    Vec3 att{0, 0, 0};
    Ray out;
    if (!scatter(m, ray, hit, rng, att, out)) return 0;
    out6[0] = att.x, out6[1] = att.y, out6[2] = att.z, out6[3] = out.dir.x, out6[4] = out.dir.y, out6[5] = out.dir.z;
    return 1;
}
// n samples of in_unit_sphere (which=0), unit(in_unit_sphere) (1) or in_unit_disk (2) with either sampler
void ref_sample(int which, int sampler, uint64_t state, int n, double* out3n) {
    Rng r{state};
    for (int i = 0; i < n; ++i) {
        Vec3 v = which == 0 ? gen_in_unit_sphere(r, sampler)
                 : which == 1 ? (sampler == 1 ? direct_unit_vector(r) : unit(gen_in_unit_sphere(r)))
                              : gen_in_unit_disk(r, sampler);
        out3n[3 * i] = v.x, out3n[3 * i + 1] = v.y, out3n[3 * i + 2] = v.z;
    }
}
double ref_reflectance(double cosine, double ri) { return reflectance(cosine, ri); }
// get_sphere_uv (surfaces.rs:108-118) of a unit outward normal: out = u, v
void ref_sphere_uv(const double* n3, double* out2) { Surface::get_sphere_uv({n3[0], n3[1], n3[2]}, out2[0], out2[1]); }
// first n uniforms of the stream for (seed, pixel, stratum)
void ref_rng_stream(uint64_t seed, uint32_t pixel, uint32_t stratum, int n, double* out) {
    Rng r = rng_init(seed, pixel, stratum);
    for (int i = 0; i < n; ++i) out[i] = gen_f64(r);
}

}  // extern "C"
