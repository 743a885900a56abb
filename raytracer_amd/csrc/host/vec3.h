// Host-side f64 3-vector and unit quaternion used while baking YAML transforms and building the BVH.
// Operation order follows the reference so baked vertices agree bit-for-bit (geom.rs:122-224, 349-470).
#pragma once
#include <cmath>

namespace rt {

struct V3 {
    double x = 0, y = 0, z = 0;
    V3() = default;
    V3(double a, double b, double c) : x(a), y(b), z(c) {}
    double get(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};

inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double c, V3 v) { return {v.x * c, v.y * c, v.z * c}; }  // geom.rs:306-313
inline V3 operator*(V3 v, double c) { return {v.x * c, v.y * c, v.z * c}; }
inline V3 operator/(V3 v, double c) { return {v.x / c, v.y / c, v.z / c}; }  // geom.rs:332-339
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   // geom.rs:177-179
inline V3 cross(V3 a, V3 b) {                                                // geom.rs:182-188
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double length(V3 v) { return std::sqrt(dot(v, v)); }
inline V3 unit(V3 v) { return v / length(v); }  // geom.rs:191-193
// Rust f64::min/max: if one operand is NaN the other is returned (geom.rs:28-43).
inline double rmin(double a, double b) { return std::fmin(a, b); }
inline double rmax(double a, double b) { return std::fmax(a, b); }
inline V3 vmin(V3 a, V3 b) { return {rmin(a.x, b.x), rmin(a.y, b.y), rmin(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {rmax(a.x, b.x), rmax(a.y, b.y), rmax(a.z, b.z)}; }

// geom.rs:351-470
struct Quat {
    double x, y, z, w;
    static Quat make(double x, double y, double z, double w) {  // geom.rs:360-368
        double inv_mag = 1.0 / std::sqrt(x * x + y * y + z * z + w * w);
        return {inv_mag * x, inv_mag * y, inv_mag * z, inv_mag * w};
    }
    static Quat rotation(V3 axis, double angle) {  // geom.rs:415-420
        double c = std::cos(angle / 2.0), s = std::sin(angle / 2.0);
        return make(s * axis.x, s * axis.y, s * axis.z, c);
    }
    Quat conj() const { return {-x, -y, -z, w}; }
};

// The 8-multiplication product the reference uses (geom.rs:454-469); kept term for term.
inline Quat operator*(Quat a, Quat b) {
    double ww = (a.z + a.x) * (b.x + b.y);
    double yy = (a.w - a.y) * (b.w + b.z);
    double zz = (a.w + a.y) * (b.w - b.z);
    double xx = ww + yy + zz;
    double qq = 0.5 * (xx + (a.z - a.x) * (b.x - b.y));
    double w = qq - ww + (a.z - a.y) * (b.y - b.z);
    double x = qq - xx + (a.x + a.w) * (b.x + b.w);
    double y = qq - yy + (a.w - a.x) * (b.y + b.z);
    double z = qq - zz + (a.z + a.y) * (b.w - b.x);
    return {x, y, z, w};
}

inline V3 rotate_vec(Quat q, V3 d) {  // geom.rs:424-434
    Quat lhs = Quat{d.x, d.y, d.z, 0.0} * q.conj();
    Quat out = q * lhs;
    return {out.x, out.y, out.z};
}
inline V3 rotate_point(Quat q, V3 origin, V3 p) { return origin + rotate_vec(q, p - origin); }  // geom.rs:437-439

// trace.rs:93-212 (the parts the loader and the BVH build use)
struct Aabb {
    V3 lo, hi;
    static Aabb empty() {
        const double inf = INFINITY;
        return {{inf, inf, inf}, {-inf, -inf, -inf}};
    }
    Aabb merge(const Aabb& o) const { return {vmin(lo, o.lo), vmax(hi, o.hi)}; }
    double surface_area() const {  // trace.rs:157-167
        if (lo.x >= hi.x || lo.y >= hi.y || lo.z >= hi.z) return 0.0;
        V3 l = hi - lo;
        return 2.0 * (l.x * l.y + l.y * l.z + l.z * l.x);
    }
    V3 centroid() const {  // trace.rs:169-174
        V3 off = 0.5 * (hi - lo);
        return lo + off;
    }
    Aabb pad(double delta) const {  // trace.rs:116-129
        double sx = std::fabs(hi.x - lo.x), sy = std::fabs(hi.y - lo.y), sz = std::fabs(hi.z - lo.z);
        double dx = sx >= delta ? 0.0 : delta / 2.0;
        double dy = sy >= delta ? 0.0 : delta / 2.0;
        double dz = sz >= delta ? 0.0 : delta / 2.0;
        return {lo - V3(dx, dy, dz), hi + V3(dx, dy, dz)};
    }
};

}  // namespace rt
