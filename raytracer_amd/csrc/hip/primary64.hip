// Exact-parity AOV kernel: closest hit of each pixel-centre ray in f64, built with -ffp-contract=off.
// It keeps the reference's arithmetic operation for operation (trace.rs:131-149, surfaces.rs:73-106,
// 246-279, 330-354) and its visiting rule — every node whose box passes the UN-narrowed interval
// (0.001, inf) is visited (bvh.rs:262-275) — so primitive ids and t agree with the CPU oracle bit for bit.
// Only the visiting ORDER differs; the reference's outcome for equal t (right subtree / later leaf entry
// wins) is reproduced by preferring the higher in-order leaf slot.
#include "device_scene.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace rt {

namespace {

struct V {
    double x, y, z;
};
__device__ inline V mk(const double* p) { return {p[0], p[1], p[2]}; }
__device__ inline V operator+(V a, V b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ inline V operator-(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline V scale(double c, V v) { return {v.x * c, v.y * c, v.z * c}; }  // geom.rs:306-313
__device__ inline double dotv(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline V crossv(V a, V b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ inline double comp(V v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

// Aabb::hit, trace.rs:131-149
__device__ inline bool aabb_hit(const double* lo, const double* hi, V o, V inv, double t_min, double t_max) {
    for (int a = 0; a < 3; ++a) {
        const double inv_d = comp(inv, a);
        double t0 = (lo[a] - comp(o, a)) * inv_d;
        double t1 = (hi[a] - comp(o, a)) * inv_d;
        if (inv_d < 0.0) {
            const double tmp = t0;
            t0 = t1;
            t1 = tmp;
        }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max < t_min) return false;
    }
    return true;
}

__device__ inline bool contains(double lo, double hi, double t) { return lo <= t && t <= hi; }

__device__ inline double prim_hit(const Prim64& p, V o, V d, double t_min, double t_max) {
    if (p.type == 3) return -1.0;  // a constant medium has no deterministic hit: the AOV sees surfaces only
    if (p.type == 1) {  // Triangle::hit
        const V v0 = mk(p.a), v1 = mk(p.a + 3), v2 = mk(p.a + 6);
        const V e1 = v1 - v0, e2 = v2 - v0;
        const V pvec = crossv(d, e2);
        const double det = dotv(e1, pvec);
        if (-1E-8 < det && det < 1E-8) return -1.0;
        const double inv_det = 1.0 / det;
        const V tvec = o - v0;
        const double u = dotv(tvec, pvec) * inv_det;
        if (!(0.0 <= u && u <= 1.0)) return -1.0;
        const V qvec = crossv(tvec, e1);
        const double v = dotv(d, qvec) * inv_det;
        if (v < 0.0 || u + v > 1.0) return -1.0;
        const double t = dotv(e2, qvec) * inv_det;
        if (t < 1E-4 || !contains(t_min, t_max, t)) return -1.0;
        return t;
    }
    if (p.type == 0) {  // Sphere::hit with ray time 0: center(0) = 1*c + 0*c = c exactly
        const V oc = o - mk(p.a);
        const double r = p.a[3];
        const double a = dotv(d, d);
        const double half_b = dotv(oc, d);
        const double c = dotv(oc, oc) - r * r;
        const double disc = half_b * half_b - a * c;
        if (disc > 0.0) {
            const double root = sqrt(disc);
            double temp = (-half_b - root) / a;
            if (contains(t_min, t_max, temp)) return temp;
            temp = (-half_b + root) / a;
            if (contains(t_min, t_max, temp)) return temp;
        }
        return -1.0;
    }
    // Quad::hit: a = normal(3) d q(3) u(3) v(3) w(3)
    const V n = mk(p.a);
    const double denom = dotv(n, d);
    if (fabs(denom) < 1E-8) return -1.0;
    const double t = (p.a[3] - dotv(n, o)) / denom;
    if (t < 1E-4 || !contains(t_min, t_max, t)) return -1.0;
    const V at = o + scale(t, d);
    const V ph = at - mk(p.a + 4);
    const double alpha = dotv(mk(p.a + 13), crossv(ph, mk(p.a + 10)));
    const double beta = dotv(mk(p.a + 13), crossv(mk(p.a + 7), ph));
    if (0.0 <= alpha && alpha < 1.0 && 0.0 <= beta && beta < 1.0) return t;
    return -1.0;
}

__global__ void primary64_kernel(DevScene sc, uint32_t width, uint32_t height, int32_t* __restrict__ prim_id,
                                 double* __restrict__ t_out) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= width * height) return;
    const uint32_t pi = idx % width;
    const uint32_t pj = height - idx / width - 1;
    const double u = (double)(pi + 1) / (double)width;   // main.rs:161
    const double v = (double)(pj + 1) / (double)height;  // main.rs:162
    const V origin = mk(sc.cam64[0]);
    // camera.rs:82-86 with a zero lens offset
    const V d = (mk(sc.cam64[1]) + scale(u, mk(sc.cam64[2])) + scale(v, mk(sc.cam64[3]))) - origin;
    const V inv = {1.0 / d.x, 1.0 / d.y, 1.0 / d.z};  // trace.rs:18
    const double t_min = 0.001, t_inf = __builtin_inf();

    int stack[kPrimary64Stack];  // rt_primary_hits rejects trees deeper than kPrimary64Stack - 1
    int sp = 0;
    stack[sp++] = 0;
    double best_t = t_inf;
    int best_slot = -1;
    while (sp > 0) {
        const int node = stack[--sp];
        if (node == kNoChild) continue;
        if (node >= 0) {
            const Node64& n = sc.nodes64[node];
            for (int c = 0; c < 2; ++c)
                if (n.child[c] != kNoChild && aabb_hit(n.lo[c], n.hi[c], origin, inv, t_min, t_inf))
                    stack[sp++] = n.child[c];
        } else {
            int slot = ~node;
            double closest = t_inf;  // BVHLeafNode::hit narrows inside the leaf only (bvh.rs:305-312)
            int leaf_slot = -1;
            while (true) {
                const Prim64& p = sc.prims64[slot];
                const double t = prim_hit(p, origin, d, t_min, closest);
                if (t >= 0.0) {
                    closest = t;
                    leaf_slot = slot;
                }
                if (p.last) break;
                ++slot;
            }
            // BVHInnerNode::hit: strictly smaller t wins, a tie goes to the right (= later) subtree
            if (leaf_slot >= 0 && (closest < best_t || (closest == best_t && leaf_slot > best_slot))) {
                best_t = closest;
                best_slot = leaf_slot;
            }
        }
    }
    prim_id[idx] = best_slot >= 0 ? sc.prims64[best_slot].surface : -1;
    t_out[idx] = best_slot >= 0 ? best_t : -1.0;
}

}  // namespace

hipError_t launch_primary64(const DevScene& sc, uint32_t width, uint32_t height, int32_t* d_prim_id, double* d_t,
                            hipStream_t stream) {
    const uint32_t n = width * height;
    hipLaunchKernelGGL(primary64_kernel, dim3((n + 63u) / 64u), dim3(64), 0, stream, sc, width, height, d_prim_id, d_t);
    return hipGetLastError();
}

}  // namespace rt
