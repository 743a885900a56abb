#include "bvh_build.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <future>

namespace rt {

namespace {

V3 v3(const double* p) { return {p[0], p[1], p[2]}; }

struct Item {
    Aabb box;
    int32_t id;
};

const double kIntersectCost = 1.0;  // bvh.rs:15
const double kTraversalCost = 2.0;  // bvh.rs:18
const size_t kMaxLeafSize = 8;      // bvh.rs:23

// Subtrees are independent (disjoint ranges of `items`, sorted in place), so the top levels hand their left
// child to another thread; each task builds into its own node array and the parent splices the arrays back in
// the order a sequential build would have produced (self, left subtree, right subtree).
constexpr size_t kParallelMin = 8192;   // smallest range worth a task
constexpr uint32_t kParallelDepth = 4;  // levels that fork: at most 2^4 tasks

int32_t splice(Bvh& dst, const Bvh& src) {
    const int32_t base = (int32_t)dst.nodes.size();
    for (BvhNode n : src.nodes) {
        if (!n.is_leaf()) n.left += base, n.right += base;
        dst.nodes.push_back(n);
    }
    dst.max_depth = std::max(dst.max_depth, src.max_depth);
    dst.max_leaf_size = std::max(dst.max_leaf_size, src.max_leaf_size);
    dst.n_leaves += src.n_leaves;
    return base;
}

struct Builder {
    std::vector<Item>& items;
    std::vector<double> right_area;             // scratch for the sweep
    std::vector<uint64_t> keys, keys_tmp;       // scratch for the sort
    std::vector<uint32_t> perm, perm_tmp;
    std::vector<Item> sorted;
    Bvh* out = nullptr;
    explicit Builder(std::vector<Item>& it, Bvh* o) : items(it), out(o) {}

    // The reference's `surfaces.sort_by(|a, b| a.min[axis].partial_cmp(b.min[axis]))` (bvh.rs:209-218): a STABLE sort,
    // whose treatment of equal keys decides the tree.  Same result as std::stable_sort with `<` on the keys, three
    // times faster on a mesh: a least-significant-digit radix sort — stable by construction — of (key, position)
    // pairs, byte by byte, skipping the bytes all keys share, then one gather of the 56-byte items.  Keys are the
    // doubles mapped to integers of the same order (-0.0 first folded into +0.0, which `<` cannot tell apart).
    void stable_sort_by_min(size_t first, size_t n, int axis) {
        if (n < 96) {
            std::stable_sort(items.begin() + first, items.begin() + first + n,
                             [axis](const Item& a, const Item& b) { return a.box.lo.get(axis) < b.box.lo.get(axis); });
            return;
        }
        keys.resize(n), keys_tmp.resize(n), perm.resize(n), perm_tmp.resize(n), sorted.resize(n);
        uint64_t all_or = 0, all_and = ~0ull;
        for (size_t i = 0; i < n; ++i) {
            const double v = items[first + i].box.lo.get(axis) + 0.0;  // -0.0 + 0.0 == +0.0
            uint64_t u;
            std::memcpy(&u, &v, 8);
            u ^= (u >> 63) ? ~0ull : 0x8000000000000000ull;
            keys[i] = u, perm[i] = (uint32_t)i;
            all_or |= u, all_and &= u;
        }
        const uint64_t varies = all_or ^ all_and;  // bits that are not the same in every key
        for (int shift = 0; shift < 64; shift += 8) {
            if (((varies >> shift) & 0xffu) == 0) continue;
            size_t count[257] = {0};
            for (size_t i = 0; i < n; ++i) count[((keys[i] >> shift) & 0xffu) + 1]++;
            for (int b = 0; b < 256; ++b) count[b + 1] += count[b];
            for (size_t i = 0; i < n; ++i) {
                const size_t dst = count[(keys[i] >> shift) & 0xffu]++;
                keys_tmp[dst] = keys[i], perm_tmp[dst] = perm[i];
            }
            keys.swap(keys_tmp), perm.swap(perm_tmp);
        }
        for (size_t i = 0; i < n; ++i) sorted[i] = items[first + perm[i]];
        std::copy(sorted.begin(), sorted.begin() + n, items.begin() + first);
    }

    int32_t make_leaf(size_t first, size_t count, uint32_t depth) {
        BvhNode n;
        n.box = Aabb::empty();  // bvh.rs:290-293
        for (size_t i = 0; i < count; ++i) n.box = n.box.merge(items[first + i].box);
        n.first = (int32_t)first;
        n.count = (int32_t)count;
        out->nodes.push_back(n);
        out->max_depth = std::max(out->max_depth, depth);
        out->max_leaf_size = std::max(out->max_leaf_size, (uint32_t)count);
        out->n_leaves++;
        return (int32_t)out->nodes.size() - 1;
    }

    // SAHSplitStrategy::split_at, bvh.rs:131-174.  Returns 0 for "do not split".  The reference builds two arrays of
    // merged boxes (prefixes from the left, from the right) and reads a pair of them per candidate; the same boxes,
    // merged in the same order, are produced here as one array of suffix AREAS and a running prefix box — an eighth
    // of the memory traffic, the very same costs.
    size_t sah_split(size_t first, size_t n) {
        if (n <= 1) return 0;
        right_area.resize(n);  // right_area[k]: surface area of the boxes k .. n-1 merged from the right end
        Aabb acc = items[first + n - 1].box;
        right_area[n - 1] = acc.surface_area();
        for (size_t k = n - 1; k-- > 0;) {
            acc = items[first + k].box.merge(acc);
            right_area[k] = acc.surface_area();
        }
        const double no_split_cost = right_area[0] * (double)n * kIntersectCost;  // all n boxes merged
        size_t best = 0;
        double best_cost = 0.0;
        Aabb left = items[first].box;  // boxes 0 .. s-1
        for (size_t s = 1; s + 1 < n; ++s) {  // (1..len-1): the last candidate is len-2
            if (s > 1) left = items[first + s - 1].box.merge(left);
            double cost = kTraversalCost + left.surface_area() * (double)s * kIntersectCost +
                          right_area[s] * (double)(n - s) * kIntersectCost;
            if (best == 0 || cost < best_cost) {  // min_by keeps the FIRST minimum
                best = s;
                best_cost = cost;
            }
        }
        if (best == 0 || !(best_cost < no_split_cost)) return 0;
        return best;
    }

    // BVHNode::new_with_strategy, bvh.rs:183-234
    int32_t build(size_t first, size_t n, uint32_t depth) {
        if (n <= kMaxLeafSize) return make_leaf(first, n, depth);
        V3 mins{INFINITY, INFINITY, INFINITY}, maxs{-INFINITY, -INFINITY, -INFINITY};
        for (size_t i = 0; i < n; ++i) {
            V3 c = items[first + i].box.centroid();
            mins = vmin(mins, c);
            maxs = vmax(maxs, c);
        }
        V3 span = maxs - mins;
        int axis = 0;
        if (span.get(axis) < span.get(1)) axis = 1;
        if (span.get(axis) < span.get(2)) axis = 2;
        stable_sort_by_min(first, n, axis);
        size_t split = sah_split(first, n);
        if (split == 0) return make_leaf(first, n, depth);
        int32_t self = (int32_t)out->nodes.size();
        out->nodes.emplace_back();
        int32_t l, r;
        if (depth <= kParallelDepth && split >= kParallelMin && n - split >= kParallelMin) {
            Bvh left_tree, right_tree;
            auto left_job = run_async([&]() {
                Builder b(items, &left_tree);
                b.build(first, split, depth + 1);
            });
            Builder b(items, &right_tree);
            b.build(first + split, n - split, depth + 1);
            left_job.get();
            l = splice(*out, left_tree);
            r = splice(*out, right_tree);
        } else {
            l = build(first, split, depth + 1);
            r = build(first + split, n - split, depth + 1);
        }
        BvhNode& node = out->nodes[self];
        node.left = l;
        node.right = r;
        node.box = out->nodes[l].box.merge(out->nodes[r].box);
        return self;
    }
};

}  // namespace

Aabb surface_bounds(const rt_scene_desc& d, const rt_surface_ref& s) {
    switch (s.type) {
        case RT_SURF_SPHERE: {
            const rt_sphere& sp = d.spheres[s.index];
            V3 c = v3(sp.center), c1 = sp.moving ? v3(sp.center1) : c, r{sp.radius, sp.radius, sp.radius};
            Aabb b{c - r, c + r};  // a negative radius yields an inverted box, as in the reference
            return b.merge(Aabb{c1 - r, c1 + r});  // surfaces.rs:57-69: the boxes at ray time 0 and 1
        }
        case RT_SURF_TRIANGLE: {
            const rt_triangle& t = d.triangles[s.index];
            V3 a = v3(t.v[0]), b = v3(t.v[1]), c = v3(t.v[2]);
            return {vmin(vmin(a, b), c), vmax(vmax(a, b), c)};
        }
        case RT_SURF_MEDIUM: {  // ConstantMedium::bounding_box = its boundary's (surfaces.rs:648-653)
            const rt_medium& m = d.media[s.index];
            Aabb box = Aabb::empty();
            for (int32_t k = 0; k < m.count; ++k) box = box.merge(surface_bounds(d, rt_surface_ref{m.boundary, m.first + k}));
            return box;
        }
        default: {
            const rt_quad& q = d.quads[s.index];
            V3 a = v3(q.q), b = a + v3(q.u) + v3(q.v);
            return Aabb{vmin(a, b), vmax(a, b)}.pad(0.0001);
        }
    }
}

Aabb true_bounds(const rt_scene_desc& d, const rt_surface_ref& s) {
    switch (s.type) {
        case RT_SURF_SPHERE: {
            const rt_sphere& sp = d.spheres[s.index];
            const double r = std::fabs(sp.radius);
            const V3 c = v3(sp.center), c1 = sp.moving ? v3(sp.center1) : c;
            return {vmin(c, c1) - V3{r, r, r}, vmax(c, c1) + V3{r, r, r}};
        }
        case RT_SURF_QUAD: {
            const rt_quad& q = d.quads[s.index];
            const V3 a = v3(q.q), u = v3(q.u), v = v3(q.v);
            return {vmin(vmin(a, a + u), vmin(a + v, a + u + v)), vmax(vmax(a, a + u), vmax(a + v, a + u + v))};
        }
        case RT_SURF_MEDIUM: {
            const rt_medium& m = d.media[s.index];
            Aabb box = Aabb::empty();
            for (int32_t k = 0; k < m.count; ++k) box = box.merge(true_bounds(d, rt_surface_ref{m.boundary, m.first + k}));
            return box;
        }
        default:
            return surface_bounds(d, s);
    }
}

void build_bvh(const rt_scene_desc& d, Bvh& out) {
    out = Bvh{};
    std::vector<Item> items(d.n_surfaces);
    for (size_t i = 0; i < d.n_surfaces; ++i) items[i] = {surface_bounds(d, d.surfaces[i]), (int32_t)i};
    Builder b(items, &out);
    out.nodes.reserve(d.n_surfaces / 2 + 16);
    if (d.use_bvh)
        b.build(0, d.n_surfaces, 1);
    else
        b.make_leaf(0, d.n_surfaces, 1);
    out.order.resize(d.n_surfaces);
    for (size_t i = 0; i < d.n_surfaces; ++i) out.order[i] = items[i].id;
}

}  // namespace rt
