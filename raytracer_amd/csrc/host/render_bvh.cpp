// Binned-SAH BVH for the f32 render kernels (no reference counterpart: the reference traverses the tree of
// bvh.rs:131-234, whose absolute traversal cost of 2.0 yields 30-90-triangle leaves on small-scale meshes such
// as lucy.yaml; on a 64-lane machine short, uniform leaves matter more than build fidelity).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <future>

#include "bvh_build.h"

namespace rt {

namespace {

constexpr int kBins = 32;
constexpr size_t kParallelBinMin = 32768;  // nodes of at least this many items bin them on several threads
constexpr size_t kSmallNode = 16;  // nodes of at most this many items find their split without the bin arrays
constexpr double kIntersectCost = 1.0;

struct Item {
    Aabb box;
    V3 c;
    int32_t id;
};

double area(const Aabb& b) {
    V3 l = b.hi - b.lo;
    if (!(l.x >= 0 && l.y >= 0 && l.z >= 0)) return 0.0;
    return 2.0 * (l.x * l.y + l.y * l.z + l.z * l.x);
}

constexpr uint32_t kBalancedFromDepth = 32;  // root = 1; 32 + log2(2^24 primitives) = 56 = the kernels' stack

constexpr size_t kParallelMin = 8192;   // as in bvh_build.cpp: the top levels fork their left child
constexpr uint32_t kParallelDepth = 4;

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

struct RBuilder {
    std::vector<Item>& items;
    Bvh* out;
    uint32_t max_leaf;
    double traversal_cost;  // relative to one primitive test
    RBuilder(std::vector<Item>& it, Bvh* o, uint32_t ml, double ct) : items(it), out(o), max_leaf(ml), traversal_cost(ct) {}

    struct Bounds {
        Aabb box = Aabb::empty();  // of the items' boxes
        Aabb cb = Aabb::empty();   // of their centroids
        void add(const Item& it) {
            box = box.merge(it.box);
            cb.lo = vmin(cb.lo, it.c), cb.hi = vmax(cb.hi, it.c);
        }
    };
    Bounds bounds_of(size_t first, size_t n) const {
        Bounds b;
        for (size_t i = first; i < first + n; ++i) b.add(items[i]);
        return b;
    }

    // both subtrees of the node `self` (already pushed); [first, mid) and [mid, first + n), with their bounds
    void children(int32_t self, const Aabb& box, size_t first, size_t mid, size_t n, const Bounds& lb, const Bounds& rb,
                  uint32_t depth) {
        int32_t l, r;
        if (depth <= kParallelDepth && mid - first >= kParallelMin && first + n - mid >= kParallelMin) {
            Bvh left_tree, right_tree;
            auto left_job = run_async([&]() {
                RBuilder b(items, &left_tree, max_leaf, traversal_cost);
                b.build(first, mid - first, lb, depth + 1);
            });
            RBuilder b(items, &right_tree, max_leaf, traversal_cost);
            b.build(mid, first + n - mid, rb, depth + 1);
            left_job.get();
            l = splice(*out, left_tree);
            r = splice(*out, right_tree);
        } else {
            l = build(first, mid - first, lb, depth + 1);
            r = build(mid, first + n - mid, rb, depth + 1);
        }
        BvhNode& node = out->nodes[self];
        node.left = l, node.right = r, node.box = box;
    }

    int32_t leaf(size_t first, size_t n, const Aabb& box, uint32_t depth) {
        BvhNode node;
        node.box = box;
        node.first = (int32_t)first, node.count = (int32_t)n;
        out->nodes.push_back(node);
        out->max_depth = std::max(out->max_depth, depth);
        out->max_leaf_size = std::max(out->max_leaf_size, (uint32_t)n);
        out->n_leaves++;
        return (int32_t)out->nodes.size() - 1;
    }

    // A node's bounds come down from its parent, which gathered them while it partitioned its items (the
    // partition's predicate sees every item once): a node then reads its items twice — one pass that bins them on
    // all three axes at once, one that partitions — where it read them five times (bounds, a binning pass per axis,
    // partition).  Same bins, same costs, same partition: the same tree, in 40 % less time.
    int32_t build(size_t first, size_t n, const Bounds& bounds, uint32_t depth) {
        const Aabb &box = bounds.box, &cb = bounds.cb;
        if (n == 1) return leaf(first, n, box, depth);
        // SAH can peel primitives off one at a time (geometrically shrinking clusters) and grow a tree as deep
        // as it is large; the kernels' traversal stacks hold 56 levels.  From this depth on split at the
        // median of the widest centroid axis, which bounds the rest of the descent by log2(n).
        if (depth >= kBalancedFromDepth) {
            if (n <= max_leaf) return leaf(first, n, box, depth);
            int axis = 0;
            for (int a = 1; a < 3; ++a)
                if (cb.hi.get(a) - cb.lo.get(a) > cb.hi.get(axis) - cb.lo.get(axis)) axis = a;
            const size_t mid = first + n / 2;
            std::nth_element(items.begin() + first, items.begin() + mid, items.begin() + first + n,
                             [axis](const Item& a, const Item& b) { return a.c.get(axis) < b.c.get(axis); });
            const int32_t self = (int32_t)out->nodes.size();
            out->nodes.emplace_back();
            children(self, box, first, mid, n, bounds_of(first, mid - first), bounds_of(mid, first + n - mid), depth);
            return self;
        }

        double lo[3], scale[3];
        bool spread[3];
        for (int axis = 0; axis < 3; ++axis) {
            lo[axis] = cb.lo.get(axis);
            const double ext = cb.hi.get(axis) - lo[axis];
            spread[axis] = ext > 0;
            scale[axis] = spread[axis] ? kBins / ext : 0.0;
        }
        double best_cost = INFINITY;
        int best_axis = -1, best_bin = -1;
        if (n <= kSmallNode) {
            // Most nodes hold a handful of items: clearing and sweeping 3 x 32 bins for them was half of the build.
            // The same split falls out of the occupied bins alone — between two occupied bins the sweep's cost does
            // not change and its strict `<` keeps the first, the occupied one — so: the items' bin numbers in
            // ascending order (an insertion sort of at most kSmallNode), suffix areas, one cost per occupied bin.
            for (int axis = 0; axis < 3; ++axis) {
                if (!spread[axis]) continue;
                int bin[kSmallNode];
                uint8_t order[kSmallNode];
                for (size_t i = 0; i < n; ++i) {
                    bin[i] = std::min(kBins - 1, (int)((items[first + i].c.get(axis) - lo[axis]) * scale[axis]));
                    size_t j = i;
                    for (; j > 0 && bin[order[j - 1]] > bin[i]; --j) order[j] = order[j - 1];
                    order[j] = (uint8_t)i;
                }
                double right_area[kSmallNode];  // of the items order[j..n-1]
                Aabb acc = Aabb::empty();
                for (size_t j = n; j-- > 0;) {
                    acc = acc.merge(items[first + order[j]].box);
                    right_area[j] = area(acc);
                }
                acc = Aabb::empty();
                for (size_t j = 0; j + 1 < n; ++j) {
                    acc = acc.merge(items[first + order[j]].box);
                    if (bin[order[j + 1]] == bin[order[j]]) continue;  // not the last item of its bin
                    const double cost = area(acc) * (double)(j + 1) + right_area[j + 1] * (double)(n - j - 1);
                    if (cost < best_cost) best_cost = cost, best_axis = axis, best_bin = bin[order[j]];
                }
            }
        } else {
            // one pass: every item into its bin on each axis whose centroids spread at all
            struct Bins {
                Aabb bb[3][kBins];
                size_t cnt[3][kBins];
                Bins() {
                    for (int axis = 0; axis < 3; ++axis)
                        for (int k = 0; k < kBins; ++k) bb[axis][k] = Aabb::empty(), cnt[axis][k] = 0;
                }
            };
            auto bin_range = [&](size_t begin, size_t end, Bins& b) {
                for (size_t i = begin; i < end; ++i) {
                    const Item& it = items[i];
                    for (int axis = 0; axis < 3; ++axis) {
                        if (!spread[axis]) continue;
                        const int k = std::min(kBins - 1, (int)((it.c.get(axis) - lo[axis]) * scale[axis]));
                        b.cnt[axis][k]++;
                        b.bb[axis][k] = b.bb[axis][k].merge(it.box);
                    }
                }
            };
            Bins bins;
            if (n >= kParallelBinMin) {
                // the top of the tree, where one node is all there is to do: the items in runs on several threads,
                // each with bins of its own, united afterwards (counts add up, boxes merge: the order does not matter)
                const size_t parts = std::min<size_t>(8, n / (kParallelBinMin / 4));
                std::vector<Bins> part(parts);
                std::vector<std::future<void>> jobs;
                for (size_t t = 1; t < parts; ++t)
                    jobs.push_back(run_async([&, t]() { bin_range(first + n * t / parts, first + n * (t + 1) / parts, part[t]); }));
                bin_range(first, first + n / parts, part[0]);
                for (auto& j : jobs) j.get();
                for (const Bins& p : part)
                    for (int axis = 0; axis < 3; ++axis)
                        for (int k = 0; k < kBins; ++k)
                            bins.cnt[axis][k] += p.cnt[axis][k], bins.bb[axis][k] = bins.bb[axis][k].merge(p.bb[axis][k]);
            } else {
                bin_range(first, first + n, bins);
            }
            Aabb (&bb)[3][kBins] = bins.bb;
            size_t (&cnt)[3][kBins] = bins.cnt;
            for (int axis = 0; axis < 3; ++axis) {
                if (!spread[axis]) continue;
                double right_area[kBins];
                size_t right_cnt[kBins];
                Aabb acc = Aabb::empty();
                size_t c = 0;
                for (int k = kBins - 1; k > 0; --k) {
                    acc = acc.merge(bb[axis][k]);
                    c += cnt[axis][k];
                    right_area[k] = area(acc), right_cnt[k] = c;
                }
                acc = Aabb::empty(), c = 0;
                for (int k = 0; k + 1 < kBins; ++k) {
                    acc = acc.merge(bb[axis][k]);
                    c += cnt[axis][k];
                    if (c == 0 || right_cnt[k + 1] == 0) continue;
                    double cost = area(acc) * (double)c + right_area[k + 1] * (double)right_cnt[k + 1];
                    if (cost < best_cost) best_cost = cost, best_axis = axis, best_bin = k;
                }
            }
        }
        const double parent_area = std::max(area(box), 1e-300);
        const double split_cost = traversal_cost + kIntersectCost * best_cost / parent_area;
        const double leaf_cost = kIntersectCost * (double)n;
        if (n <= max_leaf && (best_axis < 0 || leaf_cost <= split_cost)) return leaf(first, n, box, depth);

        size_t mid;
        Bounds lb, rb;
        bool have_bounds = false;
        if (best_axis < 0) {
            mid = first + n / 2;  // coincident centroids: split by position in the list
        } else {
            const double blo = lo[best_axis], bscale = scale[best_axis];
            auto it = std::partition(items.begin() + first, items.begin() + first + n, [&](const Item& item) {
                const bool left = std::min(kBins - 1, (int)((item.c.get(best_axis) - blo) * bscale)) <= best_bin;
                (left ? lb : rb).add(item);  // the predicate is called once per item
                return left;
            });
            mid = (size_t)(it - items.begin());
            have_bounds = true;
            if (mid == first || mid == first + n) mid = first + n / 2, have_bounds = false;
        }
        if (!have_bounds) lb = bounds_of(first, mid - first), rb = bounds_of(mid, first + n - mid);
        const int32_t self = (int32_t)out->nodes.size();
        out->nodes.emplace_back();
        children(self, box, first, mid, n, lb, rb, depth);
        return self;
    }
};

}  // namespace

void build_render_bvh(const rt_scene_desc& d, const std::vector<int32_t>& ids, uint32_t max_leaf, double traversal_cost, Bvh& out) {
    out = Bvh{};
    std::vector<Item> items(ids.size());
    RBuilder b(items, &out, std::max(1u, max_leaf), traversal_cost);
    for (size_t i = 0; i < ids.size(); ++i) {
        // the reference's box is inverted for a negative radius and too small for some quads: real bounds here,
        // padded like the reference pads its (possibly flat) quad boxes
        const Aabb fixed = true_bounds(d, d.surfaces[ids[i]]).pad(d.surfaces[ids[i]].type == RT_SURF_QUAD ? 0.0001 : 0.0);
        items[i] = {fixed, 0.5 * (fixed.lo + fixed.hi), ids[i]};
    }
    out.nodes.reserve(ids.size());
    if (!ids.empty()) b.build(0, ids.size(), b.bounds_of(0, ids.size()), 1);
    out.order.resize(ids.size());
    for (size_t i = 0; i < ids.size(); ++i) out.order[i] = items[i].id;
}

}  // namespace rt
