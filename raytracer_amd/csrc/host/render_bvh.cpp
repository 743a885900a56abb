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
constexpr size_t kTeamPartitionMin = 65536;  // ... and partition them on a team (the root, mostly: see kTeamSortMin in bvh_build.cpp)
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

constexpr size_t kParallelMin = 2048;   // as in bvh_build.cpp: the top levels fork their left child
constexpr uint32_t kParallelDepth = 8;

using Items = std::vector<Item, Uninit<Item>>;

struct RBuilder {
    Items& items;
    TreePiece* out;
    uint32_t max_leaf;
    double traversal_cost;  // relative to one primitive test
    RBuilder(Items& it, TreePiece* o, uint32_t ml, double ct) : items(it), out(o), max_leaf(ml), traversal_cost(ct) {}

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

    // both subtrees of the node in own[slot] (already pushed); [first, mid) and [mid, first + n), with their bounds
    void children(size_t slot, const Aabb& box, size_t first, size_t mid, size_t n, const Bounds& lb, const Bounds& rb,
                  uint32_t depth) {
        int32_t l, r;
        if (depth <= kParallelDepth && mid - first >= kParallelMin && first + n - mid >= kParallelMin) {
            auto left_tree = std::make_unique<TreePiece>(mid - first), right_tree = std::make_unique<TreePiece>(first + n - mid);
            fork_join(
                [&]() {
                    RBuilder b(items, left_tree.get(), max_leaf, traversal_cost);
                    b.build(first, mid - first, lb, depth + 1);
                },
                [&]() {
                    RBuilder b(items, right_tree.get(), max_leaf, traversal_cost);
                    b.build(mid, first + n - mid, rb, depth + 1);
                });
            l = out->attach(std::move(left_tree));
            r = out->attach(std::move(right_tree));
        } else {
            l = build(first, mid - first, lb, depth + 1);
            r = build(mid, first + n - mid, rb, depth + 1);
        }
        BvhNode& node = out->own[slot];
        node.left = l, node.right = r, node.box = box;
    }

    int32_t leaf(size_t first, size_t n, const Aabb& box, uint32_t depth) {
        BvhNode node;
        node.box = box;
        node.first = (int32_t)first, node.count = (int32_t)n;
        out->max_depth = std::max(out->max_depth, depth);
        out->max_leaf_size = std::max(out->max_leaf_size, (uint32_t)n);
        out->n_leaves++;
        return out->push(node);
    }

    // The partition of a node at the top of the tree, where it is all there is to do, on a team of threads: every
    // member counts the items of its run that go left and gathers both sides' bounds, then moves its items to their
    // places in a second array — the left ones of all runs first, run by run, then the right ones — and copies its
    // share back.  Stable, so the outcome does not depend on how many threads there are.  Returns the left count.
    template <class Pred>
    size_t partition_on_a_team(size_t first, size_t n, Pred&& goes_left, Bounds& lb, Bounds& rb) {
        const size_t members = std::min<size_t>(8, n / 16384);
        Items moved(n);
        std::vector<size_t> lefts(members, 0);
        std::vector<Bounds> lbs(members), rbs(members);
        std::vector<uint8_t, Uninit<uint8_t>> side(n);
        run_team(members, [&](size_t m, Team& team) {
            const auto [begin, end] = team.chunk(m, n);
            Bounds my_lb, my_rb;
            size_t my_left = 0;
            for (size_t i = begin; i < end; ++i) {
                const Item& item = items[first + i];
                const bool left = goes_left(item);
                side[i] = left, my_left += left;
                (left ? my_lb : my_rb).add(item);
            }
            lefts[m] = my_left, lbs[m] = my_lb, rbs[m] = my_rb;
            team.barrier();
            size_t all_left = 0, l = 0, r = 0;  // l, r: left / right items of the runs before this one
            for (size_t k = 0; k < team.members; ++k) {
                all_left += lefts[k];
                if (k < m) l += lefts[k], r += team.chunk(k, n).second - team.chunk(k, n).first - lefts[k];
            }
            r += all_left;
            for (size_t i = begin; i < end; ++i) moved[side[i] ? l++ : r++] = items[first + i];
            team.barrier();
            std::copy(moved.begin() + begin, moved.begin() + end, items.begin() + first + begin);
        });
        size_t all_left = 0;
        for (size_t k = 0; k < members; ++k) {
            all_left += lefts[k];
            lb.box = lb.box.merge(lbs[k].box), lb.cb = lb.cb.merge(lbs[k].cb);
            rb.box = rb.box.merge(rbs[k].box), rb.cb = rb.cb.merge(rbs[k].cb);
        }
        return all_left;
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
            const size_t slot = out->own.size();
            const int32_t self = out->push(BvhNode{});
            children(slot, box, first, mid, n, bounds_of(first, mid - first), bounds_of(mid, first + n - mid), depth);
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
                std::vector<TaskRef> jobs;
                for (size_t t = 1; t < parts; ++t)
                    jobs.push_back(pool_submit([&, t]() { bin_range(first + n * t / parts, first + n * (t + 1) / parts, part[t]); }));
                bin_range(first, first + n / parts, part[0]);
                for (auto& j : jobs) pool_wait(j);
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
            auto goes_left = [&](const Item& item) {
                return std::min(kBins - 1, (int)((item.c.get(best_axis) - blo) * bscale)) <= best_bin;
            };
            bool done = false;
            if (n >= kTeamPartitionMin) {
                try {
                    mid = first + partition_on_a_team(first, n, goes_left, lb, rb);
                    done = true;
                } catch (const std::system_error&) {  // no threads to be had: partition on this one
                    lb = Bounds{}, rb = Bounds{};
                }
            }
            if (!done) {
                // (the big nodes keep their items in order, team or no team, so that the tree does not depend on it)
                auto split = [&](auto&& pred) {
                    return n >= kTeamPartitionMin ? std::stable_partition(items.begin() + first, items.begin() + first + n, pred)
                                                  : std::partition(items.begin() + first, items.begin() + first + n, pred);
                };
                auto it = split([&](const Item& item) {
                    const bool left = goes_left(item);
                    (left ? lb : rb).add(item);  // the predicate is called once per item
                    return left;
                });
                mid = (size_t)(it - items.begin());
            }
            have_bounds = true;
            if (mid == first || mid == first + n) mid = first + n / 2, have_bounds = false;
        }
        if (!have_bounds) lb = bounds_of(first, mid - first), rb = bounds_of(mid, first + n - mid);
        const size_t slot = out->own.size();
        const int32_t self = out->push(BvhNode{});
        children(slot, box, first, mid, n, lb, rb, depth);
        return self;
    }
};

}  // namespace

void build_render_bvh(const rt_scene_desc& d, const std::vector<int32_t>& ids, uint32_t max_leaf, double traversal_cost, Bvh& out) {
    out = Bvh{};
    Items items(ids.size());  // (filled below, in runs, by the threads that then own the pages)
    TreePiece piece(ids.size());
    RBuilder b(items, &piece, std::max(1u, max_leaf), traversal_cost);
    // (the node array's pages are touched for the first time beside the build, not after it: at most 2n - 1 nodes)
    ScopedTask node_room([&]() { out.nodes.resize(2 * ids.size()); });
    constexpr size_t kChunks = 8;
    RBuilder::Bounds part[kChunks];
    std::atomic<size_t> next_part{0};
    parallel_chunks(ids.size(), 8192, kChunks, [&](size_t begin, size_t end) {
        RBuilder::Bounds mine;
        for (size_t i = begin; i < end; ++i) {
            // the reference's box is inverted for a negative radius and too small for some quads: real bounds here,
            // padded like the reference pads its (possibly flat) quad boxes
            const Aabb fixed = true_bounds(d, d.surfaces[ids[i]]).pad(d.surfaces[ids[i]].type == RT_SURF_QUAD ? 0.0001 : 0.0);
            items[i] = {fixed, 0.5 * (fixed.lo + fixed.hi), ids[i]};
            mine.add(items[i]);
        }
        part[next_part.fetch_add(1)] = mine;
    });
    RBuilder::Bounds all;
    for (const RBuilder::Bounds& p : part) all.box = all.box.merge(p.box), all.cb = all.cb.merge(p.cb);
    if (!ids.empty()) b.build(0, ids.size(), all, 1);
    node_room.get();
    piece.finish(out);
    out.order.resize(ids.size());
    for (size_t i = 0; i < ids.size(); ++i) out.order[i] = items[i].id;
}

}  // namespace rt
