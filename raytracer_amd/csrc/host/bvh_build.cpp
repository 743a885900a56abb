#include "bvh_build.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <future>
#include <stdexcept>

namespace rt {

void TreePiece::lay_out(BvhNode* dst, int32_t base) const {
    // the big attached pieces as tasks of their own, the rest here
    std::vector<TaskRef> jobs;
    for (const Attached& a : attached) {
        if (a.piece->size >= 4096) jobs.push_back(pool_submit([&a, dst, base]() { a.piece->lay_out(dst, base + a.at); }));
        else a.piece->lay_out(dst, base + a.at);
    }
    for (size_t i = 0; i < own.size(); ++i) {
        BvhNode n = own[i];
        if (!n.is_leaf()) n.left += base, n.right += base;
        dst[base + pos[i]] = n;
    }
    for (auto& j : jobs) pool_wait(j);
}

void TreePiece::finish(Bvh& out) const {
    out.nodes.resize((size_t)size);
    lay_out(out.nodes.data(), 0);
    out.max_depth = max_depth, out.max_leaf_size = max_leaf_size, out.n_leaves = n_leaves;
}

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
// child to another thread; each task builds into a piece of its own, which the parent attaches where a sequential
// build would have produced its nodes (self, left subtree, right subtree): TreePiece.
constexpr size_t kParallelMin = 2048;   // smallest range worth a task
// Ranges of at least this many items are sorted by a team of threads: the one or two nodes at the very top, while
// nothing else of this build runs.  (Further down, several nodes are in flight beside the render-tree build; two
// teams at 40-60 000 items each then waited at their barriers for members that had no core: 5-7 ms a sort, not 1.5.)
constexpr size_t kTeamSortMin = 65536;
constexpr uint32_t kParallelDepth = 8;  // levels that fork (with kParallelMin: some fifty tasks on a 100 000-triangle mesh)

struct Builder {
    std::vector<Item>& items;
    // scratch (grown without being filled: every element is written before it is read)
    std::vector<double, Uninit<double>> right_area;           // for the sweep
    std::vector<uint64_t, Uninit<uint64_t>> keys, keys_tmp;  // for the sort
    std::vector<uint32_t, Uninit<uint32_t>> perm, perm_tmp;
    std::vector<Item, Uninit<Item>> sorted;
    TreePiece* out = nullptr;
    explicit Builder(std::vector<Item>& it, TreePiece* o) : items(it), out(o) {}

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
        if (n >= kTeamSortMin) {
            try {
                return team_sort_by_min(first, n, axis);
            } catch (const std::system_error&) {  // no threads to be had: the same sort on this one
            }
        }
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

    // (make_leaf and build return the node's index in the piece and leave its box in `box`)
    // The same sort — the same digits, least significant first, each pass stable — for the ranges at the top of the
    // tree, where it is all there is to do, on a team of threads: every member takes one run of the array, counts its
    // run's digits, finds where its share of each bucket starts (after the smaller buckets and after the same bucket's
    // items of the runs before it) and moves its items there in order.  A stable sort has one outcome: this one.
    static uint64_t sort_key(double v) {
        v += 0.0;  // -0.0 + 0.0 == +0.0
        uint64_t u;
        std::memcpy(&u, &v, 8);
        return u ^ ((u >> 63) ? ~0ull : 0x8000000000000000ull);
    }
    void team_sort_by_min(size_t first, size_t n, int axis) {
        const size_t members = std::min<size_t>(8, n / 16384);
        std::vector<uint64_t> ors(members, 0), ands(members, ~0ull);
        std::vector<std::array<uint32_t, 256>> counts(members);
        run_team(members, [&](size_t m, Team& team) {
            const auto [begin, end] = team.chunk(m, n);
            uint64_t *k_in = keys.data(), *k_out = keys_tmp.data();  // every member swaps its own
            uint32_t *p_in = perm.data(), *p_out = perm_tmp.data();
            uint64_t my_or = 0, my_and = ~0ull;
            for (size_t i = begin; i < end; ++i) {
                const uint64_t u = sort_key(items[first + i].box.lo.get(axis));
                k_in[i] = u, p_in[i] = (uint32_t)i;
                my_or |= u, my_and &= u;
            }
            ors[m] = my_or, ands[m] = my_and;
            team.barrier();
            uint64_t all_or = 0, all_and = ~0ull;
            for (size_t k = 0; k < team.members; ++k) all_or |= ors[k], all_and &= ands[k];
            const uint64_t varies = all_or ^ all_and;  // bits that are not the same in every key
            for (int shift = 0; shift < 64; shift += 8) {
                if (((varies >> shift) & 0xffu) == 0) continue;
                std::array<uint32_t, 256> mine{};
                for (size_t i = begin; i < end; ++i) mine[(k_in[i] >> shift) & 0xffu]++;
                counts[m] = mine;
                team.barrier();
                uint32_t at = 0;  // mine[b] becomes where this run's items of bucket b go
                for (int b = 0; b < 256; ++b) {
                    uint32_t before = 0, all = 0;
                    for (size_t k = 0; k < team.members; ++k) {
                        if (k < m) before += counts[k][b];
                        all += counts[k][b];
                    }
                    mine[b] = at + before, at += all;
                }
                for (size_t i = begin; i < end; ++i) {
                    const uint32_t dst = mine[(k_in[i] >> shift) & 0xffu]++;
                    k_out[dst] = k_in[i], p_out[dst] = p_in[i];
                }
                team.barrier();  // (also: nobody writes counts[] for the next digit before everybody has read this one's)
                std::swap(k_in, k_out), std::swap(p_in, p_out);
            }
            for (size_t i = begin; i < end; ++i) sorted[i] = items[first + p_in[i]];
            team.barrier();
            std::copy(sorted.begin() + begin, sorted.begin() + end, items.begin() + first + begin);
        });
    }

    int32_t make_leaf(size_t first, size_t count, uint32_t depth, Aabb& box) {
        BvhNode n;
        n.box = Aabb::empty();  // bvh.rs:290-293
        for (size_t i = 0; i < count; ++i) n.box = n.box.merge(items[first + i].box);
        n.first = (int32_t)first;
        n.count = (int32_t)count;
        box = n.box;
        out->max_depth = std::max(out->max_depth, depth);
        out->max_leaf_size = std::max(out->max_leaf_size, (uint32_t)count);
        out->n_leaves++;
        return out->push(n);
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
    int32_t build(size_t first, size_t n, uint32_t depth, Aabb& box) {
        if (n <= kMaxLeafSize) return make_leaf(first, n, depth, box);
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
        if (split == 0) return make_leaf(first, n, depth, box);
        const size_t slot = out->own.size();
        const int32_t self = out->push(BvhNode{});
        int32_t l, r;
        Aabb lbox, rbox;
        if (depth <= kParallelDepth && split >= kParallelMin && n - split >= kParallelMin) {
            auto left_tree = std::make_unique<TreePiece>(split), right_tree = std::make_unique<TreePiece>(n - split);
            fork_join(
                [&]() {
                    Builder b(items, left_tree.get());
                    b.build(first, split, depth + 1, lbox);
                },
                [&]() {
                    Builder b(items, right_tree.get());
                    b.build(first + split, n - split, depth + 1, rbox);
                });
            l = out->attach(std::move(left_tree));
            r = out->attach(std::move(right_tree));
        } else {
            l = build(first, split, depth + 1, lbox);
            r = build(first + split, n - split, depth + 1, rbox);
        }
        BvhNode& node = out->own[slot];
        node.left = l;
        node.right = r;
        node.box = box = lbox.merge(rbox);
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

namespace {

// ---- nested trees (rt_bvh_group; Surface::Bvh, surfaces.rs:374) ------------------------------------------------
// Every group — and the scene's own list, "group -1" — gets a tree of its own over its members, a member being a
// surface or a whole sub-group with one box, exactly as the reference's builds see them.  The trees are then laid
// into ONE node array in in-order, so that Bvh::order is the reference's in-order leaf sequence with every nested
// tree expanded in place.  A reference leaf [a, b, G, c] becomes the chain ((leaf[a, b], tree(G)), leaf[c]): an inner
// node prefers its right child on a tie (bvh.rs:272-273) as a leaf prefers its later member (bvh.rs:308), so the
// chain resolves every tie as the leaf does.  Node boxes are the world-space bounds of what lies below them.
struct Nested {
    const rt_scene_desc& d;
    Bvh& out;
    struct Tree {
        Bvh t;                      // over the group's members; t.order: member codes (>= 0 surface id, < 0 ~group)
        std::vector<int32_t> members;
        Aabb build_box = Aabb::empty();  // union of the members' boxes in this group's build
    };
    std::vector<Tree> trees;  // [0] = the scene's own list, [g + 1] = group g

    Nested(const rt_scene_desc& desc, Bvh& o) : d(desc), out(o), trees(desc.n_groups + 1) {}

    Aabb member_box(int32_t code) const {
        if (code < 0) {
            const rt_bvh_group& g = d.groups[~code];
            if (!(g.box[0] > g.box[3])) return Aabb{V3{g.box[0], g.box[1], g.box[2]}, V3{g.box[3], g.box[4], g.box[5]}};
            return trees[(size_t)(~code) + 1].build_box;
        }
        if (d.surface_build_box) {
            const double* b = d.surface_build_box + 6 * (size_t)code;
            if (!(b[0] > b[3])) return Aabb{V3{b[0], b[1], b[2]}, V3{b[3], b[4], b[5]}};
        }
        return surface_bounds(d, d.surfaces[code]);
    }

    void collect_members() {
        std::vector<uint8_t> listed(d.n_groups, 0);
        for (size_t i = 0; i < d.n_surfaces; ++i) {
            const int32_t g = d.surface_group[i];
            // the groups on the way up that this surface is the first member of join their parents' lists first
            std::vector<int32_t> chain;
            for (int32_t k = g; k >= 0 && !listed[k]; k = d.groups[k].parent) chain.push_back(k);
            for (size_t c = chain.size(); c-- > 0;) {
                listed[chain[c]] = 1;
                trees[(size_t)(d.groups[chain[c]].parent + 1)].members.push_back(~chain[c]);
            }
            trees[(size_t)(g + 1)].members.push_back((int32_t)i);
        }
    }

    // the group's own tree (children first: a sub-group without a given box has its members' union)
    void build_tree(int32_t g) {
        Tree& tr = trees[(size_t)(g + 1)];
        for (int32_t code : tr.members)
            if (code < 0) build_tree(~code);
        std::vector<Item> items(tr.members.size());
        for (size_t i = 0; i < items.size(); ++i) {
            items[i] = {member_box(tr.members[i]), tr.members[i]};
            tr.build_box = tr.build_box.merge(items[i].box);
        }
        TreePiece piece;
        Builder b(items, &piece);
        const bool one_leaf = g >= 0 ? d.groups[g].leaf != 0 : !d.use_bvh;
        Aabb box;
        if (one_leaf) b.make_leaf(0, items.size(), 1, box);
        else b.build(0, items.size(), 1, box);
        piece.finish(tr.t);
        tr.t.order.resize(items.size());
        for (size_t i = 0; i < items.size(); ++i) tr.t.order[i] = items[i].id;
    }

    int32_t new_inner(int32_t l, int32_t r) {
        BvhNode n;
        n.left = l, n.right = r;
        n.box = out.nodes[l].box.merge(out.nodes[r].box);
        out.nodes.push_back(n);
        return (int32_t)out.nodes.size() - 1;
    }

    // lays node `node` of group g's tree into `out`; depth: the reference's (root of the scene = 1), flat: the array's
    int32_t emit(int32_t g, int32_t node, uint32_t depth, uint32_t flat) {
        const Bvh& t = trees[(size_t)(g + 1)].t;
        const BvhNode& n = t.nodes[node];
        out.max_depth = std::max(out.max_depth, depth);
        out.flat_depth = std::max(out.flat_depth, flat);
        if (!n.is_leaf()) {
            const int32_t self = (int32_t)out.nodes.size();
            out.nodes.emplace_back();
            const int32_t l = emit(g, n.left, depth + 1, flat + 1);
            const int32_t r = emit(g, n.right, depth + 1, flat + 1);
            BvhNode& me = out.nodes[self];
            me.left = l, me.right = r;
            me.box = out.nodes[l].box.merge(out.nodes[r].box);
            return self;
        }
        out.n_leaves++;
        out.max_leaf_size = std::max(out.max_leaf_size, (uint32_t)n.count);
        // runs of surfaces and sub-trees, left to right
        std::vector<int32_t> parts, runs;
        int32_t k = n.first;
        const int32_t end = n.first + n.count;
        // (chain nodes sit above the parts: a part's flat depth is at most flat + number of parts - 1)
        uint32_t n_parts = 0;
        for (int32_t i = n.first; i < end;) {
            if (t.order[i] < 0) ++n_parts, ++i;
            else { ++n_parts; while (i < end && t.order[i] >= 0) ++i; }
        }
        const uint32_t part_flat = flat + (n_parts > 1 ? n_parts - 1 : 0);
        do {
            if (k < end && t.order[k] < 0) {
                parts.push_back(emit(~t.order[k], 0, depth + 1, part_flat + 0));
                ++k;
                continue;
            }
            BvhNode leaf;
            leaf.box = Aabb::empty();
            leaf.first = (int32_t)out.order.size();
            while (k < end && t.order[k] >= 0) {
                leaf.box = leaf.box.merge(surface_bounds(d, d.surfaces[t.order[k]]));
                out.order.push_back(t.order[k++]);
            }
            leaf.count = (int32_t)out.order.size() - leaf.first;
            out.nodes.push_back(leaf);
            out.flat_depth = std::max(out.flat_depth, part_flat);
            parts.push_back((int32_t)out.nodes.size() - 1);
            runs.push_back(parts.back());
        } while (k < end);
        int32_t top = parts[0];
        for (size_t p = 1; p < parts.size(); ++p) top = new_inner(top, parts[p]);
        if (parts.size() > 1)
            for (int32_t r : runs) out.nodes[r].gate = top;
        return top;
    }

    void run() {
        for (size_t g = 0; g < d.n_groups; ++g)
            if (d.groups[g].parent < -1 || d.groups[g].parent >= (int32_t)g)
                throw std::runtime_error("rt_bvh_group: a group's parent must be listed before it");
        for (size_t i = 0; i < d.n_surfaces; ++i)
            if (d.surface_group[i] < -1 || d.surface_group[i] >= (int32_t)d.n_groups)
                throw std::runtime_error("surface_group: group index out of range");
        collect_members();
        for (size_t g = 0; g < d.n_groups; ++g)
            if (trees[g + 1].members.empty()) throw std::runtime_error("rt_bvh_group: a group without members");
        build_tree(-1);
        for (const Tree& tr : trees) out.nodes.reserve(out.nodes.size() + tr.t.nodes.size());
        out.order.reserve(d.n_surfaces);
        // the root must be node 0: an inner root reserves its slot first (emit does); a chain's top is made last
        const int32_t root = emit(-1, 0, 1, 1);
        if (root != 0) {  // the scene's own list is one leaf holding groups: move the chain's top to the front
            std::swap(out.nodes[0], out.nodes[root]);
            auto fix = [&](int32_t& r) { r = r == 0 ? root : (r == root ? 0 : r); };
            for (BvhNode& n : out.nodes) {
                if (!n.is_leaf()) fix(n.left), fix(n.right);
                if (n.gate >= 0) fix(n.gate);
            }
        }
        if (out.order.size() != d.n_surfaces) throw std::runtime_error("nested BVH build lost surfaces (groups not contiguous?)");
        // the count of nodes as the reference's recursion sees them: every tree's own nodes
        size_t n_ref_nodes = 0;
        for (const Tree& tr : trees) n_ref_nodes += tr.t.nodes.size();
        out.ref_nodes = n_ref_nodes;
    }
};

}  // namespace

void build_bvh(const rt_scene_desc& d, Bvh& out) {
    out = Bvh{};
    if (d.n_groups > 0 && d.surface_group) {
        Nested nested(d, out);
        nested.run();
        return;
    }
    std::vector<Item> items(d.n_surfaces);
    parallel_chunks(d.n_surfaces, 8192, 8, [&](size_t begin, size_t end) {
        for (size_t i = begin; i < end; ++i) items[i] = {surface_bounds(d, d.surfaces[i]), (int32_t)i};
    });
    TreePiece piece(d.n_surfaces);
    Builder b(items, &piece);
    Aabb box;
    if (d.use_bvh)
        b.build(0, d.n_surfaces, 1, box);
    else
        b.make_leaf(0, d.n_surfaces, 1, box);
    piece.finish(out);
    out.order.resize(d.n_surfaces);
    for (size_t i = 0; i < d.n_surfaces; ++i) out.order[i] = items[i].id;
}

}  // namespace rt
