// Host BVH build with the reference's topology (bvh.rs:131-234): widest-centroid axis, STABLE sort by
// bbox.min[axis], full SAH sweep, leaf <= 8 unless SAH declines to split.  Produces an index-based tree
// (no pointer chasing) whose leaves are contiguous ranges of `order`, so `order` IS the in-order leaf
// sequence that decides equal-t ties in the reference traversal (bvh.rs:272-273, 308).
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#include "../../../include/rt.h"
#include "vec3.h"
#include "worker.h"

namespace rt {

struct BvhNode {
    Aabb box;
    int32_t left = -1, right = -1;  // inner: node indices; leaf: -1
    int32_t first = 0, count = 0;   // leaf: range in Bvh::order
    // A leaf of the reference that holds nested trees among its surfaces (rt_bvh_group) is laid out as a chain of
    // inner nodes over its runs of surfaces and its sub-trees (build_bvh); `gate` then names the chain's top node,
    // whose box is the reference leaf's (what a surface's gate is taken from, device_rt.h leaf_gate_hit); -1: its own.
    int32_t gate = -1;
    bool is_leaf() const { return left < 0; }
};

struct Bvh {
    // (OsBlocks: the render tree's arrays are made on a worker thread and live as long as the scene — see worker.h)
    std::vector<BvhNode, OsBlocks<BvhNode>> nodes;  // nodes[0] is the root
    std::vector<int32_t, OsBlocks<int32_t>> order;  // surface ids, in-order over the leaves
    uint32_t max_depth = 0;       // root = 1
    uint32_t max_leaf_size = 0;
    uint32_t n_leaves = 0;
    // With nested groups the figures above count as the reference's recursion does (a nested tree's nodes and
    // leaves are counted, its root one level below the leaf that holds it, a group one member of that leaf); the
    // node array itself is then deeper by the chains: `flat_depth` (0: same as max_depth).
    uint32_t flat_depth = 0;
    size_t ref_nodes = 0;  // nodes as the reference counts them (0: nodes.size())
};

// Where a build puts its nodes.  The top levels of both builders hand one child to another thread, and every task
// builds into a piece of its own; a piece that was built elsewhere is ATTACHED where a sequential build would have
// laid its nodes (self, left subtree, right subtree), i.e. it takes its range of indices, and its nodes are copied
// once, at the end, into the one array (lay_out, big pieces as tasks of their own).  Indices inside a piece count
// from the piece's first node.  (The pieces used to be copied into their parent at every join: a node of lucy's
// tree was copied four times, the last copies on one thread while the others had ended — 6 of the render tree's 19 ms.)
struct TreePiece {
    std::vector<BvhNode, OsBlocks<BvhNode>> own;  // the nodes this task made, in the order it made them
    std::vector<int32_t, OsBlocks<int32_t>> pos;  // own[i]'s index in the piece
    struct Attached {
        int32_t at;
        std::unique_ptr<TreePiece> piece;
    };
    std::vector<Attached> attached;
    int32_t size = 0;  // nodes of the piece, attached pieces included
    uint32_t max_depth = 0, max_leaf_size = 0, n_leaves = 0;
    TreePiece() = default;
    // a piece over n_items items: at most 2 n - 1 nodes, room for which is taken once (address space; pages come as they are written)
    explicit TreePiece(size_t n_items) { own.reserve(2 * n_items), pos.reserve(2 * n_items); }

    // the index of the new node; own.back() is the node (the caller keeps own.size() - 1 to come back to it)
    int32_t push(const BvhNode& n) {
        own.push_back(n), pos.push_back(size);
        return size++;
    }
    int32_t attach(std::unique_ptr<TreePiece> p) {
        const int32_t at = size;
        size += p->size;
        max_depth = std::max(max_depth, p->max_depth), max_leaf_size = std::max(max_leaf_size, p->max_leaf_size);
        n_leaves += p->n_leaves;
        attached.push_back({at, std::move(p)});
        return at;
    }
    void lay_out(BvhNode* dst, int32_t base) const;
    // the finished tree: nodes in `out.nodes` (root first), the figures in `out`
    void finish(Bvh& out) const;
};

// Bounding box of one surface: surfaces.rs:57-69 (sphere), 281-291 (triangle), 356-359 (quad).
Aabb surface_bounds(const rt_scene_desc& d, const rt_surface_ref& s);

// What the surface really occupies (the reference's boxes are inverted for negative radii and too small for
// parallelograms whose sides share an axis): the render tree is built over these.
Aabb true_bounds(const rt_scene_desc& d, const rt_surface_ref& s);

// use_bvh == false gives the single-leaf tree of Bvh::new_leaf (bvh.rs:63-68).
void build_bvh(const rt_scene_desc& d, Bvh& out);

}  // namespace rt

namespace rt {

// Render-path tree: the f32 kernels are free to use their own topology (only the f64 parity kernel and the
// equal-t tie ranks need the reference's).  Binned SAH (32 bins, all three axes), scale-invariant cost,
// leaves of at most `max_leaf` surfaces, built over the surface ids in `ids`.  `order` holds surface ids.
// `traversal_cost`: SAH cost of a node visit relative to one primitive test (BuildKnobs::traversal_cost).
void build_render_bvh(const rt_scene_desc& d, const std::vector<int32_t>& ids, uint32_t max_leaf, double traversal_cost, Bvh& out);
// The same builder on the GPU (hip/bvh_device.hip), on the current device.  bounds6: lo.xyz hi.xyz per primitive,
// f32, rounded outward; ids[i]: the surface id of primitive i.  Throws on a HIP error.
void build_render_bvh_device(const std::vector<float>& bounds6, const std::vector<int32_t>& ids, uint32_t max_leaf,
                             const BuildKnobs& knobs, Bvh& out);

}  // namespace rt
