// Host BVH build with the reference's topology (bvh.rs:131-234): widest-centroid axis, STABLE sort by
// bbox.min[axis], full SAH sweep, leaf <= 8 unless SAH declines to split.  Produces an index-based tree
// (no pointer chasing) whose leaves are contiguous ranges of `order`, so `order` IS the in-order leaf
// sequence that decides equal-t ties in the reference traversal (bvh.rs:272-273, 308).
#pragma once
#include <cstdint>
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
    std::vector<BvhNode> nodes;   // nodes[0] is the root
    std::vector<int32_t> order;   // surface ids, in-order over the leaves
    uint32_t max_depth = 0;       // root = 1
    uint32_t max_leaf_size = 0;
    uint32_t n_leaves = 0;
    // With nested groups the figures above count as the reference's recursion does (a nested tree's nodes and
    // leaves are counted, its root one level below the leaf that holds it, a group one member of that leaf); the
    // node array itself is then deeper by the chains: `flat_depth` (0: same as max_depth).
    uint32_t flat_depth = 0;
    size_t ref_nodes = 0;  // nodes as the reference counts them (0: nodes.size())
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
