// Host-side scene ownership: the vectors behind an rt_scene_desc, plus the loader entry point.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../../include/rt.h"
#include "vec3.h"

namespace rt {

// resize() of a vector with this allocator leaves new elements of a plain C struct as they are (no zero fill): a
// mesh's triangles are then written once, by the threads that fill them, instead of twice with the first on one thread.
template <class T>
struct DefaultInit : std::allocator<T> {
    template <class U>
    struct rebind {
        using other = DefaultInit<U>;
    };
    using std::allocator<T>::allocator;
    template <class U>
    void construct(U* p) noexcept {
        ::new ((void*)p) U;
    }
    template <class U, class... A>
    void construct(U* p, A&&... a) {
        ::new ((void*)p) U(std::forward<A>(a)...);
    }
};

// Owns the arrays an rt_scene_desc points at.  `desc` must stay the first member: rt_load_scene hands
// out &desc and rt_scene_desc_free casts it back.
struct HostScene {
    rt_scene_desc desc{};
    std::vector<rt_material> materials;
    std::vector<rt_sphere> spheres;
    std::vector<rt_triangle, DefaultInit<rt_triangle>> triangles;
    std::vector<rt_quad> quads;
    std::vector<rt_surface_ref, DefaultInit<rt_surface_ref>> surfaces;
    std::vector<rt_medium> media;
    std::vector<rt_perlin> perlin;
    std::vector<rt_image> images;
    std::vector<std::vector<uint8_t>> image_data;  // pixels behind `images`
    // nesting of the reference-topology tree (rt_bvh_group): empty for scene files
    std::vector<rt_bvh_group> groups;
    std::vector<int32_t> surface_group;      // per surface once any group exists (kept in step by the example generators)
    std::vector<double> surface_build_box;   // 6 per surface, or empty
    void sync_desc();  // refresh desc pointers/counts after the vectors changed
};

// load.rs:39-160.  Throws std::runtime_error with an anyhow-like "context: cause" message.
void load_scene_file(const std::string& yaml_path, double aspect_ratio, bool use_bvh, HostScene& out);

// scene/example.rs:69-79 (one-weekend, cornell).  Throws std::runtime_error for the others.
void example_scene(const std::string& name, double aspect_ratio, uint64_t seed, bool use_bvh, HostScene& out);

// camera.rs:43-74
rt_camera build_camera(double vfov_degrees, double aspect_ratio, V3 from, V3 towards, V3 vup, double aperture,
                       bool has_focus, double focus_dist);

// surfaces.rs:313-327
rt_quad make_quad(V3 q, V3 u, V3 v, int material);

struct ObjMesh {
    std::string name;
    std::vector<float> positions;  // xyz per vertex, f32 as tobj yields them (load.rs:258-261)
    std::vector<float> normals;    // xyz per normal, may be empty
    std::vector<uint32_t> indices; // 3 per triangle
};
// First model of an OBJ file, gunzipped iff the extension is "gz" (load.rs:202-229, 315-325).
void load_obj_first_model(const std::string& path, ObjMesh& out);

}  // namespace rt
