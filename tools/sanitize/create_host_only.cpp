// Sanitizer driver: the host-only scene path (load + both tree builds + both flattenings, RT_DEVICE_NONE) of
// one scene file, several times in one process.  Built and run by tools/sanitize/run.sh under
// ThreadSanitizer and AddressSanitizer in the build container (no GPU is touched).
#include <cstdio>
#include <cstdlib>

#include "../../include/rt.h"

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "scenes/lucy.yaml";
    const int rounds = argc > 2 ? atoi(argv[2]) : 3;
    for (int r = 0; r < rounds; ++r) {
        rt_scene_desc* desc = nullptr;
        if (rt_load_scene(path, 16.0 / 9.0, 1, &desc)) {
            fprintf(stderr, "load: %s\n", rt_last_error());
            return 1;
        }
        rt_scene* scene = nullptr;
        if (rt_scene_create(desc, RT_DEVICE_NONE, &scene)) {
            fprintf(stderr, "create: %s\n", rt_last_error());
            return 1;
        }
        rt_scene_info info;
        rt_scene_get_info(scene, &info);
        printf("round %d: %llu surfaces, reference tree depth %u, render tree depth %u, %.1f ms\n", r,
               (unsigned long long)info.n_surfaces, info.max_depth, info.render_depth, info.build_ms);
        rt_scene_destroy(scene);
        rt_scene_desc_free(desc);
    }
    return 0;
}
