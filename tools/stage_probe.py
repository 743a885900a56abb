import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
for name in sys.argv[1:]:
    d = ra.load_scene(f"scenes/{name}.yaml", 16 / 9)
    s = ra.Scene(d, 0)
    i = s.info()
    print(name, "render tree nodes", i.render_nodes, "depth", i.render_depth, "maxleaf", i.render_max_leaf, "loose", i.n_loose, flush=True)
    img, st = s.render(1920, 1080, 4, collect_stats=True)
    print(f"  seg/sample {st.segments/st.samples:.2f} nodes/seg {st.node_visits/st.segments:.2f} tris/seg {st.tri_tests/st.segments:.2f} sph/seg {st.sphere_tests/st.segments:.2f} quad/seg {st.quad_tests/st.segments:.2f}", flush=True)
    for sn in (8, 16):
        img, st = s.render(1920, 1080, sn)
        print(f"  sn={sn}: {st.kernel_ms:.1f} ms {st.samples/st.kernel_ms/1e3:.0f} Msamples/s", flush=True)
