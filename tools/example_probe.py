"""Stage statistics of an --example scene: tools/example_probe.py NAME [width height sn]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
name = sys.argv[1]
w, h, sn = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (800, 800, 8)
s = ra.Scene(ra.example_scene(name, w / h), 0)
i = s.info()
print(name, "nodes", i.render_nodes, "depth", i.render_depth, "loose", i.n_loose, flush=True)
img, st = s.render(w, h, sn, collect_stats=True)
print(f"  seg/sample {st.segments/st.samples:.2f} nodes/seg {st.node_visits/st.segments:.2f} tris/seg {st.tri_tests/st.segments:.2f} sph/seg {st.sphere_tests/st.segments:.2f} quad/seg {st.quad_tests/st.segments:.2f}", flush=True)
for k in (sn, 2 * sn):
    ms = min(s.render(w, h, k)[1].kernel_ms for _ in range(2))
    print(f"  sn={k}: {ms:.1f} ms {w*h*k*k/ms/1e3:.0f} Msamples/s", flush=True)
