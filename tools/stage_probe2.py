import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
name, sn, sampler = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
d = ra.load_scene(f"scenes/{name}.yaml", 16 / 9)
s = ra.Scene(d, 0)
img, st = s.render(1920, 1080, sn, collect_stats=True, sampler=sampler)
print(f"{name} sn={sn} sampler={sampler} STATS build: {st.kernel_ms:.1f} ms seg {st.segments} nodes {st.node_visits} tris {st.tri_tests}", flush=True)
