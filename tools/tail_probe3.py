import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
name, step = sys.argv[1], int(sys.argv[2])
s = ra.Scene(ra.load_scene(f"scenes/{name}.yaml", 16 / 9), 0)
s.render(1920, 1080, 8)
w, h = (int(a) for a in os.environ.get("RT_WH", "1920x1080").split("x"))
img, st = s.render(w, h, 32, stratum_begin=0, stratum_step=step, collect_stats=True)
print(name, step, f"{st.kernel_ms:.1f} ms (counting build)")
