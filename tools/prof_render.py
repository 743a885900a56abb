"""Profiling driver: one scene, one render call per repetition, through the C ABI.
usage: python tools/prof_render.py SCENE WIDTH HEIGHT SN KERNEL(0|1) [REPS]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra

name, w, h, sn, kernel = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 1
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = ra.load_scene(os.path.join(root, "scenes", f"{name}.yaml"), w / h)
s = ra.Scene(d, 0)
for _ in range(reps):
    img, st = s.render(w, h, sn, kernel=kernel)
    print(f"{name} {w}x{h} sn={sn} kernel={kernel}: {st.kernel_ms:.2f} ms, {st.samples / st.kernel_ms / 1e3:.1f} Msamples/s", flush=True)
