"""Largest per-pixel deviations of a 16-spp render from the oracle's golden image.  usage: golden_diff.py scene kernel sampler"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as ra
name, kernel, sampler = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
g = np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))
want = (g["image_sum"] if sampler == 0 else g["image_sum_direct"]) / 16
ih, iw, _ = want.shape
s = ra.Scene(ra.load_scene(os.path.join(ROOT, "scenes", f"{name}.yaml"), iw / ih), 0)
img, st = s.render(iw, ih, 4, max_depth=40, seed=1, kernel=kernel, sampler=sampler, collect_stats=True)
got = img.astype(np.float64) / 16
dev = np.abs(got - want).max(-1)
order = np.argsort(dev.ravel())[::-1][:12]
print(name, iw, ih, "rmse", np.sqrt(np.mean((got - want) ** 2, axis=(0, 1))), "segments", st.segments, "golden", int(g["segments"]))
for o in order:
    y, x = divmod(int(o), iw)
    print(f"  pixel ({y},{x}): got {got[y, x].round(4)} want {want[y, x].round(4)}")
print("pixels with dev > 0.01:", int((dev > 0.01).sum()), " > 0.001:", int((dev > 0.001).sum()))
