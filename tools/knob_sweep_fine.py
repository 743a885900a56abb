"""Shade threshold in steps of 4 at the default node bias (see knob_sweep.py).  usage: python tools/knob_sweep_fine.py SCENE..."""
import os, sys
os.environ["RT_TUNE"] = "1"  # the library reads its tuning knobs only then (hip/kernels.h tuning_env)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as ra
w, h, sn = 1920, 1080, 23
for name in sys.argv[1:]:
    s = ra.Scene(ra.load_scene(os.path.join(ROOT, "scenes", f"{name}.yaml"), w / h), 0)
    s.render(w, h, 4)
    row = []
    for T in (24, 28, 32, 36, 40, 44, 48, 52):
        os.environ["RT_SHADE_T"] = str(T)
        ms = min(s.render(w, h, sn)[1].kernel_ms for _ in range(3))
        row.append(f"T={T}: {w * h * sn * sn / ms / 1e3:6.0f}")
    print(name, "  ".join(row), flush=True)
    s.close()
