"""Sweep the persistent kernel's scheduling knobs (read with getenv at every launch) per scene, in one process.
usage: python tools/knob_sweep.py SCENE...   (on the GPU box)"""
import os, sys
os.environ["RT_TUNE"] = "1"  # the library reads its tuning knobs only then (hip/kernels.h tuning_env)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as ra
w, h, sn = 1920, 1080, 16
for name in sys.argv[1:]:
    s = ra.Scene(ra.load_scene(os.path.join(ROOT, "scenes", f"{name}.yaml"), w / h), 0)
    s.render(w, h, 4)
    for T in (16, 24, 32, 40, 48, 56):
        row = []
        for B in (1, 2, 3, 4, 6):
            os.environ["RT_SHADE_T"], os.environ["RT_NODE_BIAS"] = str(T), str(B)
            ms = min(s.render(w, h, sn)[1].kernel_ms for _ in range(3))
            row.append(f"B={B}: {w * h * sn * sn / ms / 1e3:6.0f}")
        print(f"{name} T={T}: " + "  ".join(row), flush=True)
    s.close()
