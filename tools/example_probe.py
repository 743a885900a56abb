"""Throughput of chosen --example scenes at 1200 px, 1024 spp (best of 3).  usage: example_probe.py name..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
for name in sys.argv[1:]:
    aspect = 1.0 if name in ("cornell", "cornell-smoke", "final-scene") else 1.5
    w = 1200
    h = int(w / aspect)
    s = ra.Scene(ra.example_scene(name, aspect), 0)
    s.render(w, h, 4)
    ms = min(s.render(w, h, 32)[1].kernel_ms for _ in range(3))
    print(f"{name}: {w * h * 1024 / ms / 1e3:.0f} Msamples/s", flush=True)
    s.close()
