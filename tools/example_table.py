"""Throughput of the eight --example scenes: 1200 pixels wide, 256 and 1024 spp."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
print("| example | size | surfaces | render-tree nodes | seg/sample | Msamples/s at 256 spp | at 1024 spp |")
print("|---|---|---|---|---|---|---|")
for name in ["one-weekend", "random-spheres", "two-spheres", "two-perlin", "earth", "cornell", "cornell-smoke", "final-scene"]:
    aspect = 1.0 if name in ("cornell", "cornell-smoke", "final-scene") else 1.5
    w = 1200
    h = int(w / aspect)
    s = ra.Scene(ra.example_scene(name, aspect), 0)
    i = s.info()
    _, c = s.render(w, h, 4, collect_stats=True)
    s.render(w, h, 4)
    r = []
    for sn in (16, 32):
        ms = min(s.render(w, h, sn)[1].kernel_ms for _ in range(2))
        r.append(w * h * sn * sn / ms / 1e3)
    print(f"| {name} | {w}x{h} | {i.n_surfaces} | {i.render_nodes} | {c.segments / c.samples:.2f} | {r[0]:.0f} | {r[1]:.0f} |", flush=True)
    s.close()
