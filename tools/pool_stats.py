import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
kernel = int(os.environ.get("KERNEL", "2"))
for name in sys.argv[1:]:
    d = ra.load_scene(f"scenes/{name}.yaml", 16 / 9)
    s = ra.Scene(d, 0)
    print(name, "kernel", kernel, flush=True)
    img, st = s.render(1920, 1080, 6, kernel=kernel, collect_stats=True)
    print(f"  stats build {st.kernel_ms:.1f} ms", flush=True)
    for sn in (8, 16):
        t = s.render(1920, 1080, sn, kernel=kernel)[1]
        print(f"  sn={sn}: {t.kernel_ms:.1f} ms {t.samples/t.kernel_ms/1e3:.0f} Ms/s", flush=True)
