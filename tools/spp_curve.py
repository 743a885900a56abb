"""Throughput against samples per pixel (fixed per-launch costs show up at low spp): tools/spp_curve.py scene..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
for name in sys.argv[1:]:
    s = ra.Scene(ra.load_scene(f"scenes/{name}.yaml", 16 / 9), 0)
    s.render(1920, 1080, 8)
    for sn in (2, 4, 8, 11, 16, 23, 32):
        ms = []
        for _ in range(3):
            img, st = s.render(1920, 1080, sn)
            ms.append(st.kernel_ms)
        best = min(ms)
        print(f"{name} sn={sn:2d} spp={sn*sn:4d}: {best:8.2f} ms  {st.samples/best/1e3:7.0f} Msamples/s   (runs {' '.join(f'{m:.2f}' for m in ms)})", flush=True)
    # strata slices as the 8-GPU run sees them: 1/8 of the 1024 strata
    for step in (2, 4, 8):
        ms = []
        for _ in range(3):
            img, st = s.render(1920, 1080, 32, stratum_begin=0, stratum_step=step)
            ms.append(st.kernel_ms)
        best = min(ms)
        print(f"{name} sn=32 every {step}th stratum: {best:8.2f} ms  {st.samples/best/1e3:7.0f} Msamples/s", flush=True)
