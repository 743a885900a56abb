"""Pool kernel vs persistent kernel: same image (up to f32 sum order) and timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra

names = sys.argv[1:] or ["lucy", "teapot", "cornell", "dual"]
for name in names:
    d = ra.load_scene(f"scenes/{name}.yaml", 16 / 9)
    s = ra.Scene(d, 0)
    a, sa = s.render(480, 270, 4, kernel=0, collect_stats=True)
    b, sb = s.render(480, 270, 4, kernel=2, collect_stats=True)
    an, bn = a.astype(np.float64) / 16, b.astype(np.float64) / 16
    differs = np.any(np.abs(an - bn) > 1e-5 * np.maximum(np.abs(an), 1.0), axis=-1)
    print(f"{name}: differing pixels {differs.mean():.2e} mean {an.mean():.6f} vs {bn.mean():.6f}; "
          f"segments {sa.segments} vs {sb.segments} nodes {sa.node_visits} vs {sb.node_visits}", flush=True)
    for sn in (8, 16):
        t0 = s.render(1920, 1080, sn, kernel=0)[1]
        t2 = s.render(1920, 1080, sn, kernel=2)[1]
        print(f"  sn={sn}: persistent {t0.kernel_ms:.1f} ms {t0.samples/t0.kernel_ms/1e3:.0f} Ms/s | pool {t2.kernel_ms:.1f} ms "
              f"{t2.samples/t2.kernel_ms/1e3:.0f} Ms/s", flush=True)
