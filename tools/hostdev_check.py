"""Host-built against device-built render tree on one scene: pixels that differ beyond f32 summation order, per kernel."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as ra
name = sys.argv[1] if len(sys.argv) > 1 else "lucy"
path = os.path.join(ROOT, "scenes", f"{name}.yaml")
host = ra.Scene(ra.load_scene(path, 16 / 9), 0)
dev = ra.Scene(ra.load_scene(path, 16 / 9, use_bvh="device"), 0)
for kernel in (0, 1, 2):
    a, sa = host.render(240, 135, 4, kernel=kernel, collect_stats=True)
    b, sb = dev.render(240, 135, 4, kernel=kernel, collect_stats=True)
    a, b = a.astype(np.float64) / 16, b.astype(np.float64) / 16
    d = np.any(np.abs(a - b) > 1e-5 * np.maximum(np.abs(a), 1.0), axis=-1)
    print(f"kernel {kernel}: {int(d.sum())} of {d.size} pixels differ ({d.mean():.5f}); segments {sa.segments} vs {sb.segments}; "
          f"nodes {sa.node_visits} vs {sb.node_visits}; tris {sa.tri_tests} vs {sb.tri_tests}")
