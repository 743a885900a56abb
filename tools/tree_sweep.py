"""Sweep the render tree's build parameters (read with getenv at scene creation): leaf size RT_LEAF and the SAH
traversal cost RT_CT.   usage: python tools/tree_sweep.py SCENE...   (on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as ra
w, h, sn = 1920, 1080, 16
for name in sys.argv[1:]:
    for leaf in (1, 2, 3, 4):
        row = []
        for ct in (0.25, 0.5, 1.0, 2.0):
            os.environ["RT_LEAF"], os.environ["RT_CT"] = str(leaf), str(ct)
            s = ra.Scene(ra.load_scene(os.path.join(ROOT, "scenes", f"{name}.yaml"), w / h), 0)
            s.render(w, h, 4)
            ms = min(s.render(w, h, sn)[1].kernel_ms for _ in range(3))
            row.append(f"ct={ct}: {w * h * sn * sn / ms / 1e3:6.0f} ({s.info().render_nodes} n)")
            s.close()
        print(f"{name} leaf={leaf}: " + "  ".join(row), flush=True)
