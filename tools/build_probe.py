"""Render-tree build on the host and on the GPU, side by side: build time, tree size, and what the render kernel
makes of each tree.   usage: python tools/build_probe.py [SCENE...]   (on the GPU box; RT_BUILD_DEBUG=1 for the
device builder's own phase times)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

import raytracer_amd as ra  # noqa: E402

w, h, sn = 1920, 1080, 16
for name in sys.argv[1:] or ["teapot", "dual", "lucy", "cornell_lucy"]:
    path = os.path.join(ROOT, "scenes", f"{name}.yaml")
    for how in (True, "device", True, "device"):
        s = ra.Scene(ra.load_scene(path, w / h, use_bvh=how), 0)
        i = s.info()
        fb = torch.empty((h, w, 3), dtype=torch.float32, device="cuda:0")
        ms = min(s.render_device(fb.data_ptr(), w, h, sn).kernel_ms for _ in range(3))
        print(f"{name:13s} {'device' if i.device_build else 'host  '} build: render tree {i.render_build_ms:7.2f} ms, scene create "
              f"{i.build_ms:7.2f} ms, {i.render_nodes} nodes depth {i.render_depth}; frame at {sn * sn} spp {ms:.2f} ms "
              f"({w * h * sn * sn / ms / 1e6:.3f} Gsamples/s)", flush=True)
        s.close()
