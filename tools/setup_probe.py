"""Where scene set-up time goes: OBJ/YAML load, tree builds + flatten (info.build_ms), device context + code-object
load (first call only), upload.   usage: python tools/setup_probe.py [SCENE...]   (on the GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
import torch  # noqa: E402
t1 = time.perf_counter()
import raytracer_amd as ra  # noqa: E402
ra.lib()
t2 = time.perf_counter()
print(f"import torch {t1 - t0:.3f} s, load librt_amd.so {t2 - t1:.3f} s", flush=True)
if "--torch-first" in sys.argv:
    t = time.perf_counter()
    torch.zeros(1, device="cuda:0")
    torch.cuda.synchronize()
    print(f"torch context {time.perf_counter() - t:.3f} s", flush=True)
w, h = 1920, 1080
for name in [a for a in sys.argv[1:] if not a.startswith("--")] or ["lucy", "lucy", "teapot"]:
    t = time.perf_counter()
    d = ra.load_scene(os.path.join(ROOT, "scenes", f"{name}.yaml"), w / h)
    tl = time.perf_counter()
    s = ra.Scene(d, 0)
    tc = time.perf_counter()
    fb = torch.empty((h, w, 3), dtype=torch.float32, device="cuda:0")
    s.render_device(fb.data_ptr(), w, h, 1)
    torch.cuda.synchronize()
    tr = time.perf_counter()
    i = s.info()
    print(f"{name}: load {1e3 * (tl - t):.1f} ms, scene create {1e3 * (tc - tl):.1f} ms (of which builds+flatten {i.build_ms:.1f}, "
          f"render tree {i.render_build_ms:.1f}), first 1-spp frame {1e3 * (tr - tc):.1f} ms", flush=True)
    s.close()
