"""What one of N ranks does per frame, on one GPU: its 1/N strata slice of the frame, frames back to back
 (a) on one stream, each awaited (round 1's pipeline), (b) on two streams in turn without host synchronisation
 (rt_render_device's two scratch sets: the next frame's render fills the CUs the draining frame leaves idle).
usage: python tools/overlap_probe.py [--slice 8] [--frames 12] [--tail 0,1,4] SCENE...   (on the GPU box)"""
import argparse
import os
os.environ["RT_TUNE"] = "1"  # the library reads its tuning knobs only then (hip/kernels.h tuning_env)
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import raytracer_amd as ra  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("scenes", nargs="+")
ap.add_argument("--slice", type=int, default=8)
ap.add_argument("--frames", type=int, default=12)
ap.add_argument("--streams", type=int, default=2, help="renders kept in flight (the library holds that many scratch sets)")
ap.add_argument("--tail", default="", help="comma list of RT_TAIL_K values to sweep (default: the library's own)")
a = ap.parse_args()
w, h, sn = 1920, 1080, 32


def probe(s, name, tail):
    if tail is not None:
        os.environ["RT_TAIL_K"] = tail
    fbs = [torch.empty((h, w, 3), dtype=torch.float32, device="cuda:0") for _ in range(a.streams)]
    streams = [torch.cuda.Stream() for _ in range(a.streams)]
    full = min(s.render_device(fbs[0].data_ptr(), w, h, sn).kernel_ms for _ in range(2))
    one = min(s.render_device(fbs[0].data_ptr(), w, h, sn, stratum_step=a.slice).kernel_ms for _ in range(3))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.frames):
        s.render_device(fbs[0].data_ptr(), w, h, sn, stratum_step=a.slice)
    torch.cuda.synchronize()
    serial = (time.perf_counter() - t0) / a.frames * 1e3
    t0 = time.perf_counter()
    for i in range(a.frames):
        s.render_device(fbs[i % a.streams].data_ptr(), w, h, sn, stratum_step=a.slice, stream=streams[i % a.streams].cuda_stream,
                        want_stats=False)
    torch.cuda.synchronize()
    overlapped = (time.perf_counter() - t0) / a.frames * 1e3
    print(f"{name} tail {tail}: full frame {full:.2f} ms; slice kernel {one:.2f} ms; per frame awaited {serial:.2f} ms, "
          f"{a.streams} streams {overlapped:.2f} ms", flush=True)


for name in a.scenes:
    scene = ra.Scene(ra.load_scene(os.path.join(ROOT, "scenes", f"{name}.yaml"), w / h), 0)
    for tail in (a.tail.split(",") if a.tail else [None]):
        probe(scene, name, tail)
    scene.close()
