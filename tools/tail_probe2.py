"""Back-to-back launches of a strata slice into a resident framebuffer (no host gaps between them)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as ra
name, step = sys.argv[1], int(sys.argv[2])
s = ra.Scene(ra.load_scene(f"scenes/{name}.yaml", 16 / 9), 0)
fb = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms = [s.render_device(fb.data_ptr(), 1920, 1080, 32, stratum_begin=0, stratum_step=step, stream=stream).kernel_ms for _ in range(12)]
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    print(name, f"1/{step}", "kernel_ms:", " ".join(f"{m:.2f}" for m in ms), f"| wall/launch {wall/12:.2f} ms", flush=True)
