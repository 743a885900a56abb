"""Launch-tail probe: the 1/8 strata slice an 8-GPU run gives each rank, against the full frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
for name in sys.argv[1:]:
    s = ra.Scene(ra.load_scene(f"scenes/{name}.yaml", 16 / 9), 0)
    s.render(1920, 1080, 8)
    out = []
    for step in (1, 4, 8):
        ms = min(s.render(1920, 1080, 32, stratum_begin=0, stratum_step=step)[1].kernel_ms for _ in range(3))
        out.append(f"1/{step}: {ms:7.2f} ms {1920*1080*1024/step/ms/1e3:6.0f} Ms/s")
    ms = min(s.render(1920, 1080, 8)[1].kernel_ms for _ in range(3))
    out.append(f"64spp: {ms:6.2f} ms")
    print(name, os.environ.get("RT_TAIL_K", "-"), " | ".join(out), flush=True)
