"""Per-scene throughput table at BASELINE resolution: device-resident rate, host-buffer
(PCIe-inclusive) rate and per-segment work counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra

w, h, sn = 1920, 1080, 32
print("| scene | Msamples/s (device) | s/frame | Msamples/s (host buffer) | seg/sample | nodes/seg | tris/seg | loose/seg | Grays/s |")
print("|---|---|---|---|---|---|---|---|---|")
for name in sys.argv[1:] or ["lucy", "teapot", "dual", "cornell", "cornell_lucy", "distant", "light", "quads"]:
    s = ra.Scene(ra.load_scene(f"scenes/{name}.yaml", w / h), 0)
    _, c = s.render(w, h, 4, collect_stats=True)
    s.render(w, h, 4)
    st, wall = None, 0.0
    for _ in range(3):  # the best of three frames (boxes differ by a few per cent from run to run)
        t0 = time.perf_counter()
        _, one = s.render(w, h, sn)
        t1 = time.perf_counter()
        if st is None or one.kernel_ms < st.kernel_ms:
            st, wall = one, t1 - t0
    seg = c.segments / c.samples
    print(f"| {name} | {st.samples / st.kernel_ms / 1e3:.0f} | {st.kernel_ms / 1e3:.3f} | {st.samples / wall / 1e6:.0f} | {seg:.2f} | "
          f"{c.node_visits / c.segments:.2f} | {c.tri_tests / c.segments:.2f} | {(c.sphere_tests + c.quad_tests) / c.segments:.2f} | "
          f"{seg * st.samples / st.kernel_ms / 1e6:.1f} |", flush=True)
    s.close()
