"""One probe for kernel work (replaces the round-1 one-off probes).  Run on the GPU box:
    python tools/perf_probe.py [--stages] [--sn 16,32] [--kernel 0] [--size 1920x1080] SCENE...
Per scene: tree/loose info, per-segment work counts from the counting build, the stage split of the counting
build (--stages: RT_STAGE_DEBUG output on stderr), and throughput at each --sn."""
import argparse, os, sys
os.environ["RT_TUNE"] = "1"  # the library reads its tuning knobs only then (hip/kernels.h tuning_env)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("scenes", nargs="+")
ap.add_argument("--stages", action="store_true")
ap.add_argument("--sn", default="16,32")
ap.add_argument("--kernel", type=int, default=0)
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--sampler", type=int, default=1)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--stats-sn", type=int, default=4, help="sn of the counting build's frame (the stage split is that frame's)")
ap.add_argument("--slice", type=int, default=1, help="render strata k = 0 mod SLICE (the share of one of SLICE GPUs)")
a = ap.parse_args()
if a.stages:
    os.environ["RT_STAGE_DEBUG"] = "1"
import raytracer_amd as ra
w, h = (int(x) for x in a.size.split("x"))
for name in a.scenes:
    ex = name.startswith("ex:")
    d = ra.example_scene(name[3:], w / h, seed=11) if ex else ra.load_scene(os.path.join(ROOT, "scenes", f"{name}.yaml"), w / h)
    s = ra.Scene(d, 0)
    i = s.info()
    print(f"{name}: render tree nodes {i.render_nodes} depth {i.render_depth} maxleaf {i.render_max_leaf} loose {i.n_loose} "
          f"setup {i.build_ms:.0f} ms", flush=True)
    _, st = s.render(w, h, a.stats_sn, collect_stats=True, kernel=a.kernel, sampler=a.sampler)
    print(f"  seg/sample {st.segments / st.samples:.2f} nodes/seg {st.node_visits / st.segments:.2f} "
          f"tris/seg {st.tri_tests / st.segments:.2f} sph/seg {st.sphere_tests / st.segments:.2f} "
          f"quad/seg {st.quad_tests / st.segments:.2f}", flush=True)
    os.environ.pop("RT_STAGE_DEBUG", None)
    for sn in (int(x) for x in a.sn.split(",")):
        best = None
        for _ in range(a.reps):
            _, st = s.render(w, h, sn, kernel=a.kernel, sampler=a.sampler, stratum_step=a.slice)
            best = st if best is None or st.kernel_ms < best.kernel_ms else best
        print(f"  sn={sn} slice 1/{a.slice}: {best.kernel_ms:.2f} ms  {best.samples / best.kernel_ms / 1e3:.0f} Msamples/s", flush=True)
    if a.stages:
        os.environ["RT_STAGE_DEBUG"] = "1"
    s.close()
