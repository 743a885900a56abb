"""The six-wave build (deep stack levels in memory) against the default build on the same scene: images must agree up to
f32 summation order.  usage: python tools/w6_check.py [scene] (spawns itself once per build)"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1] if len(sys.argv) > 1 else "lucy"
if len(sys.argv) > 2:
    import raytracer_amd as ra
    s = ra.Scene(ra.load_scene(os.path.join(ROOT, "scenes", f"{name}.yaml"), 16 / 9), 0)
    out = {}
    for stats in (False, True):
        for rep in range(2):
            img, st = s.render(240, 135, 4, collect_stats=stats)
            out[f"{int(stats)}{rep}"] = img
            print(sys.argv[2], "stats", stats, "rep", rep, "segments", st.segments, "nodes", st.node_visits, "sum", float(img.astype(np.float64).sum()), flush=True)
    np.savez(sys.argv[2], **out)
else:
    other = {"RT_NO_W6": "1"}
    if os.environ.get("W6_CHECK_OTHER_LIB"):  # e.g. a build with RT_LDS_STACK_CAP=32: the same kernel without overflow
        other = {"RT_AMD_LIB": os.environ["W6_CHECK_OTHER_LIB"], "RT_FORCE_W6": "1"}
    for tag, env in (("w6", {}), ("default", other)):
        subprocess.run([sys.executable, __file__, name, f"/tmp/w6check_{tag}.npz"], env=dict(os.environ, **env), check=True)
    a, b = np.load("/tmp/w6check_w6.npz"), np.load("/tmp/w6check_default.npz")
    for k in a.files:
        d = np.abs(a[k].astype(np.float64) - b[k]) > 1e-5 * np.maximum(np.abs(b[k]), 1.0) * 16
        print(k, "pixels differing w6 vs default:", int(d.any(-1).sum()), "max abs", float(np.abs(a[k] - b[k]).max()))
    d = np.abs(a["00"].astype(np.float64) - a["01"]) > 1e-4
    print("w6 rep0 vs rep1:", int(d.any(-1).sum()))
