"""Where does a small render tree start to beat testing every sphere and quad per ray ("loose")?  Scenes of N spheres
and quads (tests/synth.many_prims, no mesh) at 1080p 256 spp; run once per library build:
    RT_AMD_LIB=<build with RT_MAX_LOOSE=16> python tools/loose_crossover.py     (the tree, for N > 16)
    RT_AMD_LIB=<build with RT_MAX_LOOSE=64> python tools/loose_crossover.py     (loose up to 64)"""
import os, pathlib, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT), sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracer_amd as ra
import synth
w, h = 1920, 1080
for n in (int(x) for x in (sys.argv[1:] or ["20", "24", "32", "40", "48", "64"])):
    with tempfile.TemporaryDirectory() as tmp:
        path = synth.many_prims(pathlib.Path(tmp), n_spheres=n - 8, n_quads=5, with_mesh=os.environ.get("WITH_MESH") == "1")
        s = ra.Scene(ra.load_scene(path, w / h), 0)
    i = s.info()
    best = None
    for _ in range(2):
        _, st = s.render(w, h, 16)
        best = st if best is None or st.kernel_ms < best.kernel_ms else best
    print(f"N={n}: loose {i.n_loose} tree nodes {i.render_nodes}: {best.samples / best.kernel_ms / 1e3:.0f} Msamples/s", flush=True)
    s.close()
