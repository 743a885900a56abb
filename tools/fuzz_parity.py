"""Differential run of the HIP path against the oracle over random scene files (the generators of tests/synth.py
with random counts, seeds, cameras — thin lens included — and frame sizes, odd ones among them).  Per scene:
  * first hits of the f64 parity kernel: ids and t equal to the oracle's, bit for bit;
  * first hits of the render kernels' f32 arithmetic: the swap / flip / dt budget of tests/test_gpu_edge.py;
  * images of the three kernel organisations, both samplers, same seeds, 1 to 16 spp: the pixels whose paths parted
    ways between f32 and f64 are counted against the frame's paths, the others must agree to rounding, the mean
    luminance as a whole (bars below; tests/test_gpu_parity.py's RMSE bar is for 16 spp and more).
Prints one line per scene and a summary; exit status 1 if any bar is missed.  The oracle is the checker here, as in
tests/; nothing of it is on the product's path.
usage (GPU box): python tools/fuzz_parity.py [--scenes N] [--seed S]"""
import argparse, os, pathlib, re, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (maps the HIP runtime first: raytracer_amd/api.py)
import raytracer_amd as ra  # noqa: E402
from oracle import ref  # noqa: E402
import synth  # noqa: E402
from test_gpu_parity import classify_f32  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=60)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--examples", action="store_true", help="the eight example generators (scene/example.rs) with random seeds instead of scene files")
ap.add_argument("--meshes", action="store_true", help="scene files around a random mesh of 2 000 to 150 000 triangles")
ap.add_argument("--extras", action="store_true", help="every third scene also: strata slices, device-built tree, no tree")
ap.add_argument("--only", type=int, default=-1, help="compare this scene of the sequence only (the others are drawn and skipped)")
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
ref.lib()


def random_camera(text):
    """The generator's fixed camera replaced by a random one looking at the scene's middle, one in three with a lens."""
    frm = [rng.uniform(-5, 5), rng.uniform(0.6, 4.5), rng.uniform(5, 9.5)]
    to = [rng.uniform(-1, 1), rng.uniform(0.2, 1.0), rng.uniform(-1, 1)]
    cam = f"camera:\n  fov: {rng.uniform(25, 70):.2f}\n  from: [{frm[0]:.3f}, {frm[1]:.3f}, {frm[2]:.3f}]\n  towards: [{to[0]:.3f}, {to[1]:.3f}, {to[2]:.3f}]\n"
    if rng.integers(3) == 0:
        cam += f"  aperture: {rng.uniform(0.02, 0.4):.3f}\n  focus_distance: {rng.uniform(4, 11):.3f}\n"
    return re.sub(r"^camera:\n(?:  .*\n)+", cam, text, count=1, flags=re.M)


def big_mesh(tmp, seed):
    """A lumpy globe of 2 000 to 150 000 triangles (the larger ones take the set-up's parser blocks, forked subtrees and
    the root's team sort and partition) with zero-area triangles at its poles and a seam of doubled
    vertices, gzipped or not, under a random chain of the loader's transforms, over a ground sphere and beside a
    light: both trees, the loader and the triangle test at a mesh's scale, on data no shipped scene has."""
    import gzip
    g = np.random.default_rng(seed)
    n_lat = int(g.integers(24, 200))
    n_lon = int(g.integers(40, 380))
    lat = np.linspace(0.0, np.pi, n_lat + 1)[:, None]
    lon = np.linspace(0.0, 2 * np.pi, n_lon + 1)[None, :]  # (the last column doubles the first: a seam)
    bumps = sum(a * np.sin(fa * lat + pa) * np.cos(fo * lon + po)
                for a, fa, fo, pa, po in zip(g.uniform(0, 0.12, 4), g.integers(1, 9, 4), g.integers(1, 9, 4),
                                             g.uniform(0, 6, 4), g.uniform(0, 6, 4)))
    rad = 1.0 + bumps
    pts = np.stack([rad * np.sin(lat) * np.cos(lon), rad * np.cos(lat) + 0 * lon, rad * np.sin(lat) * np.sin(lon)], -1)
    lines = ["# lumpy globe", "o globe"] + ["v %.7g %.7g %.7g" % tuple(p) for p in pts.reshape(-1, 3)]
    idx = lambda i, j: i * (n_lon + 1) + j + 1
    for i in range(n_lat):
        for j in range(n_lon):
            a, b, c, d_ = idx(i, j), idx(i, j + 1), idx(i + 1, j + 1), idx(i + 1, j)
            lines.append(f"f {a} {b} {c}")
            lines.append(f"f {a} {c} {d_}")
    text = "\n".join(lines) + "\n"
    name = "globe.obj.gz" if g.integers(2) else "globe.obj"
    (tmp / name).write_bytes(gzip.compress(text.encode(), 3) if name.endswith(".gz") else text.encode())
    chain = []
    for _ in range(int(g.integers(0, 4))):
        kind = g.integers(3)
        if kind == 0:
            chain.append(f"        - {{type: rotate, axis: {'xyz'[g.integers(3)]}, angle: {g.uniform(-180, 180):.2f}}}\n")
        elif kind == 1:
            chain.append(f"        - {{type: scale, factor: {g.uniform(0.4, 1.6):.3f}}}\n")
        else:
            chain.append(f"        - {{type: translate, dir: [{g.uniform(-1, 1):.3f}, {g.uniform(0, 1):.3f}, {g.uniform(-1, 1):.3f}]}}\n")
    s = [synth.CAM, "scene:\n  background: [0.5, 0.6, 0.8]\n  surfaces:\n",
         "    - type: sphere\n      radius: 100\n      position: [0, -101.8, 0]\n      material: {type: lambertian, albedo: [0.5, 0.5, 0.5]}\n",
         f"    - type: mesh\n      path: {name}\n      material: {synth._mat(g)}\n",
         ("      transform:\n" + "".join(chain)) if chain else "",
         "    - type: quad\n      position: [-1, 4, -1]\n      sides: [[2, 0, 0], [0, 0, 2]]\n      material: {type: diffuseLight, albedo: [4, 4, 4]}\n",
         f"    - type: sphere\n      radius: 0.5\n      position: [2.4, -1.3, 1.0]\n      material: {synth._mat(g)}\n"]
    path = tmp / "globe.yaml"
    path.write_text("".join(s))
    return str(path)


def make_scene(tmp, k):
    if args.meshes:
        seed = int(rng.integers(1 << 30))
        path = big_mesh(tmp, seed)
        text = random_camera(open(path).read())
        open(path, "w").write(text)
        return "big_mesh", seed, path
    kind = ["many_prims", "many_prims_no_mesh", "spheres_only", "room", "many_quads", "few_loose"][k % 6]
    seed = int(rng.integers(1 << 30))
    if kind == "many_prims":
        path = synth.many_prims(tmp, n_spheres=int(rng.integers(12, 60)), n_quads=int(rng.integers(1, 12)), seed=seed)
    elif kind == "many_prims_no_mesh":
        path = synth.many_prims(tmp, n_spheres=int(rng.integers(28, 70)), n_quads=int(rng.integers(3, 10)), with_mesh=False, seed=seed)
    elif kind == "spheres_only":
        path = synth.many_prims(tmp, n_spheres=int(rng.integers(34, 90)), n_quads=0, with_mesh=False, seed=seed)
    elif kind == "room":
        path = synth.room(tmp, n_spheres=int(rng.integers(4, 40)), seed=seed)
    elif kind == "many_quads":
        path = synth.many_quads(tmp, n=int(rng.integers(31, 90)), seed=seed)
    else:  # everything loose: no render tree at all, or loose primitives beside a small mesh
        path = synth.many_prims(tmp, n_spheres=int(rng.integers(1, 9)), n_quads=int(rng.integers(1, 5)), with_mesh=bool(rng.integers(2)), seed=seed)
    text = open(path).read()
    if kind != "room":  # (the room's camera has to stay inside it)
        text = random_camera(text)
    open(path, "w").write(text)
    return kind, seed, path


# Largest relative dt of an f32 first hit on the same surface.  tests/ hold the shipped scenes to 2e-4; a random camera
# finds the radius-100 ground at its horizon, where the ray's f32 direction (the render kernels' representation) moves
# a grazing root by more: 2.9e-4 at the worst pixel of 2000 scenes (seed 7, scene 1878, pixel (33, 5)).
F32_DT_MAX = 1e-3
SAME_PATH = 2e-3        # largest difference of a pixel whose paths agree (radiance, averaged over the pixel's samples)
SAME_PATH_RMSE = 2e-4   # per-channel RMSE over those pixels
PARTED_FRACTION = 2e-3  # paths of a frame that may part ways (plus three)
LUM_BAR = 6e-4          # relative mean luminance over those pixels (tests/test_gpu_parity.py's bar)
bad = 0
parted_total = paths_total = caught_total = 0
LUM_ALL_BAR = 2e-2      # relative mean luminance of the whole frame, moved pixels included (a few thousand paths: noise of the movers)
# A glass mesh (--meshes: one scene in four) sends a path through dozens of refractions and inner reflections off
# small facets: there 0.3-1.3 % of the paths part ways, the others still agree to 1e-8.
PARTED_FRACTION_MESH = 2e-2
worst = {"rmse": 0.0, "lum": 0.0, "lum_all": 0.0, "swaps": 0, "flips": 0, "rel": 0.0, "parted": 0.0}
t_start = time.time()
def compare(k, label, desc, scene, oracle, w, h, sn, flips_allowed, dt_max, dt_allowed, parted_bar=None):
    """One scene on both sides; a line of output; the summary's counters."""
    global bad, parted_total, paths_total, caught_total
    parted_bar = PARTED_FRACTION if parted_bar is None else parted_bar
    notes = []
    rid, rt = oracle.primary_hits(w, h, threads=16)
    ids, t = scene.primary_hits(w, h, precision=64)
    if not (np.array_equal(ids, rid) and np.array_equal(t, rt)):
        notes.append(f"f64 first hits differ: {int((ids != rid).sum())} ids, {int((t != rt).sum())} t")
    ids32, t32 = scene.primary_hits(w, h, precision=32)
    # The f32 triangle test takes hits within 2^-20 of an edge (device_rt.h: no ray slips between two triangles of a
    # mesh); the reference's test, and so the oracle's, loses a ray that runs exactly along a shared edge — the
    # jitter-free probe's centre column lies in the plane x = 0 exactly, and so does an edge of tests/synth.py's
    # room mesh.  Such pixels (f32: a triangle, nearer than what f64 finds behind the crack) are counted apart.
    is_tri = np.concatenate([desc.surfaces["type"] == 1, [False]])  # (id -1: no hit)
    behind = np.where(rid >= 0, rt, np.inf)
    caught = (ids32 != rid) & is_tri[ids32] & (t32 < behind * (1.0 - 1e-5))  # (nearer by more than a tie's rounding)
    caught_total += int(caught.sum())
    ids32 = np.where(caught, rid, ids32)
    t32 = np.where(caught, rt, t32)
    swaps, flips, n_rel, max_rel = classify_f32(ids32, t32, rid, rt)
    if flips > flips_allowed or swaps > max(8, 5e-3 * rid.size, flips_allowed) or n_rel > dt_allowed or max_rel > dt_max:
        notes.append(f"f32 first hits: swaps {swaps} flips {flips} n_rel {n_rel} max_rel {max_rel:.2e}")
        diff = np.argwhere(ids32 != rid)
        notes.append("ids differ at " + ", ".join(f"({i}, {j}): f32 {int(ids32[j, i])} t {t32[j, i]:.7g} / f64 {int(rid[j, i])} t {rt[j, i]:.7g}" for j, i in diff[:10]))
        same = (ids32 == rid) & (rid >= 0)
        rel = np.where(same, np.abs(t32 - rt) / np.where(rt > 0, rt, 1), 0)
        j, i = np.unravel_index(np.argmax(rel), rel.shape)
        sref = desc.surfaces[int(rid[j, i])]
        notes.append(f"worst dt at pixel ({i}, {j}): surface {int(rid[j, i])} type {int(sref['type'])} index {int(sref['index'])}, "
                     f"t64 {rt[j, i]:.9g} t32 {t32[j, i]:.9g}"
                     + (f", triangle {desc.triangles[int(sref['index'])]['v'].tolist()}" if int(sref['type']) == 1 else ""))
    worst["swaps"], worst["flips"], worst["rel"] = max(worst["swaps"], swaps), max(worst["flips"], flips), max(worst["rel"], max_rel)
    n = sn * sn
    for sampler in (0, 1):
        want, _ = oracle.render(w, h, sn, threads=16, sampler=sampler)
        want = want / n
        for kernel in (0, 1, 2):
            img, st = scene.render(w, h, sn, kernel=kernel, sampler=sampler, collect_stats=True)
            got = img.astype(np.float64) / n
            if not np.isfinite(got).all() or st.samples != w * h * n:
                notes.append(f"kernel {kernel} sampler {sampler}: non-finite radiance or {st.samples} samples")
                continue
            # A pixel whose paths are the same on both sides differs by rounding only (f32 sums against f64: below
            # SAME_PATH); one path that parts ways — a rounding-sensitive decision somewhere along it, more often the
            # more curved mirrors it has met — moves its pixel by up to its throughput / n, or by a light's
            # radiance.  So: the pixels that moved are counted against the frame's paths, and the rest must agree
            # closely, pixel by pixel and in their mean luminance.
            dev = np.abs(got - want).max(axis=-1)
            parted = dev > SAME_PATH
            keep = ~parted
            rmse = float(np.sqrt(np.mean((got[keep] - want[keep]) ** 2, axis=0)).max()) if keep.any() else 0.0
            lum = abs(got[keep].mean() - want[keep].mean()) / max(want[keep].mean(), 1e-6) if keep.any() else 0.0
            frac = parted.sum() / (w * h * n)
            worst["rmse"], worst["lum"] = max(worst["rmse"], rmse), max(worst["lum"], lum)
            worst["parted"] = max(worst["parted"], frac)
            parted_total += int(parted.sum())
            paths_total += w * h * n
            lum_all = abs(got.mean() - want.mean()) / max(want.mean(), 1e-6)  # the moved pixels included: parting ways is no bias
            worst["lum_all"] = max(worst["lum_all"], lum_all)
            if parted.sum() > 3 + parted_bar * w * h * n or rmse > SAME_PATH_RMSE or lum > LUM_BAR or lum_all > LUM_ALL_BAR:
                notes.append(f"kernel {kernel} sampler {sampler}: {int(parted.sum())} pixels moved ({frac:.1e} of the paths), "
                             f"rmse of the rest {rmse:.1e}, luminance {lum:.1e}, of the whole frame {lum_all:.1e}")
    info = scene.info()
    bad += bool(notes)
    print(f"{k:3d} {label} {w:3d}x{h:<3d} sn {sn} surfaces {info.n_surfaces:4d} loose {info.n_loose:2d} "
          f"render nodes {info.render_nodes:4d}: {'ok' if not notes else '; '.join(notes)}", flush=True)


def extras(k, path, scene, w, h, sn):
    """Further properties: strata slices add up to the frame (what several GPUs render), a render tree
    built on the GPU finds the same hits and the same frame as the host-built one; and no tree at all (--no-bvh) renders
    the oracle's --no-bvh frame."""
    global bad
    notes = []
    n = sn * sn
    whole, st = scene.render(w, h, sn, collect_stats=True)
    parts = [scene.render(w, h, sn, stratum_begin=r, stratum_step=3, collect_stats=True) for r in range(3)]
    if sum(p[1].samples for p in parts) != st.samples or not same_image(sum(p[0].astype(np.float64) for p in parts), whole, n):
        notes.append("three strata slices do not add up to the frame")
    ids, t = scene.primary_hits(w, h, precision=32)
    for mode in ("device", False):
        other = ra.Scene(ra.load_scene(path, w / h, use_bvh=mode), 0)
        if mode == "device" and other.info().device_build != 1:
            notes.append("the device build was not used")
        oi, ot = other.primary_hits(w, h, precision=32)
        # (one tree's boxes let a ray reach a primitive that another tree's boxes cull only where the hit lies on
        # the box's face to within rounding: ids may swap among equal-t hits, t may not move)
        if mode == "device" and (np.abs(np.where(ids >= 0, t, 0) - np.where(oi >= 0, ot, 0)).max() > 1e-9 or ((ids >= 0) != (oi >= 0)).any()):
            notes.append(f"use_bvh={mode}: f32 first hits differ from the host-built tree's")
        img, st2 = other.render(w, h, sn, collect_stats=True)
        if mode is False:
            # Without a tree the reference keeps everything in ONE leaf, whose box every ray of interest crosses: a
            # negative-radius sphere (the hollow glass of tests/synth.py) is then met by rays that its small leaf of
            # the tree would have kept away (Q8) — the frame is the reference's --no-bvh frame, not the tree's.
            want, _ = ref.RefScene.from_yaml(path, w / h, use_bvh=False).render(w, h, sn, threads=16, sampler=1)
            dev = np.abs(img.astype(np.float64) - want).max(axis=-1) / n
            moved = dev > SAME_PATH
            if st2.samples != st.samples or moved.sum() > 3 + (PARTED_FRACTION_MESH if args.meshes else PARTED_FRACTION) * w * h * n or \
                    np.sqrt(np.mean((img.astype(np.float64)[~moved] - want[~moved]) ** 2)) / n > SAME_PATH_RMSE:
                notes.append(f"use_bvh=False: the frame differs from the oracle's --no-bvh frame ({int(moved.sum())} pixels moved)")
        elif st2.samples != st.samples or not same_image(img, whole, n):
            notes.append(f"use_bvh={mode}: the frame differs from the host-built tree's")
        other.close()
    bad += bool(notes)
    print(f"{k:3d} extras: {'ok' if not notes else '; '.join(notes)}", flush=True)


def same_image(a, b, n):  # tests/test_gpu_parity.py _same_image: equal up to f32 summation order and the rare path that flips
    a, b = np.asarray(a, np.float64) / n, np.asarray(b, np.float64) / n
    differs = np.any(np.abs(a - b) > 1e-5 * np.maximum(np.abs(a), 1.0), axis=-1)
    return differs.mean() <= 2e-3 + 3.0 / differs.size and abs(a.mean() - b.mean()) <= 1e-3 * max(a.mean(), 1e-6)


for k in range(args.scenes):
    if args.examples:
        # (the generators' own cameras; test_examples.py's bars for the first hits: the cornell examples' camera sits on
        # the boxes' plane of symmetry, and the radius-1000 grounds show their horizon)
        from test_examples import ALL_EXAMPLES, CAMERAS
        name = ALL_EXAMPLES[k % len(ALL_EXAMPLES)]
        seed = int(rng.integers(1 << 30))
        w, h, sn = int(rng.integers(40, 140)), int(rng.integers(30, 110)), int(rng.integers(1, 4))
        if args.only >= 0 and k != args.only:
            continue
        desc = ra.example_scene(name, w / h, seed=seed)
        scene = ra.Scene(desc, 0)
        oracle = ref.RefScene.from_desc(desc, CAMERAS[name](w / h))
        # (first hits: the cornell examples' camera sits on the room's axis, so the jitter-free probe's middle column
        # and — in frames of the right shape — its diagonals run exactly along seams between two quads, which the f64
        # test lets through (a miss in a closed room) and the f32 ray direction's rounding does not: up to a fifth of
        # the frame's perimeter.  The radius-1000 grounds show their horizon: dt as in test_examples.py.)
        compare(k, f"{name:18s} seed {seed:10d}", desc, scene, oracle, w, h, sn, max(4, 3e-3 * w * h, 0.2 * (w + h)), 1e-3, max(2, 1e-3 * w * h))
        scene.close()
        continue
    with tempfile.TemporaryDirectory() as d:
        kind, seed, path = make_scene(pathlib.Path(d), k)
        w, h = int(rng.integers(40, 200)), int(rng.integers(30, 130))
        sn = int(rng.integers(1, 5))
        if args.only >= 0 and k != args.only:
            continue
        desc = ra.load_scene(path, w / h)
        scene = ra.Scene(desc, 0)
        oracle = ref.RefScene.from_yaml(path, w / h)
        compare(k, f"{kind:18s} seed {seed:10d}", desc, scene, oracle, w, h, sn, max(4, 1e-3 * w * h), F32_DT_MAX, 4,
                PARTED_FRACTION_MESH if args.meshes else None)
        if args.extras and k % 3 == 0:
            extras(k, path, scene, w, h, sn)
        scene.close()
print(f"{args.scenes} scenes in {time.time() - t_start:.0f} s, {bad} with a missed bar.  Images: {parted_total} of {paths_total} paths "
      f"parted ways ({parted_total / max(paths_total, 1):.1e}; worst frame {worst['parted']:.1e}, bar {PARTED_FRACTION_MESH if args.meshes else PARTED_FRACTION} + 3), rmse of the pixels that did "
      f"not move at most {worst['rmse']:.1e} (bar {SAME_PATH_RMSE}), mean luminance within {worst['lum']:.1e} (bar {LUM_BAR}; whole frames, movers included: "
      f"{worst['lum_all']:.1e}, bar {LUM_ALL_BAR}).  f32 first hits: "
      f"at most {worst['swaps']} tie-swaps and {worst['flips']} flips a frame, max rel dt {worst['rel']:.2e}; {caught_total} pixels where "
      f"the f32 test held a ray on a shared triangle edge that the f64 test lets through")
sys.exit(1 if bad else 0)
