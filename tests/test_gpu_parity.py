"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle and the committed
golden vectors.  Needs an MI355X:  pytest -m gpu.

Bars (BASELINE.json north_star):
  * primary-hit primitive ids EXACT and t within 1e-5 relative — met bit-for-bit by the f64 AOV kernel
    (precision=64), which is therefore also a full-resolution oracle on the device.  The f32 kernel (precision=32:
    the arithmetic and pruned traversal the render kernels use) is compared with it at the golden size and at
    BASELINE's full frame sizes, and its mismatches are split into
      tie-swaps  ids differ, both hit, |dt|/t <= 1e-5: two surfaces meet at the pixel-centre ray (a shared mesh
                 edge, a box corner) and f32 picks the neighbour; harmless, reported, loosely bounded;
      flips      hit <-> miss, or |dt|/t > 1e-5: budgeted per scene at 2x the measured count (floor 4 pixels);
    t must be within 1e-5 relative wherever ids agree, except on a budgeted handful of grazing pixels (a ray
    direction rounded to f32 moves a silhouette-grazing root by more), and within 2e-4 everywhere.
    Measured table: DESIGN.md §2 (tests/parity_table.py).
  * images at equal spp: per-channel RMSE <= IMG_RMSE and mean luminance within IMG_LUM relative.  Oracle and
    GPU draw identical variates, so most paths coincide and the residual is the f32-vs-f64 path divergence:
    measured RMSE <= 0.0015 at 16 spp and <= 0.0007 at 1024 spp, luminance within 2.7e-4 (cornell_lucy, the worst).
"""
import os

import numpy as np
import pytest

from conftest import ROOT, scene_path

pytestmark = pytest.mark.gpu

SCENES = ["cornell", "cornell_lucy", "teapot", "lucy", "dual", "distant", "light", "quads"]
T_REL = 1e-5
T_REL_MAX = 2e-4         # grazing pixels
IMG_RMSE = 0.003         # on linear radiance averaged over spp (values are O(0.1..1)); 2x the measured worst
IMG_LUM = 6e-4           # relative; 2x the measured worst
# f32 AOV vs f64 AOV, (frame, tie-swap bound, flip budget, pixels with rel dt > 1e-5 among equal ids): measured
# swaps/flips/pixels in the comment.  quads.yaml: the quads' edges lie in the plane x = 0, which the pixel
# centres of column W/2 - 1 look along exactly ((pi + 1) / W = 1/2, Q5): on-edge in f64, either side in f32.
F32_BUDGET = {
    "golden": {"cornell": (40, 4, 4), "cornell_lucy": (40, 4, 4), "teapot": (8, 4, 4), "lucy": (8, 4, 4),      # 12/0/0 12/0/0 0/0/0 0/0/0
               "dual": (8, 4, 4), "distant": (8, 4, 4), "light": (8, 4, 4), "quads": (8, 40, 4)},               # 0/0/0 0/0/0 0/0/0 0/20/0
    "full": {("lucy", 1920, 1080): (40, 4, 4), ("teapot", 1920, 1080): (40, 4, 16),                            # 9/1/1 13/2/8
             ("dual", 1920, 1080): (40, 16, 26), ("cornell_lucy", 3840, 2160): (1000, 96, 4)},                  # 16/8/13 400/48/1
}


def classify_f32(ids32, t32, ids64, t64):
    """(tie-swaps, flips, id-equal pixels with rel dt > T_REL, max rel dt among id-equal pixels)."""
    both = (ids64 >= 0) & (ids32 >= 0)
    rel = np.zeros_like(t64)
    rel[both] = np.abs(t32[both] - t64[both]) / t64[both]
    same = ids64 == ids32
    swap = ~same & both & (rel <= T_REL)
    flip = ~same & ~swap
    return int(swap.sum()), int(flip.sum()), int((rel[same & both] > T_REL).sum()), float(rel[same & both].max(initial=0.0))


def golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))


@pytest.fixture(scope="module")
def scenes(ra):
    cache = {}

    def get(name, aspect, use_bvh=True):
        key = (name, round(aspect, 6), use_bvh)
        if key not in cache:
            cache[key] = ra.Scene(ra.load_scene(scene_path(name), aspect, use_bvh), 0)
        return cache[key]

    yield get
    for s in cache.values():
        s.close()


# ------------------------------------------------------------------ primary hits
@pytest.mark.parametrize("name", SCENES)
def test_primary_hits_f64_exact_vs_golden(scenes, name):
    g = golden(name)
    h, w = g["prim_id"].shape
    ids, t = scenes(name, w / h).primary_hits(w, h, precision=64)
    assert np.array_equal(ids, g["prim_id"])
    assert np.array_equal(t, g["t"])  # same operations, no contraction: bit-identical


@pytest.mark.parametrize("name,w,h", [("lucy", 480, 270), ("cornell", 300, 300), ("teapot", 480, 270),
                                      ("dual", 320, 180), ("distant", 320, 180)])
def test_primary_hits_f64_exact_vs_oracle(scenes, ref, name, w, h):
    ids, t = scenes(name, w / h).primary_hits(w, h, precision=64)
    rid, rt = ref.RefScene.from_yaml(scene_path(name), w / h).primary_hits(w, h, threads=16)
    assert np.array_equal(ids, rid)
    hit = rid >= 0
    assert np.max(np.abs(t[hit] - rt[hit]) / rt[hit], initial=0.0) <= T_REL
    assert np.array_equal(t, rt)


@pytest.mark.parametrize("name", SCENES)
def test_primary_hits_f32_within_budget(scenes, name):
    g = golden(name)
    h, w = g["prim_id"].shape
    ids, t = scenes(name, w / h).primary_hits(w, h, precision=32)
    swaps, flips, n_rel, max_rel = classify_f32(ids, t, g["prim_id"], g["t"])
    b_swaps, b_flips, b_rel = F32_BUDGET["golden"][name]
    assert swaps <= b_swaps and flips <= b_flips and n_rel <= b_rel and max_rel <= T_REL_MAX, (swaps, flips, n_rel, max_rel)


@pytest.mark.parametrize("name,w,h", list(F32_BUDGET["full"]))
def test_primary_hits_f32_vs_f64_at_full_size(scenes, name, w, h):
    """BASELINE's frame sizes: the render arithmetic's closest hits against the bit-exact f64 kernel's."""
    s = scenes(name, w / h)
    ids64, t64 = s.primary_hits(w, h, precision=64)
    ids32, t32 = s.primary_hits(w, h, precision=32)
    swaps, flips, n_rel, max_rel = classify_f32(ids32, t32, ids64, t64)
    b_swaps, b_flips, b_rel = F32_BUDGET["full"][(name, w, h)]
    assert flips <= b_flips, f"{flips} flips of {ids64.size} pixels (budget {b_flips})"
    assert flips / ids64.size <= 1e-4  # BASELINE north_star's "exact" up to silhouette pixels; SURVEY appendix A
    assert swaps <= b_swaps and n_rel <= b_rel and max_rel <= T_REL_MAX, (swaps, flips, n_rel, max_rel)


def test_no_bvh_gives_identical_primary_hits(scenes):
    """--no-bvh (main.rs:60-65): every surface in one leaf; t must not change, ids only among exact ties."""
    for name in ("teapot", "quads", "distant"):
        a = scenes(name, 1.0, True).primary_hits(48, 48, 64)
        b = scenes(name, 1.0, False).primary_hits(48, 48, 64)
        assert np.array_equal(a[1], b[1])
        assert np.all((a[0] == b[0]) | (a[1] == b[1]))


# ------------------------------------------------------------------ images
LIGHT_OUTLIERS = {"cornell": 2, "cornell_lucy": 2}  # scenes lit by a small 15-unit emitter


def _check_image(got_sum, want_sum, n, name, rmse_bar=IMG_RMSE, lum_bar=IMG_LUM):
    """RMSE and mean luminance of the averaged images.  In the scenes lit by a small bright emitter, one path in a few
    thousand parts ways between f32 and f64 (a rounding-sensitive decision somewhere along it), and when one of the
    two then finds the 15-unit light its pixel moves by 15 * throughput / n — 0.26 at 16 spp for a throughput of 0.28, an
    RMSE of 0.0068 over the 48 x 32 golden frame all by itself (measured: pixel (3, 12) of cornell.yaml, once the
    triangle test's rounding was pinned).  Which path does so changes with every change of the kernels' rounding; at
    low spp a few such pixels (deviation above 1 / n: more than any path that does not end on a light can carry)
    are therefore set aside and counted, and the bars hold for all the rest."""
    got, want = got_sum.astype(np.float64) / n, want_sum / n
    assert np.isfinite(got).all()
    keep = np.ones(got.shape[:2], bool)
    # two, and one more per 40 000 paths of the frame (measured: 1 of 24 576 on cornell.yaml's golden frame, 6 of
    # 331 776 on cornell_lucy.yaml at 96 x 96 x 36)
    allowed = (LIGHT_OUTLIERS.get(name.split()[0], 0) + got.shape[0] * got.shape[1] * n // 40000) if n <= 64 and name.split()[0] in LIGHT_OUTLIERS else 0
    if allowed:
        dev = np.abs(got - want).max(axis=-1)
        out = dev > 1.0 / n
        assert out.sum() <= allowed, (name, int(out.sum()), np.sort(dev.ravel())[-4:])
        keep = ~out
    rmse = np.sqrt(np.mean((got[keep] - want[keep]) ** 2, axis=0))
    assert np.all(rmse <= rmse_bar), (name, rmse)
    lum = abs(got[keep].mean() - want[keep].mean()) / max(want[keep].mean(), 1e-6)
    assert lum <= lum_bar, (name, lum, got.mean(), want.mean())


@pytest.mark.parametrize("sampler", [0, 1])
@pytest.mark.parametrize("kernel", [0, 1, 2])
@pytest.mark.parametrize("name", SCENES)
def test_image_matches_golden(ra, scenes, name, kernel, sampler):
    """Same sampler on both sides (0: the reference's rejection loops, 1: the direct maps): GPU and oracle then
    follow the same paths sample by sample."""
    g = golden(name)
    want = g["image_sum"] if sampler == 0 else g["image_sum_direct"]
    ih, iw, _ = want.shape
    img, st = scenes(name, iw / ih).render(iw, ih, 4, max_depth=40, seed=1, kernel=kernel, collect_stats=True,
                                           sampler=sampler)
    _check_image(img, want, 16, name)
    assert st.samples == iw * ih * 16
    # The GPU ends a path at a light / an absorbed metal bounce without re-tracing (Q1), so it traces no more
    # segments than the reference, up to the few paths whose f32 and f64 versions part ways (2 % allowed).
    assert 0 < st.segments <= 1.02 * int(g["segments"])


@pytest.mark.parametrize("name,w,h,sn", [("teapot", 160, 90, 6), ("cornell_lucy", 96, 96, 6), ("distant", 160, 90, 8),
                                         ("teapot", 96, 54, 32), ("cornell_lucy", 96, 96, 32), ("lucy", 96, 54, 32),
                                         ("dual", 96, 54, 32)])
def test_image_matches_oracle_higher_spp(scenes, ref, name, w, h, sn):
    """Up to 1024 spp per mesh scene, all three organisations, both samplers."""
    o = ref.RefScene.from_yaml(scene_path(name), w / h)
    for sampler in (0, 1):
        want, _ = o.render(w, h, sn, threads=16, sampler=sampler)
        for kernel in (0, 1, 2):
            img, _ = scenes(name, w / h).render(w, h, sn, kernel=kernel, sampler=sampler)
            # cornell at 36 spp: one path in ~2000 parts ways in f32 and a path that finds the 15-unit light moves
            # its pixel by 0.2: measured RMSE 0.0062 and luminance 7e-4 (0.0007 and 2.7e-4 at 1024 spp)
            low = (name, sn) == ("cornell_lucy", 6)
            _check_image(img, want, sn * sn, name, rmse_bar=0.012 if low else IMG_RMSE, lum_bar=2e-3 if low else IMG_LUM)


def _same_image(a, b, n):
    """Equal up to f32 summation order, except for the rare sample whose path flips at a rounding-sensitive
    decision (the two kernel organisations are compiled separately and contract multiply-adds differently)."""
    a, b = a.astype(np.float64) / n, b.astype(np.float64) / n
    differs = np.any(np.abs(a - b) > 1e-5 * np.maximum(np.abs(a), 1.0), axis=-1)
    return differs.mean() <= 2e-3 and abs(a.mean() - b.mean()) <= 1e-4 * a.mean()


def test_kernels_agree_and_are_repeatable(scenes):
    s = scenes("teapot", 16 / 9)
    a, _ = s.render(192, 108, 4, kernel=0)
    b, _ = s.render(192, 108, 4, kernel=0)
    c, _ = s.render(192, 108, 4, kernel=1)
    d, _ = s.render(192, 108, 4, kernel=2)
    assert np.allclose(a, b, rtol=1e-5, atol=1e-6)   # same binary, same samples: only the f32 sum order may differ
    assert _same_image(a, c, 16)                     # same paths in all three kernel organisations
    assert _same_image(a, d, 16)


def test_strata_slices_sum_to_the_full_frame(scenes):
    """Multi-GPU sharding: strata k = g mod N on GPU g; the union is the single-GPU sample set."""
    s = scenes("teapot", 16 / 9)
    for kernel in (0, 1, 2):
        full, st = s.render(160, 90, 4, kernel=kernel)
        parts = [s.render(160, 90, 4, kernel=kernel, stratum_begin=g, stratum_step=3) for g in range(3)]
        assert sum(p[1].samples for p in parts) == st.samples
        assert np.allclose(sum(p[0].astype(np.float64) for p in parts), full, rtol=1e-5, atol=1e-6)


def test_seed_changes_the_image_and_stream_is_keyed_by_pixel(scenes):
    s = scenes("quads", 1.5)
    a, _ = s.render(60, 40, 2, seed=1)
    b, _ = s.render(60, 40, 2, seed=2)
    assert not np.array_equal(a, b)
    # a pixel's samples do not depend on the image's other pixels being rendered in the same launch shape
    assert abs(a.mean() - b.mean()) < 0.05 * a.mean()


def test_tonemap_bit_exact_vs_oracle(ra, ref):
    rng = np.random.default_rng(1)
    x = (rng.random((33, 47, 3)) * 40).astype(np.float32)
    x[0, 0] = [0, np.nan, -1]
    x[0, 1] = [1e9, 16.0, 15.999]
    assert np.array_equal(ra.tonemap(x, 16), ref.tonemap(x.astype(np.float64), 16))


def test_render_device_into_torch_tensor(ra, scenes):
    import torch

    s = scenes("quads", 1.5)
    fb = torch.zeros((40, 60, 3), dtype=torch.float32, device="cuda:0")
    st = s.render_device(fb.data_ptr(), 60, 40, 2, stream=torch.cuda.current_stream().cuda_stream)
    host, _ = s.render(60, 40, 2)
    assert np.allclose(fb.cpu().numpy(), host, rtol=1e-5, atol=1e-6)
    assert st.kernel_ms > 0 and st.samples == 60 * 40 * 4


def test_max_depth_and_metal_resampling_quirk(scenes, ref):
    """Q1: an absorbed metal reflection is re-sampled, one depth unit per attempt; depth-limited paths add 0."""
    name, w, h = "dual", 96, 54
    o = ref.RefScene.from_yaml(scene_path(name), w / h)
    for md in (1, 2, 5):
        want, _ = o.render(w, h, 3, max_depth=md, threads=16, sampler=0)
        got, _ = scenes(name, w / h).render(w, h, 3, max_depth=md, sampler=0)
        _check_image(got, want, 9, f"{name} max_depth={md}")


# ------------------------------------------------------------------ full-size properties (BASELINE sizes)
def test_full_size_frame_properties(scenes):
    """BASELINE-size frames at reduced spp, checked through size-independent properties: finite values; no
    pixel brighter than the sky where nothing emits (all albedos <= 1); two disjoint strata halves estimate
    the same mean; a pixel that sees only sky returns exactly background * n."""
    s = scenes("lucy", 1920 / 1080)
    a, sa = s.render(1920, 1080, 4, stratum_begin=0, stratum_step=2)
    b, sb = s.render(1920, 1080, 4, stratum_begin=1, stratum_step=2)
    assert sa.samples == sb.samples == 1920 * 1080 * 8
    assert np.isfinite(a).all() and np.isfinite(b).all()
    assert np.all(a >= 0) and np.all(a <= np.array([0.7, 0.8, 1.0]) * 8 * (1 + 1e-5))  # lucy.yaml:10
    ma, mb = a.mean() / 8, b.mean() / 8
    assert abs(ma - mb) < 0.002 * ma
    q = scenes("quads", 1920 / 1080)
    img, _ = q.render(1920, 1080, 3)
    assert np.allclose(img[0, 0] / 9, [0.7, 0.8, 1.0], rtol=1e-6)  # the top-left corner looks past every quad


def test_direct_sampler_is_distribution_equal_on_the_gpu(scenes, ref):
    """Cross check of the two samplers through the product path: GPU(direct) against oracle(REFERENCE sampler)
    at high spp — different variates, so the bound is statistical (two independent 1024-spp estimates)."""
    # cornell's small area light makes it the noisier of the two: independent 1024-spp estimates differ by ~0.03
    for name, w, h, tol in (("teapot", 96, 54, 0.03), ("cornell", 64, 64, 0.06)):
        want, _ = ref.RefScene.from_yaml(scene_path(name), w / h).render(w, h, 32, threads=16, sampler=0)
        got, _ = scenes(name, w / h).render(w, h, 32, sampler=1)
        a, b = got.astype(np.float64) / 1024, want / 1024
        assert abs(a.mean() - b.mean()) <= 0.01 * b.mean()
        assert np.sqrt(np.mean((a - b) ** 2)) <= tol


def _id_scene(ra, ref, name, aspect):
    """The scene with every surface turned into a light whose emission spells its surface id (1 + id % 256,
    1 + id / 256 % 256, 1 + id / 65536 — exact in f32), on the product's side and on the oracle's; the sky is 0.5."""
    from oracle import loader_ref
    from raytracer_amd.api import MATERIAL_DTYPE

    d = ra.load_scene(scene_path(name), aspect)
    surf = d.surfaces
    n = len(surf)
    mats = np.zeros(n, dtype=MATERIAL_DTYPE)
    sid = np.arange(n)
    mats["kind"] = ra.RT_MAT_DIFFUSE_LIGHT
    mats["albedo"] = np.stack([1 + sid % 256, 1 + sid // 256 % 256, 1 + sid // 65536], axis=1)
    sph, tri, quad = d.spheres.copy(), d.triangles.copy(), d.quads.copy()
    for arr, typ in ((sph, ra.RT_SURF_SPHERE), (tri, ra.RT_SURF_TRIANGLE), (quad, ra.RT_SURF_QUAD)):
        of_type = surf["type"] == typ
        arr["material"][surf["index"][of_type]] = sid[of_type]
    mine = ra.SceneDesc.from_arrays(mats, sph, tri, quad, surf, (0.5, 0.5, 0.5), d.camera)
    flat = loader_ref.load_scene(scene_path(name), aspect)
    flat.materials = np.column_stack([np.full(n, 3.0), mats["albedo"], np.zeros(n)])
    for arr, typ, col in ((flat.spheres, 0, 4), (flat.triangles, 1, 18), (flat.quads, 2, 9)):
        of_type = np.asarray(flat.surfaces)[:, 0] == typ
        arr[np.asarray(flat.surfaces)[of_type, 1], col] = sid[of_type]
    flat.background = np.array([0.5, 0.5, 0.5])
    return mine, ref.RefScene(flat)


def _decode_ids(img):
    img = np.asarray(img, dtype=np.float64).reshape(-1, 3)
    ids = (img[:, 0] - 1) + 256 * (img[:, 1] - 1) + 65536 * (img[:, 2] - 1)
    return np.where(img[:, 0] >= 1.0, ids, -1).astype(np.int64)


# the bar is 2x the measured count of mismatching pixels (of 2 073 600; at least 8).  Measured: lucy 10, teapot 12, dual 16,
# cornell_lucy 2 — the same pixels' worth in all three kernel organisations: f32 against f64 at shared edges and silhouettes,
# as in the id table, not scheduling
STAGE_MACHINE_BUDGET = {"lucy": 20, "teapot": 24, "dual": 32, "cornell_lucy": 8}


@pytest.mark.parametrize("name", list(STAGE_MACHINE_BUDGET))
def test_first_hits_of_the_render_kernels_at_full_size_vs_oracle(ra, ref, name):
    """What the id table above cannot say: that the persistent kernel's STAGE MACHINE (and the pool's, and the wavefront
    kernels) finds the oracle's surface for a full 1920 x 1080 frame of the render path's own rays — jittered, through
    the lens (lucy.yaml's aperture), drawn from the shared streams — and not only `primary32_kernel`'s while-while loop
    for pixel-centre rays.  Every surface emits its own id; one sample per pixel; the image IS the id map."""
    w, h = 1920, 1080
    mine, oracle = _id_scene(ra, ref, name, w / h)
    want, _ = oracle.render(w, h, 1, max_depth=1, seed=5, threads=16, sampler=1)
    want_ids = _decode_ids(want)
    assert (want_ids >= 0).mean() > 0.05
    s = ra.Scene(mine, 0)
    for kernel in (ra.RT_KERNEL_PERSISTENT, ra.RT_KERNEL_POOL, ra.RT_KERNEL_WAVEFRONT):
        got, _ = s.render(w, h, 1, max_depth=1, seed=5, kernel=kernel, sampler=1)
        got_ids = _decode_ids(got)
        bad = int((got_ids != want_ids).sum())
        print(f"\n{name} kernel {kernel}: {bad} of {w * h} first hits differ from the oracle's")
        assert bad <= STAGE_MACHINE_BUDGET[name], (name, kernel, bad)
    s.close()
