"""The oracle against closed-form truths (tests/analytic.py) that do not come from the oracle: furnace scenes,
the direct light of a sphere on a plane, the fuzzy-metal re-sampling quirk, and the six sphere (u, v) answers
the reference's source holds (surfaces.rs:112-114).  CPU only; tests/test_gpu_analytic.py holds the HIP path to
the same truths."""
import numpy as np
import pytest

import analytic as A
from oracle import loader_ref as L

SKY = np.array(A.L_SKY)


def test_sphere_uv_known_answers_of_the_reference(ref):
    """surfaces.rs:112-114, verbatim: <1 0 0> -> <0.50 0.50>, <-1 0 0> -> <0.00 0.50>, <0 1 0> -> <0.50 1.00>,
    <0 -1 0> -> <0.50 0.00>, <0 0 1> -> <0.25 0.50>, <0 0 -1> -> <0.75 0.50>."""
    for n, (u, v) in A.SPHERE_UV_KAT:
        gu, gv = ref.sphere_uv(n)
        assert gu == pytest.approx(u, abs=1e-15) and gv == pytest.approx(v, abs=1e-15), (n, gu, gv)


@pytest.mark.parametrize("sampler", [0, 1])
@pytest.mark.parametrize("material,albedo", [
    ("{type: lambertian, albedo: [0.8, 0.5, 0.2]}", (0.8, 0.5, 0.2)),
    ("{type: metal, albedo: [0.9, 0.6, 0.3], fuzz: 0}", (0.9, 0.6, 0.3)),
    ("{type: metal, albedo: [0.9, 0.6, 0.3], fuzz: 0.3}", (0.9, 0.6, 0.3)),
    ("{type: dielectric, index: 1.5}", (1.0, 1.0, 1.0)),
])
def test_furnace_sphere(ref, tmp_path, material, albedo, sampler):
    w = h = 48
    sn = 8
    img, _ = ref.RefScene.from_yaml(A.furnace_sphere_yaml(tmp_path, material), 1.0).render(w, h, sn, threads=8, sampler=sampler)
    img /= sn * sn
    inside, outside = A.furnace_sphere_masks(w, h)
    assert inside.sum() > 100 and outside.sum() > 100
    want = np.array(albedo) * SKY
    assert np.allclose(img[outside], SKY, rtol=1e-12)
    assert np.allclose(img[inside], want, rtol=1e-9), np.abs(img[inside] / want - 1).max()
    f = A.furnace_sphere_coverage()
    mean = (1 - f) * SKY + f * want
    assert np.allclose(img.mean(axis=(0, 1)), mean, rtol=2e-3, atol=1e-4)


@pytest.mark.parametrize("sampler", [0, 1])
def test_furnace_flat_triangle_mesh(ref, tmp_path, sampler):
    w = h = 48
    sn = 6
    img, _ = ref.RefScene.from_yaml(A.flat_mesh_yaml(tmp_path), 1.0).render(w, h, sn, threads=8, sampler=sampler)
    img /= sn * sn
    inside, outside = A.flat_mesh_masks(w, h)
    assert inside.sum() > 200 and outside.sum() > 200
    assert np.allclose(img[outside], SKY, rtol=1e-12)
    assert np.allclose(img[inside], np.array([0.8, 0.5, 0.2]) * SKY, rtol=1e-9)


@pytest.mark.parametrize("sampler", [0, 1])
def test_sphere_light_over_lambertian_plane(ref, tmp_path, sampler):
    w = h = 24
    sn = 32
    img, _ = ref.RefScene.from_yaml(A.light_plane_yaml(tmp_path), 1.0).render(w, h, sn, threads=8, sampler=sampler)
    n = sn * sn
    img /= n
    want, p = A.light_plane_expected(w, h)
    assert p.min() > 0.003 and p.max() < 0.08
    sigma = A.LIGHT["le"] * np.sqrt(p * (1 - p) / n)[..., None] * np.array(A.LIGHT["rho"])
    assert np.all(np.abs(img - want) <= 5 * sigma + 1e-3 * want)
    # whole image: 576 pixels' worth of samples
    s_img = np.sqrt((sigma ** 2).sum(axis=(0, 1))) / (w * h)
    assert np.all(np.abs(img.mean(axis=(0, 1)) - want.mean(axis=(0, 1))) <= 4 * s_img + 2e-3 * want.mean(axis=(0, 1)))


@pytest.mark.parametrize("sampler", [0, 1])
def test_fuzzy_metal_resampling_quirk(ref, tmp_path, sampler):
    w = h = 24
    sn = 24
    n = sn * sn
    s = ref.RefScene.from_yaml(A.metal_plane_yaml(tmp_path), 1.0)
    p_abs, sees = A.metal_plane_absorbed(w, h)
    assert sees.sum() > 200 and 0.2 < p_abs[sees].mean() < 0.45
    a = np.array(A.METAL["a"]) * SKY
    full, _ = s.render(w, h, sn, max_depth=40, threads=8, sampler=sampler)
    assert np.allclose(full[sees] / n, a, rtol=1e-9)          # Q1: every absorbed reflection is sampled again
    two, _ = s.render(w, h, sn, max_depth=2, threads=8, sampler=sampler)
    got = two[sees] / n / a                                     # fraction of first attempts accepted
    sigma = np.sqrt(p_abs[sees] * (1 - p_abs[sees]) / n)[:, None]
    assert np.all(np.abs(got - (1 - p_abs[sees])[:, None]) <= 5 * sigma + 2e-3)
    assert abs(got.mean() - (1 - p_abs[sees]).mean()) <= 4 * np.sqrt((sigma ** 2).sum()) / sees.sum() + 1e-3
    one, _ = s.render(w, h, sn, max_depth=1, threads=8, sampler=sampler)
    assert not one[sees].any()                                  # depth 1: the first hit already ends the path black


def _uv_scene_arrays(ref, tmp_path, direction, w, h):
    """Oracle scene of the uv-checker sphere (textures are not part of the YAML schema: arrays + extras)."""
    frm, to, fov = A.uv_view(direction)
    f = L.FlatScene()
    f.materials = [[0, *A.UV_ALBEDO, 0.0]]
    f.spheres = [[0, 0, 0, 1.0, 0]]
    f.triangles, f.quads = [], []
    f.surfaces = [[0, 0]]
    f.background = np.array(A.L_SKY)
    f.camera = np.array([fov, w / h, *frm, *to, 0.0, 0.0, 0.0])
    flat = f.finish()
    flat.extras = dict(mat_ext=np.array([[2, *A.UV_ODD, A.UV_SCALE, 0]], dtype=np.float64), sphere_end=np.zeros((1, 3)),
                       perlin_vec=np.zeros((0, 256, 3)), perlin_perm=np.zeros((0, 3, 256), np.int32),
                       image_bytes=np.zeros(1, np.uint8), image_dims=np.zeros((0, 3), np.int64), media=np.zeros((0, 5)))
    return flat


@pytest.mark.parametrize("direction", [n for n, _ in A.SPHERE_UV_KAT])
def test_uv_checker_sphere_pattern(ref, tmp_path, direction):
    w = h = 64
    sn = 3
    img, _ = ref.RefScene(_uv_scene_arrays(ref, tmp_path, direction, w, h)).render(w, h, sn, threads=8, sampler=1)
    img /= sn * sn
    want, safe = A.uv_checker_expected(direction, w, h)
    assert safe.sum() > 300
    assert np.allclose(img[safe], want[safe], rtol=1e-9), (direction, np.abs(img[safe] - want[safe]).max())
    assert len(np.unique(want[safe][:, 2])) == 2  # both colours are in view


# ---- thin lens (camera.rs:43-87): the headline scene lucy.yaml is the one shipped scene with an aperture
@pytest.mark.parametrize("sampler", [0, 1])
def test_thin_lens_focus_plane_is_sharp_at_any_aperture(ref, tmp_path, sampler):
    """Every lens sample of a pixel sample passes through the same point of the focus plane: an emissive board ON
    that plane renders the same — exactly the cell colour under each pixel — with aperture 0, 0.5 and 1 (lens radius
    two cells wide)."""
    w = h = 64
    sn = 4
    want, safe = A.lens_board_expected(w, h)
    assert safe.sum() > 0.5 * w * h and len(np.unique(want[safe].round(3), axis=0)) >= 30
    for aperture in (0, 0.5, 1.0):
        img, _ = ref.RefScene.from_yaml(A.lens_board_yaml(tmp_path, aperture), 1.0).render(w, h, sn, threads=8, sampler=sampler)
        img /= sn * sn
        assert np.allclose(img[safe], want[safe], rtol=1e-12), (aperture, np.abs(img[safe] - want[safe]).max())


@pytest.mark.parametrize("sampler", [0, 1])
def test_thin_lens_circle_of_confusion(ref, tmp_path, sampler):
    """Off the focus plane a colour edge blurs over the disk of radius aperture / 2 * |1 - depth / focus|."""
    w = h = 48
    sn = 24
    n = sn * sn
    for aperture in (0.0, 0.5):
        img, _ = ref.RefScene.from_yaml(A.lens_quadrants_yaml(tmp_path, aperture), 1.0).render(w, h, sn, threads=8, sampler=sampler)
        mean, var = A.lens_quadrants_expected(w, h, aperture)
        assert A.statistically_equal(img / n, mean, var, n) is None, aperture
    # the test has teeth: the blur of aperture 0.5 is not that of 0.25 or 1.0 (aperture / 2 is the lens radius)
    for wrong in (0.25, 1.0):
        mean, var = A.lens_quadrants_expected(w, h, wrong)
        assert A.statistically_equal(img / n, mean, var, n) is not None, wrong


# ---- dielectric (material.rs:140-181): Schlick's split, the front_face choice of ratio, refract's direction
@pytest.mark.parametrize("sampler", [0, 1])
@pytest.mark.parametrize("side", ["front", "back"])
def test_glass_pane_reflects_with_schlick_probability(ref, tmp_path, side, sampler):
    """A one-quad pane over a black floor under the sky: E[sample] = L * R(cos theta), R = Schlick at ratio 1 / index
    on the front face, index on the back face, and 1 beyond the critical angle."""
    g = A.GLASS_FRONT if side == "front" else A.GLASS_BACK
    w = h = 40
    sn = 24
    n = sn * sn
    img, _ = ref.RefScene.from_yaml(A.glass_pane_yaml(tmp_path, g, "black"), 1.0).render(w, h, sn, threads=8, sampler=sampler)
    mean, var, r, qe = A.glass_pane_expected(g, w, h, "black")
    assert r.min() < 0.06 and (r[r < 1] > 0.3).any()  # from near-normal to grazing incidence
    if side == "back":
        assert ((r == 1) & (mean[..., 0] > 0)).any()   # total internal reflection is in view
    assert A.statistically_equal(img / n, mean, var, n, qe) is None
    # teeth: the other face's ratio does not fit
    other = dict(g, up=not g["up"])
    mean, var, _, qe = A.glass_pane_expected(other, w, h, "black")
    assert A.statistically_equal(img / n, mean, var, n, qe) is not None


@pytest.mark.parametrize("sampler", [0, 1])
@pytest.mark.parametrize("side", ["front", "back"])
def test_glass_pane_refracts_by_snell(ref, tmp_path, side, sampler):
    """Emissive cells one unit under the pane: the colour mixed in with probability 1 - R is the cell under the
    Snell-displaced point (refract, material.rs:175-181)."""
    g = A.GLASS_FRONT if side == "front" else A.GLASS_BACK
    w = h = 40
    sn = 24
    n = sn * sn
    img, _ = ref.RefScene.from_yaml(A.glass_pane_yaml(tmp_path, g, "cells"), 1.0).render(w, h, sn, threads=8, sampler=sampler)
    mean, var, _, qe = A.glass_pane_expected(g, w, h, "cells")
    assert A.statistically_equal(img / n, mean, var, n, qe) is None
    # teeth: a ray that went straight on (ratio 1), or bent the other way, would light other cells
    for wrong in (1.0, 1.0 / g["ior"]):
        mean, var, _, qe = A.glass_pane_expected(dict(g, ior=wrong), w, h, "cells")
        assert A.statistically_equal(img / n, mean, var, n, qe) is not None, wrong


# ---- the widened path: constant media, moving spheres, the checker texture
def test_smoke_ball_transmits_exp_of_minus_density_times_chord(ra, ref, tmp_path):
    """ConstantMedium::hit's free path (surfaces.rs:611-651): a ball of black smoke lets through exp(-density * chord)."""
    w = h = 40
    sn = 24
    n = sn * sn
    desc = A.smoke_ball_desc(ra, tmp_path)
    mean, var = A.smoke_ball_expected(w, h)
    assert mean[h // 2, w // 2, 0] < 0.25 * SKY[0] and np.allclose(mean[0, 0], SKY)  # e^-1.6 through the middle, sky at the corner
    for sampler in (0, 1):
        img, _ = ref.RefScene.from_desc(desc, A.wide_camera_params(1.0)).render(w, h, sn, threads=8, sampler=sampler)
        assert A.statistically_equal(img / n, mean, var, n) is None, sampler
    # teeth: another density does not fit
    keep = dict(A.SMOKE)
    try:
        A.SMOKE["density"] = 0.6
        wrong, wvar = A.smoke_ball_expected(w, h)
    finally:
        A.SMOKE.update(keep)
    assert A.statistically_equal(img / n, wrong, wvar, n) is not None


def test_moving_light_is_seen_for_the_share_of_the_shutter_it_covers(ra, ref, tmp_path):
    """Sphere::center (surfaces.rs:52-54) and the uniform shutter time (camera.rs:80)."""
    w = h = 40
    sn = 24
    n = sn * sn
    desc = A.moving_light_desc(ra, tmp_path)
    mean, var = A.moving_light_expected(w, h)
    assert 0.2 < mean[..., 0].max() / A.MOVER["colour"][0] < 0.6  # nowhere lit for the whole shutter: the sphere moves 4 radii
    for sampler in (0, 1):
        img, _ = ref.RefScene.from_desc(desc, A.wide_camera_params(1.0)).render(w, h, sn, threads=8, sampler=sampler)
        assert A.statistically_equal(img / n, mean, var, n) is None, sampler


def test_checker_plane_pattern(ra, ref, tmp_path):
    """Checker::color_at (material.rs:237-247) on a Lambertian plane under the uniform sky: colour(p) * L exactly."""
    w = h = 64
    sn = 3
    desc = A.checker_plane_desc(ra, tmp_path)
    want, safe = A.checker_plane_expected(w, h)
    assert safe.sum() > 600 and len(np.unique(want[safe][:, 1])) == 2
    img, _ = ref.RefScene.from_desc(desc, A.checker_camera_params(1.0)).render(w, h, sn, threads=8, sampler=1)
    img /= sn * sn
    assert np.allclose(img[safe], want[safe], rtol=1e-9), np.abs(img[safe] - want[safe]).max()
