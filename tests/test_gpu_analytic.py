"""The HIP path, through the C ABI, against closed-form truths that do not come from the oracle
(tests/analytic.py; tests/test_analytic.py holds the oracle to the same ones): furnace scenes on all three kernel
organisations and both samplers at 4096 spp, the direct light of a sphere on a plane, the fuzzy-metal
re-sampling quirk (Q1, Q2), and the reference's six sphere (u, v) answers (surfaces.rs:112-114) through a
uv-checker whose colour the answer decides.  f32 bars: a pixel sums 4096 values of about 0.5 with f32 atomics, so
"exactly albedo * L" is held to 5e-6 relative for the persistent kernel (per-chunk LDS sums, then one global add
per chunk: measured 1.6e-6) and 1.2e-4 for the wavefront and pool organisations, which add sample by sample into
one float (measured 4.0e-5).  Needs an MI355X:  pytest -m gpu."""
import numpy as np
import pytest

import analytic as A

pytestmark = pytest.mark.gpu

SKY = np.array(A.L_SKY)
F32 = {0: 5e-6, 1: 1.2e-4, 2: 1.2e-4}  # by kernel organisation


@pytest.mark.parametrize("material,albedo", [
    ("{type: lambertian, albedo: [0.8, 0.5, 0.2]}", (0.8, 0.5, 0.2)),
    ("{type: metal, albedo: [0.9, 0.6, 0.3], fuzz: 0}", (0.9, 0.6, 0.3)),
    ("{type: metal, albedo: [0.9, 0.6, 0.3], fuzz: 0.3}", (0.9, 0.6, 0.3)),
    ("{type: dielectric, index: 1.5}", (1.0, 1.0, 1.0)),
])
def test_furnace_sphere(ra, tmp_path, material, albedo):
    w = h = 64
    sn = 64  # 4096 spp
    scene = ra.Scene(ra.load_scene(A.furnace_sphere_yaml(tmp_path, material), 1.0), 0)
    inside, outside = A.furnace_sphere_masks(w, h)
    want = np.array(albedo) * SKY
    f = A.furnace_sphere_coverage()
    mean = (1 - f) * SKY + f * want
    for kernel in (0, 1, 2):
        for sampler in (0, 1):
            img, st = scene.render(w, h, sn, kernel=kernel, sampler=sampler)
            img = img.astype(np.float64) / (sn * sn)
            assert st.samples == w * h * sn * sn
            assert np.allclose(img[outside], SKY, rtol=F32[kernel]), (kernel, sampler)
            assert np.allclose(img[inside], want, rtol=F32[kernel]), (kernel, sampler, np.abs(img[inside] / want - 1).max())
            assert np.allclose(img.mean(axis=(0, 1)), mean, rtol=1e-3), (kernel, sampler)
    scene.close()


def test_furnace_flat_triangle_mesh(ra, tmp_path):
    """1152 coplanar triangles through the render tree and the f32 Möller–Trumbore test: inside the outline every
    sample must be albedo * L.  A ray that slips between two triangles would return L — more than albedo * L: none
    may.  A scattered ray that hits the plane it left returns albedo^2 * L: f32 places the hit point up to
    |t d| 2^-24 ~ 3e-7 under the plane, and a scatter direction n + unit vector of length s < 0.025 (probability
    s^2 / 4 = 1.5e-4) then still crosses the plane beyond t = 0.001 — expected on half of those, 7.5e-5 of the
    samples; f64 places the point 1e-15 off and never does (tests/test_analytic.py).  Budget: 1.5e-4."""
    w = h = 64
    sn = 64
    n = sn * sn
    scene = ra.Scene(ra.load_scene(A.flat_mesh_yaml(tmp_path), 1.0), 0)
    assert scene.info().n_loose == 0 and scene.info().render_nodes > 500
    inside, outside = A.flat_mesh_masks(w, h)
    rho = np.array([0.8, 0.5, 0.2])
    want = rho * SKY
    for kernel in (0, 1, 2):
        for sampler in (0, 1):
            img, _ = scene.render(w, h, sn, kernel=kernel, sampler=sampler)
            img = img.astype(np.float64) / n
            assert np.allclose(img[outside], SKY, rtol=F32[kernel])
            dev = img[inside] / want - 1
            assert dev.max() <= F32[kernel], (kernel, sampler, dev.max())  # no leak between triangles
            # most pixels have no re-hit: their common deviation is the organisation's f32 summation bias
            base = np.median(dev, axis=0)
            assert np.all(np.abs(base) <= F32[kernel]), (kernel, sampler, base)
            events = (dev - base) * n / (rho - 1)  # re-hits per pixel, the same whole number on every channel
            assert np.all(np.abs(events - np.round(events[:, 2:])) <= 0.1), (kernel, sampler)
            rate = np.round(events[:, 2]).sum() / (inside.sum() * n)
            assert rate <= 1.5e-4, (kernel, sampler, rate)
    scene.close()


@pytest.mark.parametrize("sampler", [0, 1])
def test_sphere_light_over_lambertian_plane(ra, tmp_path, sampler):
    w = h = 32
    sn = 64
    n = sn * sn
    scene = ra.Scene(ra.load_scene(A.light_plane_yaml(tmp_path), 1.0), 0)
    want, p = A.light_plane_expected(w, h)
    sigma = A.LIGHT["le"] * np.sqrt(p * (1 - p) / n)[..., None] * np.array(A.LIGHT["rho"])
    s_img = np.sqrt((sigma ** 2).sum(axis=(0, 1))) / (w * h)
    for kernel in (0, 1, 2):
        img, _ = scene.render(w, h, sn, kernel=kernel, sampler=sampler)
        img = img.astype(np.float64) / n
        assert np.all(np.abs(img - want) <= 5 * sigma + 1e-3 * want), kernel
        assert np.all(np.abs(img.mean(axis=(0, 1)) - want.mean(axis=(0, 1))) <= 4 * s_img + 2e-3 * want.mean(axis=(0, 1))), kernel
    scene.close()


@pytest.mark.parametrize("sampler", [0, 1])
def test_fuzzy_metal_resampling_quirk(ra, tmp_path, sampler):
    w = h = 32
    sn = 64
    n = sn * sn
    scene = ra.Scene(ra.load_scene(A.metal_plane_yaml(tmp_path), 1.0), 0)
    p_abs, sees = A.metal_plane_absorbed(w, h)
    a = np.array(A.METAL["a"]) * SKY
    sigma = np.sqrt(p_abs[sees] * (1 - p_abs[sees]) / n)[:, None]
    for kernel in (0, 1, 2):
        full, _ = scene.render(w, h, sn, max_depth=40, kernel=kernel, sampler=sampler)
        assert np.allclose(full[sees].astype(np.float64) / n, a, rtol=F32[kernel]), kernel
        two, _ = scene.render(w, h, sn, max_depth=2, kernel=kernel, sampler=sampler)
        got = two[sees].astype(np.float64) / n / a
        assert np.all(np.abs(got - (1 - p_abs[sees])[:, None]) <= 5 * sigma + 2e-3), kernel
        assert abs(got.mean() - (1 - p_abs[sees]).mean()) <= 4 * np.sqrt((sigma ** 2).sum()) / sees.sum() + 1e-3
        one, _ = scene.render(w, h, sn, max_depth=1, kernel=kernel, sampler=sampler)
        assert not one[sees].any()
    scene.close()


@pytest.mark.parametrize("direction", [n for n, _ in A.SPHERE_UV_KAT])
def test_uv_checker_sphere_pattern(ra, tmp_path, direction):
    """The reference's six (u, v) answers decide the colour at the centre of each view; the pattern around them is
    the same closed form."""
    w = h = 64
    sn = 8
    frm, to, fov = A.uv_view(direction)
    text = (A.camera_text(frm, to, fov)
            + f"scene:\n  background: [{A.L_SKY[0]}, {A.L_SKY[1]}, {A.L_SKY[2]}]\n  surfaces:\n"
            + "    - type: sphere\n      radius: 1\n      position: [0, 0, 0]\n"
            + "      material: {type: lambertian, albedo: [0.9, 0.9, 0.9]}\n")
    path = tmp_path / "uv.yaml"
    path.write_text(text)
    base = ra.load_scene(str(path), w / h)
    mats = base.materials
    mats["texture"][0] = ra.RT_TEX_UV_CHECKER
    mats["albedo"][0], mats["albedo2"][0], mats["tex_scale"][0] = A.UV_ALBEDO, A.UV_ODD, A.UV_SCALE
    desc = ra.SceneDesc.from_arrays(mats, base.spheres, base.triangles, base.quads, base.surfaces, base.background,
                                    base.camera)
    scene = ra.Scene(desc, 0)
    img, _ = scene.render(w, h, sn)
    img = img.astype(np.float64) / (sn * sn)
    want, safe = A.uv_checker_expected(direction, w, h)
    assert safe.sum() > 300
    assert np.allclose(img[safe], want[safe], rtol=F32[0]), (direction, np.abs(img[safe] - want[safe]).max())
    # the centre of the view is the KAT point itself (the two polar views see the pole, where u is undefined)
    if direction[1] == 0:
        c = img[h // 2 - 1:h // 2 + 1, w // 2 - 1:w // 2 + 1].reshape(-1, 3)
        u, v = dict(A.SPHERE_UV_KAT)[direction]
        # the -x view straddles the seam u = 0 | 1: floor(5 * 0.0) = 0 and floor(5 * 0.999) = 4 are both even
        even = (int(np.floor(A.UV_SCALE * u)) + int(np.floor(A.UV_SCALE * v))) % 2 == 0
        colour = np.array(A.UV_ALBEDO if even else A.UV_ODD) * SKY
        assert np.allclose(c, colour, rtol=F32[0]), (direction, c)
    scene.close()


# ---- thin lens (camera.rs:43-87; the headline scene lucy.yaml is the one shipped scene with an aperture)
def test_thin_lens_focus_plane_is_sharp_at_any_aperture(ra, tmp_path):
    """An emissive board ON the focus plane renders exactly the cell colour under each pixel whatever the aperture:
    every lens sample of a pixel sample passes through the same point of that plane."""
    w = h = 64
    sn = 8
    want, safe = A.lens_board_expected(w, h)
    assert safe.sum() > 0.5 * w * h
    for aperture in (0, 0.5, 1.0):
        scene = ra.Scene(ra.load_scene(A.lens_board_yaml(tmp_path, aperture), 1.0), 0)
        for kernel in (0, 1, 2):
            for sampler in (0, 1):
                img, _ = scene.render(w, h, sn, kernel=kernel, sampler=sampler)
                img = img.astype(np.float64) / (sn * sn)
                assert np.allclose(img[safe], want[safe], rtol=F32[kernel]), (aperture, kernel, sampler,
                                                                               np.abs(img[safe] - want[safe]).max())
        scene.close()


@pytest.mark.parametrize("sampler", [0, 1])
def test_thin_lens_circle_of_confusion(ra, tmp_path, sampler):
    """Off the focus plane the colour edges blur over the disk of radius aperture / 2 * |1 - depth / focus|."""
    w = h = 48
    sn = 64
    n = sn * sn
    for aperture in (0.0, 0.5):
        scene = ra.Scene(ra.load_scene(A.lens_quadrants_yaml(tmp_path, aperture), 1.0), 0)
        mean, var = A.lens_quadrants_expected(w, h, aperture)
        for kernel in (0, 1, 2):
            img, _ = scene.render(w, h, sn, kernel=kernel, sampler=sampler)
            img = img.astype(np.float64) / n
            assert A.statistically_equal(img, mean, var, n) is None, (aperture, kernel)
        scene.close()
    for wrong in (0.4, 0.6):  # teeth at 4096 spp: a lens radius 20 % off does not fit
        mean, var = A.lens_quadrants_expected(w, h, wrong)
        assert A.statistically_equal(img, mean, var, n) is not None, wrong


# ---- dielectric (material.rs:140-181): Schlick's split, the front_face choice of ratio, refract's direction
@pytest.mark.parametrize("sampler", [0, 1])
@pytest.mark.parametrize("side", ["front", "back"])
@pytest.mark.parametrize("floor", ["black", "cells"])
def test_glass_pane(ra, tmp_path, floor, side, sampler):
    """black floor: E[sample] = L * R(cos theta) — Schlick at ratio 1 / index on the front face, index on the back,
    1 beyond the critical angle; emissive cells under the pane: the colour mixed in with probability 1 - R is the
    cell under the Snell-displaced point.  The cells (264 quads) live in the render tree, the pane stays loose."""
    g = A.GLASS_FRONT if side == "front" else A.GLASS_BACK
    w = h = 40
    sn = 64
    n = sn * sn
    scene = ra.Scene(ra.load_scene(A.glass_pane_yaml(tmp_path, g, floor), 1.0), 0)
    if floor == "cells":
        assert scene.info().render_nodes > 100 and scene.info().n_loose >= 1
    mean, var, _, qe = A.glass_pane_expected(g, w, h, floor)
    for kernel in (0, 1, 2):
        img, _ = scene.render(w, h, sn, kernel=kernel, sampler=sampler)
        img = img.astype(np.float64) / n
        assert A.statistically_equal(img, mean, var, n, qe) is None, kernel
    scene.close()
    wrong = dict(g, up=not g["up"]) if floor == "black" else dict(g, ior=1.0)
    mean, var, _, qe = A.glass_pane_expected(wrong, w, h, floor)
    assert A.statistically_equal(img, mean, var, n, qe) is not None


# ---- the widened path: constant media, moving spheres, the checker texture — closed forms, not the oracle
@pytest.mark.parametrize("sampler", [0, 1])
def test_smoke_ball_transmits_exp_of_minus_density_times_chord(ra, tmp_path, sampler):
    """ConstantMedium::hit's free path (surfaces.rs:611-651) through the MEDIA build of the kernel."""
    w = h = 40
    sn = 64
    n = sn * sn
    scene = ra.Scene(A.smoke_ball_desc(ra, tmp_path), 0)
    mean, var = A.smoke_ball_expected(w, h)
    for kernel in (0, 1, 2):  # the other organisations defer to the persistent kernel for media
        img, _ = scene.render(w, h, sn, kernel=kernel, sampler=sampler)
        img = img.astype(np.float64) / n
        assert A.statistically_equal(img, mean, var, n) is None, kernel
    scene.close()
    keep = dict(A.SMOKE)
    try:
        A.SMOKE["density"] = 0.72  # teeth at 4096 spp: a density 10 % off does not fit
        wrong, wvar = A.smoke_ball_expected(w, h)
    finally:
        A.SMOKE.update(keep)
    assert A.statistically_equal(img, wrong, wvar, n) is not None


@pytest.mark.parametrize("sampler", [0, 1])
def test_moving_light_is_seen_for_the_share_of_the_shutter_it_covers(ra, tmp_path, sampler):
    """Sphere::center at the ray's time (surfaces.rs:52-54), the shutter time drawn uniformly (camera.rs:80): the EXT
    build of the kernel."""
    w = h = 40
    sn = 64
    n = sn * sn
    scene = ra.Scene(A.moving_light_desc(ra, tmp_path), 0)
    mean, var = A.moving_light_expected(w, h)
    img, _ = scene.render(w, h, sn, sampler=sampler)
    img = img.astype(np.float64) / n
    assert A.statistically_equal(img, mean, var, n) is None
    scene.close()


def test_checker_plane_pattern(ra, tmp_path):
    """Checker::color_at (material.rs:237-247) in f64 on the hit point: colour(p) * L exactly, cell by cell."""
    w = h = 64
    sn = 8
    scene = ra.Scene(A.checker_plane_desc(ra, tmp_path), 0)
    want, safe = A.checker_plane_expected(w, h)
    assert safe.sum() > 600
    for sampler in (0, 1):
        img, _ = scene.render(w, h, sn, sampler=sampler)
        img = img.astype(np.float64) / (sn * sn)
        assert np.allclose(img[safe], want[safe], rtol=F32[0]), np.abs(img[safe] - want[safe]).max()
    scene.close()
