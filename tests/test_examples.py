"""The eight procedural example scenes (scene/example.rs:53-79; `one-weekend` is the reference CLI's default).
The reference draws the random ones from thread_rng(), so there is no fixture to compare their content with; what
is pinned is the generator's structure (counts, fixed spheres, textures, motion, media, camera) and, on the GPU,
render parity with the oracle over the very arrays the generator produced."""
import subprocess

import numpy as np
import pytest

from conftest import ROOT

CAMERAS = {  # [vfov, aspect, from, towards, aperture, has_focus, focus] as each example builds its camera
    "one-weekend": lambda a: [20.0, a, 13, 2, 3, 0, 0, 0, 0.1, 1.0, 10.0],       # example.rs:85-90
    "random-spheres": lambda a: [20.0, a, 13, 2, 3, 0, 0, 0, 0.1, 1.0, 10.0],    # example.rs:146-151
    "two-spheres": lambda a: [20.0, a, 13, 2, 3, 0, 0, 0, 0.0, 0.0, 0.0],        # example.rs:209-212
    "two-perlin": lambda a: [20.0, a, 13, 2, 3, 0, 0, 0, 0.0, 0.0, 0.0],         # example.rs:237-240
    "cornell": lambda a: [40.0, a, 278, 278, -800, 278, 278, 0, 0.0, 0.0, 0.0],  # example.rs:263-266
    "earth": lambda a: [20.0, a, 0, 0, 12, 0, 0, 0, 0.0, 0.0, 0.0],              # example.rs:406-409
    "cornell-smoke": lambda a: [40.0, a, 278, 278, -800, 278, 278, 0, 0.0, 0.0, 0.0],  # example.rs:332-335
    "final-scene": lambda a: [40.0, a, 478, 278, -600, 278, 278, 0, 0.0, 0.0, 0.0],    # example.rs:467-470
}


def test_one_weekend_structure(ra):
    d = ra.example_scene("one-weekend", 1.5, seed=3)
    s, m = d.spheres, d.materials
    assert 400 < len(s) <= 488 and len(m) == len(s)           # 22 x 22 grid minus the cells near (4, 0.2, 0)
    assert s["center"][0].tolist() == [0, -1000, 0] and s["radius"][0] == 1000
    assert s["center"][-3:].tolist() == [[-4, 1, 0], [0, 1, 0], [4, 1, 0]] and np.all(s["radius"][-3:] == 1.0)
    assert m["kind"][-3:].tolist() == [0, 2, 1] and m["param"][-2] == 1.5 and m["param"][-1] == 0.0
    small = s[1:-3]
    assert np.all(small["radius"] == 0.2) and np.all(small["center"][:, 1] == 0.2)
    assert np.all(np.linalg.norm(small["center"] - [4.0, 0.2, 0.0], axis=1) > 0.9)
    kinds = m["kind"][1:-3]
    frac = [(kinds == k).mean() for k in (0, 1, 2)]            # 80 % diffuse, 15 % metal, 5 % glass
    assert abs(frac[0] - 0.8) < 0.08 and abs(frac[1] - 0.15) < 0.06 and abs(frac[2] - 0.05) < 0.04
    metal = m[1:-3][kinds == 1]
    assert np.all((metal["albedo"] >= 0.5) & (metal["albedo"] < 1.0)) and np.all((metal["param"] >= 0) & (metal["param"] < 0.5))
    assert np.array_equal(d.background, [0.7, 0.8, 1.0]) and d.camera.lens_radius == 0.05
    # the seed fixes the draws
    again = ra.example_scene("one-weekend", 1.5, seed=3)
    other = ra.example_scene("one-weekend", 1.5, seed=4)
    assert np.array_equal(again.spheres["center"], s["center"]) and not np.array_equal(other.spheres["center"][:50], s["center"][:50])


def test_cornell_structure(ra):
    d = ra.example_scene("cornell", 1.0)
    q = d.quads
    assert len(q) == 6 + 12 and len(d.spheres) == 0 and np.array_equal(d.background, [0, 0, 0])
    assert d.materials["kind"].tolist() == [0, 0, 0, 3] and d.materials["albedo"][3].tolist() == [15, 15, 15]
    # tall box: 165 x 333 x 165 rotated 15 degrees about its centre, moved by (265, 0, 295)
    pts = np.concatenate([q["q"][6:12], q["q"][6:12] + q["u"][6:12], q["q"][6:12] + q["v"][6:12],
                          q["q"][6:12] + q["u"][6:12] + q["v"][6:12]])
    assert np.allclose(pts[:, 1].min(), 0.001) and np.allclose(pts[:, 1].max(), 333.0)
    assert np.allclose(pts[:, [0, 2]].mean(0), [265 + 82.5, 295 + 82.5])
    edges = np.linalg.norm(q["u"][6:12], axis=1)
    assert np.allclose(np.sort(edges), [165] * 6)
    c, s_ = np.cos(np.radians(15)), np.sin(np.radians(15))
    assert np.allclose(np.abs(q["u"][6]), [165 * c, 0, 165 * s_])  # the front face's dx after the turn


def test_textured_and_moving_examples_structure(ra):
    d = ra.example_scene("random-spheres", 1.5, seed=2)
    m, s = d.materials, d.spheres
    assert m["texture"][0] == ra.RT_TEX_CHECKER and m["tex_scale"][0] == 0.32           # example.rs:157-161
    assert m["albedo"][0].tolist() == [0.2, 0.3, 0.1] and m["albedo2"][0].tolist() == [0.9, 0.9, 0.9]
    moving = s["moving"] == 1
    assert np.all(m["kind"][s["material"][moving]] == 0) and moving.sum() > 300        # only the diffuse ones bounce
    lift = s["center1"][moving] - s["center"][moving]
    assert np.all(lift[:, [0, 2]] == 0) and np.all((lift[:, 1] >= 0) & (lift[:, 1] < 0.5))
    assert np.array_equal(s["center1"][~moving], s["center"][~moving])
    d = ra.example_scene("two-spheres", 1.5)
    assert d.materials["texture"].tolist() == [ra.RT_TEX_UV_CHECKER] and d.materials["tex_scale"][0] == 30.0
    assert d.spheres["center"].tolist() == [[0, -10, 0], [0, 10, 0]] and np.all(d.spheres["material"] == 0)
    d = ra.example_scene("two-perlin", 1.5, seed=5)
    p = d.perlin
    assert len(p) == 1 and d.materials["texture"][0] == ra.RT_TEX_MARBLE and d.materials["tex_scale"][0] == 4.0
    assert np.allclose(np.linalg.norm(p["ranvec"][0], axis=1), 1.0)
    for a in range(3):
        assert sorted(p["perm"][0][a].tolist()) == list(range(256))                     # a permutation each
    assert not np.array_equal(p["perm"][0][0], p["perm"][0][1])
    d = ra.example_scene("earth", 1.5)
    assert d.materials["texture"][0] == ra.RT_TEX_IMAGE and len(d.images) == 1 and d.images[0].size == 0


def test_media_examples_structure(ra):
    d = ra.example_scene("cornell-smoke", 1.0)
    med, mats = d.media, d.materials
    assert len(med) == 2 and med["boundary"].tolist() == [2, 2] and med["count"].tolist() == [6, 6]
    assert med["density"].tolist() == [0.01, 0.01]
    assert mats["kind"][med["material"]].tolist() == [4, 4]                       # Isotropic phase functions
    assert mats["albedo"][med["material"]].tolist() == [[0, 0, 0], [1, 1, 1]]    # smoke and fog
    assert len(d.quads) == 18 and len(d.surfaces) == 8                            # boundary quads are not surfaces
    assert d.surfaces["type"].tolist() == [2] * 6 + [3, 3]
    assert mats["albedo"][3].tolist() == [7, 7, 7]
    d = ra.example_scene("final-scene", 1.0, seed=9)
    assert len(d.quads) == 2400 + 1 and len(d.spheres) == 6 + 2 + 1000 and len(d.media) == 2  # 6 shown, 2 boundary copies, the cluster
    assert d.media["density"].tolist() == [0.2, 0.0001] and d.spheres["radius"][d.media["first"]].tolist() == [70, 5000]
    assert np.array_equal(d.background, [0, 0, 0]) and d.spheres["moving"].sum() == 1
    tops = (d.quads["q"][:2400, 1] + d.quads["u"][:2400, 1] + d.quads["v"][:2400, 1]).reshape(400, 6).max(1)
    assert np.all((tops >= 1.0) & (tops < 101.0))                                # box heights, example.rs:484
    cluster = d.spheres["center"][-1000:]
    assert np.all(d.spheres["radius"][-1000:] == 10) and np.all(np.abs(cluster.mean(0) - [-100 + 82.5, 270 + 82.5, 395 + 82.5]) < 12)


def test_a_fifth_medium_is_refused_with_a_message(ra):
    """The kernels hold at most four constant media (kMaxMedia, device_scene.h; the eight examples need two): a scene
    description with a fifth is refused at rt_scene_create — host-only, no device needed — with a message, not
    rendered wrongly.  Four are accepted."""
    d = ra.example_scene("cornell-smoke", 1.0)
    media = np.concatenate([d.media, d.media, d.media[:1]])  # 5 media over the same two boundaries
    surfaces = np.concatenate([d.surfaces, np.array([(3, 2), (3, 3), (3, 4)], dtype=d.surfaces.dtype)])

    def make(n_media, n_surf):
        return ra.SceneDesc.from_arrays(d.materials, d.spheres, d.triangles, d.quads, surfaces[:n_surf], d.background,
                                        d.camera, media=media[:n_media])

    ra.Scene(make(4, len(d.surfaces) + 2), ra.RT_DEVICE_NONE).close()
    with pytest.raises(ra.api.RtError, match="too many media \\(at most 4\\)"):
        ra.Scene(make(5, len(d.surfaces) + 3), ra.RT_DEVICE_NONE)


def _smoke_with_split_sides(ra):
    """cornell-smoke with the tall box's boundary given as TEN quads instead of six: every side face cut into a lower and
    an upper half (same planes, same solid).  Returns (that description, the original)."""
    d = ra.example_scene("cornell-smoke", 1.0)
    quads, media = d.quads, d.media.copy()
    first = int(media["first"][0])
    assert media["count"][0] == 6 and media["boundary"][0] == ra.RT_SURF_QUAD
    extra = []
    for side in range(4):  # front, right, back, left (make_box's order, surfaces.rs:549-578); top and bottom stay whole
        q = quads[first + side].copy()
        half = q["v"] / 2.0
        lower, upper = q.copy(), q.copy()
        lower["v"], upper["v"] = half, half
        upper["q"] = q["q"] + half
        for h in (lower, upper):
            n = np.cross(h["u"], h["v"])
            h["normal"], h["w"] = n / np.sqrt(n @ n), n / (n @ n)
            h["d"] = h["normal"] @ h["q"]
        # the very same planes, up to the last bits of a turned normal
        assert np.allclose(lower["normal"], q["normal"], rtol=0, atol=1e-15) and abs(lower["d"] - q["d"]) + abs(upper["d"] - q["d"]) < 1e-12
        extra += [lower, upper]
    extra += [quads[first + 4].copy(), quads[first + 5].copy()]
    more = np.concatenate([quads, np.array(extra, dtype=quads.dtype)])
    media["first"][0], media["count"][0] = len(quads), 10
    split = ra.SceneDesc.from_arrays(d.materials, d.spheres, d.triangles, more, d.surfaces, d.background, d.camera, media=media)
    return split, d


def test_medium_boundary_of_ten_quads_builds(ra, ref):
    """A boundary of more than six quads is accepted (the reference's ConstantMedium takes any Hittable,
    surfaces.rs:601-608), and oracle and product see the same scene."""
    split, d = _smoke_with_split_sides(ra)
    s = ra.Scene(split, ra.RT_DEVICE_NONE)
    o = ref.RefScene.from_desc(split, CAMERAS["cornell-smoke"](1.0))
    assert np.array_equal(s.leaf_order(), o.leaf_order()[0])
    s.close()


@pytest.mark.gpu
def test_medium_boundary_of_ten_quads_renders_like_the_box_it_is(ra, ref):
    """The ten-quad boundary encloses the same solid by the same planes, so a ray enters and leaves it at the very
    same t (to the last bits of a turned normal) and the free-path draws fall the same way: the image is the six-quad
    box's, bit for bit at one sample per pixel and up to the order of the additions at 25 — and within the usual bars of the oracle's image of the ten-quad scene."""
    from test_gpu_parity import _check_image, _same_image

    split, d = _smoke_with_split_sides(ra)
    w = h = 96
    a, b = ra.Scene(split, 0), ra.Scene(d, 0)
    for sampler in (0, 1):
        ia, _ = a.render(w, h, 5, sampler=sampler)
        ib, _ = b.render(w, h, 5, sampler=sampler)
        assert _same_image(ia, ib, 25)  # equal up to the order of the f32 additions into a pixel (atomics)
        one_a, _ = a.render(w, h, 1, sampler=sampler)
        one_b, _ = b.render(w, h, 1, sampler=sampler)
        assert np.array_equal(one_a, one_b)  # one sample per pixel: no additions, the very same bits
        want, _ = ref.RefScene.from_desc(split, CAMERAS["cornell-smoke"](1.0)).render(w, h, 5, threads=16, sampler=sampler)
        _check_image(ia, want, 25, f"ten-quad medium, sampler {sampler}")
    a.close(), b.close()


def test_examples_camera_matches_oracle_and_unsupported_names_are_refused(ra, ref):
    for name, aspect in (("one-weekend", 1.5), ("cornell", 1.0), ("random-spheres", 1.5), ("two-spheres", 1.5),
                         ("two-perlin", 1.5), ("earth", 1.5), ("cornell-smoke", 1.0), ("final-scene", 1.0)):
        d = ra.example_scene(name, aspect)
        cam = ref.RefScene.from_desc(d, CAMERAS[name](aspect)).camera()
        for k in ("origin", "lower_left", "horizontal", "vertical", "u", "v", "w"):
            assert np.array_equal(np.array(getattr(d.camera, k)), cam[k]), (name, k)
    with pytest.raises(ra.api.RtError, match="invalid value 'teapot'"):
        ra.example_scene("teapot", 1.5)


@pytest.mark.parametrize("name", ["two-spheres", "earth", "cornell", "cornell-smoke"])
def test_deterministic_examples_match_the_independent_restatement(ra, name):
    """The four examples that draw no random numbers, restated in oracle/loader_ref.py from example.rs on its own
    (make_box, the Rotated/Translated wrappers turned into world-space quads, the materials, the media): the
    generator's arrays must hold the very same doubles, surface by surface in the reference's order."""
    from oracle import loader_ref as L

    want = L.example_scene(name)
    d = ra.example_scene(name, 1.25)
    mats, sph, quads, surf, media = d.materials, d.spheres, d.quads, d.surfaces, d.media
    tex_names = {ra.RT_TEX_SOLID: "solid", ra.RT_TEX_UV_CHECKER: "uv_checker", ra.RT_TEX_IMAGE: "image"}

    def material(i):
        m = mats[i]
        return (int(m["kind"]), tuple(m["albedo"]), float(m["param"]), tex_names[int(m["texture"])],
                tuple(m["albedo2"]) if m["texture"] != ra.RT_TEX_SOLID else (0.0, 0.0, 0.0),
                float(m["tex_scale"]) if m["texture"] != ra.RT_TEX_SOLID else 0.0)

    assert np.array_equal(d.background, want["background"])
    shown = [s for s in surf if s["type"] != ra.api.RT_SURF_QUAD + 1]  # media are listed after the surfaces
    assert len(shown) == len(want["surfaces"]) and len(media) == len(want["media"])
    for s, w in zip(shown, want["surfaces"]):
        if w[0] == "sphere":
            assert s["type"] == ra.RT_SURF_SPHERE
            g = sph[s["index"]]
            assert np.array_equal(g["center"], w[1]) and g["radius"] == w[2] and not g["moving"]
            assert material(g["material"]) == w[3]
        else:
            assert s["type"] == ra.RT_SURF_QUAD
            g = quads[s["index"]]
            assert np.array_equal(g["q"], w[1]) and np.array_equal(g["u"], w[2]) and np.array_equal(g["v"], w[3]), (name, s["index"])
            assert material(g["material"]) == w[4]
    for k, (box, density, albedo) in enumerate(want["media"]):
        m = media[k]
        assert m["boundary"] == ra.RT_SURF_QUAD and m["count"] == 6 and m["density"] == density
        assert mats[m["material"]]["kind"] == 4 and tuple(mats[m["material"]]["albedo"]) == albedo
        for j, (q, u, v) in enumerate(box):
            g = quads[m["first"] + j]
            assert np.array_equal(g["q"], q) and np.array_equal(g["u"], u) and np.array_equal(g["v"], v), (name, k, j)
        assert surf[len(shown) + k]["type"] == 3 and surf[len(shown) + k]["index"] == k
    # nesting (Surface::Bvh): cornell's two boxes are one-leaf trees inside the scene's list, with the wrapped boxes'
    # bounds; the other three have none
    groups = want.get("groups", [])
    assert len(d.groups) == len(groups)
    if groups:
        assert d.surface_group.tolist() == want["surface_group"]
        for g, (parent, leaf, box) in zip(d.groups, groups):
            assert (int(g["parent"]), int(g["leaf"])) == (parent, leaf) and np.array_equal(g["box"], box), (g, box)
        assert d.surface_build_box is None or (d.surface_build_box[:, 0] > d.surface_build_box[:, 3]).all()
    else:
        assert d.surface_group is None
    if name == "earth":  # the missing earth.png: an image texture without pixels shades the fallback colour
        assert len(d.images) == 1 and d.images[0].size == 0
    # the camera: Camera::builder defaults (no aperture, focus distance |from - towards|), camera.rs:43-74
    fov, frm, to, aperture, focus = want["camera"]
    assert aperture == 0.0 and focus is None and d.camera.lens_radius == 0.0
    assert CAMERAS[name](1.25)[:8] == [fov, 1.25, *frm, *to]


ALL_EXAMPLES = ["one-weekend", "random-spheres", "two-spheres", "two-perlin", "earth", "cornell", "cornell-smoke", "final-scene"]


@pytest.mark.parametrize("name", ALL_EXAMPLES)
def test_examples_nest_their_trees_like_the_reference(ra, ref, name):
    """Surface::Bvh (surfaces.rs:374): a tree as ONE member of another tree's leaf.  one-weekend adds the ground to the
    scene and everything else to a Bvh of its own (example.rs:96-139); make_box is a one-leaf Bvh (surfaces.rs:591), so
    cornell's scene list holds 6 quads + 2 trees; final-scene holds a tree of 400 such boxes and, wrapped in Rotated and
    Translated, a tree of 1000 spheres built where the spheres were drawn (example.rs:474-498, 553-570).  The product's
    build (host only here) and the oracle's — two implementations, the oracle's in the reference's own recursive form —
    must give the same in-order leaf sequence and the same tree figures."""
    aspect = 1.0
    d = ra.example_scene(name, aspect, seed=11)
    s = ra.Scene(d, ra.RT_DEVICE_NONE)
    order, st = ref.RefScene.from_desc(d, CAMERAS[name](aspect)).leaf_order()
    info, mine = s.info(), s.leaf_order()
    assert np.array_equal(mine, order) and sorted(order.tolist()) == list(range(d.n_surfaces))
    assert (info.n_nodes, info.n_leaves, info.max_depth, info.max_leaf_size) == (st["nodes"], st["leaves"], st["depth"], st["max_leaf"])
    g, sg = d.groups, d.surface_group
    if name == "one-weekend":
        assert len(g) == 1 and (g["parent"][0], g["leaf"][0]) == (-1, 0) and sg[0] == -1 and np.all(sg[1:] == 0)
        assert order[0] == 0      # the scene's list is the leaf [ground, tree]: the ground loses every tie to a sphere on it
    elif name == "cornell":
        assert len(g) == 2 and g["leaf"].tolist() == [1, 1] and sg.tolist() == [-1] * 6 + [0] * 6 + [1] * 6
        assert order.tolist() == list(range(18))  # 8 members: one leaf, in listing order, the boxes' leaves in place
        assert (info.n_nodes, info.n_leaves, info.max_depth, info.max_leaf_size) == (3, 3, 2, 8)
    elif name == "final-scene":
        assert len(g) == 402 and g["parent"][0] == -1 and np.all(g["parent"][1:401] == 0) and np.all(g["leaf"][1:401] == 1)
        assert (g["parent"][401], g["leaf"][401]) == (-1, 0) and g["box"][401][0] < g["box"][401][3]  # the turned cluster's box
        pos = np.empty(d.n_surfaces, dtype=np.int64)
        pos[order] = np.arange(d.n_surfaces)
        boxes = pos[:2400].reshape(400, 6)                      # a box's six quads: adjacent and in make_box's order
        assert np.all(np.diff(boxes, axis=1) == 1)
        tree_of_boxes = np.sort(pos[:2400])
        assert tree_of_boxes[-1] - tree_of_boxes[0] == 2399     # the tree of boxes is one run of the sequence
        cluster = np.sort(pos[sg == 401])
        assert len(cluster) == 1000 and cluster[-1] - cluster[0] == 999
        bb = d.surface_build_box
        assert np.allclose(bb[sg == 401][:, 3:] - bb[sg == 401][:, :3], 20.0) and np.all(bb[sg != 401][:, 0] > bb[sg != 401][:, 3])
    else:
        assert len(g) == 0 and sg is None  # random-spheres adds every sphere to the scene itself (example.rs:153-203)
    s.close()
    # --no-bvh makes ONE leaf of the scene's own list (scene/mod.rs:60-65); the trees inside it stay trees
    flat_d = ra.example_scene(name, aspect, seed=11, use_bvh=False)
    flat_s = ra.Scene(flat_d, ra.RT_DEVICE_NONE)
    flat_order, flat_st = ref.RefScene.from_desc(flat_d, CAMERAS[name](aspect), use_bvh=False).leaf_order()
    fi = flat_s.info()
    assert np.array_equal(flat_s.leaf_order(), flat_order)
    assert (fi.n_nodes, fi.n_leaves, fi.max_depth, fi.max_leaf_size) == (flat_st["nodes"], flat_st["leaves"], flat_st["depth"], flat_st["max_leaf"])
    if name == "final-scene":
        assert fi.max_leaf_size == 11 and fi.n_nodes > 400  # the scene's 11 members in one leaf, the nested trees intact
    flat_s.close()


def _tie_scene(ra, nested):
    """20 small quads in the plane y = 0 and, LISTED LAST, one large quad in the same plane under all of them: every ray
    that meets a small quad meets the large one at the very same t.  `nested`: the small quads are a Bvh of their own."""
    from raytracer_amd.api import GROUP_DTYPE, MATERIAL_DTYPE, QUAD_DTYPE, SPHERE_DTYPE, SURFACE_DTYPE, TRIANGLE_DTYPE

    def quad(q, u, v, material):
        r = np.zeros((), dtype=QUAD_DTYPE)
        n = np.cross(u, v)
        normal = n / np.sqrt(n @ n)
        r["q"], r["u"], r["v"], r["normal"], r["d"], r["w"], r["material"] = q, u, v, normal, normal @ np.asarray(q, float), n / (n @ n), material
        return r

    quads = [quad((-2.0 + 0.8 * (k % 5), 0.0, -1.6 + 0.8 * (k // 5)), (0.5, 0.0, 0.0), (0.0, 0.0, 0.5), 0) for k in range(20)]
    quads.append(quad((-50.0, 0.0, -50.0), (100.0, 0.0, 0.0), (0.0, 0.0, 100.0), 1))
    mats = np.zeros(2, dtype=MATERIAL_DTYPE)
    mats["albedo"] = [(0.8, 0.2, 0.2), (0.2, 0.2, 0.8)]
    surfaces = np.array([(2, k) for k in range(21)], dtype=SURFACE_DTYPE)
    cam = ra.example_scene("two-spheres", 1.0).camera  # from (13, 2, 3) towards the origin, fov 20
    groups = np.array([(-1, 0, (1, 0, 0, 0, 0, 0))], dtype=GROUP_DTYPE) if nested else None
    return ra.SceneDesc.from_arrays(mats, np.zeros(0, SPHERE_DTYPE), np.zeros(0, TRIANGLE_DTYPE), np.array(quads), surfaces,
                                    (0.7, 0.8, 1.0), cam, groups=groups, surface_group=[0] * 20 + [-1] if nested else None)


def test_nesting_decides_exact_ties(ra, ref):
    """What the nesting changes: the in-order leaf sequence, hence the winner of an exact equal-t tie (the later one,
    bvh.rs:272-273, 308).  Flat, the reference's stable sort by box minimum puts the large quad (its minimum is the
    smallest) FIRST: a small quad wins the tie.  With the small quads in a Bvh of their own the scene's list is the
    leaf [tree, large quad]: the large quad comes LAST and wins."""
    w = h = 64
    for nested, winner_is_large in ((False, False), (True, True)):
        d = _tie_scene(ra, nested)
        o = ref.RefScene.from_desc(d, CAMERAS["two-spheres"](1.0))
        ids, t = o.primary_hits(w, h)
        s = ra.Scene(d, ra.RT_DEVICE_NONE)
        order = s.leaf_order()
        assert np.array_equal(order, o.leaf_order()[0])
        assert (order[-1] == 20) == winner_is_large and (order[0] == 20) == (not winner_is_large)
        # pixels whose centre ray meets a small quad (the flat case names them; the nested case must name the large quad there)
        if not nested:
            on_small = ids < 20
            assert on_small.sum() > 50 and np.all(ids[~on_small & (ids >= 0)] == 20)
            small_pixels, small_t = on_small & (ids >= 0), t.copy()
        else:
            assert np.all(ids[small_pixels] == 20) and np.array_equal(t[small_pixels], small_t[small_pixels])
        s.close()


@pytest.mark.gpu
def test_nesting_decides_exact_ties_on_the_gpu(ra, ref):
    """The same two scenes through the C ABI: the f64 parity kernel (which resolves ties by the position in the in-order
    sequence) and the f32 render path (which resolves them by rank) both name the oracle's winner."""
    w = h = 64
    for nested in (False, True):
        d = _tie_scene(ra, nested)
        want_ids, want_t = ref.RefScene.from_desc(d, CAMERAS["two-spheres"](1.0)).primary_hits(w, h)
        s = ra.Scene(d, 0)
        ids, t = s.primary_hits(w, h, precision=64)
        assert np.array_equal(ids, want_ids) and np.array_equal(t, want_t)
        ids32, _ = s.primary_hits(w, h, precision=32)
        # f32 may round the two coplanar quads' t apart, but never towards the other winner where f64 ties exactly:
        # both planes are y = 0 with the same normal, so the f32 t's are equal too
        assert np.array_equal(ids32, want_ids)
        s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h,sn", [("one-weekend", 150, 100, 4), ("cornell", 96, 96, 5),
                                         ("random-spheres", 150, 100, 4), ("two-spheres", 120, 80, 4),
                                         ("two-perlin", 120, 80, 4), ("earth", 90, 60, 3),
                                         ("cornell-smoke", 96, 96, 5), ("final-scene", 100, 100, 4)])
def test_example_renders_match_the_oracle(ra, ref, name, w, h, sn):
    from test_gpu_parity import T_REL, _check_image, classify_f32

    d = ra.example_scene(name, w / h, seed=11)
    scene = ra.Scene(d, 0)
    oracle = ref.RefScene.from_desc(d, CAMERAS[name](w / h))
    rid, rt = oracle.primary_hits(w, h, threads=16)
    ids, t = scene.primary_hits(w, h, precision=64)
    assert np.array_equal(ids, rid) and np.array_equal(t, rt)
    ids32, t32 = scene.primary_hits(w, h, precision=32)
    swaps, flips, n_rel, max_rel = classify_f32(ids32, t32, rid, rt)
    # the cornell examples' camera sits on the box's plane of symmetry: pixel-centre rays of the middle column
    # run along the boxes' edges (16 flips and 6 swaps of 9216 pixels measured), as in quads.yaml
    assert flips <= max(4, 3e-3 * rid.size) and swaps <= max(8, 5e-3 * rid.size), (name, swaps, flips)
    # f32 ray directions: a grazing hit on a radius-1000 ground sphere (the horizon row) amplifies their rounding
    assert n_rel <= 1e-3 * rid.size and max_rel <= 10 * T_REL, (name, n_rel, max_rel)
    for sampler in (0, 1):
        want, _ = oracle.render(w, h, sn, threads=16, sampler=sampler)
        for kernel in (0, 1, 2):
            img, st = scene.render(w, h, sn, kernel=kernel, sampler=sampler)
            _check_image(img, want, sn * sn, f"{name} kernel {kernel} sampler {sampler}")
    scene.close()


@pytest.mark.gpu
def test_cli_default_scene_is_one_weekend(tmp_path):
    out = tmp_path / "o.png"
    r = subprocess.run([f"{ROOT}/raytracer_amd/raytracer", "--width", "120", "-n", "4", "--seed", "5", "-o", str(out)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "Done in" in r.stderr and out.stat().st_size > 1000
