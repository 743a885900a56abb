"""The render tree built on the GPU (use_bvh = 2 / RT_BUILD=device; SURVEY §8 f2, hip/bvh_device.hip) against the
host-built one.  Any valid tree over the same primitives finds the same closest hits (the leaf tests do not depend
on the tree, boxes only cull, equal-t ties go by the reference's rank), so the f32 primary hits must be identical
and the images equal up to summation order; the library itself refuses a device-built tree that is not a tree over
exactly the scene's primitives (check_device_tree in api.cpp).  Needs an MI355X:  pytest -m gpu."""
import numpy as np
import pytest

import synth
from conftest import scene_path
from test_gpu_parity import F32_BUDGET, _check_image, _same_image, classify_f32, golden

pytestmark = pytest.mark.gpu


def _pair(ra, path, aspect):
    host = ra.Scene(ra.load_scene(path, aspect), 0)
    dev = ra.Scene(ra.load_scene(path, aspect, use_bvh="device"), 0)
    hi, di = host.info(), dev.info()
    assert hi.device_build == 0 and di.device_build == 1
    assert di.n_surfaces == hi.n_surfaces and di.n_loose == hi.n_loose
    # the reference-topology tree (tie ranks, parity kernel) is the host's in both
    assert (di.n_nodes, di.n_leaves, di.max_depth) == (hi.n_nodes, hi.n_leaves, hi.max_depth)
    assert np.array_equal(host.leaf_order(), dev.leaf_order())
    return host, dev


@pytest.mark.parametrize("name", ["teapot", "lucy", "cornell_lucy", "dual", "quads"])
def test_device_built_tree_finds_the_same_hits(ra, name):
    w, h = 480, 270
    host, dev = _pair(ra, scene_path(name), w / h)
    hi, di = host.info(), dev.info()
    print(f"\n{name}: render tree host {hi.render_nodes} nodes depth {hi.render_depth} in {hi.render_build_ms:.1f} ms; "
          f"device {di.render_nodes} nodes depth {di.render_depth} in {di.render_build_ms:.1f} ms")
    assert di.render_max_leaf <= max(2, hi.render_max_leaf) and di.render_depth <= 56
    # leaves of at most two primitives: a tree over n primitives has between n/2 and n leaves
    assert hi.render_nodes / 2 <= di.render_nodes <= 2 * hi.render_nodes + 2
    ids_h, t_h = host.primary_hits(w, h, precision=32)
    ids_d, t_d = dev.primary_hits(w, h, precision=32)
    assert np.array_equal(ids_h, ids_d) and np.array_equal(t_h, t_d)
    # the f64 parity kernel walks the reference-topology tree: untouched
    a, b = host.primary_hits(w, h, precision=64), dev.primary_hits(w, h, precision=64)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for kernel in (0, 1, 2):
        img_h, st_h = host.render(240, 135, 4, kernel=kernel, collect_stats=True)
        img_d, st_d = dev.render(240, 135, 4, kernel=kernel, collect_stats=True)
        assert st_h.samples == st_d.samples
        assert _same_image(img_h, img_d, 16), (name, kernel)
    host.close(), dev.close()


@pytest.mark.parametrize("name", ["lucy", "cornell_lucy", "dual"])
def test_host_built_trees_do_not_depend_on_the_build_threads(ra, name, monkeypatch):
    """Scene set-up runs on many threads (OBJ blocks parsed beside the inflater, forked subtrees attached as pieces,
    the root's sort and partition on a team) and says the result is the one a single thread produces:
    RT_BUILD_THREADS=1 does all of it on the calling thread, and the two scenes must be the same to the bit — the
    figures of both trees, the reference tree's leaf order, the first hits of both kernels' arithmetic, and a
    one-sample-per-pixel frame (one path per pixel: no summation order to differ by)."""
    w, h = 320, 180
    many = ra.Scene(ra.load_scene(scene_path(name), w / h), 0)
    monkeypatch.setenv("RT_BUILD_THREADS", "1")
    one = ra.Scene(ra.load_scene(scene_path(name), w / h), 0)
    monkeypatch.delenv("RT_BUILD_THREADS")
    mi, oi = many.info(), one.info()
    for field in ("n_surfaces", "n_nodes", "n_leaves", "max_depth", "max_leaf_size", "render_nodes", "render_depth",
                  "render_max_leaf", "n_loose"):
        assert getattr(mi, field) == getattr(oi, field), field
    assert np.array_equal(many.leaf_order(), one.leaf_order())
    for precision in (32, 64):
        a, b = many.primary_hits(w, h, precision=precision), one.primary_hits(w, h, precision=precision)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), precision
    img_m, st_m = many.render(w, h, 1, collect_stats=True)
    img_o, st_o = one.render(w, h, 1, collect_stats=True)
    assert np.array_equal(img_m, img_o)
    # the same tree is walked the same way: not one node visit or triangle test more
    assert (st_m.node_visits, st_m.tri_tests) == (st_o.node_visits, st_o.tri_tests)
    many.close(), one.close()


TREE_SCENES = ["cornell", "cornell_lucy", "teapot", "lucy", "dual"]  # the golden scenes that have a render tree


@pytest.fixture(scope="module")
def device_built(ra):
    cache = {}

    def get(name, aspect):
        key = (name, round(aspect, 6))
        if key not in cache:
            cache[key] = ra.Scene(ra.load_scene(scene_path(name), aspect, use_bvh="device"), 0)
            assert cache[key].info().device_build == 1
        return cache[key]

    yield get
    for s in cache.values():
        s.close()


@pytest.mark.parametrize("name", TREE_SCENES)
def test_device_built_tree_primary_hits_vs_oracle_golden(device_built, name):
    """The device-built tree against the ORACLE's golden vectors directly (not against the host-built tree): the f32
    closest hits within the same budget test_gpu_parity.py holds the host-built tree to."""
    g = golden(name)
    h, w = g["prim_id"].shape
    ids, t = device_built(name, w / h).primary_hits(w, h, precision=32)
    swaps, flips, n_rel, max_rel = classify_f32(ids, t, g["prim_id"], g["t"])
    b_swaps, b_flips, b_rel = F32_BUDGET["golden"][name]
    assert swaps <= b_swaps and flips <= b_flips and n_rel <= b_rel and max_rel <= 2e-4, (swaps, flips, n_rel, max_rel)


@pytest.mark.parametrize("sampler", [0, 1])
@pytest.mark.parametrize("kernel", [0, 1, 2])
@pytest.mark.parametrize("name", TREE_SCENES)
def test_device_built_tree_image_matches_oracle_golden(device_built, name, kernel, sampler):
    """test_image_matches_golden with the render tree built on the GPU: the oracle's image, same bars."""
    g = golden(name)
    want = g["image_sum"] if sampler == 0 else g["image_sum_direct"]
    ih, iw, _ = want.shape
    img, st = device_built(name, iw / ih).render(iw, ih, 4, max_depth=40, seed=1, kernel=kernel, collect_stats=True,
                                                 sampler=sampler)
    _check_image(img, want, 16, name)
    assert st.samples == iw * ih * 16 and 0 < st.segments <= 1.02 * int(g["segments"])


def test_device_build_of_synthetic_scenes(ra, tmp_path):
    """Spheres and quads inside the tree, a degenerate deep tree, scenes small enough for one thread to finish (fewer
    primitives than the small-node threshold: no level pass runs at all), a single triangle."""
    cases = [synth.many_prims(tmp_path), synth.deep_tree(tmp_path, 200), synth.deep_tree(tmp_path, 40, name="small"),
             synth.deep_tree(tmp_path, 6, name="tiny"), synth.deep_tree(tmp_path, 1, name="one")]
    for path in cases:
        w, h = 160, 120
        host, dev = _pair(ra, path, w / h)
        ids_h, t_h = host.primary_hits(w, h, precision=32)
        ids_d, t_d = dev.primary_hits(w, h, precision=32)
        assert np.array_equal(ids_h, ids_d) and np.array_equal(t_h, t_d), path
        img_h, _ = host.render(w, h, 3)
        img_d, _ = dev.render(w, h, 3)
        assert _same_image(img_h, img_d, 9), path
        host.close(), dev.close()


def test_env_switch_and_no_bvh(ra, monkeypatch):
    monkeypatch.setenv("RT_BUILD", "device")
    s = ra.Scene(ra.load_scene(scene_path("teapot"), 1.5), 0)
    assert s.info().device_build == 1
    s.close()
    # --no-bvh keeps its single leaf whatever the switch says
    s = ra.Scene(ra.load_scene(scene_path("quads"), 1.5, use_bvh=False), 0)
    assert s.info().render_nodes <= 1 or s.info().render_depth == 1
    s.close()


def test_cli_device_build_writes_the_same_image(tmp_path):
    import subprocess

    from PIL import Image

    from conftest import ROOT

    cli = f"{ROOT}/raytracer_amd/raytracer"
    args = [cli, "--scene", scene_path("teapot"), "--width", "160", "--height", "90", "-n", "16", "--seed", "5", "--no-progress"]
    a, b = tmp_path / "host.png", tmp_path / "device.png"
    for out, extra in ((a, []), (b, ["--device-build"])):
        r = subprocess.run(args + extra + ["-o", str(out)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
    pa = np.asarray(Image.open(a).convert("RGB"), dtype=np.int32)
    pb = np.asarray(Image.open(b).convert("RGB"), dtype=np.int32)
    assert np.abs(pa - pb).max() <= 1 and (pa != pb).mean() <= 0.01
