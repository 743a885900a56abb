"""Host-side logic of the product library, CPU only: the C ABI loads and exports every declared symbol,
the loader (load.rs) and BVH build (bvh.rs) agree with the oracle, error behaviour mirrors the reference."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, scene_path
from oracle import loader_ref

ALL_SCENES = ["cornell", "cornell_lucy", "teapot", "lucy", "dual", "distant", "light", "quads"]


def test_library_exports_every_declared_symbol(ra):
    lib = ra.lib()
    header = open(os.path.join(ROOT, "include", "rt.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", header))
    assert declared == set(ra.api.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.rt_version()


@pytest.mark.parametrize("name", ALL_SCENES)
def test_loader_matches_oracle_loader_bit_for_bit(ra, name):
    d = ra.load_scene(scene_path(name), 1.5)
    f = loader_ref.load_scene(scene_path(name), 1.5)
    sf = d.surfaces
    assert np.array_equal(np.stack([sf["type"], sf["index"]], 1), f.surfaces)
    tri = d.triangles
    assert len(tri) == len(f.triangles)
    if len(tri):
        assert np.array_equal(tri["v"].reshape(-1, 9), f.triangles[:, :9])           # baked vertices
        assert np.array_equal(tri["n"].reshape(-1, 9), f.triangles[:, 9:18], equal_nan=True)  # Mesh::new normals
        assert np.array_equal(tri["material"], f.triangles[:, 18].astype(np.int32))
    sp = d.spheres
    if len(sp):
        assert np.array_equal(np.column_stack([sp["center"], sp["radius"], sp["material"]]), f.spheres)
    q = d.quads
    if len(q):
        assert np.array_equal(np.column_stack([q["q"], q["u"], q["v"], q["material"]]), f.quads)
    m = d.materials
    assert np.array_equal(np.column_stack([m["kind"], m["albedo"], m["param"]]), f.materials)
    assert np.array_equal(d.background, f.background)


def test_loader_survey_appendix_a_numbers(ra):
    """bounding boxes of the baked meshes (computed from the data files)."""
    cases = {
        "teapot": ((-3.21700001, -4.768e-08, -2), (3.21700001, 3.15000005, 2), 6320),
        "lucy": ((-1.86024805, -1.90970602, 0.18949188), (1.86024805, 4.47918192, 2.32456787), 99970),
        "cornell": ((147.78263672, 1.38892212, 205.27234039), (408.21736328, 448.61107788, 354.72765961), 99970),
    }
    for name, (lo, hi, n) in cases.items():
        v = ra.load_scene(scene_path(name), 1.5).triangles["v"].reshape(-1, 3)
        assert len(v) == 3 * n
        assert np.allclose(v.min(0), lo, rtol=0, atol=5e-8)
        assert np.allclose(v.max(0), hi, rtol=0, atol=5e-8)
    dual = ra.load_scene(scene_path("dual"), 1.5).triangles["v"]
    assert np.allclose(dual[:6320].reshape(-1, 3).min(0), (-3.217, -4.768e-08, -5), atol=1e-6)
    assert np.allclose(dual[6320:].reshape(-1, 3).max(0), (3.65100002, 3.15000005, 5), atol=1e-6)


def test_hex_albedo_divides_by_256(ra):
    m = ra.load_scene(scene_path("teapot"), 1.5).materials
    assert np.array_equal(m["albedo"][1], np.array([0x33, 0x33, 0x33]) / 256.0)   # '333333'
    assert np.array_equal(m["albedo"][2], np.array([0x1a, 0x49, 0xa9]) / 256.0)   # 1a49a9
    assert m["kind"].tolist() == [0, 1, 0, 2, 1, 1, 0] and m["param"][3] == 1.5 and m["param"][1] == 0.2


@pytest.mark.parametrize("name", ALL_SCENES)
@pytest.mark.parametrize("use_bvh", [True, False])
def test_bvh_matches_reference_topology(ra, ref, name, use_bvh):
    if not use_bvh and name not in ("teapot", "quads", "distant"):
        pytest.skip("single-leaf case covered on three scenes")
    d = ra.load_scene(scene_path(name), 1.5, use_bvh)
    s = ra.Scene(d, ra.RT_DEVICE_NONE)
    o = ref.RefScene.from_yaml(scene_path(name), 1.5, use_bvh)
    order, st = o.leaf_order()
    info = s.info()
    assert np.array_equal(s.leaf_order(), order)      # same in-order leaf sequence => same ties
    assert (info.n_nodes, info.n_leaves, info.max_depth, info.max_leaf_size) == (
        st["nodes"], st["leaves"], st["depth"], st["max_leaf"])
    assert sorted(order.tolist()) == list(range(d.n_surfaces))  # every surface in exactly one leaf


def test_camera_matches_oracle(ra, ref):
    for name, aspect in (("lucy", 16 / 9), ("cornell", 1.0), ("teapot", 1.5)):
        c = ra.load_scene(scene_path(name), aspect).camera
        oc = ref.RefScene.from_yaml(scene_path(name), aspect).camera()
        for k in ("origin", "lower_left", "horizontal", "vertical", "u", "v", "w"):
            assert np.array_equal(np.array(list(getattr(c, k))), oc[k]), (name, k)
        assert c.lens_radius == oc["lens_radius"]
    assert ra.load_scene(scene_path("lucy"), 1.5).camera.lens_radius == 0.05  # aperture 0.1 / 2


def test_reference_rejects_its_own_dual_and_distant(ra):
    """The untouched reference files lack the required `background` (load.rs:379): hard error there and here."""
    for f in ("dual_original.yaml", "distant_original.yaml"):
        with pytest.raises(ra.api.RtError, match="missing field `background`"):
            ra.load_scene(os.path.join(ROOT, "tests", "golden", "invalid", f), 1.5)


def _write(tmp_path, text, name="s.yaml"):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


CAM = "camera:\n  fov: 40\n  from: [0, 0, 5]\n  towards: [0, 0, 0]\n"


def test_loader_errors(ra, tmp_path):
    with pytest.raises(ra.api.RtError, match="load scene file"):
        ra.load_scene(str(tmp_path / "nope.yaml"), 1.5)
    with pytest.raises(ra.api.RtError, match="scene is empty"):           # load.rs:50-52
        ra.load_scene(_write(tmp_path, CAM + "scene:\n  background: [0,0,0]\n  surfaces: []\n"), 1.5)
    bad_hex = CAM + ("scene:\n  background: [0,0,0]\n  surfaces:\n    - type: sphere\n      radius: 1\n"
                     "      position: [0,0,0]\n      material: {type: lambertian, albedo: abc}\n")
    with pytest.raises(ra.api.RtError, match="expected hex color format aabbcc"):  # load.rs:475-477
        ra.load_scene(_write(tmp_path, bad_hex), 1.5)
    digits = bad_hex.replace("albedo: abc", "albedo: 333333")  # unquoted digits are an integer to serde_yaml
    with pytest.raises(ra.api.RtError, match="albedo"):
        ra.load_scene(_write(tmp_path, digits), 1.5)
    unknown = bad_hex.replace("type: sphere", "type: torus")
    with pytest.raises(ra.api.RtError, match="unknown variant `torus`"):
        ra.load_scene(_write(tmp_path, unknown), 1.5)
    mesh = CAM + ("scene:\n  background: [0,0,0]\n  surfaces:\n    - type: mesh\n      path: ./missing.obj\n"
                  "      material: {type: lambertian, albedo: [1,1,1]}\n")
    with pytest.raises(ra.api.RtError, match="could not load mesh at"):   # load.rs:68-69
        ra.load_scene(_write(tmp_path, mesh), 1.5)


def test_yaml_subset_forms(ra, tmp_path):
    """Appendix C forms: flow maps, quoted scalars, '.12' / '-.45' numbers, comments, dashes at key indent."""
    text = ("# leading comment\ncamera: {fov: 20.0, from: [-2, 2, 1], towards: [0, 0, -1], aperture: .1}\n"
            "scene:\n  background: [.7, .8, 1.0]   # sky\n  surfaces:\n"
            "  - type: sphere\n    radius: -.45\n    position: [ -1,0, -1 ]\n"
            "    material: { type: \"dielectric\", index: 1.5 }\n"
            "  - type: quad\n    position: [0,0,0]\n    sides: [[1,0,0],\n            [0,1,0]]\n"
            "    material:\n      type: metal\n      albedo: 'ae8975'\n      fuzz: 0\n")
    d = ra.load_scene(_write(tmp_path, text), 2.0)
    assert d.spheres["radius"][0] == -0.45 and d.camera.lens_radius == 0.05
    assert d.quads["v"][0].tolist() == [0, 1, 0]
    assert d.materials["kind"].tolist() == [2, 1]
    assert np.array_equal(d.background, [0.7, 0.8, 1.0])


def test_box_expands_to_six_transformed_quads(ra, tmp_path):
    text = CAM + ("scene:\n  background: [0,0,0]\n  surfaces:\n    - type: box\n      corners: [[1,2,3],[0,0,0]]\n"
                  "      material: {type: lambertian, albedo: [1,1,1]}\n      transform:\n"
                  "        - {type: rotate, axis: y, angle: 90}\n        - {type: translate, dir: [10, 0, 0]}\n")
    p = _write(tmp_path, text)
    d = ra.load_scene(p, 1.0)
    f = loader_ref.load_scene(p, 1.0)
    q = d.quads
    assert len(q) == 6
    assert np.array_equal(np.column_stack([q["q"], q["u"], q["v"], q["material"]]), f.quads)
    pts = np.concatenate([q["q"], q["q"] + q["u"] + q["v"]])
    assert np.allclose(pts.min(0), [9, 0, 1], atol=1e-12) and np.allclose(pts.max(0), [12, 2, 2], atol=1e-12)


def test_obj_parsing(ra, tmp_path):
    (tmp_path / "m.obj").write_text("# c\no first\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 9 9 9\nvt 0 0\n"
                                    "f 3/1 1/1 2/1\nf -2 -4 -3\no second\nv 5 5 5\nf 1 2 5\n")
    text = CAM + ("scene:\n  background: [0,0,0]\n  surfaces:\n    - type: mesh\n      path: m.obj\n"
                  "      material: {type: lambertian, albedo: [1,1,1]}\n")
    d = ra.load_scene(_write(tmp_path, text), 1.0)
    assert len(d.triangles) == 2  # first model only (load.rs:225-229); the unreferenced vertex 4 is dropped
    v = d.triangles["v"]
    # re-centred on the bbox centre of the referenced vertices (0.5, 0.5, 0)
    assert np.array_equal(v[0], np.array([[0, 1, 0], [0, 0, 0], [1, 0, 0]]) - [0.5, 0.5, 0])
    assert np.array_equal(v[1], v[0])  # negative (relative) indices resolve to the same vertices
    (tmp_path / "quad.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n")
    with pytest.raises(ra.api.RtError, match="not divisible by 3"):
        ra.load_scene(_write(tmp_path, text.replace("m.obj", "quad.obj")), 1.0)


def _long_obj(n_faces=30000, second_model_at=None, bad_line_at=None):
    """An OBJ of `n_faces` triangles in well over three 256 KB blocks of text, vertices and faces interleaved, indices
    absolute and relative by turns, comments and `vt` lines in between; optionally an object statement (and a second
    model after it) or a malformed vertex at a given face."""
    rng = np.random.default_rng(11)
    lines, n_v = ["# long mesh", "o long", "g part"], 0
    for f in range(n_faces):
        if f == second_model_at:
            lines.append("o second")
        pts = rng.uniform(-3, 3, (3, 3))
        for k, p in enumerate(pts):
            lines.append("v %.9g %.9g %.9g" % tuple(p) if not (f == bad_line_at and k == 1) else "v 1.0 oops 2.0")
        n_v += 3
        if f % 7 == 0:
            lines.append("vt 0.5 0.5")
        if f % 11 == 0:
            lines.append("# face %d" % f)
        # a corner of some earlier face every now and then, so that blocks refer back across their borders
        far = 1 + int(rng.integers(0, n_v - 2)) if f % 5 == 0 else n_v - 2
        lines.append("f %d %d %d" % (far, n_v - 1, n_v) if f % 2 else "f %d -2 -1" % (far - n_v - 1))
    return "\n".join(lines) + "\n"


def test_obj_blocks_parsed_beside_the_inflater_match_the_independent_loader(ra, tmp_path):
    """The reader parses a large file block by block while it is being inflated (obj_reader.cpp): relative indices,
    line numbers and the first model's end must come out as if the file had been read line by line — checked against
    oracle/loader_ref.py's sequential reader, gzipped and plain, with the second model starting in a late block."""
    import gzip
    text = CAM + ("scene:\n  background: [0,0,0]\n  surfaces:\n    - type: mesh\n      path: m.obj\n"
                  "      material: {type: lambertian, albedo: [1,1,1]}\n")
    wants = {}
    for second in (None, 26000):
        obj = _long_obj(second_model_at=second)
        assert len(obj) > 4 * (256 << 10)
        (tmp_path / "m.obj").write_text(obj)
        (tmp_path / "m.obj.gz").write_bytes(gzip.compress(obj.encode(), 6))
        want = loader_ref.load_scene(_write(tmp_path, text), 1.0).triangles
        assert len(want) == (second or 30000)
        wants[second] = want
        for name in ("m.obj", "m.obj.gz"):
            got = ra.load_scene(_write(tmp_path, text.replace("m.obj", name)), 1.0).triangles
            assert len(got) == len(want), name
            assert np.array_equal(got["v"].reshape(-1, 9), want[:, :9]), name
            assert np.array_equal(got["n"].reshape(-1, 9), want[:, 9:18]), name
    # two gzip members in one file (the trailer then understates the size) read as one text
    obj = _long_obj()
    half = obj.index("\n", len(obj) // 2) + 1
    (tmp_path / "m.obj.gz").write_bytes(gzip.compress(obj[:half].encode()) + gzip.compress(obj[half:].encode()))
    got = ra.load_scene(_write(tmp_path, text.replace("m.obj", "m.obj.gz")), 1.0).triangles
    assert len(got) == 30000 and np.array_equal(got["v"].reshape(-1, 9), wants[None][:, :9])
    # a malformed line in a late block is reported with its line number in the file; after the first model's end it is not read
    bad = _long_obj(bad_line_at=25000)
    line = 1 + bad[:bad.index("oops")].count("\n")
    (tmp_path / "m.obj").write_text(bad)
    with pytest.raises(ra.api.RtError, match=r"bad 'v' at line %d\b" % line):
        ra.load_scene(_write(tmp_path, text), 1.0)
    (tmp_path / "m.obj").write_text(_long_obj(second_model_at=20000, bad_line_at=25000))
    assert len(ra.load_scene(_write(tmp_path, text), 1.0).triangles) == 20000
    # a truncated gzip file is an error, not a shorter mesh
    whole = gzip.compress(obj.encode())
    (tmp_path / "m.obj.gz").write_bytes(whole[: len(whole) // 2])
    with pytest.raises(ra.api.RtError, match="could not read"):
        ra.load_scene(_write(tmp_path, text.replace("m.obj", "m.obj.gz")), 1.0)


def test_obj_bare_group_line_ends_the_first_model(ra, tmp_path):
    """tobj finds the keyword with split_whitespace: a bare `g` (or `o`) line — `g\\r\\n` in a CRLF file included — is
    an object/group statement, so after a face it ends the first model (load.rs:225-229 uses the first model only).
    Product loader and the independent loader agree, and neither merges the second model's faces in."""
    crlf = "v 0 0 0\r\nv 1 0 0\r\nv 0 1 0\r\nf 1 2 3\r\ng\r\nv 5 5 5\r\nv 6 5 5\r\nv 5 6 5\r\nf 4 5 6\r\n"
    (tmp_path / "crlf.obj").write_bytes(crlf.encode())
    (tmp_path / "tabs.obj").write_text("\tv 0 0 0\nv\t1 0 0\nv 0 1 0\nf 1 2 3\n  o  \nv 5 5 5\nf 2 3 4\n")
    text = CAM + ("scene:\n  background: [0,0,0]\n  surfaces:\n    - type: mesh\n      path: crlf.obj\n"
                  "      material: {type: lambertian, albedo: [1,1,1]}\n")
    for name in ("crlf.obj", "tabs.obj"):
        path = _write(tmp_path, text.replace("crlf.obj", name))
        d = ra.load_scene(path, 1.0)
        f = loader_ref.load_scene(path, 1.0)
        assert len(d.triangles) == 1 and len(f.triangles) == 1, name
        assert np.array_equal(d.triangles["v"].reshape(-1, 9), f.triangles[:, :9])
        assert np.abs(d.triangles["v"]).max() <= 0.5  # re-centred on the first model's own box, not the merged one


def test_reference_tree_does_not_depend_on_the_build_threads(ra, monkeypatch):
    """The threaded set-up (forked subtrees attached as pieces, team sort at the root, OBJ blocks) against the same
    work on the calling thread alone (RT_BUILD_THREADS=1): same triangles, same reference-topology tree."""
    path = scene_path("lucy")
    many_desc = ra.load_scene(path, 1.5)
    many = ra.Scene(many_desc, ra.RT_DEVICE_NONE)
    monkeypatch.setenv("RT_BUILD_THREADS", "1")
    one_desc = ra.load_scene(path, 1.5)
    one = ra.Scene(one_desc, ra.RT_DEVICE_NONE)
    assert np.array_equal(many_desc.triangles["v"], one_desc.triangles["v"])
    assert np.array_equal(many_desc.triangles["n"], one_desc.triangles["n"])
    mi, oi = many.info(), one.info()
    for field in ("n_nodes", "n_leaves", "max_depth", "max_leaf_size", "render_nodes", "render_depth", "render_max_leaf"):
        assert getattr(mi, field) == getattr(oi, field), field
    assert np.array_equal(many.leaf_order(), one.leaf_order())


def test_a_forked_child_builds_scenes_with_a_pool_of_its_own(ra):
    """The set-up's tasks run on a pool of threads that sleeps between scenes (host/worker.cpp).  fork() copies none of
    them into the child: the child must start its own at its first task — not queue work for threads that do not
    exist, nor wait on the parent's condition variables — and both processes must end without joining anything."""
    import os
    ra.Scene(ra.load_scene(scene_path("teapot"), 1.5), ra.RT_DEVICE_NONE)  # the parent's pool exists and is asleep
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        try:
            s = ra.Scene(ra.load_scene(scene_path("lucy"), 1.5), ra.RT_DEVICE_NONE)
            os.write(w, str(s.info().n_nodes).encode())
        finally:
            os._exit(0)
    os.close(w)
    import select
    ready, _, _ = select.select([r], [], [], 120)
    assert ready, "the forked child hangs"
    assert os.read(r, 64) == b"6227"
    os.waitpid(pid, 0)
    os.close(r)
    # and the parent's pool still works
    assert ra.Scene(ra.load_scene(scene_path("lucy"), 1.5), ra.RT_DEVICE_NONE).info().n_nodes == 6227


def test_scenes_set_up_from_several_threads_at_once(ra):
    """Callers on several host threads (ctypes releases the GIL) share the set-up's one pool: their tasks queue side
    by side, every waiter helps with whatever is queued — and each gets the scene it would have got alone."""
    import hashlib
    import threading

    def signature(name):
        s = ra.Scene(ra.load_scene(scene_path(name), 16 / 9), ra.RT_DEVICE_NONE)
        i = s.info()
        return (i.n_nodes, i.n_leaves, i.render_nodes, i.render_depth, hashlib.sha1(s.leaf_order().tobytes()).hexdigest())

    names = ("lucy", "teapot", "dual", "cornell_lucy")
    alone = {n: signature(n) for n in names}
    wrong = []

    def caller(k):
        for it in range(4):
            n = names[(k + it) % len(names)]
            if signature(n) != alone[n]:
                wrong.append((k, it, n))

    threads = [threading.Thread(target=caller, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not wrong


def test_host_only_scene_refuses_to_render(ra):
    s = ra.Scene(ra.load_scene(scene_path("quads"), 1.5), ra.RT_DEVICE_NONE)
    with pytest.raises(ra.api.RtError, match="host-only"):
        s.render(8, 8, 1)
    with pytest.raises(ra.api.RtError, match="host-only"):
        s.primary_hits(8, 8)


def test_scene_desc_validation(ra):
    d = ra.load_scene(scene_path("quads"), 1.5)
    sf = d.surfaces.copy()
    sf["index"][0] = 99
    bad = ra.SceneDesc.from_arrays(d.materials, d.spheres, d.triangles, d.quads, sf, d.background, d.camera)
    with pytest.raises(ra.api.RtError, match="out of range"):
        ra.Scene(bad, ra.RT_DEVICE_NONE)


def test_png_writer_roundtrip(ra, tmp_path):
    from PIL import Image

    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    p = str(tmp_path / "o.png")
    ra.write_png(p, img)
    back = Image.open(p)
    assert back.mode == "RGB" and back.size == (53, 37)
    assert np.array_equal(np.asarray(back), img)  # row 0 = top (main.rs:158)


def test_cli_flags(ra):
    exe = os.path.join(ROOT, "raytracer_amd", "raytracer")
    r = subprocess.run([exe, "--height", "10", "--aspect-ratio", "2.0", "--scene", "x"], capture_output=True, text=True)
    assert r.returncode == 2 and "cannot be used with" in r.stderr       # conflicts_with, main.rs:44
    r = subprocess.run([exe, "--bogus"], capture_output=True, text=True)
    assert r.returncode == 2 and "unexpected argument" in r.stderr
    r = subprocess.run([exe, "--help"], capture_output=True, text=True)
    for flag in ("--num-samples", "--output", "--threads", "--width", "--height", "--aspect-ratio", "--seed",
                 "--scene", "--example", "--no-progress", "--no-bvh"):
        assert flag in r.stdout
    r = subprocess.run([exe, "--scene", "/nonexistent.yaml"], capture_output=True, text=True)
    assert r.returncode == 1 and "load scene file '/nonexistent.yaml'" in r.stderr   # main.rs:101


def test_render_tree_depth_is_capped_and_small_prims_move_into_the_tree(ra, tmp_path):
    import synth

    deep = ra.Scene(ra.load_scene(synth.deep_tree(tmp_path, 3000, 0.5), 1.5), ra.RT_DEVICE_NONE).info()
    assert 41 <= deep.render_depth <= 56  # SAH alone would go ~1500 levels deep on this cluster
    many = ra.Scene(ra.load_scene(synth.many_prims(tmp_path), 1.5), ra.RT_DEVICE_NONE).info()
    assert many.n_loose == 1 and many.n_surfaces == 1 + 28 + 2 + 5 + 200  # the ground sphere, a giant, stays loose
    room = ra.Scene(ra.load_scene(synth.room(tmp_path), 1.5), ra.RT_DEVICE_NONE).info()
    assert room.n_loose == 6 and room.n_surfaces == 6 + 20 + 4  # the walls stay loose, the 20 small spheres do not


def test_obj_normals_and_first_model_only(ra, tmp_path):
    """An OBJ with `vn` normals and two objects (load.rs:225-229, 300-312): the loader takes the first
    model only, uses the file's normals as they are — looked up by position index (surfaces.rs:231-243), not turned
    by the rotate transform, which moves vertices only — and agrees with the independent loader bit for bit."""
    obj = ("o first\n"
           "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0.25\n"
           "vn 0 0 1\nvn 0.1 0 0.9\nvn 0 0.2 0.8\nvn -0.3 0 0.7\n"
           "f 1//1 2//2 3//3\nf 1//1 3//3 4//4\n"
           "o second\n"
           "v 5 5 5\nv 6 5 5\nv 6 6 5\nvn 1 0 0\n"
           "f 5//5 6//5 7//5\n")
    (tmp_path / "m.obj").write_text(obj)
    (tmp_path / "s.yaml").write_text(
        "camera:\n  fov: 40\n  from: [0, 0, 5]\n  towards: [0, 0, 0]\n"
        "scene:\n  background: [0.5, 0.5, 0.5]\n  surfaces:\n"
        "    - type: mesh\n      path: m.obj\n      material: {type: lambertian, albedo: [0.5, 0.5, 0.5]}\n"
        "      transform:\n        - {type: rotate, axis: y, angle: 30}\n        - {type: translate, dir: [0, 1, 0]}\n")
    d = ra.load_scene(str(tmp_path / "s.yaml"), 1.0)
    f = loader_ref.load_scene(str(tmp_path / "s.yaml"), 1.0)
    tri = d.triangles
    assert len(tri) == 2 and len(f.triangles) == 2                                   # the second object is not loaded
    assert np.array_equal(tri["v"].reshape(-1, 9), f.triangles[:, :9])
    assert np.array_equal(tri["n"].reshape(-1, 9), f.triangles[:, 9:18])
    want = np.array([[0, 0, 1], [0.1, 0, 0.9], [0, 0.2, 0.8], [-0.3, 0, 0.7]], dtype=np.float32).astype(np.float64)
    assert np.array_equal(tri["n"][0], want[[0, 1, 2]]) and np.array_equal(tri["n"][1], want[[0, 2, 3]])
    assert not np.allclose(tri["v"][0][1], [1, 0, 0])                                # the vertices did turn
