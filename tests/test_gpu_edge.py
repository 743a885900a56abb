"""Edge cases of the render path that no shipped scene reaches, on synthetic scene files (tests/synth.py):
spheres and quads living in the render tree (scenes with more than 16 of them beside a mesh, 32 without one), trees deeper than the
smallest traversal-stack build, --no-bvh renders, ranks without strata, 1 spp and odd frame sizes.
Same bars as test_gpu_parity.py.  Needs an MI355X:  pytest -m gpu."""
import numpy as np
import pytest

import synth
from conftest import ROOT, scene_path
from test_gpu_parity import T_REL, T_REL_MAX, _check_image, _same_image, classify_f32

pytestmark = pytest.mark.gpu


def _primary_parity(scene, oracle, w, h):
    rid, rt = oracle.primary_hits(w, h, threads=16)
    ids, t = scene.primary_hits(w, h, precision=64)
    assert np.array_equal(ids, rid) and np.array_equal(t, rt)
    ids32, t32 = scene.primary_hits(w, h, precision=32)
    swaps, flips, n_rel, max_rel = classify_f32(ids32, t32, rid, rt)
    assert flips <= max(4, 1e-3 * rid.size) and swaps <= max(8, 5e-3 * rid.size), (swaps, flips)
    assert n_rel <= 4 and max_rel <= T_REL_MAX, (n_rel, max_rel)


def test_spheres_and_quads_inside_the_render_tree(ra, ref, tmp_path):
    path = synth.many_prims(tmp_path)
    w, h, sn = 120, 80, 4
    scene = ra.Scene(ra.load_scene(path, w / h), 0)
    info = scene.info()
    # 36 spheres+quads > 16 beside a mesh: in the tree, all but the ground sphere (a giant stays loose, api.cpp loose_surfaces)
    assert info.n_loose == 1 and info.n_surfaces > 200
    oracle = ref.RefScene.from_yaml(path, w / h)
    _primary_parity(scene, oracle, w, h)
    for sampler in (0, 1):
        want, _ = oracle.render(w, h, sn, threads=16, sampler=sampler)
        for kernel in (0, 1, 2):
            img, st = scene.render(w, h, sn, kernel=kernel, sampler=sampler, collect_stats=True)
            _check_image(img, want, sn * sn, f"many_prims kernel {kernel} sampler {sampler}")
            assert st.sphere_tests > 0 and st.quad_tests > 0 and st.tri_tests > 0
    scene.close()


def test_spheres_only_render_tree(ra, ref, tmp_path):
    """A tree of nothing but spheres — with the hollow glass ball: a negative radius behind its leaf's gate, Q8 —
    runs a build of its own (TREE = 2), like the triangles-only mesh scenes and the quads-only one below."""
    path = synth.many_prims(tmp_path, n_spheres=40, n_quads=0, with_mesh=False)
    w, h, sn = 120, 80, 4
    scene = ra.Scene(ra.load_scene(path, w / h), 0)
    info = scene.info()
    assert info.n_surfaces == 43 and info.n_loose == 1  # more than 32 and no mesh: a tree (the ground stays loose)
    oracle = ref.RefScene.from_yaml(path, w / h)
    _primary_parity(scene, oracle, w, h)
    for sampler in (0, 1):
        want, _ = oracle.render(w, h, sn, threads=16, sampler=sampler)
        img, st = scene.render(w, h, sn, sampler=sampler, collect_stats=True)
        _check_image(img, want, sn * sn, f"spheres only, sampler {sampler}")
        assert st.tri_tests == 0 and st.quad_tests == 0 and st.sphere_tests > 0
    scene.close()


@pytest.mark.parametrize("n,min_depth,max_depth", [(120, 25, 32), (300, 33, 40), (3000, 41, 56)])
def test_deep_render_trees(ra, ref, tmp_path, n, min_depth, max_depth):
    """Depth 28 runs the 32-level stack build, depth 38 the 40-level one, depth 43 the 56-level one (which also makes
    the pool kernel fall back)."""
    path = synth.deep_tree(tmp_path, n, 0.5)
    w, h, sn = 96, 64, 3
    scene = ra.Scene(ra.load_scene(path, w / h), 0)
    assert min_depth <= scene.info().render_depth <= max_depth
    oracle = ref.RefScene.from_yaml(path, w / h)
    _primary_parity(scene, oracle, w, h)
    want, _ = oracle.render(w, h, sn, threads=16, sampler=1)
    for kernel in (0, 1, 2):
        img, _ = scene.render(w, h, sn, kernel=kernel)
        _check_image(img, want, sn * sn, f"deep_tree({n}) kernel {kernel}")
    scene.close()


def test_no_bvh_render_equals_bvh_render(ra):
    """--no-bvh (main.rs:60-65): one leaf with every surface; the closest hit and hence every path is the same."""
    w, h = 64, 36
    a = ra.Scene(ra.load_scene(scene_path("teapot"), w / h, True), 0)
    b = ra.Scene(ra.load_scene(scene_path("teapot"), w / h, False), 0)
    assert b.info().render_max_leaf >= 6320
    for kernel in (0, 1, 2):
        ia, _ = a.render(w, h, 2, kernel=kernel)
        ib, _ = b.render(w, h, 2, kernel=kernel)
        assert _same_image(ia, ib, 4)
    a.close(), b.close()


def test_rank_without_strata_renders_zeros(ra):
    """8 ranks, 4 strata: ranks 4..7 have nothing to do and must contribute an all-zero framebuffer."""
    s = ra.Scene(ra.load_scene(scene_path("quads"), 1.5), 0)
    for kernel in (0, 1, 2):
        img, st = s.render(60, 40, 2, kernel=kernel, stratum_begin=5, stratum_step=8)
        assert st.samples == 0 and not img.any()
        part = [s.render(60, 40, 2, kernel=kernel, stratum_begin=g, stratum_step=8)[0] for g in range(8)]
        full, _ = s.render(60, 40, 2, kernel=kernel)
        assert np.allclose(sum(p.astype(np.float64) for p in part), full, rtol=1e-5, atol=1e-6)
    s.close()


@pytest.mark.parametrize("w,h,sn", [(13, 7, 1), (1, 1, 3), (8, 8, 1), (257, 3, 2)])
def test_small_and_odd_frames(ra, ref, w, h, sn):
    s = ra.Scene(ra.load_scene(scene_path("distant"), w / h), 0)
    want, _ = ref.RefScene.from_yaml(scene_path("distant"), w / h).render(w, h, sn, threads=4, sampler=1)
    for kernel in (0, 1, 2):
        img, st = s.render(w, h, sn, kernel=kernel)
        assert st.samples == w * h * sn * sn and img.shape == (h, w, 3)
        got, ref_img = img.astype(np.float64) / (sn * sn), want / (sn * sn)
        # few samples: compare sample by sample (same streams), allowing the rare f32/f64 path flip
        close = np.all(np.abs(got - ref_img) <= 1e-3 * np.maximum(ref_img, 1.0), axis=-1)
        assert close.mean() >= 0.98 or (w * h <= 64 and (~close).sum() <= 1)
    s.close()


def test_repeated_renders_of_different_sizes_share_one_scene(ra):
    """The scene keeps its scratch accumulator between calls and grows it on demand."""
    s = ra.Scene(ra.load_scene(scene_path("quads"), 1.5), 0)
    a1, _ = s.render(60, 40, 2)
    big, _ = s.render(300, 200, 1)
    a2, _ = s.render(60, 40, 2)
    assert big.shape == (200, 300, 3) and np.isfinite(big).all()
    assert np.allclose(a1, a2, rtol=1e-5, atol=1e-6)
    s.close()


def test_baseline_config_c1_in_full_on_both_sides(ra, ref):
    """BASELINE.json configs[0] / SURVEY §8 C1: cornell.yaml 400x400 at 100 spp, the one configuration small
    enough to run in full through the oracle (~5 s on 16 host threads) and the GPU."""
    w = h = 400
    s = ra.Scene(ra.load_scene(scene_path("cornell"), 1.0), 0)
    o = ref.RefScene.from_yaml(scene_path("cornell"), 1.0)
    for sampler in (0, 1):
        want, _ = o.render(w, h, 10, threads=16, sampler=sampler)
        img, st = s.render(w, h, 10, sampler=sampler)
        assert st.samples == 16_000_000
        _check_image(img, want, 100, f"C1 sampler {sampler}")
    s.close()


def test_4k_frame_and_eight_strata_slices(ra):
    """SURVEY §8 C4 geometry (cornell_lucy.yaml 3840x2160, 8 GPUs) at reduced spp: the frame renders, stays
    finite, and the eight ranks' strata slices add up to the single-GPU frame."""
    s = ra.Scene(ra.load_scene(scene_path("cornell_lucy"), 3840 / 2160), 0)
    full, st = s.render(3840, 2160, 4)
    assert st.samples == 3840 * 2160 * 16 and np.isfinite(full).all() and full.min() >= 0
    parts = sum(s.render(3840, 2160, 4, stratum_begin=g, stratum_step=8)[0].astype(np.float64) for g in range(8))
    assert np.allclose(parts, full, rtol=1e-5, atol=1e-5)
    s.close()


def test_very_high_sample_counts_per_pixel(ra):
    """sn >= 1600 leaves the multiply-high stratum split for a division (sn^3 no longer fits 32 bits), and the
    hand-out then runs to 250 000 chunks per tile: the image must be the same estimate as at sn = 1500, and a
    strata slice of it must hold its share of the samples."""
    s = ra.Scene(ra.load_scene(scene_path("quads"), 1.0), 0)
    a, sa = s.render(8, 8, 1500)
    b, sb = s.render(8, 8, 2000)
    assert sa.samples == 64 * 1500 * 1500 and sb.samples == 64 * 2000 * 2000
    ma, mb = a.astype(np.float64) / 1500 ** 2, b.astype(np.float64) / 2000 ** 2
    assert np.isfinite(mb).all()
    assert np.abs(ma - mb).max() <= 0.01 * max(ma.max(), 1e-6)  # ~2e6 samples per pixel on both sides
    c, sc = s.render(8, 8, 2000, stratum_begin=3, stratum_step=8)
    assert sc.samples == 64 * (2000 * 2000 // 8)
    mc = c.astype(np.float64) / (2000 * 2000 // 8)
    assert np.abs(mc - mb).max() <= 0.02 * max(mb.max(), 1e-6)
    s.close()


def test_library_before_torch_shares_one_hip_runtime():
    """A fresh process that loads librt_amd.so first and imports torch afterwards must still see the GPU from
    torch (api._share_torch_hip_runtime): two HIP runtimes in one process leave torch without devices."""
    import subprocess
    import sys

    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import raytracer_amd as ra\n"
        "s = ra.Scene(ra.load_scene('scenes/quads.yaml', 1.5), 0)\n"
        "img, st = s.render(60, 40, 2)\n"
        "import torch\n"
        "fb = torch.zeros((40, 60, 3), dtype=torch.float32, device='cuda:0')\n"
        "s.render_device(fb.data_ptr(), 60, 40, 2, stream=torch.cuda.current_stream().cuda_stream)\n"
        "import numpy as np\n"
        "assert np.allclose(fb.cpu().numpy(), img, rtol=1e-5, atol=1e-6)\n"
        "print('ok')\n"
    ) % ROOT
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_frame_pipeline_with_rccl_work_objects():
    """bench.py's FramePipeline on the GPU as several ranks run it — a stream per framebuffer, renders enqueued
    without host synchronisation (the next frame's render beside this frame's drain), real RCCL work objects (a
    one-rank group: the box has one GPU), stream-ordered wait, finish: every finished frame equals a plain render
    of the same strata slice."""
    import subprocess
    import sys

    code = (
        "import os, sys; sys.path.insert(0, %r)\n"
        "import numpy as np, torch, torch.distributed as dist\n"
        "import raytracer_amd as ra, bench\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29571')\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "s = ra.Scene(ra.load_scene('scenes/teapot.yaml', 1.5), 0)\n"
        "want, _ = s.render(192, 128, 8, stratum_begin=1, stratum_step=2)\n"
        "streams = [torch.cuda.Stream() for _ in range(2)]\n"
        "fbs = [torch.zeros((128, 192, 3), dtype=torch.float32, device='cuda:0') for _ in range(2)]\n"
        "done = []\n"
        "def render(buf, k):\n"
        "    s.render_device(buf.data_ptr(), 192, 128, 8, stratum_begin=1, stratum_step=2, stream=streams[k].cuda_stream, want_stats=False)\n"
        "def finish(buf, k):\n"
        "    assert torch.cuda.current_stream() == streams[k]\n"
        "    done.append(buf.clone())\n"
        "pipe = bench.FramePipeline(fbs, 2, render, finish, streams)\n"
        "for _ in range(7): pipe.step()\n"
        "pipe.drain(); torch.cuda.synchronize()\n"
        "assert len(done) == 7\n"
        "for d in done: assert np.allclose(d.cpu().numpy(), want, rtol=1e-5, atol=1e-6)\n"
        "dist.destroy_process_group()\n"
        "print('ok')\n"
    ) % ROOT
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-3000:]


def test_two_renders_of_one_scene_in_flight_on_two_streams(ra):
    """rt_render_device keeps two scratch sets per scene (include/rt.h): two renders enqueued back to back on two
    streams, neither awaited in between, give the frames serial renders give; so does a third and fourth call
    that re-use the sets (ordered behind their previous use by an event wait), with a different frame size that
    makes one set grow."""
    import torch

    s = ra.Scene(ra.load_scene(scene_path("teapot"), 1.5), 0)
    w, h, sn = 384, 256, 6
    want_a, _ = s.render(w, h, sn, seed=1)
    want_b, _ = s.render(w, h, sn, seed=2)
    want_c, _ = s.render(w + 64, h, 2, seed=3)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    fa = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    fb = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    fc = torch.zeros((h, w + 64, 3), dtype=torch.float32, device="cuda:0")
    fd = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    s.render_device(fa.data_ptr(), w, h, sn, seed=1, stream=s1.cuda_stream, want_stats=False)
    s.render_device(fb.data_ptr(), w, h, sn, seed=2, stream=s2.cuda_stream, want_stats=False)
    s.render_device(fc.data_ptr(), w + 64, h, 2, seed=3, stream=s2.cuda_stream, want_stats=False)  # set 0 again, from s2
    s.render_device(fd.data_ptr(), w, h, sn, seed=1, stream=s1.cuda_stream, want_stats=False)      # set 1 again, from s1
    torch.cuda.synchronize()
    assert np.allclose(fa.cpu().numpy(), want_a, rtol=1e-5, atol=1e-6)
    assert np.allclose(fb.cpu().numpy(), want_b, rtol=1e-5, atol=1e-6)
    assert np.allclose(fc.cpu().numpy(), want_c, rtol=1e-5, atol=1e-6)
    assert np.allclose(fd.cpu().numpy(), want_a, rtol=1e-5, atol=1e-6)
    s.close()


def test_walls_stay_loose_beside_spheres_in_the_tree(ra, ref, tmp_path):
    """A closed room: its six walls are giants and stay loose, 20 small spheres share the render tree with a mesh."""
    path = synth.room(tmp_path)
    w, h, sn = 120, 80, 4
    scene = ra.Scene(ra.load_scene(path, w / h), 0)
    assert scene.info().n_loose == 6
    oracle = ref.RefScene.from_yaml(path, w / h)
    _primary_parity(scene, oracle, w, h)
    for sampler in (0, 1):
        want, _ = oracle.render(w, h, sn, threads=16, sampler=sampler)
        for kernel in (0, 1, 2):
            img, st = scene.render(w, h, sn, kernel=kernel, sampler=sampler, collect_stats=True)
            _check_image(img, want, sn * sn, f"room, kernel {kernel} sampler {sampler}")
            assert st.quad_tests > 0 and st.sphere_tests > 0 and st.tri_tests > 0
    scene.close()


def test_quads_only_render_tree(ra, ref, tmp_path):
    """A tree of nothing but quads (TREE = 3): 42 of them — up to 32 spheres and quads without a mesh are tested loose,
    as the cornell example's 18 are."""
    path = synth.many_quads(tmp_path, 40)
    w, h, sn = 120, 80, 4
    scene = ra.Scene(ra.load_scene(path, w / h), 0)
    info = scene.info()
    assert info.n_surfaces == 42 and info.n_loose == 1 and info.render_nodes > 20  # the floor stays loose
    oracle = ref.RefScene.from_yaml(path, w / h)
    _primary_parity(scene, oracle, w, h)
    for sampler in (0, 1):
        want, _ = oracle.render(w, h, sn, threads=16, sampler=sampler)
        for kernel in (0, 1, 2):
            img, st = scene.render(w, h, sn, kernel=kernel, sampler=sampler, collect_stats=True)
            _check_image(img, want, sn * sn, f"quads only, kernel {kernel} sampler {sampler}")
            assert st.tri_tests == 0 and st.sphere_tests == 0 and st.quad_tests > 0
    scene.close()


def test_up_to_32_loose_without_a_mesh(ra, ref, tmp_path):
    """31 spheres and no triangle: no render tree at all, every sphere tested by every ray (the shade-only build)."""
    path = synth.many_prims(tmp_path, n_spheres=28, n_quads=0, with_mesh=False)
    w, h, sn = 120, 80, 4
    scene = ra.Scene(ra.load_scene(path, w / h), 0)
    info = scene.info()
    assert info.n_surfaces == 31 and info.n_loose == 31 and info.render_nodes == 0
    oracle = ref.RefScene.from_yaml(path, w / h)
    _primary_parity(scene, oracle, w, h)
    for sampler in (0, 1):
        want, _ = oracle.render(w, h, sn, threads=16, sampler=sampler)
        for kernel in (0, 1, 2):
            img, st = scene.render(w, h, sn, kernel=kernel, sampler=sampler, collect_stats=True)
            _check_image(img, want, sn * sn, f"31 loose, kernel {kernel} sampler {sampler}")
            assert st.node_visits == 0 and st.sphere_tests > 0
    scene.close()


def test_single_leaf_trees_of_one_kind(ra, ref, tmp_path):
    """--no-bvh on a spheres-only tree: the synthetic root's dummy slot is read by the spheres-only build as a
    sphere — a NaN-radius one that no ray can hit (api.cpp flatten_render)."""
    path = synth.many_prims(tmp_path, n_spheres=40, n_quads=0, with_mesh=False)
    w, h, sn = 96, 64, 3
    a = ra.Scene(ra.load_scene(path, w / h, True), 0)
    b = ra.Scene(ra.load_scene(path, w / h, False), 0)
    assert b.info().render_max_leaf == 42
    ia, _ = a.render(w, h, sn)
    ib, _ = b.render(w, h, sn)
    assert _same_image(ia, ib, sn * sn)
    a.close(), b.close()
