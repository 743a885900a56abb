"""The oracle against the committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).
CPU only.  The vectors are oracle outputs (the Rust reference cannot run here and ships no fixtures), so this
pins the oracle against drift; the GPU tests then compare the HIP path with the same vectors."""
import os

import numpy as np
import pytest

from conftest import ROOT, scene_path

SCENES = ["cornell", "cornell_lucy", "teapot", "lucy", "dual", "distant", "light", "quads"]


def golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))


@pytest.mark.parametrize("name", SCENES)
def test_oracle_primary_hits_match_golden(ref, name):
    g = golden(name)
    h, w = g["prim_id"].shape
    o = ref.RefScene.from_yaml(scene_path(name), w / h)
    ids, t = o.primary_hits(w, h)
    assert np.array_equal(ids, g["prim_id"])
    assert np.array_equal(t, g["t"])
    order, st = o.leaf_order()
    assert [st["nodes"], st["leaves"], st["depth"], st["max_leaf"]] == g["tree"].tolist()
    assert np.array_equal(order[:4096], g["leaf_order_head"])
    assert int(np.sum(order.astype(np.int64) * np.arange(1, len(order) + 1))) == int(g["leaf_order_sum"])


@pytest.mark.parametrize("name", ["teapot", "quads", "distant", "light", "cornell"])
def test_oracle_image_matches_golden(ref, name):
    g = golden(name)
    ih, iw, _ = g["image_sum"].shape
    o = ref.RefScene.from_yaml(scene_path(name), iw / ih)
    img, cnt = o.render(iw, ih, 4, max_depth=40, seed=1)
    assert np.array_equal(img, g["image_sum"])      # deterministic: counter-seeded streams, fixed fold order
    assert cnt["segments"] == int(g["segments"])
    img, cnt = o.render(iw, ih, 4, max_depth=40, seed=1, sampler=1)
    assert np.allclose(img, g["image_sum_direct"], rtol=1e-12, atol=1e-12)  # libm sin/cos may differ in the last ulp
    assert abs(cnt["segments"] - int(g["segments_direct"])) <= 2


def test_bvh_vs_brute_force_primary_hits(ref):
    """--no-bvh puts every surface in one leaf (main.rs:60-65): identical hits (the reference's debugging aid)."""
    for name in ("teapot", "quads", "distant"):
        a = ref.RefScene.from_yaml(scene_path(name), 1.0, use_bvh=True).primary_hits(24, 24)
        b = ref.RefScene.from_yaml(scene_path(name), 1.0, use_bvh=False).primary_hits(24, 24)
        same = a[0] == b[0]
        # a different leaf order can only change the winner among exactly equal t
        assert np.all(same | (a[1] == b[1]))
        assert np.array_equal(a[1], b[1])


def test_strata_slices_partition_the_sample_set(ref):
    o = ref.RefScene.from_yaml(scene_path("quads"), 1.5)
    full, _ = o.render(30, 20, 3)
    parts = [o.render(30, 20, 3, stratum_begin=g, stratum_step=4)[0] for g in range(4)]
    assert np.allclose(sum(parts), full, rtol=1e-13, atol=0)


def test_both_samplers_have_the_same_distribution(ref):
    """The direct sampler (RT_SAMPLER_DIRECT) must be distribution-equal to the reference's rejection loops
    (util.rs:15-35): moments of 400k samples, then converged images of two scenes."""
    for which, r_mean in ((0, 0.75), (1, 1.0), (2, 2.0 / 3.0)):  # ball, sphere surface, disk
        a, b = ref.sample(which, 0, 99, 400000), ref.sample(which, 1, 99, 400000)
        for v in (a, b):
            assert np.all(np.abs(v.mean(0)) < 5e-3)
            assert abs(np.linalg.norm(v, axis=1).mean() - r_mean) < 3e-3
        assert np.allclose((a ** 2).mean(0), (b ** 2).mean(0), atol=3e-3)
        assert np.allclose(np.abs(a).mean(0), np.abs(b).mean(0), atol=3e-3)
    for name in ("teapot", "lucy"):
        o = ref.RefScene.from_yaml(scene_path(name), 1.5)
        a, _ = o.render(60, 40, 16, sampler=0)
        b, _ = o.render(60, 40, 16, sampler=1)
        a, b = a / 256, b / 256
        assert abs(a.mean() - b.mean()) < 0.005 * a.mean()
        assert np.sqrt(np.mean((a - b) ** 2)) < 0.02   # two independent 256-spp estimates
