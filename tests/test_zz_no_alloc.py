"""rt_render_device_noalloc (include/rt.h): the entry point that allocates nothing — static device scratch
instead of the scene's hipMalloc'ed accumulator — against rt_render_device, through the C ABI.  Two calls in a
row, the second with a smaller frame (the scratch keeps the first call's contents).  Bar: the two entry points
run the same kernel on the same samples, so at 1 spp (one add per pixel) the frames are bit-identical, and at
16 spp they agree up to the order of the f32 atomic adds (rtol 1e-5).  Needs an MI355X:  pytest -m gpu."""
import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu


def _both(scene, torch, w, h, sn, **kw):
    stream = torch.cuda.current_stream().cuda_stream
    a = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda:0")
    b = torch.full((h, w, 3), -2.0, dtype=torch.float32, device="cuda:0")
    scene.render_device(a.data_ptr(), w, h, sn, stream=stream, **kw)
    scene.render_device_noalloc(b.data_ptr(), w, h, sn, stream=stream, **kw)
    torch.cuda.synchronize()
    return a.cpu().numpy(), b.cpu().numpy()


@pytest.mark.parametrize("name", ["teapot", "quads"])
def test_noalloc_matches_render_device_twice_second_call_smaller(ra, name):
    import torch

    scene = ra.Scene(ra.load_scene(scene_path(name), 1.5), 0)
    for w, h in ((240, 160), (96, 64)):  # the second call is smaller: stale sums beyond its tiles must not leak
        a, b = _both(scene, torch, w, h, 1)
        assert np.array_equal(a, b), f"{name} {w}x{h} 1 spp: {(a != b).sum()} values differ"
        assert a.any()
        a, b = _both(scene, torch, w, h, 4, stratum_begin=1, stratum_step=2, sampler=ra.RT_SAMPLER_REFERENCE)
        assert np.isfinite(b).all()
        assert np.allclose(a, b, rtol=1e-5, atol=1e-6)
    scene.close()


def test_noalloc_limits_are_errors(ra):
    import torch

    scene = ra.Scene(ra.load_scene(scene_path("quads"), 1.5), 0)
    fb = torch.zeros((8, 8, 3), dtype=torch.float32, device="cuda:0")
    with pytest.raises(ra.RtError, match="exceeds the static scratch"):
        scene.render_device_noalloc(fb.data_ptr(), 3840, 2160, 1)  # 129 600 tiles > 32 768 (nothing is launched)
    with pytest.raises(ra.RtError, match="RT_KERNEL_PERSISTENT"):
        scene.render_device_noalloc(fb.data_ptr(), 8, 8, 1, kernel=ra.RT_KERNEL_WAVEFRONT)
    # the largest frame BASELINE's headline uses fits: 1920x1080 = 32 400 tiles
    big = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
    scene.render_device_noalloc(big.data_ptr(), 1920, 1080, 1)
    torch.cuda.synchronize()
    want, _ = scene.render(1920, 1080, 1)
    assert np.array_equal(big.cpu().numpy(), want)
    scene.close()


def test_noalloc_calls_on_two_streams_and_two_scenes_are_ordered(ra):
    """Every call on a GPU shares that GPU's one scratch array.  Calls issued by one host thread on different
    streams — and for different scenes — are ordered by the library (an event per GPU: each call's stream waits
    for the previous call's end), so a long frame on one stream followed at once by a short one on another gives
    both frames right."""
    import torch

    a_scene = ra.Scene(ra.load_scene(scene_path("teapot"), 1.5), 0)
    b_scene = ra.Scene(ra.load_scene(scene_path("quads"), 1.5), 0)
    s1, s2 = torch.cuda.Stream("cuda:0"), torch.cuda.Stream("cuda:0")
    fa = torch.zeros((320, 480, 3), dtype=torch.float32, device="cuda:0")
    fb = torch.zeros((64, 96, 3), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    for _ in range(3):
        a_scene.render_device_noalloc(fa.data_ptr(), 480, 320, 8, stream=s1.cuda_stream)   # ~10 M samples in flight ...
        b_scene.render_device_noalloc(fb.data_ptr(), 96, 64, 1, stream=s2.cuda_stream)     # ... when this one is issued
    torch.cuda.synchronize()
    want_a, _ = a_scene.render(480, 320, 8)
    want_b, _ = b_scene.render(96, 64, 1)
    assert np.allclose(fa.cpu().numpy(), want_a, rtol=1e-5, atol=1e-6)
    assert np.array_equal(fb.cpu().numpy(), want_b)
    a_scene.close(), b_scene.close()
