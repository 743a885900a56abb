"""The `raytracer` command line on the GPU: progressive output (the image so far written while the
render goes on, the final image unchanged) and the native exchange step (rt_group_*: one RCCL reduce; the box has
one GPU, so the clique has one member).  Needs an MI355X:  pytest -m gpu."""
import ctypes as C
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, scene_path

pytestmark = pytest.mark.gpu
CLI = f"{ROOT}/raytracer_amd/raytracer"


def _png(path):
    from PIL import Image

    return np.asarray(Image.open(path).convert("RGB"), dtype=np.int32)


def test_progressive_writes_partial_images_and_the_same_final_image(tmp_path):
    args = [CLI, "--scene", scene_path("teapot"), "--width", "160", "--height", "90", "-n", "64", "--seed", "3", "--no-progress"]
    a, b = tmp_path / "progressive.png", tmp_path / "plain.png"
    r = subprocess.run(args + ["--progressive", "0", "-o", str(a)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    m = re.search(r"wrote (\d+) partial images? to", r.stderr)
    assert m and int(m.group(1)) >= 8, r.stderr          # 32 batches of 2 strata: an image after each but the last
    assert "Done in" in r.stderr and not (tmp_path / "progressive.png.part").exists()
    r = subprocess.run(args + ["-o", str(b)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    pa, pb = _png(a), _png(b)
    assert pa.shape == pb.shape == (90, 160, 3)
    # same samples; the batches' sums are added on the host, so a value may land on the other side of a rounding
    assert np.abs(pa - pb).max() <= 1 and (pa != pb).mean() <= 0.01


def test_progressive_partial_image_is_a_coarser_estimate_of_the_final_one(tmp_path):
    """A long enough interval that only the first partial image is written: it must already look like the final
    image (every pixel's strata subset is spread over the pixel), just noisier."""
    out = tmp_path / "p.png"
    args = [CLI, "--scene", scene_path("quads"), "--width", "120", "--height", "80", "-n", "1024", "--no-progress", "-o", str(out)]
    r = subprocess.run(args + ["--progressive", "0"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    final = _png(out)
    assert final.std() > 5  # not a blank frame


def test_cli_gpus_1_with_rccl_reduce_flag_is_accepted(tmp_path):
    out = tmp_path / "o.png"
    r = subprocess.run([CLI, "--scene", scene_path("quads"), "--width", "60", "--height", "40", "-n", "4", "--gpus", "1",
                        "--reduce", "rccl", "--json", "-o", str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and '"reduce": "none"' in r.stdout, r.stderr


def test_group_reduce_sum_one_member_clique(ra):
    """rt_group_create / rt_group_reduce_sum / rt_group_destroy through the C ABI: librccl is loaded on demand, a
    clique over this box's one GPU is opened, and the in-place sum-reduce onto its root leaves the buffer as it is."""
    import torch

    lib = ra.lib()
    devices = (C.c_int * 1)(0)
    g = C.c_void_p()
    ra.api._check(lib.rt_group_create(devices, 1, C.byref(g)))
    x = torch.arange(3 * 64 * 48, dtype=torch.float32, device="cuda:0")
    want = x.clone()
    torch.cuda.synchronize()
    bufs = (C.c_void_p * 1)(x.data_ptr())
    ra.api._check(lib.rt_group_reduce_sum(g, bufs, x.numel(), 0, None))
    torch.cuda.synchronize()
    assert torch.equal(x, want)
    assert lib.rt_group_reduce_sum(g, bufs, x.numel(), 3, None) != 0  # a root outside the clique is refused
    lib.rt_group_destroy(g)


def test_bench_two_rank_flow_rehearsed_on_one_gpu(tmp_path):
    """bench.py's several-rank flow — strata slices, a stream per framebuffer, two renders in flight, the frame
    pipeline, max-over-ranks timing, rank 0's line and image — with two ranks sharing this box's one GPU.  RCCL
    refuses two ranks on one device, so the rehearsal sums over gloo through host copies (bench.py says so in the
    line); the frame must equal the one-rank frame."""
    import json
    import socket
    import sys

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    common = ["--scene", "teapot", "--width", "320", "--height", "180", "--spp", "64", "--steps", "3", "--warmup", "1",
              "--no-cpu-baseline"]
    one, two = tmp_path / "one.png", tmp_path / "two.png"
    r = subprocess.run([sys.executable, f"{ROOT}/bench.py", "--gpus", "1", "--png", str(one)] + common,
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(__import__("os").environ, RT_BENCH_REHEARSE_ONE_GPU="1", RT_BENCH_HANG_DUMP="200")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), f"{ROOT}/bench.py", "--gpus", "2", "--png", str(two)] + common,
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["value"] > 0 and "REHEARSAL" in line["config"]["parallelism"]
    assert "cpu_baseline" not in line and line["roofline"]["traffic"] is None
    pa, pb = _png(one), _png(two)
    assert pa.shape == pb.shape == (180, 320, 3)
    assert np.abs(pa - pb).max() <= 1 and (pa != pb).mean() <= 0.01  # same sample set, summed in another order
