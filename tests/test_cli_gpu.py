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
    assert m and int(m.group(1)) >= 8, r.stderr          # ~30 batches of 2 strata: an image after each but the last
    assert "Done in" in r.stderr and not (tmp_path / "progressive.png.part").exists()
    r = subprocess.run(args + ["-o", str(b)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    pa, pb = _png(a), _png(b)
    assert pa.shape == pb.shape == (90, 160, 3)
    # same samples; the batches' sums are added on the host, so a value may land on the other side of a rounding
    assert np.abs(pa - pb).max() <= 1 and (pa != pb).mean() <= 0.01


def test_progressive_partial_image_is_a_coarser_estimate_of_the_final_one(tmp_path):
    """The partial images themselves (--keep-partials keeps every one beside the output): 1024 spp in N ~ 30 batches, so
    partial k is the image at 1024 k / N spp.  Every batch's strata are a lattice spread over the pixel's 32 x 32 grid
    (the step between them is coprime to 32), so the first partial already is the final image, only noisier, and the
    partials close in on it: the RMS distance to the final image falls like sqrt(1/k - 1/N).  (This test, once it
    read the partials, found the batches confined to one column of the grid — a strip a 32nd of the pixel high.)"""
    out = tmp_path / "p.png"
    args = [CLI, "--scene", scene_path("teapot"), "--width", "120", "--height", "80", "-n", "1024", "--seed", "7", "--no-progress",
            "-o", str(out)]
    r = subprocess.run(args + ["--progressive", "0", "--keep-partials"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    m = re.search(r"wrote (\d+) partial images to", r.stderr)
    assert m and 24 <= int(m.group(1)) <= 31, r.stderr
    last = int(m.group(1))
    final = _png(out)
    assert final.std() > 5  # not a blank frame
    dist = {}
    for k in (1, 4, 16, last):
        part = _png(f"{out}.partial{k}.png")
        assert part.shape == final.shape == (80, 120, 3)
        dist[k] = np.sqrt(np.mean((part - final).astype(np.float64) ** 2))  # RMS difference in 8-bit steps
    print("\nRMS distance of partial k to the final image, 8-bit steps:", {k: round(v, 2) for k, v in dist.items()})
    model = {k: np.sqrt(1.0 / k - 1.0 / (last + 1)) for k in dist}
    # measured: 4.30, 2.07, 0.84, 0.29 steps for k = 1, 4, 16, 28 of 29 (with the one-column batches: 10.5, 8.8, 5.1)
    for k in (4, 16):  # the fall follows the model within a factor 1.4 either way
        assert 1 / 1.4 < (dist[k] / dist[1]) / (model[k] / model[1]) < 1.4, (dist, model)
    assert 1.0 < dist[1] < 8.0 and dist[last] < 0.6, dist  # the last partial: the 8-bit rounding's own 0.29 and little more
    # and the mean brightness is right from the first partial on (normalised by the samples so far)
    assert abs(_png(f"{out}.partial1.png").mean() - final.mean()) < 0.01 * final.mean() + 0.5


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
    # refused before any RCCL call: a null member buffer, an empty reduce — and the clique still works afterwards
    assert lib.rt_group_reduce_sum(g, (C.c_void_p * 1)(None), x.numel(), 0, None) != 0
    assert b"null buffer" in lib.rt_last_error()
    assert lib.rt_group_reduce_sum(g, bufs, 0, 0, None) != 0
    ra.api._check(lib.rt_group_reduce_sum(g, bufs, x.numel(), 0, None))
    torch.cuda.synchronize()
    assert torch.equal(x, want)
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
    # several ranks: the line says that its rate is pipelined and carries one frame's awaited latency and the same
    # frame on one GPU alone beside it (the scaling figures mean nothing here: the ranks share the card)
    cfg = line["config"]
    assert cfg["frames_in_flight"] == 3 and "pipelined" in cfg["timing"]
    assert cfg["frame_latency_s"] > 0 and cfg["full_frame_one_gpu_s"] > 0
    assert cfg["frame_latency_s"] >= 0.5 * cfg["seconds_per_frame"]
    assert 0 < cfg["scaling_efficiency_vs_n1"] and 0 < cfg["latency_efficiency_vs_n1"]
    pa, pb = _png(one), _png(two)
    assert pa.shape == pb.shape == (180, 320, 3)
    assert np.abs(pa - pb).max() <= 1 and (pa != pb).mean() <= 0.01  # same sample set, summed in another order
