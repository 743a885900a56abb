"""world_size-2 gloo rehearsal of the multi-GPU path (bench.py): strata are dealt round-robin over ranks, each
rank renders its slice of the FULL frame, and one sum-reduce to rank 0 combines the float framebuffers
(SURVEY §8e).  On CPU the oracle stands in for the HIP kernel; the sharding and reduce code is bench.py's own."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, scene_path


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from oracle import ref

    w, h, sn = 36, 24, 3
    begin, step = bench.strata_slice(rank, world)
    o = ref.RefScene.from_yaml(scene_path("teapot"), w / h)
    part, _ = o.render(w, h, sn, stratum_begin=begin, stratum_step=step, threads=2)
    fb = torch.from_numpy(part.astype(np.float32))
    bench.reduce_framebuffer(fb, world)
    dist.barrier()
    if rank == 0:
        np.save(out_path, fb.numpy())
    dist.destroy_process_group()


def test_two_rank_strata_shard_and_reduce(tmp_path, ref):
    out = str(tmp_path / "fb.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    full, _ = ref.RefScene.from_yaml(scene_path("teapot"), 36 / 24).render(36, 24, 3, threads=2)
    assert np.allclose(got, full, rtol=2e-6, atol=1e-6)  # same sample set; f32 rounding of the two partial sums


def test_strata_slice_covers_each_stratum_once():
    import bench

    for world in (1, 2, 3, 4, 8):
        for n in (1, 4, 9, 1024):
            seen = []
            for r in range(world):
                b, s = bench.strata_slice(r, world)
                seen += list(range(b, n, s))
            assert sorted(seen) == list(range(n))


def _pipeline_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench

    frame = [0]
    finished = []

    def render(buf, k):
        frame[0] += 1
        buf.fill_(float((rank + 1) * frame[0]))  # rank r's "slice" of frame f
        return frame[0]

    def finish(buf, k):
        finished.append(buf.clone())

    pipe = bench.FramePipeline([torch.zeros(4, 3), torch.zeros(4, 3)], world, render, finish)
    order = [pipe.step() for _ in range(5)]
    assert order == [1, 2, 3, 4, 5]
    assert len(finished) == 4  # the last frame's reduce is still open
    pipe.drain()
    dist.barrier()
    if rank == 0:
        np.save(out_path, torch.stack(finished).numpy())
    dist.destroy_process_group()


def test_frame_pipeline_overlapped_reduce_two_ranks(tmp_path):
    """bench.py's FramePipeline: frame i's reduce is awaited after frame i+1's render was issued; every frame
    is finished exactly once, in order, holding the sum over ranks."""
    out = str(tmp_path / "frames.npy")
    mp.spawn(_pipeline_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    assert got.shape == (5, 4, 3)
    for f in range(5):
        assert np.all(got[f] == 3.0 * (f + 1))  # (1 + 2) * frame


def test_frame_pipeline_single_rank_finishes_in_step():
    import bench

    log = []
    pipe = bench.FramePipeline([torch.zeros(2)], 1, lambda b, k: log.append("render") or len(log), lambda b, k: log.append("finish"))
    pipe.step(), pipe.step()
    pipe.drain()
    assert log == ["render", "finish", "render", "finish"]


def test_handout_groups_matches_the_kernel_plan():
    import bench

    # 1024 strata: 896 in 32s, then 64 in 16s, 32 in 8s, 16 in 4s, 16 in 2s
    assert bench.handout_groups(1024) == 28 + 4 + 4 + 4 + 8
    # 128 strata: 64 in 16s, 32 in 8s, 16 in 4s, 8 in 2s, 8 in 2s
    assert bench.handout_groups(128) == 4 + 4 + 4 + 4 + 4
    assert bench.handout_groups(4) == 2 and bench.handout_groups(1) == 1 and bench.handout_groups(0) == 0
