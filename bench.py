"""Headline benchmark: whole-frame path tracing throughput in Msamples/s (BASELINE.json), one process per GPU.

A "step" is one full frame of the workload: every rank renders its strata slice of the FULL frame through the
C ABI (rt_render_device) into a device framebuffer owned by torch, the slices are combined by one RCCL
sum-reduce to rank 0, and rank 0 tonemaps.  Default workload: scenes/lucy.yaml 1920x1080 at
1024 spp — the configuration BASELINE.json's metric names ("s/frame lucy.yaml 1080p 1024spp").  The scene is
resident in HBM before the timed region; timing excludes scene load, BVH build and PNG encode.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_VALU_TFLOPS = 157.3   # MI355X_MICROARCH.md, "Peak FP32 (vector)"
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md, "HBM3E peak BW" (spec)


def strata_slice(rank: int, world: int):
    """Rank r of N renders strata k = r, r+N, r+2N, ... (k = si*sn + sj, main.rs:165-166): the union over
    ranks is exactly the single-GPU sample set, and every rank sees every pixel (perfect balance)."""
    return rank, world


def reduce_framebuffer(fb, world: int) -> None:
    """The path's one exchange step: sum the per-rank float framebuffers onto rank 0."""
    if world > 1:
        import torch.distributed as dist

        dist.reduce(fb, dst=0, op=dist.ReduceOp.SUM)


class FramePipeline:
    """Two framebuffers in turn: frame i's sum-reduce runs (on RCCL's own stream) beside frame i+1's render, and
    frame i is finished — reduce awaited, `finish(buffer)` called, e.g. the tonemap — after frame i+1's render
    was issued.  A buffer is rendered into again only after the frame it held was finished, in stream order.
    With one rank there is nothing to overlap: a frame is finished in its own step."""

    def __init__(self, buffers, world: int, render, finish):
        self.buffers, self.world, self.render, self.finish = buffers, world, render, finish
        self.work = [None] * len(buffers)
        self.open = [False] * len(buffers)
        self.i = 0

    def _finish(self, k: int) -> None:
        if self.work[k] is not None:
            self.work[k].wait()  # the current stream waits for the reduce; the host does not
            self.work[k] = None
        if self.open[k]:
            self.finish(self.buffers[k])
            self.open[k] = False

    def step(self):
        k = self.i % len(self.buffers)
        self.i += 1
        self._finish(k)  # no-op unless the pipeline is deeper than its buffers
        out = self.render(self.buffers[k])
        self.open[k] = True
        if self.world > 1:
            import torch.distributed as dist

            self.work[k] = dist.reduce(self.buffers[k], dst=0, op=dist.ReduceOp.SUM, async_op=True)
            self._finish((k - 1) % len(self.buffers))  # the frame before: its reduce ran beside this render
        else:
            self._finish(k)
        return out

    def drain(self) -> None:
        n = len(self.buffers)
        for j in range(n):  # oldest frame first
            self._finish((self.i + j) % n)


def handout_groups(strata: int, tail_chunks: int = 4) -> int:
    """Chunks per tile of the persistent kernel's guided hand-out (render_persistent.hip: five phases of shrinking
    chunks — 32 strata first for frames of 512 strata and more, else 16, then halving down to 2; a phase leaves
    four of its chunks' worth — one in scenes without a tree —, at most half of the rest, to the finer ones)."""
    first = 32 if strata >= 512 else 16
    done = groups = 0
    for p in range(5):
        per, left = max(2, first >> p), strata - done
        take = left if p == 4 else (left - min(left // 2, tail_chunks * per)) // per * per
        groups += (take + per - 1) // per
        done += take
    return groups


def flops_per_sample(st) -> float:
    """sum over bounces [N_box*24 + N_tri*60 + N_sph*30 + N_quad*45 + shade*60] + raygen*40,
    with N_box = 2 child boxes per visited node."""
    n = max(st.samples, 1)
    return (2 * st.node_visits * 24 + st.tri_tests * 60 + st.sphere_tests * 30 + st.quad_tests * 45
            + st.segments * 60) / n + 40


def cache_bytes_per_sample(st) -> float:
    """SURVEY §8(d) cache-level bytes: N_node*64 + N_tri*84 + N_sph*16 + N_quad*72 per sample."""
    n = max(st.samples, 1)
    return (st.node_visits * 64 + st.tri_tests * 84 + st.sphere_tests * 16 + st.quad_tests * 72) / n


def pmc_traffic(workload: str):
    """Memory-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE + WRITE_SIZE, KB -> bytes), when they were taken on this very workload; else None.  The
    counters cannot be read from inside the timed process, so this is the profile of the same command."""
    path = os.path.join(ROOT, "profiles", "r1_final_pmc.json")
    try:
        with open(path) as fh:
            p = json.load(fh)
    except OSError:
        return None
    if p.get("workload") != workload:
        return None
    return round((p["FETCH_SIZE_KB"] + p["WRITE_SIZE_KB"]) * 1024.0)


def cpu_baseline(scene_file: str, width: int, height: int, sn: int, threads: int):
    """The oracle (the f64 CPU restatement of the reference's render loop, Q1 re-trace cost included) timed on
    this box's host cores on a bounded sample of the same workload: same scene and resolution, sn*sn spp."""
    from oracle import ref

    o = ref.RefScene.from_yaml(scene_file, width / height)
    t0 = time.perf_counter()
    o.render(width, height, sn, max_depth=40, seed=1, threads=threads)
    dt = time.perf_counter() - t0
    n = width * height * sn * sn
    return {"value": round(n / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{os.path.basename(scene_file)} {width}x{height} at {sn * sn} spp ({n} samples, {dt:.1f} s)"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="lucy")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--kernel", choices=["persistent", "wavefront", "pool"], default="persistent")
    ap.add_argument("--sampler", choices=["direct", "reference"], default="direct",
                    help="unit-sphere/disk sampling: loop-free direct maps (default) or the reference's rejection loops; "
                         "same distributions (tests/test_golden.py, tests/test_gpu_parity.py)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-sn", type=int, default=10, help="strata per axis of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--png", default=None, help="rank 0 writes the last frame here")
    args = ap.parse_args()

    import numpy as np
    import torch

    import raytracer_amd as ra

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=dev)

    scene_file = os.path.join(ROOT, "scenes", f"{args.scene}.yaml")
    w, h = args.width, args.height
    sn = int(np.floor(np.sqrt(args.spp)))  # main.rs:142
    kernel = {"persistent": ra.RT_KERNEL_PERSISTENT, "wavefront": ra.RT_KERNEL_WAVEFRONT,
              "pool": ra.RT_KERNEL_POOL}[args.kernel]
    sampler = ra.RT_SAMPLER_DIRECT if args.sampler == "direct" else ra.RT_SAMPLER_REFERENCE
    t0 = time.perf_counter()
    desc = ra.load_scene(scene_file, w / h)
    scene = ra.Scene(desc, local_rank)
    setup_s = time.perf_counter() - t0
    info = scene.info()

    fbs = [torch.empty((h, w, 3), dtype=torch.float32, device=dev) for _ in range(2 if world > 1 else 1)]
    fb = fbs[0]
    rgb8 = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
    begin, step_ = strata_slice(rank, world)
    stream = torch.cuda.current_stream(dev).cuda_stream
    kernel_ms = []

    def render(buf):
        st = scene.render_device(buf.data_ptr(), w, h, sn, max_depth=40, seed=args.seed, stratum_begin=begin,
                                 stratum_step=step_, kernel=kernel, stream=stream, sampler=sampler)
        kernel_ms.append(st.kernel_ms)  # HIP events on `stream`, around this step's render kernels
        return st

    def tonemap(buf):
        if rank == 0:
            ra.api._check(ra.lib().rt_tonemap_rgb8_device(buf.data_ptr(), w, h, sn * sn, rgb8.data_ptr(), stream))

    # a step = render this rank's strata slice of one frame + the frame's sum-reduce + tonemap on rank 0; with
    # several ranks the reduce of one frame overlaps the render of the next (FramePipeline)
    frames = FramePipeline(fbs, world, render, tonemap)
    one_step = frames.step

    def fence():
        frames.drain()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        one_step()
    fence()
    kernel_ms.clear()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = one_step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    samples_per_step = w * h * sn * sn
    value = samples_per_step * args.steps / elapsed / 1e6

    # Algorithmic work per sample from the counting build of the same kernel, outside the timed region,
    # at reduced spp (per-sample counts do not depend on spp).
    cnt = scene.render_device(fb.data_ptr(), w, h, min(sn, 4), max_depth=40, seed=args.seed, kernel=kernel,
                              collect_stats=True, stream=stream, sampler=sampler)
    fl = flops_per_sample(cnt)
    my_samples = last.samples  # samples this rank rendered per step
    avg_kernel_s = float(np.mean(kernel_ms)) / 1e3
    achieved_tflops = fl * my_samples / avg_kernel_s / 1e12
    # HBM-side traffic the algorithm needs per launch (persistent kernels): clear of the tile-planar accumulator
    # (12 B/pixel), one 768-byte flush of a wave's tile sums per chunk (1024 samples, fewer at the tail), and the resolve pass
    # (12 B/pixel read + 12 B/pixel written).  The wavefront form instead adds 12 B per sample directly and
    # streams its path-state queues (~216 B per segment).
    seg_per_sample = cnt.segments / max(cnt.samples, 1)
    if kernel == ra.RT_KERNEL_WAVEFRONT:
        hbm_bytes = 12.0 * w * h + 12.0 * my_samples + 216.0 * seg_per_sample * my_samples
    else:
        tiles = ((w + 7) // 8) * ((h + 7) // 8)
        hbm_bytes = 36.0 * w * h + 768.0 * tiles * handout_groups(my_samples // (w * h), 4 if info.render_nodes else 1)

    if rank == 0:
        workload = (f"scenes/{args.scene}.yaml {w}x{h} {sn * sn} spp, max_depth 40, "
                    f"{args.kernel} kernel, {args.sampler} sampler")
        line = {
            "metric": "Msamples/sec (whole node)", "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "seconds_per_frame": round(elapsed / args.steps, 4),
                       "parallelism": f"strata-sharded x{world}, 1 RCCL sum-reduce/frame" if world > 1 else "1 GPU",
                       "scene_surfaces": int(info.n_surfaces), "render_bvh_nodes": int(info.render_nodes),
                       "render_bvh_depth": int(info.render_depth), "loose_prims": int(info.n_loose),
                       "scene_setup_s": round(setup_s, 3)},
            "roofline": {"bound": "valu_fp32", "achieved": round(achieved_tflops, 4), "peak": PEAK_FP32_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved_tflops / PEAK_FP32_VALU_TFLOPS, 5),
                         "traffic": pmc_traffic(workload) if world == 1 else None,
                         "kernel": {ra.RT_KERNEL_PERSISTENT: "render_persistent_kernel", ra.RT_KERNEL_POOL: "render_pool_kernel",
                                    ra.RT_KERNEL_WAVEFRONT: "wf_extend+wf_shade"}[kernel],
                         "kernel_ms_avg": round(avg_kernel_s * 1e3, 3), "flops_per_sample": round(fl, 1),
                         "segments_per_sample": round(seg_per_sample, 3),
                         "node_visits_per_segment": round(cnt.node_visits / max(cnt.segments, 1), 2),
                         "tri_tests_per_segment": round(cnt.tri_tests / max(cnt.segments, 1), 2),
                         "cache_bytes_per_sample": round(cache_bytes_per_sample(cnt), 1),
                         "rays_per_s": round(seg_per_sample * my_samples / avg_kernel_s, 0),
                         "hbm": {"achieved": round(hbm_bytes / avg_kernel_s / 1e9, 3), "peak": PEAK_HBM_GBS,
                                 "unit": "GB/s", "frac": round(hbm_bytes / avg_kernel_s / 1e9 / PEAK_HBM_GBS, 6)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            line["cpu_baseline"] = cpu_baseline(scene_file, w, h, args.cpu_sn, threads)
        if args.png:
            ra.write_png(args.png, rgb8.cpu().numpy())
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
