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
    """Framebuffers in turn, each with a stream of its own when `streams` is given: frame i+1's render is
    enqueued on the next stream while frame i's render drains (rt_render_device keeps three scratch sets per scene
    for exactly this), frame i's sum-reduce runs on RCCL's stream beside frame i+1's render, and frame i is finished
    — reduce awaited on its stream, `finish(buffer, k)` called, e.g. the tonemap — after frame i+1's render was
    issued.  A buffer is rendered into again only after the frame it held was finished, in its stream's order.
    `render(buffer, k)` and `finish(buffer, k)` get the buffer's index (= its stream's).  With one rank there is
    nothing to overlap: a frame is finished in its own step."""

    def __init__(self, buffers, world: int, render, finish, streams=None, reduce=None):
        self.buffers, self.world, self.render, self.finish, self.streams = buffers, world, render, finish, streams
        self.reduce = reduce  # stand-in for the asynchronous sum-reduce (the one-GPU rehearsal's host reduce)
        self.work = [None] * len(buffers)
        self.open = [False] * len(buffers)
        self.i = 0

    def _on(self, k: int):
        if self.streams is None:
            import contextlib

            return contextlib.nullcontext()
        import torch

        return torch.cuda.stream(self.streams[k])

    def _finish(self, k: int) -> None:
        with self._on(k):
            if self.work[k] is not None:
                self.work[k].wait()  # buffer k's stream waits for the reduce; the host does not
                self.work[k] = None
            if self.open[k]:
                self.finish(self.buffers[k], k)
                self.open[k] = False

    def step(self):
        k = self.i % len(self.buffers)
        self.i += 1
        self._finish(k)  # no-op unless the pipeline is deeper than its buffers
        with self._on(k):
            out = self.render(self.buffers[k], k)
            self.open[k] = True
            if self.world > 1:
                import torch.distributed as dist

                # RCCL's stream first waits for what buffer k's stream holds so far: this frame's render
                if self.reduce is not None:
                    self.work[k] = self.reduce(self.buffers[k], k)
                else:
                    self.work[k] = dist.reduce(self.buffers[k], dst=0, op=dist.ReduceOp.SUM, async_op=True)
        if self.world > 1:
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
    four of its chunks' worth — eight with a tree of up to 18 levels, one in scenes without a tree —, at most half
    of the rest, to the finer ones)."""
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


PMC_PROFILE = os.path.join("profiles", "r3_final_pmc.json")


def kernel_sources_sha() -> str:
    """Hash of the device sources the dominant kernel is built from; profiles/*_pmc.json records the one it was
    taken on (tools/pmc_to_json.py), so a profile of older kernels is not passed off as this run's."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "raytracer_amd", "csrc", "hip")
    for name in ("device_scene.h", "device_rt.h", "render_persistent.hip"):
        with open(os.path.join(d, name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(workload: str):
    """(bytes, source): memory-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC
    passes (FETCH_SIZE + WRITE_SIZE, KB -> bytes; MI355X_MICROARCH.md, HBM section) — only when they were taken on
    this very workload AND on these very kernel sources; else (None, reason).  The counters cannot be read from
    inside the timed process: this is the profile of the same command, labelled as such."""
    try:
        with open(os.path.join(ROOT, PMC_PROFILE)) as fh:
            p = json.load(fh)
    except OSError:
        return None, f"{PMC_PROFILE} missing"
    if p.get("workload") != workload:
        return None, f"{PMC_PROFILE} is of another workload"
    if p.get("kernel_sources_sha") != kernel_sources_sha():
        return None, f"{PMC_PROFILE} was taken on other kernel sources ({p.get('kernel_sources_sha')})"
    return round((p["FETCH_SIZE_KB"] + p["WRITE_SIZE_KB"]) * 1024.0), f"{PMC_PROFILE} (rocprofv3 --pmc, same command, commit {p.get('commit', '?')})"


def host_cpus():
    """(threads to use, CPUs this process may run on, CPUs of the machine, model name).  The reference's rayon pool
    uses every CPU it may run on (main.rs:122-127); so does the baseline — bounded by the scheduler affinity, the
    cgroup CPU quota and, on the GPU box, the per-GPU CPU share the harness exports as OMP_NUM_THREADS."""
    allowed = len(os.sched_getaffinity(0))
    use, why = allowed, "sched_getaffinity"
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max" and int(quota) // int(period) >= 1 and int(quota) // int(period) < use:
            use, why = int(quota) // int(period), "cgroup cpu.max"
    except (OSError, ValueError):
        pass
    omp = os.environ.get("OMP_NUM_THREADS", "")
    if omp.isdigit() and 0 < int(omp) < use:
        use, why = int(omp), "OMP_NUM_THREADS (the harness's CPU share per GPU)"
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return use, allowed, os.cpu_count() or allowed, model, why


def cpu_baseline(scene_file: str, width: int, height: int, seconds: float):
    """The oracle (the f64 CPU restatement of the reference's render loop, Q1 re-trace cost included) timed on
    this box's host CPUs on a bounded sample of the same workload: same scene and resolution, at the spp that a
    short calibration run says takes about `seconds` (>= 30 s of CPU work).  Always with the REFERENCE
    sampler — the rejection loops of util.rs:15-35 are what the reference's CPU renderer runs — whatever sampler the
    GPU headline uses; the direct maps' CPU figure, on a quarter of the sample, is reported beside it."""
    from oracle import ref

    threads, allowed, total, model, why = host_cpus()
    o = ref.RefScene.from_yaml(scene_file, width / height)

    def timed(sn, sampler):
        t0 = time.perf_counter()
        o.render(width, height, sn, max_depth=40, seed=1, threads=threads, sampler=sampler)
        return time.perf_counter() - t0

    per_spp = timed(2, 0) / 4.0
    sn = max(2, min(64, int(round((seconds / per_spp) ** 0.5))))
    dt = timed(sn, 0)
    n = width * height * sn * sn
    sn_d = max(2, sn // 2)
    dt_d = timed(sn_d, 1)
    return {"value": round(n / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "cores_allowed": allowed,
            "cores_total": total, "cores_bound_by": why, "cpu_model": model, "kind": "port",
            "sample": f"{os.path.basename(scene_file)} {width}x{height} at {sn * sn} spp ({n} samples, {dt:.1f} s of "
                      f"{threads} threads), reference sampler (the rejection loops of util.rs:15-35)",
            "direct_sampler": {"value": round(width * height * sn_d * sn_d / dt_d / 1e6, 4), "unit": "Msamples/s",
                               "sample": f"{sn_d * sn_d} spp, {dt_d:.1f} s"}}


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
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="CPU work of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--png", default=None, help="rank 0 writes the last frame here")
    args = ap.parse_args()

    import numpy as np
    import torch

    import raytracer_amd as ra

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # Where is every thread after N seconds? (then exit non-zero).  Armed by default with several ranks, so that a
    # collective that never completes (a first RCCL run on new hardware) ends with stacks on stderr instead of being
    # killed at the caller's limit; RT_BENCH_HANG_DUMP=SECONDS sets it for any run, 0 disarms.
    hang_after = float(os.environ.get("RT_BENCH_HANG_DUMP", "600" if world > 1 else "0"))
    if hang_after > 0:
        import faulthandler

        faulthandler.dump_traceback_later(hang_after, exit=True)
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # Rehearsal of the several-rank flow on a one-GPU box (RT_BENCH_REHEARSE_ONE_GPU=1): every rank on GPU 0, the
    # collectives over gloo instead of RCCL (which refuses two ranks on one device).  The figures mean nothing — the
    # ranks share the card — and the line says so; what it exercises is the pipeline, the streams and the timing code.
    rehearsal = world > 1 and os.environ.get("RT_BENCH_REHEARSE_ONE_GPU") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        import datetime

        limit = datetime.timedelta(seconds=float(os.environ.get("RT_BENCH_DIST_TIMEOUT", "180")))
        if rehearsal:
            dist.init_process_group("gloo", timeout=limit)
        else:
            # the reduce's blocks queue behind CU-filling render launches: let them go first when slots free up
            os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
            dist.init_process_group("nccl", device_id=dev, timeout=limit)

    scene_file = os.path.join(ROOT, "scenes", f"{args.scene}.yaml")
    w, h = args.width, args.height
    sn = int(np.floor(np.sqrt(args.spp)))  # main.rs:142
    kernel = {"persistent": ra.RT_KERNEL_PERSISTENT, "wavefront": ra.RT_KERNEL_WAVEFRONT,
              "pool": ra.RT_KERNEL_POOL}[args.kernel]
    sampler = ra.RT_SAMPLER_DIRECT if args.sampler == "direct" else ra.RT_SAMPLER_REFERENCE
    # A cold process spends ~0.12-0.18 s creating the device context (HIP runtime, code objects): process start-up,
    # not scene set-up, so it is taken first and reported on its own.
    t0 = time.perf_counter()
    torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    context_s = time.perf_counter() - t0
    # The library's own first use costs as much again (its 1.8 MB of code objects are loaded at its first kernel
    # launch or symbol lookup): process start-up too, taken by one 1-pixel tonemap and reported on its own.
    t0 = time.perf_counter()
    one = torch.zeros(3, dtype=torch.float32, device=dev)
    ra.api._check(ra.lib().rt_tonemap_rgb8_device(one.data_ptr(), 1, 1, 1, torch.empty(3, dtype=torch.uint8, device=dev).data_ptr(),
                                                  torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize()
    library_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    desc = ra.load_scene(scene_file, w / h)
    load_s = time.perf_counter() - t0
    scene = ra.Scene(desc, local_rank)
    setup_s = time.perf_counter() - t0
    info = scene.info()

    fbs = [torch.empty((h, w, 3), dtype=torch.float32, device=dev) for _ in range(3 if world > 1 else 1)]
    fb = fbs[0]
    rgb8 = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
    begin, step_ = strata_slice(rank, world)
    # one rank: everything on torch's current stream, the library times its kernels with HIP events on it.
    # several ranks: a stream per framebuffer, so that the next frame's render fills the CUs this frame's draining
    # waves leave idle; the library is then not asked for statistics (that would synchronise the stream) and the
    # launches are timed with events recorded on their own streams.
    streams = [torch.cuda.Stream(dev) for _ in fbs] if world > 1 else None
    stream = torch.cuda.current_stream(dev).cuda_stream
    kernel_ms, kernel_events = [], []
    my_strata = len(range(begin, sn * sn, step_))

    def render(buf, k):
        if streams is None:
            st = scene.render_device(buf.data_ptr(), w, h, sn, max_depth=40, seed=args.seed, stratum_begin=begin,
                                     stratum_step=step_, kernel=kernel, stream=stream, sampler=sampler)
            kernel_ms.append(st.kernel_ms)  # HIP events on `stream`, around this step's render kernels
            return st
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(streams[k])
        scene.render_device(buf.data_ptr(), w, h, sn, max_depth=40, seed=args.seed, stratum_begin=begin,
                            stratum_step=step_, kernel=kernel, stream=streams[k].cuda_stream, sampler=sampler,
                            want_stats=False)
        e1.record(streams[k])
        kernel_events.append((e0, e1))
        return None

    def tonemap(buf, k):
        if rank == 0:
            on = stream if streams is None else streams[k].cuda_stream
            ra.api._check(ra.lib().rt_tonemap_rgb8_device(buf.data_ptr(), w, h, sn * sn, rgb8.data_ptr(), on))

    # a step = render this rank's strata slice of one frame + the frame's sum-reduce + tonemap on rank 0; with
    # several ranks the reduce of one frame and the drain of its render overlap the render of the next
    def host_reduce(buf, k):
        # rehearsal only: gloo's reduce of device tensors hangs on this image when GPU work is in flight (with any
        # kernels, not just these), so the ranks' frames are summed over host copies, awaited at once
        streams[k].synchronize()
        host = buf.cpu()
        dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
        if rank == 0:
            buf.copy_(host)
        return None

    def max_over_ranks(seconds):
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    frames = FramePipeline(fbs, world, render, tonemap, streams, host_reduce if rehearsal else None)
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
    kernel_ms.clear(), kernel_events.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    elapsed = time.perf_counter() - t0
    # overlapped launches share the GPU: their event-bracketed times overlap too, so with several ranks the
    # average launch duration is taken as the step time (render-bound: the reduce and tonemap hide beside it)
    if kernel_events:
        kernel_ms = [elapsed / args.steps * 1e3] * args.steps
        kernel_events.clear()
    if world > 1:
        elapsed = max_over_ranks(elapsed)

    samples_per_step = w * h * sn * sn
    value = samples_per_step * args.steps / elapsed / 1e6
    png_image = rgb8.cpu().numpy() if (args.png and rank == 0) else None  # the headline's last frame, not a later leg's

    # The same steps with the other sampler (outside `value`'s timed region): the headline runs the direct maps,
    # the reference's own rejection loops are reported beside it.
    other = ra.RT_SAMPLER_REFERENCE if sampler == ra.RT_SAMPLER_DIRECT else ra.RT_SAMPLER_DIRECT
    main_sampler, sampler = sampler, other
    one_step()
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    elapsed_other = time.perf_counter() - t1
    if world > 1:
        elapsed_other = max_over_ranks(elapsed_other)
    sampler = main_sampler
    kernel_ms = kernel_ms[:args.steps]
    kernel_events.clear()

    # With several ranks `value` is pipelined throughput (three frames in flight).  What one frame takes with nothing
    # else in flight — render this rank's slice, sum-reduce, tonemap on rank 0, awaited; the max over ranks of the
    # best of three — and what the SAME frame takes on one GPU alone (every rank renders all strata, no reduce; rank
    # 0's tonemap included), timed in this process, give the scaling the line reports beside it.
    frame_latency_s = full_frame_one_gpu_s = None
    if world > 1:
        def awaited(fn):
            best = float("inf")
            for _ in range(3):
                fence()
                t = time.perf_counter()
                fn()
                torch.cuda.synchronize(dev)
                best = min(best, time.perf_counter() - t)
            return max_over_ranks(best)

        def one_frame():
            with torch.cuda.stream(streams[0]):
                render(fbs[0], 0)
                if rehearsal:
                    host_reduce(fbs[0], 0)
                else:
                    dist.reduce(fbs[0], dst=0, op=dist.ReduceOp.SUM)
                tonemap(fbs[0], 0)

        def full_frame():
            scene.render_device(fbs[0].data_ptr(), w, h, sn, max_depth=40, seed=args.seed, kernel=kernel,
                                stream=streams[0].cuda_stream, sampler=sampler, want_stats=False)
            tonemap(fbs[0], 0)

        frame_latency_s = awaited(one_frame)
        full_frame_one_gpu_s = awaited(full_frame)
        kernel_events.clear()

    # Algorithmic work per sample from the counting build of the same kernel, outside the timed region,
    # at reduced spp (per-sample counts do not depend on spp).
    cnt = scene.render_device(fb.data_ptr(), w, h, min(sn, 4), max_depth=40, seed=args.seed, kernel=kernel,
                              collect_stats=True, stream=stream, sampler=sampler)
    fl = flops_per_sample(cnt)
    my_samples = w * h * my_strata  # samples this rank rendered per step
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
        tail = (8 if info.render_depth <= 18 else 4) if info.render_nodes else 1  # render_persistent.hip: tail_chunks
        hbm_bytes = 36.0 * w * h + 768.0 * tiles * handout_groups(my_samples // (w * h), tail)

    if rank == 0:
        workload = (f"scenes/{args.scene}.yaml {w}x{h} {sn * sn} spp, max_depth 40, "
                    f"{args.kernel} kernel, {args.sampler} sampler")
        traffic, traffic_source = pmc_traffic(workload) if world == 1 else (None, "single-GPU profile only")
        line = {
            "metric": "Msamples/sec (whole node)", "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": f"the reference's own scene file scenes/{args.scene}.yaml + its OBJ mesh (not synthetic)",
            "config": {"workload": workload, "seconds_per_frame": round(elapsed / args.steps, 4),
                       **({"timing": "pipelined throughput: 3 frames in flight on 3 streams (seconds_per_frame = elapsed / steps)",
                           "frames_in_flight": 3,
                           "frame_latency_s": round(frame_latency_s, 5),
                           "full_frame_one_gpu_s": round(full_frame_one_gpu_s, 5),
                           "scaling_efficiency_vs_n1": round(full_frame_one_gpu_s / (world * elapsed / args.steps), 4),
                           "latency_efficiency_vs_n1": round(full_frame_one_gpu_s / (world * frame_latency_s), 4)}
                          if world > 1 else {"timing": "one frame at a time, awaited"}),
                       "parallelism": (f"REHEARSAL: {world} ranks sharing one GPU, gloo" if rehearsal else
                                       f"strata-sharded x{world}, 1 RCCL sum-reduce/frame") if world > 1 else "1 GPU",
                       "scene_surfaces": int(info.n_surfaces), "render_bvh_nodes": int(info.render_nodes),
                       "render_bvh_depth": int(info.render_depth), "loose_prims": int(info.n_loose),
                       "scene_setup_s": round(setup_s, 3), "scene_load_s": round(load_s, 3),
                       "tree_builds_and_flatten_s": round(info.build_ms / 1e3, 3), "device_context_s": round(context_s, 3), "library_first_use_s": round(library_s, 3),
                       "other_sampler": {"sampler": "reference" if main_sampler == ra.RT_SAMPLER_DIRECT else "direct",
                                         "value": round(samples_per_step * args.steps / elapsed_other / 1e6, 3),
                                         "unit": "Msamples/s", "ms_per_step": round(elapsed_other / args.steps * 1e3, 3)}},
            "roofline": {"bound": "valu_fp32", "achieved": round(achieved_tflops, 4), "peak": PEAK_FP32_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved_tflops / PEAK_FP32_VALU_TFLOPS, 5),
                         "traffic": traffic, "traffic_source": traffic_source,
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
            line["cpu_baseline"] = cpu_baseline(scene_file, w, h, args.cpu_seconds)
        if png_image is not None:
            ra.write_png(args.png, png_image)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
