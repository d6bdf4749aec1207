#!/usr/bin/env python3
"""bench.py — headline benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): Mrays/sec @1920x1080, 8 bounces.  A *ray* is one Trace() invocation
(RayTraceShader.compute:454), counted by the kernels themselves; a *step* is one frame of the reference's
frame protocol (RayTraceMaster.OnRenderImage: set uniforms -> Dispatch -> AdditionShader blit -> present blit to
`destination`, RM:806-820), with every input already resident in HBM.  Workload at N=1: config C3 of BASELINE.json (bunny-class 69,600-triangle
mesh + triangle BVH, 1920x1080, numBounces 8, numRays 1, synthetic scene from unityraytracer_amd.scenes).

Multi-GPU (weak scaling): one process per GPU; the frame grows to N x (1920x1080) pixels at the same
aspect and camera (N=4 is 3840x2160, BASELINE's multi-GPU resolution), cut into 8-row strips dealt
round-robin to the ranks (global pixel ids -> pixels identical to a single-GPU frame); each rank
accumulates its strips locally and ONE gather per frame (RCCL via torch.distributed) brings them to rank 0.

`--scaling strong` keeps the frame FIXED (C3: 1920x1080; C4 / C5: BASELINE's 3840x2160 "tiled across 8 GPUs") and divides its strips
among the N ranks; the line then says `"scaling": "strong"`.

Rank 0 prints one JSON line.  Extra objects:
  `roofline` — the dominant kernel (the trace kernel, named by the library: urt_debug_launch_info) against the roof that BINDS it.
      SURVEY 8(d)'s algorithmic byte rate saturates on this path (the BVH top and the object tables are in LDS, the rest is
      L2 / Infinity-Cache resident: it exceeds the HBM peak), so it is reported as `algorithmic` (with `frac_of_hbm_peak` and
      `saturated`), the measured HBM side as `hbm` (counter traffic / launch time / 8 TB/s), and `bound` / `achieved` / `peak` / `frac`
      are the VALU issue roof: VALU wave-instructions of the launch (committed SQ_INSTS_VALU pass of the same launch shape,
      profiles/pmc_traffic.json) x 128 flop (a 64-lane fma) / the launch's HIP-event duration measured live here, against the
      157.3 TFLOP/s f32 vector peak (= one wave64 VALU instruction per 2 cycles per SIMD, 1024 SIMDs, 2.4 GHz).  frac <= 1 by construction.
  `cpu_baseline` — the scalar C++ oracle (a port, not the reference) timed on the host cores for a bounded sample of the same workload.
URT_BENCH_VERIFY=1: after the timed region (outside it) the presented image is compared bit for bit with the oracle's accumulation of the
same W + K frames (N = 1), or with a single-rank render (N > 1).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3       # same guide, "F32 (f32 in) 157.3 TF (= vector peak)": 256 CUs x 4 SIMDs x 32 lanes/cycle x 2 flop x 2.4 GHz,
                               # i.e. one wave64 VALU instruction (priced as a 64-lane fma = 128 flop) per 2 cycles per SIMD

# ALGORITHMIC bytes per counted event — SURVEY.md §8(d)'s table, the figure `roofline.achieved` / `frac` are computed from:
#   28 B object-level BVHNode, S_node = 64 B triangle-BVH node (this build's node: two child boxes), S_tri = 36 B world-space
#   triangle, 16 B sphere test, 76 B triangle closest hit (3 normals + material), 40 B sphere hit, 16 B per pixel written.
BYTES_8D = {"tlas_nodes": 28, "blas_nodes": 64, "tri_tests": 36, "sphere_tests": 16, "hit_tri": 76, "hit_sphere": 40, "pixels": 16}
# what the kernel's records really are (reported separately as `padded_record_bytes_per_launch`): 48-B padded triangle
# records (v0,e1,e2 as float4), 48 B of normals + 40 B material per triangle hit, and the 4 sky texels of a path's last bounce
BYTES_PADDED = {"tlas_nodes": 28, "blas_nodes": 64, "tri_tests": 48, "sphere_tests": 16, "hit_tri": 88, "hit_sphere": 40,
                "hit_sky": 64, "pixels": 16}


def algorithmic_bytes(c: dict, table=BYTES_8D) -> int:
    return sum(table[k] * int(c[k]) for k in table)


def usable_cores() -> int:
    """Host cores this process may really use: affinity mask, capped by the cgroup CPU quota (cpu.max)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 256 timed frames = four launches of 64 deferred frames (about 75 ms of GPU time at 1080p; round 1's 20-frame
    # default timed 15 ms) after 8 warm-up frames; still a few seconds end to end
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default="C3", help="BASELINE.json config to trace (C2..C5, C3D = C3 in close-up); the metric is quoted on C3")
    ap.add_argument("--kernel-mode", type=int, default=None, help="0 per-pixel, 1 per-bounce queues, 2 persistent, 3 persistent + phase-scheduled lanes (default), 4 path pool in LDS, 5 shared traversal service")
    ap.add_argument("--frames-per-launch", type=int, default=None, help="library option frames_per_launch (0 auto, 1 = one launch per frame, 2..64)")
    ap.add_argument("--gather-every", type=int, default=1, help="multi-GPU: gather the accumulated strips to rank 0 after every K-th frame (and after the last); "
                    "1 = the frame-end gather of every frame (default); a gather overwrites the whole image, so K > 1 only lowers the rate at which rank 0 could present it")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="urt_set_option NAME VALUE before the run (A/B of library options with the bench's own metric; reported in config.options)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="multi-GPU: weak = the frame grows with N (default, the driver's scaling bench); strong = the config's own frame "
                         "(C4 / C5: 3840x2160, BASELINE configs 4 / 5) is divided among the N ranks")
    ap.add_argument("--gather-rgb", type=int, default=1,
                    help="multi-GPU: 1 (default) = the frame-end gather moves RGB only (12 B per pixel): the alpha channel of the running mean is a "
                         "function of the sample index alone (AS:40) and rank 0 writes it itself; 0 = all four channels (16 B per pixel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--clock-warmup-ms", type=float, default=60.0,
                    help="before the W warm-up steps: run the same step untimed for this long so that the GPU has left its idle clocks "
                         "(a step is 0.25 ms: W = 5 steps are 1.3 ms of GPU work, far less than the clock governor's reaction time; measured: "
                         "the 20-frame launch takes 5.2 ms from idle clocks, 4.75 ms after >= 15 ms of load).  Reported as `clock_warmup`; 0 = off")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` as issued by hand or by a driver that does not wrap it in torch.distributed.run: this process
        # has made no GPU call yet (torch is not even imported), so it may start the N ranks itself — as a CHILD process, never
        # an exec — forward rank 0's JSON line (the child's stdout is inherited) and exit with the launcher's code.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    import torch
    import torch.distributed as dist

    from unityraytracer_amd import Context, RayTraceMaster, scenes, strips

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # URT_DIST_BACKEND=gloo is a REHEARSAL mode for boxes with fewer GPUs than ranks (ranks share a card, the gather is
    # staged through host memory); the driver's real multi-GPU runs use nccl (= RCCL over xGMI), one GPU per rank.
    backend = os.environ.get("URT_DIST_BACKEND", "nccl")
    if backend != "gloo" and world > torch.cuda.device_count():
        raise SystemExit(f"bench.py --gpus {world}: this node shows {torch.cuda.device_count()} GPU(s); the nccl (RCCL) backend needs one GPU per rank. "
                         "(URT_DIST_BACKEND=gloo is the rehearsal mode in which ranks share a card.)")
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # ---- workload --------------------------------------------------------------------------------
    base_w, base_h = (1920, 1080) if args.config in ("C2", "C3", "C3D") else (3840, 2160)
    if args.scaling == "strong":
        s = 1.0                                                # the config's own frame, whatever N
    else:
        s = math.sqrt(world) if args.config in ("C2", "C3", "C3D") else math.sqrt(world / 8.0) if world > 1 else 1.0
    width, height = int(round(base_w * s)), int(round(base_h * s))
    scene = scenes.CONFIGS[args.config](width, height)

    ctx = Context(dev_index)
    main_stream = None
    if world > 1:
        # ONE explicit (non-default) stream shared by torch and the library: the HIP kernels, the pack/unpack kernels and the
        # events that order the collective all live on it.  (torch's default stream has handle 0, which the C ABI reads as
        # "use the library's own stream" — the two would then be unordered.)
        main_stream = torch.cuda.Stream(device=device)
        torch.cuda.set_stream(main_stream)
        assert main_stream.cuda_stream != 0
        ctx.set_stream(main_stream.cuda_stream)
    # N = 1: the library works on its own stream, where it batches consecutive frames into one persistent launch
    # (include/urt.h "frame batching"); fence() below submits and waits through the C ABI.
    if args.frames_per_launch is not None:
        ctx.set_option("frames_per_launch", args.frames_per_launch)
    if args.kernel_mode is not None:
        ctx.set_option("kernel_mode", args.kernel_mode)
    for kv in args.opt:
        k, v = kv.split("=", 1)
        ctx.set_option(k, int(v))
    ctx.set_option("time_dispatch", 1)
    master = RayTraceMaster(ctx, scene, rank=rank, world_size=world)

    # ---- multi-GPU frame-end gather, pipelined in bursts --------------------------------------------------------------
    # Every frame ends with ONE collective (gather of the rank's accumulated strips to rank 0).  The library batches B
    # consecutive frames into one persistent launch per rank (frames_per_launch, explicit because the stream is shared with
    # torch); frame i's strips are packed into their own buffer by a pack kernel queued behind the batch.  After B frames
    # the batch is submitted and its B gathers run on a second stream — overlapping with the NEXT batch's rendering — where
    # rank 0 also de-interleaves them (urt_texture_unpack_rows_on), so that its render stream carries exactly what the other
    # ranks' do.  Pack buffers form a ring of 2B, every reuse is ordered by events.  URT_BENCH_NO_OVERLAP=1 serialises.
    overlap = world > 1 and os.environ.get("URT_BENCH_NO_OVERLAP") != "1"
    burst = 1
    if world > 1:
        burst = max(1, min(16, args.frames_per_launch if args.frames_per_launch else int(os.environ.get("URT_BENCH_BURST", "16"))))
        ctx.set_option("frames_per_launch", burst)
        rgb = bool(args.gather_rgb)
        n_floats = strips.packed_rows(height, world) * width * (3 if rgb else 4)
        alpha_of = {}                                             # sample index -> alpha of the running mean after it (strips.running_mean_alpha)
        slot_alpha = {}

        def mean_alpha(sample):
            """alpha of the running mean after samples 0..sample (one blend step per new sample: the table grows with the run)"""
            if sample not in alpha_of:
                last = max((k for k in alpha_of if k < sample), default=-1)
                w = alpha_of.get(last, 0.0)
                for k in range(last + 1, sample + 1):
                    w = strips.running_mean_alpha([k], start=w)
                    alpha_of[k] = w
            return alpha_of[sample]
        ring = 2 * burst
        packed = [torch.zeros(n_floats, dtype=torch.float32, device=device) for _ in range(ring)]
        gathered = [[torch.empty(n_floats, dtype=torch.float32, device=device) for _ in range(world)] for _ in range(ring)] if rank == 0 else [None] * ring
        comm_stream = torch.cuda.Stream(device=device) if overlap else main_stream
        ev_batch = torch.cuda.Event()
        ev_gather = [torch.cuda.Event() for _ in range(ring)]
        full = None
        if rank == 0:
            from unityraytracer_amd import RenderTexture
            full = RenderTexture(ctx, width, height)
    # The step is the reference's LITERAL frame (RM:806-820): Dispatch -> Blit(_target, _converged, additionMaterial) ->
    # Blit(_converged, destination).  N = 1: `destination` is a RenderTexture the frame is presented to every step (the library
    # queues the present with the deferred frames and fuses it into the blend pass).  N > 1: the presented image is `full` on
    # rank 0 — the gather's de-interleave writes it every frame, that IS the present of the gathered frame.
    destination = None
    if world == 1:
        from unityraytracer_amd import RenderTexture
        destination = RenderTexture(ctx, width, height)
    state = {"i": 0, "pending": []}

    def do_gather(slot):
        if backend == "nccl":
            dist.gather(packed[slot], gathered[slot], dst=0)              # the ONE collective of the frame
        else:                                                             # rehearsal: same data path through host memory
            host = packed[slot].cpu()
            parts = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
            dist.gather(host, parts, dst=0)
            if rank == 0:
                for r in range(world):
                    gathered[slot][r].copy_(parts[r])

    def submit():
        """Submit the deferred batch (trace launch + blends + packs) and run its gathers on the communication stream."""
        ctx.flush()
        if not state["pending"]:
            return
        ev_batch.record(main_stream)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ev_batch)
            for slot in state["pending"]:
                do_gather(slot)
                if rank == 0:
                    for r in range(world):
                        if rgb:
                            full.unpack_rows_rgb(r, world, gathered[slot][r].data_ptr(), slot_alpha[slot], stream=comm_stream.cuda_stream)
                        else:
                            full.unpack_rows(r, world, gathered[slot][r].data_ptr(), stream=comm_stream.cuda_stream)
                ev_gather[slot].record(comm_stream)
        if not overlap:
            main_stream.wait_event(ev_gather[state["pending"][-1]])
        state["pending"] = []

    def step():
        if world == 1:
            master.OnRenderImage(destination)
            return
        i = state["i"]
        slot = i % ring
        if i >= ring:
            main_stream.wait_event(ev_gather[slot])                       # the gather that last used this pack buffer is done with it
        master.OnRenderImage()                                            # deferred: dispatch of this rank's strips + accumulate
        state["i"] = i + 1
        state["frames"] = state.get("frames", 0) + 1
        if (i + 1) % max(1, args.gather_every) == 0:
            master._converged.pack_rows(rank, world, packed[slot].data_ptr(), rgb=rgb)   # deferred behind them
            slot_alpha[slot] = mean_alpha(master._currentSample - 1) if rgb else None
            state["pending"].append(slot)
            state["stale"] = False
        else:
            state["stale"] = True                                          # rank 0's image is behind: drain() gathers the last frame
        if state["frames"] >= burst:
            state["frames"] = 0
            submit()

    def drain():
        if world > 1:
            if state.get("stale"):
                slot = (state["i"] - 1) % ring
                master._converged.pack_rows(rank, world, packed[slot].data_ptr(), rgb=rgb)
                slot_alpha[slot] = mean_alpha(master._currentSample - 1) if rgb else None
                state["pending"].append(slot)
                state["stale"] = False
            submit()
            for e in ev_gather:
                main_stream.wait_event(e)

    def fence():
        ctx.synchronize()                     # submits the library's deferred frames and waits for its stream
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    # ---- clock warm-up (untimed, reported): the step itself, repeated until the GPU has been busy for --clock-warmup-ms ----
    clock_warmup = {"steps": 0, "ms": 0.0}
    if args.clock_warmup_ms > 0:
        tw = time.perf_counter()
        # N > 1: every rank must run the SAME number of steps (each carries a collective): a fixed count there, the clock decides at N = 1
        rounds = None if world == 1 else max(1, int(math.ceil(args.clock_warmup_ms / 20.0)))
        while (rounds is None and (time.perf_counter() - tw) * 1e3 < args.clock_warmup_ms) or (rounds is not None and rounds > 0):
            for _ in range(64):
                step()
            drain()
            fence()
            clock_warmup["steps"] += 64
            if rounds is not None:
                rounds -= 1
        clock_warmup["ms"] = round((time.perf_counter() - tw) * 1e3, 1)
        master._frame = 0                      # the W warm-up and K timed steps are frames 0 .. W+K-1 of the documented sequence, as without it
        master._currentSample = 0              # (sample 0 blends with alpha 1: the accumulation restarts)
    for _ in range(args.warmup):
        step()
    drain()
    fence()
    ctx.reset_counters()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                                   # the last frame's strips are on rank 0 and de-interleaved inside the timed region
    fence()
    elapsed = time.perf_counter() - t0
    c = ctx.counters()

    red_dev = device if backend == "nccl" else torch.device("cpu")
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    rays = torch.tensor([float(c["rays"])], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_rays = float(rays.item())

    # ---- N = 1 verification against the oracle (URT_BENCH_VERIFY=1), outside the timed region: what `destination` holds now is the
    # present of the last timed step = the running mean of frames 0 .. W+K-1 (the clock warm-up restarted the sequence) ----
    verify_img = None
    if world == 1 and os.environ.get("URT_BENCH_VERIFY") == "1":
        verify_img = (destination.GetPixels(), master._converged.GetPixels())

    # ---- roofline of the dominant kernel (rank 0): replay the same frames with traversal counters on -----------
    roofline = None
    issue = None
    cpu = None
    if rank == 0:
        # trace_ms = sum of the trace launches' own HIP-event durations (recorded by the library on the stream it launches
        # on).  A batched launch traces several frames, so "per frame" is the launch time divided by its frames.
        kernel_ms = c["trace_ms"] / max(1, c["dispatches"])
        launch_ms = c["trace_ms"] / max(1, c["launches"])
        frames_per_launch = c["dispatches"] / max(1, c["launches"])
        linfo = ctx.launch_info()                                      # the instantiation the library really launched (not a guess from the scene)
        ctx.set_option("count_stats", 1)
        ctx.set_option("time_dispatch", 0)
        master._frame = args.warmup
        ctx.reset_counters()
        for _ in range(args.steps):
            master.OnRenderImage(destination)
        cc = ctx.counters()
        ctx.set_option("count_stats", 0)
        assert cc["rays"] == c["rays"], "counting replay traced a different number of rays"
        alg_frame = algorithmic_bytes(cc) / max(1, cc["dispatches"])
        alg = alg_frame * frames_per_launch                            # per LAUNCH, like launch_ms
        padded = algorithmic_bytes(cc, BYTES_PADDED) / max(1, cc["dispatches"]) * frames_per_launch
        alg_rate = alg / (launch_ms * 1e-3) / 1e9
        # Counter-side figures are NOT measured in this run (PMC needs rocprofv3): they are the committed passes of the last profiled
        # build, per launch of the SAME shape (config and frames per launch), or null when no such profile exists
        traffic, traffic_source, issue, valu_insts, pmc_kernel = None, None, None, None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")       # written from rocprofv3 --pmc passes (profiles/README.md)
        if os.path.exists(pmc) and world == 1 and args.kernel_mode in (None, 3) and not args.opt:
            try:
                j = json.load(open(pmc)).get("configs", {}).get(args.config) or {}
                j = (j.get("by_frames_per_launch") or {}).get(str(int(round(frames_per_launch))))     # the entry of THIS launch shape
                if j:
                    traffic = j.get("hbm_bytes_per_launch")
                    traffic_source = f"profiles/pmc_traffic.json ({j.get('round')}: separate rocprofv3 --pmc passes of `{j.get('command')}`, an earlier run of this launch shape — not measured in this run)"
                    issue = j.get("issue")
                    valu_insts = (issue or {}).get("valu_insts_per_launch")
                    if valu_insts is None and issue and issue.get("valu_insts_per_frame"):
                        valu_insts = issue["valu_insts_per_frame"] * frames_per_launch
                    pmc_kernel = j.get("kernel")
            except Exception:
                traffic = None
        kname = linfo["kernel"]
        if pmc_kernel and kname not in pmc_kernel:                     # the committed counters are of another instantiation: not this kernel's
            traffic, traffic_source, issue, valu_insts = None, None, None, None
        algorithmic = {"byte_table": "SURVEY.md 8(d): 28 B object node, 64 B BVH node, 36 B triangle, 16 B sphere, 76 B triangle hit, 40 B sphere hit, 16 B pixel",
                       "bytes_per_launch": int(alg), "bytes_per_ray": round(alg_frame * cc["dispatches"] / max(1, cc["rays"]), 1),
                       "rate": round(alg_rate, 1), "unit": "GB/s", "frac_of_hbm_peak": round(alg_rate / HBM_PEAK_GBS, 4),
                       "saturated": bool(alg_rate >= 0.8 * HBM_PEAK_GBS),
                       "note": "every counted node / triangle / hit record priced as if it came from HBM; the BVH top and the object tables are in LDS, "
                               "the rest is L2 / Infinity-Cache resident, so this rate can exceed the HBM peak: it is a work rate, not a roofline",
                       "padded_record_bytes_per_launch": int(padded)}
        hbm = None
        if traffic:
            hbm = {"achieved": round(traffic / (launch_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "bytes_per_launch": traffic,
                   "over_algorithmic": round(traffic / max(1.0, alg), 3)}
        common = {"algorithmic_frac": algorithmic["frac_of_hbm_peak"],     # SURVEY 8(d)'s byte fraction — saturates (> 1 possible): a work rate, see `algorithmic`
                  "kernel": kname, "kernel_mode": linfo["kernel_mode"], "launch_ms": round(launch_ms, 4), "frames_per_launch": round(frames_per_launch, 2),
                  "kernel_ms_per_frame": round(kernel_ms, 4), "traffic": traffic, "traffic_source": traffic_source, "hbm": hbm, "algorithmic": algorithmic,
                  "launch": {k: linfo[k] for k in ("front_mode", "n_blocks", "block_threads", "lds_bytes", "waves_per_cu", "xcd_run", "frame_group", "top_nodes", "slab_frames_max")}}
        if valu_insts:
            # the binding roof: VALU issue.  wave-instructions (committed SQ_INSTS_VALU of this launch shape) x 128 flop / live launch time
            tf = valu_insts * 128.0 / (launch_ms * 1e-3) / 1e12
            roofline = {"bound": "valu", "achieved": round(tf, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / VALU_PEAK_TFLOPS, 4),
                        "what": "VALU wave-instructions of the launch, each priced as a 64-lane fma (128 flop), against the f32 vector peak = one wave64 VALU "
                                "instruction per 2 cycles per SIMD; the lanes that do useful work are `lane_util` of that",
                        "valu_insts_per_launch": int(valu_insts), "lane_util": (issue or {}).get("lane_util"), **common}
        else:
            # no committed counters for this launch shape: the byte rate alone, flagged as saturated when it is; frac stays <= 1 or null
            roofline = {"bound": "hbm", "achieved": round(alg_rate, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg_rate / HBM_PEAK_GBS, 4) if alg_rate < HBM_PEAK_GBS else None,
                        "what": "algorithmic bytes / launch time (no committed PMC passes for this launch shape); null frac = the byte rate exceeds the HBM peak (cache-resident data)",
                        **common}

        # ---- CPU baseline: the oracle (scalar C++ port) on the host cores, bounded sample -------------------------
        if world == 1 and not args.no_cpu_baseline:
            from oracle import pyoracle
            from unityraytracer_amd import debug_build_blas
            o = pyoracle.Oracle(scene)
            if len(scene.mesh_objects):
                nodes, tri, root, _, _ = debug_build_blas(scene.mesh_objects, scene.vertices, scene.indices)
                o.set_blas(nodes, tri, root)
            cores = min(pyoracle.hardware_threads(), usable_cores())
            # bounded sample: whole frames of the same workload (frame uniforms 0, 1, 2, ...) until ~6 s of wall time on
            # all usable cores (= cores x 6 s of CPU work), then a single-thread figure on a crop of frame 0
            cpu_rays, cpu_frames, tc = 0, 0, time.perf_counter()
            while True:
                ox, oy, sd = scenes.frame_uniforms(cpu_frames)
                o.set_frame((ox, oy), sd)
                _, oc = o.render(mode=1, threads=cores, counters=True)
                cpu_rays += oc["rays"]; cpu_frames += 1
                dt = time.perf_counter() - tc
                if dt >= 6.0 or cpu_frames >= 400:
                    break
            o.set_frame((0.5, 0.5), 0.5)
            y0 = height // 2 - height // 32
            t1 = time.perf_counter()
            _, oc1 = o.render(rect=(0, y0, width, y0 + height // 16), mode=1, threads=1, counters=True)
            dt1 = time.perf_counter() - t1
            cpu = {"value": round(cpu_rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
                   "value_1_thread": round(oc1["rays"] / dt1 / 1e6, 3),
                   "sample": f"{cpu_frames} frames of {scene.name} {width}x{height} ({dt:.1f} s wall on {cores} std::threads), BVH-culled scalar oracle; "
                             f"1-thread figure: a {height // 16}-row band of frame 0 ({dt1:.1f} s)"}

    verified = None
    if verify_img is not None:
        # the oracle's accumulation of the same W + K frames (frame uniforms 0 .. W+K-1, AS:9,39-41 blend), bit for bit
        import numpy as np
        from oracle import pyoracle
        from unityraytracer_amd import debug_build_blas
        o = pyoracle.Oracle(scene)
        if len(scene.mesh_objects):
            nodes, tri, root, _, _ = debug_build_blas(scene.mesh_objects, scene.vertices, scene.indices)
            o.set_blas(nodes, tri, root)
        acc = None
        for i in range(args.warmup + args.steps):
            ox, oy, sd = (scene.pixel_offset[0], scene.pixel_offset[1], scene.seed) if i == 0 else scenes.frame_uniforms(i)
            o.set_frame((ox, oy), sd)
            img = o.render(mode=1, threads=min(pyoracle.hardware_threads(), usable_cores()))
            acc = pyoracle.accumulate(img, acc if acc is not None else np.zeros_like(img), i)
        verified = bool(np.array_equal(verify_img[0].view(np.uint32), acc.view(np.uint32)) and np.array_equal(verify_img[1].view(np.uint32), acc.view(np.uint32)))
        print(f"[verify] destination and _converged after {args.warmup + args.steps} frames == the oracle's running mean: {verified}", file=sys.stderr, flush=True)
        if not verified:
            raise SystemExit("the presented image differs from the oracle's accumulation of the same frames")
    if world > 1 and os.environ.get("URT_BENCH_VERIFY") == "1":
        # rank 0 re-renders the LAST frame alone and compares the gathered, accumulated image bit for bit
        fence()
        if rank == 0:
            import numpy as np
            got = full.GetPixels()
            solo = RayTraceMaster(ctx, scene)
            for _ in range(args.warmup + args.steps):      # `full` holds the gather of the last TIMED step
                solo.OnRenderImage()
            want = solo._converged.GetPixels()
            solo.OnDisable()
            same = bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)))
            print(f"[verify] gathered {world}-rank frame == single-rank frame: {same}", file=sys.stderr, flush=True)
            if not same:
                raise SystemExit("multi-rank image differs from the single-rank image")
        fence()
    if rank == 0:
        out = {
            "metric": "Mrays/sec @1920x1080, 8 bounces", "value": round(total_rays / elapsed / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "present": True,     # the step includes RM:819's Graphics.Blit(_converged, destination) (N > 1: the gather into rank 0's image)
            "clock_warmup": clock_warmup,   # untimed steps run BEFORE the W warm-up steps to leave idle clocks (--clock-warmup-ms; 0 disables)
            "config": {"workload": f"{args.config}: {scene.name}, {scene.n_triangles} triangles + triangle BVH, {len(scene.spheres)} spheres, "
                                   f"ground plane, equirect sky; numBounces {scene.num_bounces}, numRays {scene.num_rays}",
                       "frame": [width, height], "pixels_per_gpu": width * height // world,
                       "partition": (f"8-row strips round-robin over ranks, one gather per frame ({'RGB, 12' if rgb else 'RGBA, 16'} B per pixel); {burst} frames per launch, each burst's gathers" + (" overlap the next burst's rendering" if overlap else " serialised")) if world > 1 else "single GPU",
                       "rays_per_step": int(total_rays / args.steps), "kernel_mode": args.kernel_mode if args.kernel_mode is not None else 3,
                       **({"gather_every": args.gather_every} if world > 1 else {}), **({"options": args.opt} if args.opt else {})},
            "roofline": roofline, "issue": issue, "cpu_baseline": cpu,
            **({"verified_against_oracle": verified} if verified is not None else {}),
        }
        print(json.dumps(out), flush=True)
    master.OnDisable()
    if destination is not None:
        destination.Release()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
