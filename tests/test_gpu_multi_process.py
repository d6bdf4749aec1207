"""GPU: the one-process-per-GPU path with REAL processes — two ranks share the box's one card, each with its own context:
strip dispatch (global pixel ids), local accumulation over several frames, RayTraceMaster.gather_converged (pack kernel ->
one torch.distributed gather -> de-interleave kernel on rank 0).  The collective runs over gloo here (host staging); the
RCCL transport itself is only exercised by the driver's multi-GPU bench.  Result must be bit-identical to one context."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path, frames):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from unityraytracer_amd import Context, RayTraceMaster, scenes
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    sc = scenes.mixed_test_scene(136, 100)                     # 13 group rows, ragged last strip
    with Context(0) as ctx:
        m = RayTraceMaster(ctx, sc, rank=rank, world_size=world)
        for _ in range(frames):
            m.OnRenderImage()
        img = m.gather_converged(dist, torch.device("cuda", 0))
        if rank == 0:
            np.save(out_path, img)
        m.OnDisable()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_processes_on_one_card_gather_equals_single_context(gpu_ctx, tmp_path):
    import torch.multiprocessing as mp
    from unityraytracer_amd import RayTraceMaster, scenes
    frames = 3
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), out, frames), nprocs=2, join=True)
    sc = scenes.mixed_test_scene(136, 100)
    m = RayTraceMaster(gpu_ctx, sc)
    for _ in range(frames):
        m.OnRenderImage()
    want = m._converged.GetPixels()
    m.OnDisable()
    got = np.load(out)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.timeout(900)
def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 ...` with no launcher around it (the shape of the driver's command): bench.py must start the
    two ranks itself, run the multi-GPU step (strips, local accumulation, pack, one gather per frame, de-interleave) and print
    ONE JSON line.  Two ranks share this box's card; the collective runs over gloo (URT_DIST_BACKEND) and rank 0 verifies the
    gathered image against a single-rank render bit for bit (URT_BENCH_VERIFY)."""
    import json
    import subprocess
    env = dict(os.environ, URT_DIST_BACKEND="gloo", URT_BENCH_VERIFY="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--config", "C2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "[verify] gathered 2-rank frame == single-rank frame: True" in r.stderr, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["value"] > 0 and j["present"] is True


def _nccl_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from unityraytracer_amd import Context, RayTraceMaster, RenderTexture, scenes, strips
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=dev)
    sc = scenes.mixed_test_scene(136, 100)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    with Context(0) as ctx:
        ctx.set_stream(st.cuda_stream)
        m = RayTraceMaster(ctx, sc, rank=rank, world_size=world)
        m.OnRenderImage()
        n_floats = strips.packed_rows(sc.height, world) * sc.width * 4
        mine = torch.zeros(n_floats, dtype=torch.float32, device=dev)
        parts = [torch.empty_like(mine) for _ in range(world)]
        m._converged.pack_rows(rank, world, mine.data_ptr())
        ctx.flush()
        dist.gather(mine, parts, dst=0)                       # RCCL, on the stream the pack kernel ran on
        full = RenderTexture(ctx, sc.width, sc.height)
        for r in range(world):
            full.unpack_rows(r, world, parts[r].data_ptr())
        np.save(out_path, full.GetPixels())
        full.Release(); m.OnDisable()
        ctx.set_stream(None)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_one_rank_nccl_gather_smoke(gpu_ctx, tmp_path):
    """The RCCL calls of the multi-GPU step (init_process_group("nccl"), dist.gather of the packed strips on the render
    stream) executed once on this one-GPU box: world_size 1, so no xGMI traffic — but the same library, stream hand-over
    (urt_context_set_stream), pack -> gather -> unpack order and buffers as an N-rank run."""
    import torch.multiprocessing as mp
    from unityraytracer_amd import RayTraceMaster, scenes
    out = str(tmp_path / "nccl.npy")
    mp.spawn(_nccl_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    sc = scenes.mixed_test_scene(136, 100)
    m = RayTraceMaster(gpu_ctx, sc)
    m.OnRenderImage()
    want = m._converged.GetPixels()
    m.OnDisable()
    assert np.array_equal(np.load(out).view(np.uint32), want.view(np.uint32))


@pytest.mark.timeout(900)
def test_bench_strong_scaling_two_ranks_rgb_gather(tmp_path):
    """`python bench.py --scaling strong --config C4 --gpus 2`: the FIXED 3840x2160 frame of BASELINE config 4 divided among two ranks
    (sharing this box's card, gloo rehearsal), the frame-end gather moving RGB only (12 B per pixel; rank 0 writes the alpha of the
    running mean itself); rank 0 verifies all four channels of the gathered image against a single-rank render bit for bit."""
    import json
    import subprocess
    env = dict(os.environ, URT_DIST_BACKEND="gloo", URT_BENCH_VERIFY="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--config", "C4",
                        "--scaling", "strong", "--clock-warmup-ms", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "[verify] gathered 2-rank frame == single-rank frame: True" in r.stderr, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["scaling"] == "strong" and j["n_gpus"] == 2 and j["config"]["frame"] == [3840, 2160] and j["config"]["pixels_per_gpu"] == 3840 * 2160 // 2
    assert "RGB, 12 B" in j["config"]["partition"]


@pytest.mark.timeout(900)
def test_bench_rgba_gather_still_available():
    import subprocess
    env = dict(os.environ, URT_DIST_BACKEND="gloo", URT_BENCH_VERIFY="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "C2", "--gather-rgb", "0",
                        "--gather-every", "2", "--clock-warmup-ms", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "[verify] gathered 2-rank frame == single-rank frame: True" in r.stderr, r.stderr[-3000:]


@pytest.mark.timeout(900)
def test_the_drivers_bench_command_verified_against_the_oracle():
    """The driver's own command, `python bench.py --gpus 1 --steps 20 --warmup 5`, with URT_BENCH_VERIFY=1: after the timed region the
    presented image (`destination`, and `_converged`) must equal the oracle's running mean of the same 25 frames bit for bit; the line
    must name the kernel the library launched and carry a roofline fraction that cannot exceed 1."""
    import json
    import subprocess
    env = dict(os.environ, URT_BENCH_VERIFY="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "URT_LIB_PATH", "URT_ALLOW_EXPERIMENT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "== the oracle's running mean: True" in r.stderr, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["verified_against_oracle"] is True and j["n_gpus"] == 1 and j["steps"] == 20 and j["scaling"] == "weak"
    rl = j["roofline"]
    assert rl["kernel"] == "k_sched<false, 256, 0, false, false>" and rl["frames_per_launch"] == 20
    assert rl["frac"] is None or 0 < rl["frac"] <= 1.0, rl
    assert rl["algorithmic"]["bytes_per_launch"] > 0 and "saturated" in rl["algorithmic"]
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["value"] > 0
