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
