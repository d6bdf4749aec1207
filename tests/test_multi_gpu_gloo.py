"""The N>1 path on CPU: world_size-2 gloo.  Each rank renders ITS strips (global pixel ids) — here with the oracle
standing in for the GPU, since this container has none — packs them, ONE gather brings them to rank 0, which
de-interleaves and compares with the single-process full frame.  Exercises unityraytracer_amd.strips, the code
the GPU ranks run around the HIP dispatch_rows/pack_rows calls."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import pyoracle
    from unityraytracer_amd import scenes, strips
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    sc = scenes.mixed_test_scene(72, 52)                     # 52 rows: 7 strips, the last one ragged (4 rows)
    o = pyoracle.Oracle(sc)
    o.build_own_blas()
    local = np.zeros((sc.height, sc.width, 4), np.float32)   # this rank's render target: only its strips get written
    for y0, y1 in strips.strip_row_ranges(sc.height, rank, world):
        local[y0:y1] = o.render(rect=(0, y0, sc.width, y1), mode=1)
    mine = torch.from_numpy(strips.pack_rows_host(local, rank, world).reshape(-1).copy())
    parts = strips.gather_to_root(dist, mine, rank, world)   # the single frame-end collective
    if rank == 0:
        img = strips.unpack_rows_host([p.numpy() for p in parts], sc.width, sc.height)
        np.save(out_path, img)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_strips_gather_equals_single_frame(tmp_path):
    import torch.multiprocessing as mp
    from oracle import pyoracle
    from unityraytracer_amd import scenes
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    sc = scenes.mixed_test_scene(72, 52)
    o = pyoracle.Oracle(sc)
    o.build_own_blas()
    full = o.render(mode=1)
    got = np.load(out)
    assert np.array_equal(got.view(np.uint32), full.view(np.uint32))


def test_running_mean_alpha_is_a_function_of_the_sample_sequence():
    """The RGB gather (urt_texture_pack_rows_rgb / unpack_rows_rgb) leaves the alpha channel of `_converged` at home: AdditionShader's
    source alpha is a = 1 / (_Sample + 1) and is blended like the colours (AS:39-41), so every pixel holds the same value after sample
    n.  strips.running_mean_alpha must be exactly what the oracle's blend leaves in the alpha channel, whatever the colours."""
    from oracle import pyoracle
    from unityraytracer_amd import strips
    rng = np.random.default_rng(7)
    conv = np.zeros((3, 5, 4), np.float32)
    for n in range(70):
        frame = rng.random((3, 5, 4), dtype=np.float32) * 3.0
        frame[..., 3] = 1.0                                   # CSMain writes alpha 1 (RS:468) — the blend does not read it
        conv = pyoracle.accumulate(frame, conv, n)
        w = np.float32(strips.running_mean_alpha(range(n + 1)))
        assert np.all(conv[..., 3].view(np.uint32) == w.view(np.uint32)), n
    # a reset (sample 0 blends with alpha 1) restarts the sequence, whatever came before
    conv = pyoracle.accumulate(frame, conv, 0)
    conv = pyoracle.accumulate(frame, conv, 1)
    assert np.all(conv[..., 3] == np.float32(strips.running_mean_alpha([0, 1])))
