"""GPU: the launch shape bench.py TIMES, against the oracle ("the timed configuration is not the tested one").

What the driver's `python bench.py --gpus 1 --steps 20 --warmup 5` (and the default 256-step command) really runs is RM:798-821
repeated — uniforms -> Dispatch -> Blit(_target, _converged, additionMaterial) -> Blit(_converged, destination) — on C3 at 1920x1080
with default options: the library defers the frames and traces 20 (or 64) of them in ONE persistent launch whose tile runs are
interleaved across the frames (`frame_group` 64, `xcd_run` auto by launch size), every frame's Result in its own slot of a slab, and the
blends + the present fused into one `k_blit_add_multi` pass.  The other full-size tests render one frame per launch; the batching
tests use 200x120 scenes that leave most of the resident waves without work.  Here:

  (a) C3 1080p, 20 x OnRenderImage(destination), defaults: ONE launch; `destination`, `_converged` == the oracle's 20-frame running
      mean (frame uniforms 0..19, AS:9,39-41), `_target` == the oracle's frame 19, rays == the oracle's 20-frame sum — bit for bit;
      64 frames (the 64-slot slab, xcd_run > 1) == one launch per frame;
  (b) 3840x2160 with a batch whose Result slab exceeds 4 GiB (40 frames x 132.7 MB: 64-bit slot offsets): `_converged` ==
      frames_per_launch 1 and 16, the last `_target` == the oracle;
  (c) BASELINE config 5's "1024-spp progressive accumulate" as a property test: finite, watchdog 0, same bits for two batch sizes,
      and the mean really is a mean of 1,024 frames (it moved away from frame 0 and its noise fell).
"""
import hashlib
import os

import numpy as np
import pytest

from oracle import pyoracle
from unityraytracer_amd import RayTraceMaster, RenderTexture, debug_build_blas, scenes

pytestmark = pytest.mark.gpu

_scenes = {}


def scene(cfg):
    if cfg not in _scenes:
        _scenes[cfg] = scenes.CONFIGS[cfg]()
    return _scenes[cfg]


def threads():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(p)))
    except Exception:
        pass
    return max(1, min(n, 32))


def bits_equal(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def product_oracle(sc):
    o = pyoracle.Oracle(sc)
    if len(sc.mesh_objects):
        nodes, tri, root, _, _ = debug_build_blas(sc.mesh_objects, sc.vertices, sc.indices)
        o.set_blas(nodes, tri, root)
    return o


def oracle_frame(o, sc, i, counters=False):
    """Frame i of the documented sequence (RayTraceMaster.SetShaderParameters: frame 0 = the scene's fixture uniforms)."""
    ox, oy, sd = (sc.pixel_offset[0], sc.pixel_offset[1], sc.seed) if i == 0 else scenes.frame_uniforms(i)
    o.set_frame((ox, oy), sd)
    return o.render(mode=1, threads=threads(), counters=counters)


def render(ctx, sc, n, fpl, present=True):
    """n frames of RM's protocol with the present, default kernel; returns (destination, _converged, last _target, counters, launch info)."""
    ctx.set_option("kernel_mode", 3)
    ctx.set_option("frames_per_launch", fpl)
    ctx.set_option("count_stats", 0)
    m = RayTraceMaster(ctx, sc)
    dest = RenderTexture(ctx, sc.width, sc.height) if present else None
    m.OnRenderImage(dest)                          # (scene preparation happens at the first dispatch)
    ctx.synchronize()
    m._frame = 0
    m._currentSample = 0                           # sample 0 blends with alpha 1: the accumulation restarts at frame 0
    ctx.reset_counters()
    for _ in range(n):
        m.OnRenderImage(dest)
    c = ctx.counters()                             # submits the deferred frames
    info = ctx.launch_info()
    out = (dest.GetPixels() if present else None, m._converged.GetPixels(), m._target.GetPixels(), c, info)
    m.OnDisable()
    if dest is not None:
        dest.Release()
    ctx.set_option("frames_per_launch", 0)
    return out


@pytest.mark.timeout(900)
def test_c3_1080p_the_drivers_20_frame_launch_equals_the_oracle(gpu_ctx):
    sc = scene("C3")
    n = 20
    dest, conv, last, c, info = render(gpu_ctx, sc, n, 0)
    assert c["launches"] == 1 and c["dispatches"] == n and c["watchdog_trips"] == 0, c
    assert info["kernel"] == "k_sched<false, 256, 0, false, false>" and info["n_frames"] == n and info["frame_group"] == n, info
    assert info["slab_frames"] >= n and info["experiment"] == 0, info
    o = product_oracle(sc)
    acc, rays, img = None, 0, None
    for i in range(n):
        img, oc = oracle_frame(o, sc, i, counters=True)
        rays += oc["rays"]
        acc = pyoracle.accumulate(img, acc if acc is not None else np.zeros_like(img), i)
    assert c["rays"] == rays, (c["rays"], rays)
    for name, got, want in (("_target of frame 19", last, img), ("_converged", conv, acc), ("destination", dest, acc)):
        if not bits_equal(got, want):
            bad = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
            d = float(np.nanmax(np.abs(got.astype(np.float64) - want.astype(np.float64))))
            raise AssertionError(f"{name}: {bad} of {sc.width * sc.height} pixels differ from the oracle's 20-frame result, max |d| = {d:.3e} (north_star tolerance 1e-4)")


@pytest.mark.timeout(900)
def test_c3_1080p_64_frame_launch_equals_one_launch_per_frame(gpu_ctx):
    """The default command's shape: 64 frames = the full 64-slot slab, tile runs of `xcd_run` > 1 interleaved over 64 frames."""
    sc = scene("C3")
    n = 64
    d64, c64, t64, k64, info = render(gpu_ctx, sc, n, 0)
    assert k64["launches"] == 1 and k64["dispatches"] == n and k64["watchdog_trips"] == 0, k64
    assert info["n_frames"] == 64 and info["xcd_run"] > 1 and info["frame_group"] == 64, info
    d1, c1, t1, k1, _ = render(gpu_ctx, sc, n, 1)
    assert k1["launches"] == n and k1["rays"] == k64["rays"]
    assert sha(d64) == sha(d1) and sha(c64) == sha(c1) and sha(t64) == sha(t1)
    assert bits_equal(d64, c64)                    # the present of the last frame IS the running mean


@pytest.mark.timeout(1500)
def test_2160p_batch_with_a_result_slab_beyond_4_gib(gpu_ctx):
    """40 frames of C4 at 3840x2160 in one launch: 40 x 132.7 MB = 5.3 GB of Result slots, i.e. slot offsets beyond 2^32 bytes
    (kernels.hip: frame x frame_stride in 64 bits), the masked object-level phase (`k_sched<.., 3, ..>`), 129,600 tiles per frame."""
    sc = scene("C4")
    n = 40
    dest, conv, last, c, info = render(gpu_ctx, sc, n, 64)
    assert c["launches"] == 1 and c["dispatches"] == n and c["watchdog_trips"] == 0, c
    assert info["kernel"] == "k_sched<false, 256, 3, false, false>" and info["n_frames"] == n, info
    assert info["slab_frames"] * sc.width * sc.height * 16 > (4 << 30), info
    for fpl, launches in ((1, n), (16, 3)):
        d2, c2, t2, k2, _ = render(gpu_ctx, sc, n, fpl)
        assert k2["launches"] == launches and k2["rays"] == c["rays"], (fpl, k2)
        assert sha(c2) == sha(conv) and sha(t2) == sha(last) and sha(d2) == sha(dest), fpl
    o = product_oracle(sc)
    want = oracle_frame(o, sc, n - 1)              # the frame that landed in the LAST slot (offset 39 x 132.7 MB)
    assert bits_equal(last, want)
    assert bits_equal(dest, conv)


@pytest.mark.timeout(1500)
def test_c5_1024_spp_progressive_accumulation(gpu_ctx):
    """BASELINE config 5: 983,040 triangles, 3840x2160, 16 bounces, 1,024 accumulated frames (sixteen 64-frame launches)."""
    sc = scene("C5")
    n = 1024
    dest, conv, last, c, info = render(gpu_ctx, sc, n, 0)
    assert c["dispatches"] == n and c["launches"] == 16 and c["watchdog_trips"] == 0, c
    assert info["kernel"] == "k_sched<false, 256, 3, false, false>", info
    assert np.isfinite(conv).all() and np.isfinite(last).all()
    assert bits_equal(dest, conv)
    _, conv32, last32, c32, _ = render(gpu_ctx, sc, n, 32)
    assert c32["launches"] == 32 and c32["rays"] == c["rays"]
    assert sha(conv32) == sha(conv) and sha(last32) == sha(last)
    # it IS a mean of 1,024 frames: the alpha channel (AS:39-41 blend the fragment's alpha a = 1 / (_Sample + 1) like the colours) holds the
    # value the sample sequence 0..1023 leaves in every pixel, and the frame-to-frame noise of a single frame (|frame 1023 - mean|) is far
    # above the noise left in the mean (|mean of 1024 - mean of the first 512|)
    from unityraytracer_amd import strips
    assert np.all(conv[..., 3].view(np.uint32) == np.float32(strips.running_mean_alpha(range(n))).view(np.uint32))
    m = RayTraceMaster(gpu_ctx, sc)
    for _ in range(512):
        m.OnRenderImage()
    half = m._converged.GetPixels()
    m.OnDisable()
    crop = (slice(0, None, 4), slice(0, None, 4))  # every fourth pixel of the frame
    noise_frame = float(np.mean(np.abs(last[crop][..., :3] - conv[crop][..., :3])))
    noise_mean = float(np.mean(np.abs(half[crop][..., :3] - conv[crop][..., :3])))
    assert noise_mean < 0.2 * noise_frame, (noise_mean, noise_frame)
