"""GPU: frame batching (include/urt.h "frame batching") is invisible.  The library defers dispatches and the AdditionShader
blits that follow them and traces several consecutive frames with ONE persistent launch; whatever a caller can observe
(`_target` of any frame, the running mean `_converged`, strips, counters) must be bit-identical to one launch per frame, and
to the oracle."""
import numpy as np
import pytest

from oracle import pyoracle
from unityraytracer_amd import Graphics, RayTraceMaster, RenderTexture, debug_build_blas, scenes

pytestmark = pytest.mark.gpu


def bits_equal(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


def run_frames(ctx, sc, n, fpl, peek_at=()):
    """n frames of RM's protocol under frames_per_launch = fpl; returns (_converged, _target of the last frame, peeks, counters)."""
    ctx.set_option("kernel_mode", 3)
    ctx.set_option("frames_per_launch", fpl)
    ctx.reset_counters()
    m = RayTraceMaster(ctx, sc)
    peeks = {}
    for i in range(n):
        m.OnRenderImage()
        if i in peek_at:                                      # a readback in the middle of a batch submits it and sees frame i
            peeks[i] = (m._target.GetPixels(), m._converged.GetPixels())
    conv, last = m._converged.GetPixels(), m._target.GetPixels()
    c = ctx.counters()
    m.OnDisable()
    ctx.set_option("frames_per_launch", 0)
    return conv, last, peeks, c


@pytest.mark.parametrize("scene_name", ["mixed", "multi_ray"])
def test_batched_frames_equal_single_launches_and_oracle(gpu_ctx, scene_name):
    sc = scenes.mixed_test_scene(200, 120)
    if scene_name == "multi_ray":
        sc.num_rays, sc.num_bounces = 3, 4                    # the _numRays > 1 instantiation: _Seed carries over between a pixel's rays
    n = 11                                                    # not a multiple of any batch size: the last launch is a partial batch
    ref_conv, ref_last, ref_peeks, c1 = run_frames(gpu_ctx, sc, n, 1, peek_at=(2, 6))
    assert c1["launches"] == n and c1["dispatches"] == n
    for fpl in (2, 4, 8, 16, 0):
        conv, last, peeks, c = run_frames(gpu_ctx, sc, n, fpl, peek_at=(2, 6))
        assert bits_equal(conv, ref_conv) and bits_equal(last, ref_last), fpl
        for i in ref_peeks:
            assert bits_equal(peeks[i][0], ref_peeks[i][0]) and bits_equal(peeks[i][1], ref_peeks[i][1]), (fpl, i)
        assert c["rays"] == c1["rays"] and c["dispatches"] == n and c["watchdog_trips"] == 0
        assert c["launches"] < n, (fpl, c["launches"])        # frames really shared launches
    # ... and the running mean equals the oracle's (frame uniforms 0..n-1, AS:9,39-41 blend)
    o = pyoracle.Oracle(sc)
    nodes, tri, root, _, _ = debug_build_blas(sc.mesh_objects, sc.vertices, sc.indices)
    o.set_blas(nodes, tri, root)
    acc = None
    for i in range(n):
        ox, oy, sd = scenes.frame_uniforms(i)
        o.set_frame((ox, oy), sd)
        img = o.render(mode=1, threads=8)
        acc = pyoracle.accumulate(img, acc if acc is not None else np.zeros_like(img), i)
    assert bits_equal(ref_conv, acc)


def test_long_batches_up_to_64_frames_per_launch(gpu_ctx):
    """The frame table of a launch lives in device memory (staged through pinned host slots, four launches deep): up to 64
    frames share one launch.  70 frames as 64 + 6 (auto), 40 + 30 and 7 x 10 — more launches than staging slots — equal 70
    single launches bit for bit."""
    sc = scenes.mixed_test_scene(96, 56)
    n = 70
    ref_conv, ref_last, _, c1 = run_frames(gpu_ctx, sc, n, 1, peek_at=())
    assert c1["launches"] == n
    for fpl, launches in ((0, 2), (64, 2), (40, 2), (10, 7)):
        conv, last, _, c = run_frames(gpu_ctx, sc, n, fpl, peek_at=())
        assert bits_equal(conv, ref_conv) and bits_equal(last, ref_last), fpl
        assert c["launches"] == launches and c["dispatches"] == n and c["rays"] == c1["rays"] and c["watchdog_trips"] == 0, (fpl, c)
    with pytest.raises(Exception):
        gpu_ctx.set_option("frames_per_launch", 65)


def test_frame_interleaving_is_invisible(gpu_ctx):
    """`frame_group` = frames of a launch whose tile runs are interleaved (default: all of them — the same tiles of consecutive
    frames are traced back to back while their BVH subtrees are hot).  1, a group size that does not divide the batch, and the
    default give the same pixels as single launches; the ragged last run of a frame (tiles not a multiple of the run length)
    and the ragged last group only produce empty slots."""
    sc = scenes.mixed_test_scene(200, 120)                     # 25 x 15 tiles: not a multiple of any run length
    n = 11
    ref_conv, ref_last, _, c1 = run_frames(gpu_ctx, sc, n, 1, peek_at=())
    try:
        for fg, run in ((1, 0), (2, 0), (5, 4), (64, 0), (64, 16), (3, 64)):
            gpu_ctx.set_option("frame_group", fg); gpu_ctx.set_option("xcd_run", run)
            conv, last, _, c = run_frames(gpu_ctx, sc, n, 8, peek_at=())
            assert bits_equal(conv, ref_conv) and bits_equal(last, ref_last), (fg, run)
            assert c["rays"] == c1["rays"] and c["launches"] == 2 and c["watchdog_trips"] == 0, (fg, run, c)
    finally:
        gpu_ctx.set_option("frame_group", 64); gpu_ctx.set_option("xcd_run", 0)


def test_batch_of_strips_equals_full_frames(gpu_ctx):
    """Multi-GPU strips (partial-coverage dispatches) are batched too: the pixels outside a rank's strips stay the zeros of
    the freshly created RenderTexture in every renamed slot."""
    sc = scenes.mixed_test_scene(160, 104)
    n, world = 6, 3
    ref_conv, _, _, _ = run_frames(gpu_ctx, sc, n, 1)
    gpu_ctx.set_option("frames_per_launch", 4)
    union = np.zeros_like(ref_conv)
    for rank in range(world):
        gpu_ctx.reset_counters()
        m = RayTraceMaster(gpu_ctx, sc, rank=rank, world_size=world)
        for _ in range(n):
            m.OnRenderImage()
        conv = m._converged.GetPixels()
        tgt = m._target.GetPixels()
        c = gpu_ctx.counters()
        assert c["launches"] < n
        m.OnDisable()
        rows = np.zeros(sc.height, bool)
        for g in range(rank, (sc.height + 7) // 8, world):
            rows[g * 8:(g + 1) * 8] = True
        assert (tgt[~rows] == 0).all()                        # nothing outside the rank's strips was ever written
        union[rows] = conv[rows]
    gpu_ctx.set_option("frames_per_launch", 0)
    assert bits_equal(union, ref_conv)


def test_observers_inside_a_batch(gpu_ctx):
    """Blit of the Result to a third image, SetPixels into the accumulator and a pointer hand-out in the middle of deferred
    frames: each sees / affects exactly the frame it follows in program order."""
    sc = scenes.mixed_test_scene(96, 64)

    def protocol(fpl):
        gpu_ctx.set_option("frames_per_launch", fpl)
        m = RayTraceMaster(gpu_ctx, sc)
        snap = RenderTexture(gpu_ctx, sc.width, sc.height)
        outs = []
        for i in range(7):
            m.OnRenderImage()
            if i == 2:
                Graphics.Blit(m._target, snap)                # copy of frame 2's Result
            if i == 4:
                m._converged.SetPixels(np.full((sc.height, sc.width, 4), 0.25, np.float32))   # clobber the running mean
            if i == 5:
                assert m._target.device_ptr() != 0            # raw pointer handed out: the image keeps its own storage from now on
        outs = [snap.GetPixels(), m._converged.GetPixels(), m._target.GetPixels()]
        snap.Release(); m.OnDisable()
        gpu_ctx.set_option("frames_per_launch", 0)
        return outs

    ref = protocol(1)
    for fpl in (3, 8):
        got = protocol(fpl)
        for a, b in zip(got, ref):
            assert bits_equal(a, b), fpl


def test_scene_change_inside_a_batch(gpu_ctx):
    """SetData between deferred frames: the frames dispatched before it are traced against the OLD scene."""
    sc = scenes.config1(64, 64)

    def protocol(fpl):
        gpu_ctx.set_option("frames_per_launch", fpl)
        m = RayTraceMaster(gpu_ctx, sc)
        for _ in range(3):
            m.OnRenderImage()
        moved = sc.spheres.copy()
        moved["position"][:, 1] += 0.5
        m._sphereBuffer.SetData(moved)                        # RM:250 — same buffer object, new contents
        for _ in range(3):
            m.OnRenderImage()
        out = m._converged.GetPixels()
        m.OnDisable()
        gpu_ctx.set_option("frames_per_launch", 0)
        return out

    assert bits_equal(protocol(8), protocol(1))


def test_checkpoint_resume_of_progressive_accumulation(gpu_ctx, tmp_path):
    """10 accumulated frames == 6 frames, checkpoint to disk, a NEW master resumed from it, 4 more frames (bit for bit)."""
    sc = scenes.mixed_test_scene(120, 72)
    ref_conv, _, _, _ = run_frames(gpu_ctx, sc, 10, 0)
    m = RayTraceMaster(gpu_ctx, sc)
    for _ in range(6):
        m.OnRenderImage()
    ck = str(tmp_path / "accum.npz")
    m.SaveCheckpoint(ck)
    m.OnDisable()
    m2 = RayTraceMaster(gpu_ctx, sc)
    m2.LoadCheckpoint(ck)
    assert m2._currentSample == 6
    for _ in range(4):
        m2.OnRenderImage()
    got = m2._converged.GetPixels()
    m2.OnDisable()
    assert bits_equal(got, ref_conv)


def test_present_blit_is_deferred_and_invisible(gpu_ctx):
    """The reference's literal frame — Dispatch, Blit(_target, _converged, mat), Blit(_converged, destination), RM:810-820 —
    keeps batching: the present is queued behind the deferred frames and fused into the blend pass.  `destination` read in the
    middle of a batch, after a batch boundary and at the end equals one launch per frame and the oracle's running mean; a second
    destination that is presented to only now and then (an earlier present elided under the as-if rule, a run that ends on a
    blend) sees exactly the frame it was last presented with."""
    sc = scenes.mixed_test_scene(136, 88)
    n = 13

    def protocol(fpl, peek_at):
        gpu_ctx.set_option("kernel_mode", 3)
        gpu_ctx.set_option("frames_per_launch", fpl)
        gpu_ctx.reset_counters()
        m = RayTraceMaster(gpu_ctx, sc)
        dest = RenderTexture(gpu_ctx, sc.width, sc.height)
        other = RenderTexture(gpu_ctx, sc.width, sc.height)
        peeks = {}
        for i in range(n):
            m.OnRenderImage(dest)
            if i in (3, 4, 9):
                Graphics.Blit(m._converged, other)             # a second present target, written three times
            if i in peek_at:
                peeks[i] = dest.GetPixels()
        out = (dest.GetPixels(), other.GetPixels(), m._converged.GetPixels(), peeks)
        c = gpu_ctx.counters()
        dest.Release(); other.Release(); m.OnDisable()
        gpu_ctx.set_option("frames_per_launch", 0)
        return out, c

    ref, c1 = protocol(1, (2, 7))
    assert c1["launches"] == n
    assert bits_equal(ref[0], ref[2])                          # destination == _converged after the last frame
    for fpl in (4, 5, 16, 0):
        got, c = protocol(fpl, (2, 7))
        assert bits_equal(got[0], ref[0]) and bits_equal(got[1], ref[1]) and bits_equal(got[2], ref[2]), fpl
        for i in ref[3]:
            assert bits_equal(got[3][i], ref[3][i]), (fpl, i)
        assert c["launches"] < n and c["rays"] == c1["rays"] and c["watchdog_trips"] == 0, (fpl, c)
    # without observers the 13 presented frames share launches as if nothing were presented (64 per launch by default)
    got, c = protocol(0, ())
    assert c["launches"] == 1 and bits_equal(got[0], ref[0]) and bits_equal(got[1], ref[1])
    # the oracle's running mean after frame 9 is what `other` shows, after frame 12 what `destination` shows
    o = pyoracle.Oracle(sc)
    nodes, tri, root, _, _ = debug_build_blas(sc.mesh_objects, sc.vertices, sc.indices)
    o.set_blas(nodes, tri, root)
    acc = None
    for i in range(n):
        ox, oy, sd = scenes.frame_uniforms(i)
        o.set_frame((ox, oy), sd)
        img = o.render(mode=1, threads=8)
        acc = pyoracle.accumulate(img, acc if acc is not None else np.zeros_like(img), i)
        if i == 9:
            assert bits_equal(ref[1], acc)
    assert bits_equal(ref[0], acc)


def test_present_of_the_result_and_into_odd_targets(gpu_ctx):
    """Deferred plain blits that are NOT the present pattern: a copy of the Result itself (a slab slot), a copy between two
    unrelated images, a copy whose destination is the sky (must not be deferred: the deferred launch reads the sky)."""
    sc = scenes.mixed_test_scene(96, 64)

    def protocol(fpl):
        gpu_ctx.set_option("frames_per_launch", fpl)
        m = RayTraceMaster(gpu_ctx, sc)
        a = RenderTexture(gpu_ctx, sc.width, sc.height)
        b = RenderTexture(gpu_ctx, sc.width, sc.height)
        half = RenderTexture(gpu_ctx, sc.sky.shape[1], sc.sky.shape[0])
        half.SetPixels(sc.sky * np.float32(0.5))
        for i in range(6):
            m.OnRenderImage()
            if i == 1:
                Graphics.Blit(m._target, a)                    # frame 1's Result
                Graphics.Blit(a, b)                            # chained: reads what the deferred copy above wrote
            if i == 3:
                Graphics.Blit(half, m.SkyboxTexture)           # frames 4, 5 see the darker sky, frames 0..3 the old one
        out = (a.GetPixels(), b.GetPixels(), m._converged.GetPixels())
        a.Release(); b.Release(); half.Release(); m.OnDisable()
        gpu_ctx.set_option("frames_per_launch", 0)
        return out

    ref = protocol(1)
    assert bits_equal(ref[0], ref[1])
    for fpl in (4, 0):
        got = protocol(fpl)
        for x, y in zip(got, ref):
            assert bits_equal(x, y), fpl


def test_watchdog_trip_is_an_error_not_a_silent_hole(gpu_ctx):
    """A wave that leaves the persistent kernel through its iteration cap has not written its pixels.  With the cap forced
    absurdly low ("watchdog_cap", a test hook) the launch ends early: the next synchronising call must fail with
    URT_ERR_WATCHDOG, the counter must show the trips, and the context must be usable afterwards."""
    from unityraytracer_amd import UrtError
    sc = scenes.mixed_test_scene(160, 96)
    gpu_ctx.set_option("kernel_mode", 3)
    gpu_ctx.reset_counters()
    m = RayTraceMaster(gpu_ctx, sc)
    m.OnRenderImage()
    good = m._target.GetPixels()
    try:
        gpu_ctx.set_option("watchdog_cap", 3)
        m.OnRenderImage(); m.OnRenderImage()                   # deferred: nothing has run yet
        with pytest.raises(UrtError) as ei:
            m._target.GetPixels()                              # submits, waits, and must refuse the incomplete image
        assert ei.value.code == 9 and "cap" in str(ei.value)
        assert gpu_ctx.counters()["watchdog_trips"] > 0
        gpu_ctx.synchronize()                                  # reported once
    finally:
        gpu_ctx.set_option("watchdog_cap", 0)
    gpu_ctx.reset_counters()
    m2 = RayTraceMaster(gpu_ctx, sc)
    m2.OnRenderImage()
    again = m2._target.GetPixels()
    assert gpu_ctx.counters()["watchdog_trips"] == 0
    m.OnDisable(); m2.OnDisable()
    assert bits_equal(again, good)
    # the automatic cap grows with the launch: 64 frames x 25 rays x 10 bounces must not trip (SampleScene's settings)
    sc2 = scenes.mixed_test_scene(48, 32)
    sc2.num_rays, sc2.num_bounces = 25, 10
    m3 = RayTraceMaster(gpu_ctx, sc2)
    for _ in range(8):
        m3.OnRenderImage()
    gpu_ctx.synchronize()
    assert gpu_ctx.counters()["watchdog_trips"] == 0
    m3.OnDisable()
