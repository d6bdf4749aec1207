"""GPU: urt_group_* — one host thread drives N ranks through the C ABI (include/urt.h "device groups").  The 1-GPU box
hosts the ranks on one card (the device list repeats ordinal 0); strip dispatch, local accumulation, pack kernels, the peer
copies to rank 0 and the de-interleave all run for real.  Whatever rank 0 holds after a gather must be bit-identical to
the same frames rendered by a single context."""
import numpy as np
import pytest

from unityraytracer_amd import DeviceGroup, RayTraceMaster, RenderTexture, UrtError, scenes

pytestmark = pytest.mark.gpu


def bits_equal(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


def single_context_frames(ctx, sc, n):
    ctx.set_option("kernel_mode", 3)
    m = RayTraceMaster(ctx, sc)
    out = []
    for _ in range(n):
        m.OnRenderImage()
        out.append(m._converged.GetPixels())
    m.OnDisable()
    return out


@pytest.mark.parametrize("ranks,fpl", [(2, 0), (3, 0), (4, 1), (3, 5)])
def test_group_gather_equals_single_context(gpu_ctx, ranks, fpl):
    sc = scenes.mixed_test_scene(176, 120)                 # 15 group rows: ragged over 2, 3 and 4 ranks
    n = 7
    ref = single_context_frames(gpu_ctx, sc, n)
    with DeviceGroup([0] * ranks) as g:
        assert g.size == ranks
        g.set_option("frames_per_launch", fpl)
        m = RayTraceMaster(g, sc)                          # the reference's driver, unchanged, on a group
        full = RenderTexture(g, sc.width, sc.height)
        peeks = {}
        for i in range(n):
            m.OnRenderImage()
            g.gather(m._converged, full)                   # ONE exchange per frame
            if i in (2, n - 1):
                peeks[i] = full.GetPixels()                # rank 0's full image after frame i
        c = g.counters()
        assert c["dispatches"] == n and c["watchdog_trips"] == 0
        if fpl != 1:
            assert c["launches"] < n                       # the ranks really batched frames under a gather-every-frame protocol
        for i, img in peeks.items():
            assert bits_equal(img, ref[i]), (ranks, fpl, i)
        full.Release()
        m.OnDisable()


def test_group_counts_rays_once_and_rejects_foreign_objects(gpu_ctx):
    sc = scenes.config1(96, 64)
    gpu_ctx.reset_counters()
    single_context_frames(gpu_ctx, sc, 2)
    want = gpu_ctx.counters()["rays"]
    with DeviceGroup([0, 0]) as g:
        m = RayTraceMaster(g, sc)
        for _ in range(2):
            m.OnRenderImage()
        assert g.counters()["rays"] == want                # every pixel is traced by exactly one rank
        m.OnDisable()
        # an object created on ONE rank's context behind the group's back breaks the shared handle numbering: reported, not ignored
        import ctypes as C
        raw = g._raw
        h = C.c_uint64()
        assert raw.urt_buffer_create(C.c_void_p(raw.urt_group_context(g._h, 1)), 4, 12, C.byref(h)) == 0
        with pytest.raises(UrtError) as e:
            RenderTexture(g, 8, 8)
        assert e.value.code == 2


@pytest.mark.parametrize("fpl", [0, 1])
def test_group_present_after_gather_is_ordered(gpu_ctx, fpl):
    """gather(_converged -> full) followed by the present Blit(full, destination) (RM:819), a larger image gathered while
    gathers are still queued (the staging buffers grow), SetPixels into a gather target: every call sees the gathers before it
    in program order.  (Round 2 forwarded these calls past the queued gathers: the present showed a stale image.)"""
    from unityraytracer_amd import Graphics
    sc = scenes.mixed_test_scene(176, 120)
    n = 6
    ref = single_context_frames(gpu_ctx, sc, n)
    with DeviceGroup([0, 0, 0]) as g:
        g.set_option("frames_per_launch", fpl)
        m = RayTraceMaster(g, sc)
        full = RenderTexture(g, sc.width, sc.height)
        dest = RenderTexture(g, sc.width, sc.height)
        peeks = {}
        for i in range(n):
            m.OnRenderImage()
            g.gather(m._converged, full)
            Graphics.Blit(full, dest)                      # the present of the gathered frame
            if i == 3:
                peeks[i] = dest.GetPixels()
        assert bits_equal(peeks[3], ref[3]) and bits_equal(dest.GetPixels(), ref[n - 1]), fpl
        if fpl != 1:
            assert g.counters()["launches"] < n            # the presents did not break the ranks' batching
        # a second, larger scene on the same group while gathers of the first are queued: staging is re-allocated safely
        sc2 = scenes.mixed_test_scene(264, 168)
        ref2 = single_context_frames(gpu_ctx, sc2, 2)
        m.OnRenderImage()
        g.gather(m._converged, full)                       # queued (frames are deferred when fpl != 1)
        m2 = RayTraceMaster(g, sc2)
        full2 = RenderTexture(g, sc2.width, sc2.height)
        for _ in range(2):
            m2.OnRenderImage()
            g.gather(m2._converged, full2)
        assert bits_equal(full2.GetPixels(), ref2[1])
        # SetPixels into a gather target after a queued gather wins over the gather
        m2.OnRenderImage()
        g.gather(m2._converged, full2)
        flat = np.full((sc2.height, sc2.width, 4), 0.5, np.float32)
        full2.SetPixels(flat)
        assert bits_equal(full2.GetPixels(), flat)
        full.Release(); dest.Release(); full2.Release()
        m.OnDisable(); m2.OnDisable()


@pytest.mark.parametrize("fpl", [0, 1])
def test_group_blit_of_a_gather_source_keeps_program_order(gpu_ctx, fpl):
    """gather(_converged -> full), then Blit(_converged, snap), then two more frames: `snap` must hold the mean after the frame the
    blit followed — a blit that reads a gather's SOURCE is ordered by the ranks' own deferred operations and must not be queued
    behind the late unpack, where it would run after frames dispatched later (group.cpp urt_group_blit)."""
    from unityraytracer_amd import Graphics
    sc = scenes.mixed_test_scene(176, 120)
    n = 5
    ref = single_context_frames(gpu_ctx, sc, n)
    with DeviceGroup([0, 0, 0]) as g:
        g.set_option("frames_per_launch", fpl)
        m = RayTraceMaster(g, sc)
        full = RenderTexture(g, sc.width, sc.height)
        snap = RenderTexture(g, sc.width, sc.height)
        for i in range(n):
            m.OnRenderImage()
            g.gather(m._converged, full)
            if i == 2:
                Graphics.Blit(m._converged, snap)          # every rank copies ITS strips of the mean after frame 2
        g.gather(snap, full)                               # bring the snapshot's strips together
        assert bits_equal(full.GetPixels(), ref[2]), fpl
        full.Release(); snap.Release()
        m.OnDisable()
