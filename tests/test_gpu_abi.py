"""GPU: behaviour of the C ABI at the boundary (include/urt.h) — error codes, Unity-like leniency, edge cases of the
dispatch (empty scene, zero loops, partial dispatch, buffer re-upload), all through the Python mirror of the reference's calls."""
import numpy as np
import pytest

from oracle import pyoracle
from unityraytracer_amd import ComputeBuffer, ComputeShader, Graphics, Material, RayTraceMaster, RenderTexture, UrtError, scenes

pytestmark = pytest.mark.gpu


def bits_equal(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_error_codes_and_unity_leniency(gpu_ctx):
    sh = ComputeShader(gpu_ctx)
    b = ComputeBuffer(gpu_ctx, 4, 12)
    with pytest.raises(UrtError) as e:
        sh.SetBuffer(0, "_Spheres", b)                       # stride 12 != 56 (RM:744)
    assert e.value.code == 6
    with pytest.raises(UrtError) as e:
        sh.SetBuffer(0, "_NoSuchBuffer", b)
    assert e.value.code == 1
    with pytest.raises(UrtError):
        ComputeBuffer(gpu_ctx, 0, 12)                        # Unity: count must be > 0
    with pytest.raises(UrtError):
        ComputeBuffer(gpu_ctx, 4, 10)                        # stride must be a multiple of 4
    with pytest.raises(UrtError) as e:
        b.SetData(np.zeros(5 * 3, np.float32))               # more elements than the buffer holds
    assert e.value.code == 1
    sh.SetInt("_MeshBVH_len", 123)                           # static const in the shader: accepted, ignored (RS:73-74)
    sh.SetInt("_SphereBVH_len", 7)
    sh.SetFloat("_NotAUniform", 1.0)                         # Unity ignores undeclared names
    sh.SetVector("_AlsoNot", (1, 2, 3, 4))
    b.Release()
    with pytest.raises(UrtError) as e:
        gpu_ctx.check(gpu_ctx.lib.urt_buffer_release(gpu_ctx._h, 987654321))
    assert e.value.code == 2
    sh.SetTexture(0, "Result", None)
    with pytest.raises(UrtError) as e:
        sh.Dispatch(0, 1, 1, 1)                              # no Result bound
    assert e.value.code == 5
    with pytest.raises(UrtError):
        sh.Dispatch(1, 1, 1, 1)                              # only kernel 0 exists
    a, c = RenderTexture(gpu_ctx, 8, 8), RenderTexture(gpu_ctx, 16, 8)
    with pytest.raises(UrtError):
        Graphics.Blit(a, c)                                  # size mismatch
    a.Release(); c.Release()


def test_empty_scene_and_zero_loops(gpu_ctx):
    """No buffers bound at all: ground plane + sky only (RS:375-379 treat missing buffers as count 0)."""
    sc = scenes.Scene("empty", 100, 60, 3, 2, sky=scenes.make_sky(64, 32))
    ref = pyoracle.Oracle(sc).render(threads=4)
    for mode in (0, 1, 2, 3, 4, 5):
        gpu_ctx.set_option("kernel_mode", mode)
        m = RayTraceMaster(gpu_ctx, sc)
        m.OnRenderImage()
        assert bits_equal(m._target.GetPixels(), ref), mode
        m.OnDisable()
    # numBounces = 0: the bounce loop never runs -> (0,0,0,1); numRays = 0: 0/0 = NaN in rgb, alpha 1 (RS:444-468 literally)
    sc0 = scenes.Scene("zero-bounces", 40, 24, 0, 1, sky=scenes.make_sky(64, 32))
    m = RayTraceMaster(gpu_ctx, sc0); m.OnRenderImage(); img = m._target.GetPixels(); m.OnDisable()
    assert bits_equal(img, pyoracle.Oracle(sc0).render()) and (img[..., :3] == 0).all() and (img[..., 3] == 1).all()
    sc1 = scenes.Scene("zero-rays", 40, 24, 2, 0, sky=scenes.make_sky(64, 32))
    m = RayTraceMaster(gpu_ctx, sc1); m.OnRenderImage(); img = m._target.GetPixels(); m.OnDisable()
    assert np.isnan(img[..., :3]).all() and (img[..., 3] == 1).all()


def test_partial_dispatch_writes_only_its_groups(gpu_ctx):
    sc = scenes.mixed_test_scene(96, 64)
    full = pyoracle.Oracle(sc)
    full.build_own_blas()
    ref = full.render(mode=1, threads=8)
    m = RayTraceMaster(gpu_ctx, sc)
    m.RebuildTrees(); m._treesNeedRebuilding = False
    m.SetShaderParameters()
    m.InitRenderTexture()
    m.RayTraceShader.SetTexture(0, "Result", m._target)
    m.RayTraceShader.Dispatch(0, 5, 3, 1)                    # 40 x 24 pixels of the 96 x 64 frame
    img = m._target.GetPixels()
    assert bits_equal(img[:24, :40], ref[:24, :40])
    assert not img[24:].any() and not img[:, 40:].any()      # threads outside the dispatched groups wrote nothing
    m.RayTraceShader.Dispatch(0, 1000, 1000, 1)              # more groups than pixels: clipped to the texture (RS:468)
    assert bits_equal(m._target.GetPixels(), ref)
    m.OnDisable()


def test_setdata_after_bind_is_picked_up(gpu_ctx):
    """SetData on a bound buffer (RM:250 runs every rebuild) must invalidate the derived device scene."""
    sc = scenes.config1(96, 96, sky=scenes.make_sky(64, 32))
    m = RayTraceMaster(gpu_ctx, sc)
    m.OnRenderImage()
    before = m._target.GetPixels()
    moved = sc.spheres.copy()
    moved["position"][:, 1] += 1.5
    sc.spheres = moved
    sc.sphere_bvh = scenes.build_object_bvh(*scenes.sphere_bounds(moved))
    m._sphereBuffer.SetData(moved)                           # same ComputeBuffer objects, new contents
    m._sphereBVHBuffer.SetData(sc.sphere_bvh)
    m._frame = 0; m._currentSample = 0
    m.OnRenderImage()
    after = m._target.GetPixels()
    assert not bits_equal(before, after)
    assert bits_equal(after, pyoracle.Oracle(sc).render(threads=4))
    m.OnDisable()


def test_moving_one_mesh_rebuilds_only_its_bvh(gpu_ctx):
    """The reference re-uploads every buffer when one object moves (RM:262-336).  The library then rebuilds the triangle BVH
    of the MeshObjects that changed only — and the frame is the same as if everything had been rebuilt."""
    from unityraytracer_amd import debug_build_blas
    sc = scenes.mixed_test_scene(96, 64)
    gpu_ctx.set_option("refit", 0)                            # (the default refits the moved MeshObject on the GPU: tests/test_gpu_refit.py)
    m = RayTraceMaster(gpu_ctx, sc)
    m.OnRenderImage()
    reused0, built0 = gpu_ctx.blas_cache_stats()
    n = len(sc.mesh_objects)
    mo = sc.mesh_objects.copy()
    mat = np.asarray(mo[1]["localToWorldMatrix"], np.float32).copy()
    mat[12] += 0.75; mat[13] += 0.25                          # column-major translation of MeshObject 1
    mo[1]["localToWorldMatrix"] = mat
    sc.mesh_objects = mo
    lo, hi = scenes.mesh_bounds(mo, sc.vertices, sc.indices)
    sc.mesh_bvh = scenes.build_object_bvh(lo, hi)
    for buf, data in ((m._meshObjectBuffer, mo), (m._vertexBuffer, sc.vertices), (m._indexBuffer, sc.indices),
                      (m._normalBuffer, sc.normals), (m._meshObjectBVHBuffer, sc.mesh_bvh)):
        buf.SetData(data)                                    # everything re-uploaded, as RebuildTrees does
    m._frame = 0; m._currentSample = 0
    m.OnRenderImage()
    reused1, built1 = gpu_ctx.blas_cache_stats()
    assert built1 - built0 == 1 and reused1 - reused0 == n - 1, (reused0, built0, reused1, built1)
    o = pyoracle.Oracle(sc)
    nodes, tri, root, _, _ = debug_build_blas(sc.mesh_objects, sc.vertices, sc.indices)   # a from-scratch build of the moved scene
    o.set_blas(nodes, tri, root)
    assert bits_equal(m._target.GetPixels(), o.render(mode=1, threads=4))
    m.OnDisable()
    gpu_ctx.set_option("refit", 1)


def test_resize_resets_accumulation_and_external_texture(gpu_ctx):
    sc = scenes.mixed_test_scene(64, 40)
    m = RayTraceMaster(gpu_ctx, sc)
    m.OnRenderImage(); m.OnRenderImage()
    assert m._currentSample == 2
    m.screen_width, m.screen_height = 48, 32                 # Screen resized: RM:826-845 recreate the targets, reset the sample
    m.scene = sc.resized(48, 32)
    m.OnRenderImage()
    assert m._currentSample == 1 and m._target.width == 48
    m.OnDisable()
    # a RenderTexture over caller-owned device memory (what the multi-GPU gather uses): here the "caller" memory is another
    # texture's storage (torch is deliberately not imported here: two HIP runtimes initialised in the wrong order in one
    # process do not share the GPU; bench.py imports torch first)
    owner = RenderTexture(gpu_ctx, 64, 40)
    ext = RenderTexture(gpu_ctx, 64, 40, external_ptr=owner.device_ptr())
    m2 = RayTraceMaster(gpu_ctx, sc)
    m2.OnRenderImage()
    Graphics.Blit(m2._target, ext)
    assert bits_equal(owner.GetPixels(), m2._target.GetPixels())
    ext.Release()                                            # releasing the alias does not free caller-owned memory
    assert bits_equal(owner.GetPixels(), m2._target.GetPixels())
    owner.Release()
    m2.OnDisable()


def test_counters_and_watchdog(gpu_ctx):
    sc = scenes.mixed_test_scene(64, 40)
    gpu_ctx.set_option("kernel_mode", 3); gpu_ctx.set_option("count_stats", 1); gpu_ctx.set_option("time_dispatch", 1)
    gpu_ctx.reset_counters()
    m = RayTraceMaster(gpu_ctx, sc)
    for _ in range(3):
        m.OnRenderImage()
    c = gpu_ctx.counters()
    assert c["dispatches"] == 3 and c["pixels"] == 3 * 64 * 40 and c["rays"] >= c["pixels"] and c["trace_ms"] > 0
    assert c["hit_sky"] + c["hit_ground"] + c["hit_sphere"] + c["hit_tri"] == c["rays"] and c["watchdog_trips"] == 0
    gpu_ctx.reset_counters()
    assert gpu_ctx.counters()["rays"] == 0
    gpu_ctx.set_option("count_stats", 0); gpu_ctx.set_option("time_dispatch", 0)
    with pytest.raises(UrtError):
        gpu_ctx.set_option("kernel_mode", 9)
    with pytest.raises(UrtError):
        gpu_ctx.set_option("no_such_option", 1)
    m.OnDisable()


def test_two_hosts_interleaved_on_one_context(gpu_ctx):
    """Two RayTraceMasters alive on ONE context, rendering alternately: every uniform and binding really changes hands each
    frame (the Python mirror skips calls that would set a name to the value the context already holds — it must not skip
    these), and the library's frame batching has to break its batches at every switch.  Each master's running mean equals
    what it produces alone."""
    sa = scenes.mixed_test_scene(96, 64)
    sb = scenes.mixed_test_scene(64, 40)                       # (both bind all seven buffers: a null buffer is never bound, RM:252-259,
    sb.num_rays, sb.num_bounces = 2, 3                         #  so a scene without meshes would inherit the other one's)
    sb.spheres = sb.spheres.copy()
    sb.spheres["position"][:, 0] += 0.7                        # other spheres, other heap, other resolution, other loop counts
    sb.sphere_bvh = scenes.build_object_bvh(*scenes.sphere_bounds(sb.spheres))
    def alone(sc, n):
        m = RayTraceMaster(gpu_ctx, sc)
        for _ in range(n):
            m.OnRenderImage()
        img = m._converged.GetPixels()
        m.OnDisable()
        return img
    want_a, want_b = alone(sa, 5), alone(sb, 5)
    ma, mb = RayTraceMaster(gpu_ctx, sa), RayTraceMaster(gpu_ctx, sb)
    for _ in range(5):
        ma.OnRenderImage()
        mb.OnRenderImage()
    got_a, got_b = ma._converged.GetPixels(), mb._converged.GetPixels()
    ma.OnDisable(); mb.OnDisable()
    assert bits_equal(got_a, want_a) and bits_equal(got_b, want_b)


def test_rgb_strip_pack_and_unpack(gpu_ctx):
    """urt_texture_pack_rows_rgb / urt_texture_unpack_rows_rgb: three channels per pixel travel, the de-interleave writes the alpha it is
    given; ragged last strip; the packed size is 12 B per pixel of the rank's padded strips.  (The dense buffer is the device memory of a
    third image: no torch in this process — its bundled HIP runtime does not share a process with a library that initialised the GPU first.)"""
    import ctypes as C
    from unityraytracer_amd import RenderTexture, strips
    w, h, world = 40, 52, 3                                   # 7 group rows, the last with 4 pixel rows
    rng = np.random.default_rng(3)
    img = rng.random((h, w, 4), dtype=np.float32)
    src = RenderTexture(gpu_ctx, w, h)
    src.SetPixels(img)
    dst = RenderTexture(gpu_ctx, w, h)
    dst.SetPixels(np.zeros_like(img))
    n_floats = strips.packed_rows(h, world) * w * 3
    dense = RenderTexture(gpu_ctx, (n_floats + 3) // 4, 1)
    for rank in range(world):
        nb = C.c_uint64()
        gpu_ctx.check(gpu_ctx.lib.urt_texture_pack_rows_rgb(gpu_ctx._h, src.handle, rank, world, None, C.byref(nb)))
        assert nb.value == strips.n_strips(h, rank, world) * 8 * w * 12
        dense.SetPixels(np.full((1, dense.width, 4), -1.0, np.float32))
        src.pack_rows(rank, world, dense.device_ptr(), rgb=True)
        dst.unpack_rows_rgb(rank, world, dense.device_ptr(), 0.25)
        host = dense.GetPixels().reshape(-1)[:n_floats].reshape(-1, w, 3)
        for j, (y0, y1) in enumerate(strips.strip_row_ranges(h, rank, world)):
            assert np.array_equal(host[8 * j: 8 * j + (y1 - y0)], img[y0:y1, :, :3])
            assert (host[8 * j + (y1 - y0): 8 * j + 8] == 0).all()          # rows beyond the image are zero-filled
    got = dst.GetPixels()
    assert np.array_equal(got[..., :3], img[..., :3]) and (got[..., 3] == 0.25).all()
    src.Release(); dst.Release(); dense.Release()


def test_launch_info_names_the_kernel_that_ran(gpu_ctx):
    from unityraytracer_amd import RayTraceMaster, scenes
    sc = scenes.mixed_test_scene(96, 64)
    m = RayTraceMaster(gpu_ctx, sc)
    try:
        for mode, name in ((0, "k_mega<false>"), (2, "k_persist<false>"), (3, "k_sched<false, 256, ")):
            gpu_ctx.set_option("kernel_mode", mode)
            m.OnRenderImage()
            info = gpu_ctx.launch_info()
            assert info["kernel"].startswith(name) and info["kernel_mode"] == mode and info["experiment"] == 0, info
            assert info["n_blocks"] > 0 and info["lds_bytes"] > 0
        gpu_ctx.set_option("count_stats", 1)
        m.OnRenderImage()
        assert gpu_ctx.launch_info()["kernel"].startswith("k_sched<true, 256, ")
    finally:
        gpu_ctx.set_option("count_stats", 0)
        gpu_ctx.set_option("kernel_mode", 3)
        m.OnDisable()


def test_pipelined_readback_sees_the_frame_it_was_asked_for(gpu_ctx):
    """urt_texture_read_begin / _end: every ticket delivers the image as it was when ITS begin was called, while later frames — deferred
    and batched or not — render; three may be in flight, a fourth is refused, an ended ticket cannot be ended twice."""
    import ctypes as C
    from unityraytracer_amd import RayTraceMaster, UrtError, scenes
    sc = scenes.mixed_test_scene(168, 96)
    for fpl in (1, 0):
        gpu_ctx.set_option("kernel_mode", 3)
        gpu_ctx.set_option("frames_per_launch", fpl)
        try:
            m = RayTraceMaster(gpu_ctx, sc)
            want = []
            for _ in range(7):
                m.OnRenderImage()
                want.append(m._converged.GetPixels())
            m.OnDisable()
            m = RayTraceMaster(gpu_ctx, sc)
            tickets, got = [], []
            for i in range(7):
                m.OnRenderImage()
                tickets.append(m._converged.ReadBegin())
                if len(tickets) == 3:
                    with pytest.raises(UrtError):
                        m._converged.ReadBegin()              # three in flight: the oldest must be ended first
                    got.append(m._converged.ReadEnd(tickets.pop(0)))
            while tickets:
                got.append(m._converged.ReadEnd(tickets.pop(0)))
            with pytest.raises(UrtError):
                m._converged.ReadEnd(1)                        # long ended
            m.OnDisable()
            assert len(got) == 7
            for i in range(7):
                assert np.array_equal(got[i].view(np.uint32), want[i].view(np.uint32)), (fpl, i)
        finally:
            gpu_ctx.set_option("frames_per_launch", 0)
