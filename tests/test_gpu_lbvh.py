"""GPU: the triangle BVH built ON the GPU (csrc/lbvh.hip, urt_set_option("blas_builder", 1): Morton sort + Karras hierarchy +
bottom-up fit) is structurally valid, keeps the top-of-forest numbering the trace kernel relies on, gives bit-identical frames
to the host-built SAH tree and to the oracle, reports bad scene data like the host builder, and prepares the 983,040-triangle
C5 scene in milliseconds (SURVEY.md §8f row f2 "GPU BLAS build (LBVH) for dynamic scenes")."""
import numpy as np
import pytest

from oracle import pyoracle
from test_blas import validate
from unityraytracer_amd import RayTraceMaster, UrtError, debug_build_blas, scenes

pytestmark = pytest.mark.gpu


def bits_equal(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.fixture(params=[1, 2, 3], ids=["karras", "budget", "sah"])
def lbvh(gpu_ctx, request):
    """Both GPU builders: 1 = the Morton radix tree as built, 2 = the same radix tree built top-down within a depth budget (csrc/lbvh.hip k_td_level)."""
    gpu_ctx.set_option("kernel_mode", 3)
    gpu_ctx.set_option("blas_builder", request.param)
    yield gpu_ctx
    gpu_ctx.set_option("blas_builder", -1)


def render(ctx, sc, frames=1):
    m = RayTraceMaster(ctx, sc)
    for _ in range(frames):
        m.OnRenderImage()
    return m, m._target.GetPixels(), m._converged.GetPixels()


@pytest.mark.parametrize("scene_fn", [lambda: scenes.mixed_test_scene(96, 64, blob=(40, 31)), lambda: scenes.many_meshes_scene(n=70, level=1),
                                      lambda: scenes.config3(128, 72, slices=60, stacks=41, sky=scenes.make_sky(64, 32))])
def test_gpu_built_tree_is_valid_and_top_is_breadth_first(lbvh, scene_fn):
    sc = scene_fn()
    m, _, _ = render(lbvh, sc)
    nodes, tri, root, info = lbvh.read_scene_blas(len(sc.mesh_objects))
    m.OnDisable()
    assert len(tri) == sc.n_triangles and sorted(tri.tolist()) == list(range(0, 3 * sc.n_triangles, 3))     # every index slot exactly once
    assert validate(sc, nodes, tri, root) <= info["max_depth"]                # boxes nest, triangles inside their leaf boxes, depth bound holds
    # interior roots are nodes 0..k-1 in MeshObject order and a breadth-first walk meets 0, 1, 2, ... (LDS top of the forest)
    roots = [int(r) for r in root if 0 <= r != 0x7FFFFFFF]
    assert roots == list(range(len(roots)))
    queue, order = list(roots), []
    while queue and len(order) < 256:
        n = queue.pop(0)
        order.append(n)
        queue += [int(c) for c in nodes[n][12:14].view(np.int32) if c >= 0]
    assert order == list(range(len(order)))
    seen = set()
    for n in nodes:
        for c in n[12:14].view(np.int32):
            if c >= 0:
                assert int(c) not in seen and int(c) < len(nodes)
                seen.add(int(c))
    assert len(seen) == len(nodes) - len(roots)                               # every non-root node has exactly one parent


@pytest.mark.parametrize("builder", [1, 2, 3])
@pytest.mark.parametrize("scene_fn", [lambda: scenes.mixed_test_scene(160, 96), lambda: scenes.many_meshes_scene(128, 80, n=120, level=0)])
def test_frames_equal_host_built_tree_and_oracle(gpu_ctx, scene_fn, builder):
    sc = scene_fn()
    gpu_ctx.set_option("kernel_mode", 3)
    gpu_ctx.set_option("blas_builder", 0)
    m, t0, c0 = render(gpu_ctx, sc, frames=3)
    m.OnDisable()
    gpu_ctx.set_option("blas_builder", builder)
    try:
        m, t1, c1 = render(gpu_ctx, sc, frames=3)
        m.OnDisable()
        # the other kernel modes walk the same device tree
        for mode in (0, 2):
            gpu_ctx.set_option("kernel_mode", mode)
            m, t2, _ = render(gpu_ctx, sc, frames=3)
            m.OnDisable()
            assert bits_equal(t2, t0), mode
    finally:
        gpu_ctx.set_option("kernel_mode", 3)
        gpu_ctx.set_option("blas_builder", -1)
    assert bits_equal(t1, t0) and bits_equal(c1, c0)
    o = pyoracle.Oracle(sc)
    o.build_own_blas()
    ox, oy, sd = scenes.frame_uniforms(2)
    o.set_frame((ox, oy), sd)
    assert bits_equal(t1, o.render(mode=1, threads=8))


def test_dynamic_scene_rebuilds_on_the_gpu(lbvh):
    """RM:215-230 protocol: an object moves -> every buffer is re-uploaded -> the BVH is rebuilt; frames follow the scene."""
    sc = scenes.mixed_test_scene(128, 80)
    m, before, _ = render(lbvh, sc)
    moved = sc.mesh_objects.copy()
    mat = np.array(moved[0]["localToWorldMatrix"]).reshape(4, 4).T.copy()
    mat[:3, 3] += (0.4, 0.3, -0.2)
    moved[0]["localToWorldMatrix"] = mat.T.reshape(16)
    m._meshObjectBuffer.SetData(moved)
    sc2 = scenes.mixed_test_scene(128, 80)
    sc2.mesh_objects = moved
    lo, hi = scenes.mesh_bounds(moved, sc2.vertices, sc2.indices)
    sc2.mesh_bvh = scenes.build_object_bvh(lo, hi)
    m._meshObjectBVHBuffer.SetData(sc2.mesh_bvh)
    m._frame = 0; m._currentSample = 0
    m.OnRenderImage()
    after = m._target.GetPixels()
    m.OnDisable()
    o = pyoracle.Oracle(sc2)
    o.build_own_blas()
    assert not bits_equal(after, before) and bits_equal(after, o.render(mode=1, threads=8))


def test_bad_index_is_reported_like_the_host_builder(lbvh):
    sc = scenes.mixed_test_scene(32, 24)
    bad = sc.indices.copy()
    bad[7] = len(sc.vertices) + 5
    sc.indices = bad
    m = RayTraceMaster(lbvh, sc)
    with pytest.raises(UrtError) as e:
        m.OnRenderImage()
    assert e.value.code == 8 and "_Indices[7]" in str(e.value)
    m.OnDisable()


@pytest.mark.parametrize("builder,limit_ms", [(1, 10.0), (2, 24.0), (3, 24.0)])
def test_c5_scene_preparation_under_10_ms(gpu_ctx, builder, limit_ms):
    """983,040 triangles in 12 MeshObjects: upload of the raw buffers + the whole GPU build, host wall clock (builder 2 — one launch
    per tree level — may take twice the 12 ms the Karras tree needs to its first frame: VERDICT round 3, item 6)."""
    sc = scenes.CONFIGS["C5"](640, 360)
    gpu_ctx.set_option("kernel_mode", 3)
    gpu_ctx.set_option("blas_builder", builder)
    try:
        m = RayTraceMaster(gpu_ctx, sc)
        m.OnRenderImage()                                        # first preparation also loads the build kernels
        times = []
        for k in range(3):
            v = np.ascontiguousarray(sc.vertices, np.float32).copy()
            v[0, 0] = np.nextafter(v[0, 0], np.float32(np.inf if k % 2 == 0 else -np.inf))     # new CONTENTS make the scene stale (equal data would not)
            m._vertexBuffer.SetData(v)
            times.append(gpu_ctx.scene_info()["prepare_ms"])
        m._vertexBuffer.SetData(np.ascontiguousarray(sc.vertices, np.float32))
        info = gpu_ctx.scene_info()
        img = m._target.GetPixels()
        m.OnDisable()
    finally:
        gpu_ctx.set_option("blas_builder", -1)
    print(f"C5 GPU scene preparation (builder {builder}): {times} ms, {info}")
    assert info["n_tris"] == 983040 and min(times) <= limit_ms, times
    gpu_ctx.set_option("blas_builder", 0)                        # host SAH tree: same pixels
    try:
        m = RayTraceMaster(gpu_ctx, sc)
        m.OnRenderImage()
        ref = m._target.GetPixels()
        host_ms = gpu_ctx.scene_info()["prepare_ms"]
        assert gpu_ctx.launch_info()["blas_builder"] == 0
        m.OnDisable()
    finally:
        gpu_ctx.set_option("blas_builder", -1)
    print(f"C5 host SAH scene preparation: {host_ms:.1f} ms")
    assert bits_equal(img, ref)
    m = RayTraceMaster(gpu_ctx, sc)                              # the default (auto) picks the GPU's SAH builder for a scene of this size
    m.OnRenderImage()
    assert gpu_ctx.launch_info()["blas_builder"] == 3 and bits_equal(m._target.GetPixels(), ref)
    m.OnDisable()


def degenerate_scene():
    """Meshes that give a splitter nothing to split on: 300 copies of ONE triangle (every centroid the same point), 257 triangles in
    a row along x with identical y / z extents (two axes without extent), a fan of 64 triangles sharing one centroid line, next to an
    ordinary blob — the builders must terminate, stay inside their level buffers and give the pixels of the host tree."""
    b = scenes.MeshSceneBuilder()
    mat = scenes._params((0.7, 0.6, 0.5), (0.1, 0.1, 0.1), (0, 0, 0), 0.4)
    tri = np.array([[-0.5, 0.2, 0.0], [0.5, 0.2, 0.0], [0.0, 1.2, 0.0]], np.float32)
    b.add(tri, np.tile(np.array([0, 1, 2], np.int32), 300), scenes.trs(translate=(-2.5, 0.3, 0.0)), mat)
    vs, ts = [], []
    for k in range(257):
        x = 0.02 * k
        vs += [[x, 0.2, 0.0], [x + 0.015, 0.2, 0.0], [x + 0.0075, 1.0, 0.0]]
        ts += [3 * k, 3 * k + 1, 3 * k + 2]
    b.add(np.array(vs, np.float32), np.array(ts, np.int32), scenes.trs(translate=(-1.0, 0.1, 1.0)), mat)
    vs, ts = [[0.0, 1.0, 0.0]], []
    for k in range(65):
        a = 2 * np.pi * k / 64
        vs.append([np.cos(a), 1.0 + 0.3 * np.sin(3 * a), np.sin(a)])
    for k in range(64):
        ts += [0, k + 2, k + 1]
    b.add(np.array(vs, np.float32), np.array(ts, np.int32), scenes.trs(translate=(2.5, 0.0, 0.5), scale=(0.8, 0.8, 0.8)), mat)
    v, t = scenes.uv_blob(24, 17)
    b.add(v, t, scenes.trs(translate=(0.5, 1.0, 2.5)), mat)
    mo, vv, ii, nn, bvh = b.finish()
    sc = scenes.Scene("degenerate", 144, 88, 4, 1, mesh_objects=mo, vertices=vv, indices=ii, normals=nn, mesh_bvh=bvh,
                      spheres=np.zeros(0, scenes.SPHERE_DT), sphere_bvh=np.zeros(0, scenes.BVHNODE_DT), sky=scenes.make_sky(64, 32))
    return sc.resized(144, 88, position=(0.0, 1.5, -7.0), fov_deg=70.0)


def test_degenerate_meshes_through_every_builder(gpu_ctx):
    sc = degenerate_scene()
    gpu_ctx.set_option("kernel_mode", 3)
    ref = None
    try:
        for builder in (0, 1, 2, 3):
            gpu_ctx.set_option("blas_builder", builder)
            m, img, _ = render(gpu_ctx, sc, frames=2)
            nodes, tri, root, info = gpu_ctx.read_scene_blas(len(sc.mesh_objects))
            assert gpu_ctx.counters()["watchdog_trips"] == 0
            m.OnDisable()
            assert len(tri) == sc.n_triangles and sorted(tri.tolist()) == sorted(set(tri.tolist())), builder      # every index slot exactly once
            assert validate(sc, nodes, tri, root) <= info["max_depth"] <= 40, (builder, info)
            if ref is None:
                ref = img
            assert bits_equal(img, ref), builder
    finally:
        gpu_ctx.set_option("blas_builder", -1)
    o = pyoracle.Oracle(sc)
    o.build_own_blas()
    ox, oy, sd = scenes.frame_uniforms(1)
    o.set_frame((ox, oy), sd)
    assert bits_equal(ref, o.render(mode=1, threads=8))


@pytest.mark.parametrize("leaf_max", [1, 4])
def test_gpu_builders_with_other_leaf_sizes_and_a_chain_shaped_mesh(gpu_ctx, leaf_max):
    """blas_leaf_max 1 (the most nodes and bin words a level of the GPU SAH builder can need) and 4, and the chain-shaped mesh whose SAH
    tree is as deep as it is long: every GPU builder gives a valid tree and the pixels of the host tree."""
    gpu_ctx.set_option("kernel_mode", 3)
    try:
        gpu_ctx.set_option("blas_leaf_max", leaf_max)
        for sc in (scenes.mixed_test_scene(120, 72, blob=(30, 23)), scenes.deep_chain_scene(96, 64, n=40)):
            ref = None
            for builder in (0, 1, 2, 3):
                gpu_ctx.set_option("blas_builder", builder)
                m, img, _ = render(gpu_ctx, sc)
                nodes, tri, root, info = gpu_ctx.read_scene_blas(len(sc.mesh_objects))
                m.OnDisable()
                assert sorted(tri.tolist()) == list(range(0, 3 * sc.n_triangles, 3)), (sc.name, builder)
                assert validate(sc, nodes, tri, root) <= info["max_depth"], (sc.name, builder, info)
                leaves = nodes[:, 12:14].view(np.int32)
                assert int((((~leaves[leaves < 0]) & 7) + 1).max()) <= max(leaf_max, 1) or builder == 0
                if ref is None:
                    ref = img
                assert bits_equal(img, ref), (sc.name, builder)
    finally:
        gpu_ctx.set_option("blas_builder", -1)
        gpu_ctx.set_option("blas_leaf_max", 2)
