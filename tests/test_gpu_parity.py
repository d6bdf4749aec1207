"""GPU parity tests proper: the HIP path, called through the C ABI, against the scalar oracle on the same
seeded inputs.  Bar: BIT-EXACT float32 (north_star allows 1e-4 per channel; the normative arithmetic of
include/urt_math.h makes equality achievable, so the tests demand it and report the 1e-4 figure too)."""
import copy

import numpy as np
import pytest

from oracle import pyoracle
from unityraytracer_amd import RayTraceMaster, debug_build_blas, scenes

pytestmark = pytest.mark.gpu

TOL = 1e-4   # north_star: "within 1e-4 per channel"


def render_gpu(ctx, scene, mode, frames=1, count=False):
    ctx.set_option("kernel_mode", mode)
    ctx.set_option("count_stats", 1 if count else 0)
    ctx.reset_counters()
    m = RayTraceMaster(ctx, scene)
    for _ in range(frames):
        m.OnRenderImage()
    target = m._target.GetPixels()
    conv = m._converged.GetPixels()
    ctrs = ctx.counters()
    m.OnDisable()
    return target, conv, ctrs


def assert_same(gpu, ref, what):
    same = np.array_equal(gpu.view(np.uint32), ref.view(np.uint32))
    if not same:
        both_nan = np.isnan(gpu) & np.isnan(ref)
        diff = np.where(both_nan, 0, np.abs(gpu.astype(np.float64) - ref.astype(np.float64)))
        bad = np.argwhere(diff > 0)
        raise AssertionError(f"{what}: {len(bad)} values differ, max |d| = {np.nanmax(diff):.3e} (tolerance {TOL}), first at {bad[:3].tolist()}")


def oracle_for(scene, use_product_blas=True):
    o = pyoracle.Oracle(scene)
    if len(scene.mesh_objects):
        nodes, tri, root, _, _ = debug_build_blas(scene.mesh_objects, scene.vertices, scene.indices)
        o.set_blas(nodes, tri, root)
    return o


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4, 5])
def test_mixed_scene_bit_exact(gpu_ctx, mode):
    sc = scenes.mixed_test_scene(200, 120)           # ragged: not a multiple of 8
    o = oracle_for(sc)
    ref, oc = o.render(mode=1, threads=8, counters=True)
    gpu, _, gc = render_gpu(gpu_ctx, sc, mode, count=True)
    assert_same(gpu, ref, f"mixed scene mode {mode}")
    for k in ("rays", "tlas_nodes", "blas_nodes", "tri_tests", "sphere_tests", "hit_tri", "hit_sphere", "hit_ground", "hit_sky", "pixels"):
        assert gc[k] == oc[k], (k, gc[k], oc[k])


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_config1_spheres_bit_exact(gpu_ctx, mode):
    sc = scenes.config1()
    ref = pyoracle.Oracle(sc).render(mode=0, threads=8)
    gpu, _, _ = render_gpu(gpu_ctx, sc, mode)
    assert_same(gpu, ref, "C1")


def test_config3_mesh_quarter_res_bit_exact(gpu_ctx):
    sc = scenes.config3(480, 270, sky=scenes.make_sky(512, 256))
    o = oracle_for(sc)
    ref = o.render(mode=1, threads=8)
    for mode in (3, 2, 1):
        gpu, _, _ = render_gpu(gpu_ctx, sc, mode)
        assert_same(gpu, ref, f"C3 480x270 mode {mode}")
    # and the literal brute-force loop of RS:243 on a crop through the mesh
    rect = (220, 120, 252, 136)
    brute = o.render(rect=rect, mode=0, threads=8)
    assert_same(gpu[rect[1]:rect[3], rect[0]:rect[2]], brute, "C3 crop vs brute force")


SCHED_VARIANTS = [
    {"sched_block": 64, "top_nodes": 0, "lds_tlas": 0},
    {"sched_block": 64, "top_nodes": 8, "top_front": 0},
    {"sched_block": 64, "top_nodes": 16, "top_front": 1},
    {"sched_block": 256, "top_nodes": 64, "top_front": 0, "lds_tlas": 0},
    {"sched_block": 256, "top_nodes": 256, "top_front": 1},
    {"sched_block": 0, "top_nodes": 64, "top_front": -1, "tile_order": 1, "xcd_run": 5},
    {"refill_min": 1, "blas_min": 1, "blas_exit": 1, "shade_min": 1, "waves_per_cu": 3},
    {"refill_min": 64, "blas_min": 64, "blas_exit": 64, "shade_min": 64, "waves_per_cu": 32},
    {"front_list": 2},                                        # the listed FRONT (the default for this scene is the masked one)
    {"front_list": 0},                                        # neither: the heap walk with the BVH top inside it
]
SCHED_DEFAULTS = {"sched_block": 0, "top_nodes": -1, "top_front": -1, "lds_tlas": 1, "tile_order": -1, "xcd_run": 0,
                  "refill_min": 16, "blas_min": 0, "blas_exit": 0, "waves_per_cu": 0, "shade_min": 32, "front_list": -1}


@pytest.mark.parametrize("variant", range(len(SCHED_VARIANTS)))
def test_scheduler_variants_do_not_change_pixels_or_counters(gpu_ctx, variant):
    """Workgroup shape, the LDS copies (BVH top, object-level tables), where the top is walked, tile order and the voting
    thresholds only decide WHEN and WHERE a pixel's operations run: same pixels, same traversal counters."""
    sc = scenes.mixed_test_scene(200, 120, blob=(40, 31))     # a blob deep enough to have > 256 BVH nodes
    o = oracle_for(sc)
    ref, oc = o.render(mode=1, threads=8, counters=True)
    try:
        for k, v in SCHED_VARIANTS[variant].items():
            gpu_ctx.set_option(k, v)
        gpu, _, gc = render_gpu(gpu_ctx, sc, 3, count=True)
    finally:
        for k, v in SCHED_DEFAULTS.items():
            gpu_ctx.set_option(k, v)
    assert_same(gpu, ref, f"scheduler variant {SCHED_VARIANTS[variant]}")
    for k in ("rays", "tlas_nodes", "blas_nodes", "tri_tests", "sphere_tests", "hit_tri", "hit_sphere", "hit_ground", "hit_sky", "pixels"):
        assert gc[k] == oc[k], (k, gc[k], oc[k])
    assert gc["watchdog_trips"] == 0


def test_many_meshes_beyond_the_lds_tables(gpu_ctx):
    """300 MeshObjects: a 1023-node object-level heap (more than the 256 entries kept in LDS) and BVH roots on both sides of
    every `top_nodes` prefix — the LDS copies and their global-memory fallbacks must agree, in every kernel mode."""
    sc = scenes.many_meshes_scene(128, 80)
    o = oracle_for(sc)
    ref, oc = o.render(mode=1, threads=8, counters=True)
    for mode in (0, 1, 2, 3, 4, 5):
        gpu, _, gc = render_gpu(gpu_ctx, sc, mode, count=True)
        assert_same(gpu, ref, f"many meshes, mode {mode}")
        for k in ("rays", "tlas_nodes", "blas_nodes", "tri_tests", "hit_tri", "hit_ground", "hit_sky", "pixels"):
            assert gc[k] == oc[k], (mode, k, gc[k], oc[k])
    try:
        for tn, tf in ((256, 1), (256, 0), (32, 1), (1, 1), (0, 0)):
            gpu_ctx.set_option("top_nodes", tn); gpu_ctx.set_option("top_front", tf)
            gpu, _, gc = render_gpu(gpu_ctx, sc, 3, count=True)
            assert_same(gpu, ref, f"many meshes, top_nodes {tn} top_front {tf}")
            assert gc["blas_nodes"] == oc["blas_nodes"] and gc["tri_tests"] == oc["tri_tests"] and gc["watchdog_trips"] == 0
    finally:
        gpu_ctx.set_option("top_nodes", -1); gpu_ctx.set_option("top_front", -1)


def test_deep_stacks_use_the_large_lds_launch_path(gpu_ctx):
    """A very deep BVH needs more than 64 KiB of LDS per 4-wave workgroup: the launch then raises the kernel's dynamic-LDS
    limit and fewer workgroups fit a CU.  `stack_pad` inflates the stacks of a normal scene to get there."""
    sc = scenes.mixed_test_scene(160, 96, blob=(40, 31))
    ref = oracle_for(sc).render(mode=1, threads=8)
    try:
        for pad, mode in ((48, 3), (96, 3), (96, 2), (96, 1), (96, 0)):
            gpu_ctx.set_option("stack_pad", pad)
            gpu, _, gc = render_gpu(gpu_ctx, sc, mode)
            assert_same(gpu, ref, f"stack_pad {pad}, mode {mode}")
            assert gc["watchdog_trips"] == 0
    finally:
        gpu_ctx.set_option("stack_pad", 0)


def test_masked_phase_at_fewer_workgroups_per_cu(gpu_ctx):
    """Deep traversal stacks in a multi-mesh scene (C4 in small: Cornell box + 3 blobs, masked object-level phase): when 5 workgroups of
    stacks no longer fit a CU's LDS, the launch counts on 4 or 3 and keeps the masked phase and the LDS copies (context.cpp
    configure_sched) before it falls back to the large-LDS path.  `stack_pad` walks through all of these layouts; the GPU-built
    (deeper) tree is one of them for real.  Pixels and counters == oracle in each."""
    sc = scenes.config4(160, 90, slices=40, stacks=31, sky=scenes.make_sky(64, 32))
    sc.num_bounces = 4
    o = oracle_for(sc)
    ref, oc = o.render(mode=1, threads=8, counters=True)
    try:
        for pad in (0, 4, 8, 12, 16, 24, 40):
            gpu_ctx.set_option("stack_pad", pad)
            gpu, _, gc = render_gpu(gpu_ctx, sc, 3, count=True)
            assert_same(gpu, ref, f"C4 small, stack_pad {pad}")
            assert gc["watchdog_trips"] == 0
            for k in ("rays", "tlas_nodes", "blas_nodes", "tri_tests"):
                assert gc[k] == oc[k], (pad, k, gc[k], oc[k])
        gpu_ctx.set_option("stack_pad", 0)
        gpu_ctx.set_option("blas_builder", 1)                  # the GPU-built Morton tree: pixels only (its own tree, its own counts)
        gpu, _, gc = render_gpu(gpu_ctx, sc, 3)
        assert_same(gpu, ref, "C4 small, GPU-built tree")
        assert gc["watchdog_trips"] == 0
    finally:
        gpu_ctx.set_option("stack_pad", 0); gpu_ctx.set_option("blas_builder", -1)


def test_chain_shaped_triangle_bvh_fills_the_traversal_stack(gpu_ctx):
    """A triangle BVH that is a chain (one triangle per leaf, every split peels off the far end) is as deep as it has leaves, and a ray
    down its axis has the far child of EVERY level on its stack at once: the per-lane LDS stacks (depth + free slot + the sentinel
    entry of the pointer-form loop, kernels.hip blas_node_eval_ptr) are used to the last entry.  Pixels and counters == oracle in every mode."""
    sc = scenes.deep_chain_scene()
    try:
        gpu_ctx.set_option("blas_leaf_max", 1)                 # (process-wide builder setting: the oracle's copy of the BVH follows)
        m = RayTraceMaster(gpu_ctx, sc); m.OnRenderImage()
        _, _, _, info = gpu_ctx.read_scene_blas(len(sc.mesh_objects))
        m.OnDisable()
        assert info["max_depth"] >= 22, info              # (the product builder: 24 levels for the 40 triangles)
        o = oracle_for(sc)
        ref, oc = o.render(mode=1, threads=8, counters=True)
        assert oc["hit_tri"] > 0
        for mode in (3, 0, 2, 4, 5):
            gpu, _, gc = render_gpu(gpu_ctx, sc, mode, count=True)
            assert_same(gpu, ref, f"chain BVH, mode {mode}")
            assert gc["watchdog_trips"] == 0
            for k in ("rays", "blas_nodes", "tri_tests"):
                assert gc[k] == oc[k], (mode, k, gc[k], oc[k])
    finally:
        gpu_ctx.set_option("blas_leaf_max", 2)


@pytest.mark.parametrize("k", [1, 2, 3, 4])
def test_path_pool_sizes_bit_exact(gpu_ctx, k):
    sc = scenes.mixed_test_scene(200, 120)
    sc.num_rays = 2
    ref = oracle_for(sc).render(mode=1, threads=8)
    try:
        gpu_ctx.set_option("pool_k", k)
        gpu_ctx.set_option("pool_blas_exit", 16 if k > 1 else 8)
        gpu, _, gc = render_gpu(gpu_ctx, sc, 4, count=True)
    finally:
        gpu_ctx.set_option("pool_k", 2); gpu_ctx.set_option("pool_blas_exit", 8)
    assert_same(gpu, ref, f"path pool K={k}, 2 rays per pixel")
    assert gc["watchdog_trips"] == 0


def test_multi_ray_multi_frame_accumulation(gpu_ctx):
    sc = scenes.mixed_test_scene(96, 64)
    sc.num_rays, sc.num_bounces = 3, 5
    o = oracle_for(sc)
    conv_ref = np.zeros((64, 96, 4), np.float32)
    for f in range(3):
        ox, oy, seed = scenes.frame_uniforms(f)
        o.set_frame((ox, oy), seed)
        conv_ref = pyoracle.accumulate(o.render(mode=1, threads=8), conv_ref, f)
    for mode in (0, 1, 2, 3, 4, 5):
        _, conv, _ = render_gpu(gpu_ctx, sc, mode, frames=3)
        assert_same(conv, conv_ref, f"3-frame running mean, mode {mode}")


def test_long_progressive_accumulation(gpu_ctx):
    """200 frames of the running mean (AS:9,39-41 with _Sample = 0..199, per-frame _PixelOffset/_Seed): the converged image
    still equals the oracle's bit for bit — the blend is evaluated in the same order on both sides, so nothing drifts."""
    sc = scenes.mixed_test_scene(64, 40)
    o = oracle_for(sc)
    conv_ref = np.zeros((40, 64, 4), np.float32)
    for f in range(200):
        ox, oy, seed = scenes.frame_uniforms(f)
        o.set_frame((ox, oy), seed)
        conv_ref = pyoracle.accumulate(o.render(mode=1, threads=8), conv_ref, f)
    _, conv, c = render_gpu(gpu_ctx, sc, 3, frames=200)
    assert_same(conv, conv_ref, "200-frame running mean")
    assert np.isfinite(conv).all() and c["watchdog_trips"] == 0


def test_strips_union_equals_full_frame(gpu_ctx):
    """dispatch_rows(r, N) for r = 0..N-1 writes exactly the pixels of one full dispatch (global ids)."""
    sc = scenes.mixed_test_scene(120, 100)
    full, _, _ = render_gpu(gpu_ctx, sc, 2)
    world = 3
    union = np.zeros_like(full)
    for r in range(world):
        gpu_ctx.set_option("kernel_mode", 2)
        m = RayTraceMaster(gpu_ctx, sc, rank=r, world_size=world)
        m.OnRenderImage()
        part = m._target.GetPixels()
        from unityraytracer_amd import strips
        for (y0, y1) in strips.strip_row_ranges(sc.height, r, world):
            union[y0:y1] = part[y0:y1]
        # rows this rank does not own stay untouched (zero-initialised texture)
        own = np.zeros(sc.height, bool)
        for (y0, y1) in strips.strip_row_ranges(sc.height, r, world):
            own[y0:y1] = True
        assert not part[~own].any()
        m.OnDisable()
    assert_same(union, full, "union of strips")


def test_shared_traversal_service_mode5(gpu_ctx):
    """kernel_mode 5: paths post the rays that must enter a triangle BVH to their workgroup's mailbox; any wave of the workgroup
    claims them, walks them on its idle lanes, answers or suspends them.  Pixels and every counter equal the oracle's whatever
    the service thresholds — including entry thresholds below the yield threshold (a visit always advances its rays one trip)
    and refill on every idle lane — on one mesh, on several (rays suspended mid-BVH resume on another wave) and with _numRays 3."""
    cases = [(scenes.config3(256, 160), {}), (scenes.config3(256, 160), {"blas_min": 8, "blas_exit": 40, "serve_refill": 1}),
             (scenes.config3(256, 160), {"blas_min": 200, "blas_exit": 1, "serve_refill": 64}), (scenes.many_meshes_scene(128, 80), {"blas_min": 16, "blas_exit": 32})]
    multi = scenes.mixed_test_scene(96, 64); multi.num_rays = 3
    cases.append((multi, {}))
    try:
        for sc, opts in cases:
            o = oracle_for(sc)
            ref, oc = o.render(mode=1, threads=8, counters=True)
            for k, v in opts.items():
                gpu_ctx.set_option(k, v)
            gpu, _, gc = render_gpu(gpu_ctx, sc, 5, count=True)
            sv = gpu_ctx.serve_stats()
            for k, v in {"blas_min": 0, "blas_exit": 0, "serve_refill": 16}.items():
                gpu_ctx.set_option(k, v)
            assert_same(gpu, ref, f"mode 5 {sc.name} {opts}")
            for k in ("rays", "tlas_nodes", "blas_nodes", "tri_tests", "hit_tri", "hit_ground", "hit_sky", "pixels"):
                assert gc[k] == oc[k], (opts, k, gc[k], oc[k])
            assert gc["watchdog_trips"] == 0
            assert sv["visits"] > 0 and sv["claimed"] > 0 and sv["lane_trips"] >= sv["trips"] > 0, sv
            if opts.get("blas_exit", 0) >= 32:
                assert sv["suspended"] > 0, sv                   # the yield path (suspend -> resumed by any wave) really ran
    finally:
        for k, v in {"blas_min": 0, "blas_exit": 0, "serve_refill": 16}.items():
            gpu_ctx.set_option(k, v)
        gpu_ctx.set_option("kernel_mode", 3); gpu_ctx.set_option("count_stats", 0)


def _heap(nodes):
    """[(vmin, vmax, index) | None = filler] -> BVHNODE array (RM:148-152; fillers all-zero, index -1: RM:490-494)"""
    a = np.zeros(len(nodes), scenes.BVHNODE_DT)
    for i, nd in enumerate(nodes):
        if nd is None:
            a[i]["index"] = -1
        else:
            a[i]["vmin"], a[i]["vmax"], a[i]["index"] = nd
    return a


def test_masked_front_on_odd_object_heaps(gpu_ctx):
    """The masked FRONT derives the reference's object-level walk (RS:294-326) from one slab-test bit per heap node.  Heaps the
    reference's builder never makes but its shader would walk all the same: a buffer that is not a complete tree, a leaf high up
    beside a deep subtree, a hit leaf whose object id is out of range (it still sets the never-reset `tests` flag, A.5), filler
    nodes below leaves, an interior node with empty bounds, 16 MeshObjects in a 31-node heap (more than the listed form's 12).
    Pixels AND traversal counters must equal the oracle's literal walk, for the masked, the listed and the plain FRONT."""
    rng = scenes.SplitMix64(0x0DD)
    b = scenes.MeshSceneBuilder()
    shapes = []
    for k in range(16):
        if k % 4 == 3:
            s_ = 0.8 + 0.4 * rng.value()
            v, t = scenes.quad((-s_, 0, 0), (s_, 0, 0), (s_, 2 * s_, 0), (-s_, 2 * s_, 0))
        else:
            v, t = scenes.icosphere(1 if k % 2 else 0, bumps=0.1)
        x, z = (k % 4 - 1.5) * 2.2 + 0.5 * (rng.value() - 0.5), (k // 4) * 2.4 - 2.0
        b.add(v, t, scenes.trs(translate=(x, 0.2 + 0.9 * rng.value(), z), scale=0.6 + 0.5 * rng.value(), yaw_deg=360.0 * rng.value()),
              scenes._params((rng.value(), rng.value(), rng.value()), (0.1, 0.1, 0.1), (1.5, 1.2, 0.8) if k == 5 else (0, 0, 0), rng.value()))
    mo, vv, ii, nn, bvh16 = b.finish()
    lo, hi = scenes.mesh_bounds(mo, vv, ii)
    box = lambda ks: (lo[list(ks)].min(axis=0), hi[list(ks)].max(axis=0))
    leaf = lambda k: (lo[k], hi[k], k)
    inner = lambda ks: box(ks) + (-1,)
    every = range(16)
    heaps = {
        "builder, 16 objects / 31 nodes": bvh16,
        "5 nodes (not a complete tree), leaf beside a subtree": _heap([inner(every), inner((0, 1)), leaf(2), leaf(0), leaf(1)]),
        "out-of-range id sets `tests`, fillers under leaves": _heap([inner(every), inner((3, 4)), (lo[5], hi[5], 99), leaf(3), leaf(4), None, None]),
        "empty interior node, deep left spine": _heap([inner(every), inner(every), (lo[6], lo[6], -1), inner((7, 8, 9)), leaf(10), None, None,
                                                       inner((7, 8)), leaf(9), None, None, None, None, None, None, leaf(7), leaf(8)]),
        "single leaf": _heap([leaf(11)]),
    }
    base = scenes.Scene("odd-heaps", 144, 88, 4, 1, mesh_objects=mo, vertices=vv, indices=ii, normals=nn, mesh_bvh=bvh16, sky=scenes.make_sky(64, 32))
    base = base.resized(144, 88, position=(0.3, 3.5, -9.0), fov_deg=70.0)
    # ... and a MeshObject WITHOUT triangles whose heap leaf is hit: it sets `tests` like any leaf but is never tested itself (a table
    # that listed it would send the ray into a triangle BVH that does not exist)
    empty = copy.copy(base)
    mo2 = mo.copy(); mo2[9]["indices_count"] = 0; mo2[2]["indices_count"] = 0
    empty.mesh_objects = mo2
    variants = [(name, base, heap) for name, heap in heaps.items()] + [("two MeshObjects without triangles", empty, bvh16)]
    try:
        for name, scene0, heap in variants:
            sc = copy.copy(scene0)
            sc.mesh_bvh = heap
            o = oracle_for(sc)
            ref, oc = o.render(mode=1, threads=8, counters=True)
            for fl in (-1, 2, 0):
                gpu_ctx.set_option("front_list", fl)
                gpu, _, gc = render_gpu(gpu_ctx, sc, 3, count=True)
                assert_same(gpu, ref, f"{name}, front_list {fl}")
                for k in ("rays", "tlas_nodes", "blas_nodes", "tri_tests", "hit_tri", "hit_ground", "hit_sky"):
                    assert gc[k] == oc[k], (name, fl, k, gc[k], oc[k])
                assert gc["watchdog_trips"] == 0
    finally:
        gpu_ctx.set_option("front_list", -1)


@pytest.mark.parametrize("mode", [0, 2, 3])
def test_round1_fault_configurations(gpu_ctx, mode):
    """The two configurations of round 1's diagnostics (gpurun_out/diag*.log): the megakernel on a mesh scene WITHOUT
    spheres (a nil-address fault then) and the mixed scene at numBounces 1..3 (88 / 305 / 477 wrong pixels then: emission of triangle
    hits read as 0).  Cause, from the evidence: the first build uploaded every scene table with hipMemcpyAsync from short-lived pageable
    staging vectors (fixed the same day); the ISA of that build's megakernel selects the material tables correctly.  Must stay green."""
    sc = scenes.mixed_test_scene(200, 120)
    sc.spheres = np.zeros(0, scenes.SPHERE_DT)
    sc.sphere_bvh = np.zeros(0, scenes.BVHNODE_DT)
    o = oracle_for(sc)
    ref, oc = o.render(mode=1, threads=8, counters=True)
    gpu, _, gc = render_gpu(gpu_ctx, sc, mode, count=True)
    assert_same(gpu, ref, f"mesh scene without spheres, mode {mode}")
    assert gc["hit_tri"] == oc["hit_tri"] and gc["sphere_tests"] == 0
    emissive = (ref[..., :3].max(axis=2) > 3.0).sum()
    assert emissive > 50                                      # the emissive quad is seen: its emission is what round 1 lost
    for b in (1, 2, 3):
        sc2 = scenes.mixed_test_scene(200, 120)
        sc2.num_bounces = b
        ref = oracle_for(sc2).render(mode=1, threads=8)
        gpu, _, _ = render_gpu(gpu_ctx, sc2, mode)
        assert_same(gpu, ref, f"mixed scene, numBounces {b}, mode {mode}")


@pytest.mark.parametrize("mode", [0, 2, 3])
def test_object_level_cull_on_the_gpu(gpu_ctx, mode):
    """The object-level cull (include/urt_math.h tlas_cull) through every walk that has it — the literal heap walk of the per-pixel and
    persistent kernels, the listed / masked object-level phase of the default kernel: pixels == the oracle's LITERAL brute force (mode 0:
    every popped object is intersected, as the reference does), pixels and traversal counters == the oracle's culled BVH mode, and the
    cull really skips work (fewer triangle tests than with front_cull = 0).  Then the GPU verification pass (csrc/cullflags.hip): with
    one heap leaf's box shrunk so that its MeshObject pokes out, that leaf is never culled — counters again equal the oracle, whose
    restated rule clears the same flag — and the pixels still equal the literal brute force."""
    sc = scenes.CONFIGS["C4"](96, 54, slices=14, stacks=11)
    o = oracle_for(sc)
    assert o.cull_flags().all()
    lit = o.render(mode=0, threads=8)
    ref, oc = o.render(mode=1, threads=8, counters=True)
    gpu, _, gc = render_gpu(gpu_ctx, sc, mode, count=True)
    assert_same(gpu, lit, f"C4-small mode {mode} vs the literal brute force")
    assert_same(gpu, ref, f"C4-small mode {mode}")
    for k in ("rays", "tlas_nodes", "blas_nodes", "tri_tests", "hit_tri", "hit_ground", "hit_sky"):
        assert gc[k] == oc[k], (k, gc[k], oc[k])
    gpu_ctx.set_option("front_cull", 0)
    try:
        gpu0, _, gc0 = render_gpu(gpu_ctx, sc, mode, count=True)
    finally:
        gpu_ctx.set_option("front_cull", 1)
    assert_same(gpu0, lit, "front_cull = 0")
    o.set_cull(False)
    _, oc0 = o.render(mode=1, threads=8, counters=True)
    assert gc0["tri_tests"] == oc0["tri_tests"] and gc0["blas_nodes"] == oc0["blas_nodes"]
    assert gc["tri_tests"] < 0.6 * gc0["tri_tests"], (gc["tri_tests"], gc0["tri_tests"])
    # a leaf whose box does not contain its object
    bad = sc.mesh_bvh.copy()
    leaf = [i for i in range(len(bad)) if bad[i]["index"] >= 0][3]
    bad[leaf]["vmax"] = bad[leaf]["vmin"] + (bad[leaf]["vmax"] - bad[leaf]["vmin"]) * 0.5
    sc2 = copy.copy(sc)
    sc2.mesh_bvh = bad
    o2 = oracle_for(sc2)
    f2 = o2.cull_flags()
    assert f2[bad[leaf]["index"]] == 0 and f2.sum() == len(f2) - 1
    lit2 = o2.render(mode=0, threads=8)
    ref2, oc2 = o2.render(mode=1, threads=8, counters=True)
    gpu2, _, gc2 = render_gpu(gpu_ctx, sc2, mode, count=True)
    assert_same(gpu2, lit2, "shrunk leaf box vs the literal brute force")
    assert_same(gpu2, ref2, "shrunk leaf box")
    for k in ("tlas_nodes", "blas_nodes", "tri_tests"):
        assert gc2[k] == oc2[k], (k, gc2[k], oc2[k])
