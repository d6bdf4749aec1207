"""CPU: two spec decisions of DESIGN.md measured instead of asserted (VERDICT r1 "what's weak" 3).

1. The object-level slab test is normatively ONE reciprocal per axis then multiply; RS:282-283 writes two
   divisions.  `oracle.set_literal_division(True)` evaluates it as written: the test counts the pixels of C1-C3 that the
   choice moves.  The numbers asserted here are the ones recorded in DESIGN.md §2.
2. The triangle BVH pads its boxes (2^-16 of the mesh extent at build time + 2^-16 of max|origin| per ray)
   so that it never culls a triangle the float32 Moller-Trumbore test accepts.  For GRAZING rays (det just above the 1e-8
   cull threshold) the errors of u, v, t are amplified by 1/det: the test throws tens of thousands of such rays at a mesh
   and counts the cases where BVH-culled traversal (the product's BVH and the oracle's own) and literal brute force
   (RS:243-266) disagree."""
import numpy as np
import pytest

from oracle import pyoracle
from unityraytracer_amd import debug_build_blas, scenes


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_literal_division_moves_no_pixel_of_c1_c2_c3():
    moved = {}
    for name in ("C1", "C2", "C3"):
        sc = scenes.CONFIGS[name]()
        o = pyoracle.Oracle(sc)
        if len(sc.mesh_objects):
            nodes, tri, root, _, _ = debug_build_blas(sc.mesh_objects, sc.vertices, sc.indices)
            o.set_blas(nodes, tri, root)
        a = o.render(mode=1, threads=8)
        try:
            pyoracle.set_literal_division(True)
            b = o.render(mode=1, threads=8)
        finally:
            pyoracle.set_literal_division(False)
        moved[name] = int((bits(a) != bits(b)).any(axis=2).sum())
    print("pixels moved by evaluating RS:282-283 with two divisions per axis:", moved)
    assert moved == {"C1": 0, "C2": 0, "C3": 0}          # DESIGN.md §2 quotes these counts


def test_literal_division_is_a_different_function_somewhere():
    """The switch really changes arithmetic: (b - o) / d and (b - o) * (1 / d) differ in the last bit for many inputs — it is
    the hit / no-hit DECISION of the slab test that is robust, not the t values."""
    rng = np.random.default_rng(7)
    num = rng.uniform(-50, 50, 200000).astype(np.float32)
    den = (rng.uniform(-1, 1, 200000).astype(np.float32) + np.float32(1e-8)).astype(np.float32)
    a = num / den
    b = num * (np.float32(1.0) / den)
    assert int((bits(a) != bits(b)).sum()) > 10000


def _grazing_scene():
    v, t = scenes.uv_blob(40, 31)
    b = scenes.MeshSceneBuilder()
    b.add(v, t, scenes.trs(translate=(0.3, 1.5, -2.0), scale=1.7, yaw_deg=20.0), scenes._params((0.7, 0.6, 0.5), (0.1, 0.1, 0.1), (0, 0, 0), 0.4))
    mo, vv, ii, nn, bvh = b.finish()
    return scenes.Scene("graze", 64, 64, 1, 1, mesh_objects=mo, vertices=vv, indices=ii, normals=nn, mesh_bvh=bvh)


def test_grazing_rays_bvh_never_culls_a_moller_trumbore_hit():
    sc = _grazing_scene()
    prod = pyoracle.Oracle(sc)
    nodes, tri, root, _, _ = debug_build_blas(sc.mesh_objects, sc.vertices, sc.indices)
    prod.set_blas(nodes, tri, root)                       # the product's builder (binned SAH, padded boxes)
    own = pyoracle.Oracle(sc)
    own.build_own_blas()                                  # the oracle's independent median-split BVH
    W = scenes.world_vertices(sc.mesh_objects[0], sc.vertices, sc.indices).reshape(-1, 3, 3).astype(np.float64)
    rng = np.random.default_rng(1234)
    n_rays, grazed_hits, mismatches = 12000, 0, 0
    for _ in range(n_rays):
        v0, v1, v2 = W[rng.integers(len(W))]
        e1, e2 = v1 - v0, v2 - v0
        N = np.cross(e1, e2)
        Nn = np.linalg.norm(N)
        nh = N / Nn
        a = rng.uniform(0, 2 * np.pi)
        t0 = e1 / np.linalg.norm(e1)
        tang = np.cos(a) * t0 + np.sin(a) * np.cross(nh, t0)
        det_target = 10 ** rng.uniform(-8, -5)             # det = -d . (e1 x e2): RS:209-211 culls below 1e-8
        d = tang - (det_target / Nn) * nh
        d /= np.linalg.norm(d)
        w = rng.dirichlet((1, 1, 1))
        p = w[0] * v0 + w[1] * v1 + w[2] * v2
        org = p - rng.uniform(0.05, 1.5) * d
        hit, tuv = pyoracle.probe_triangle(org, d, v0, v1, v2)
        if hit and tuv[0] > 0:
            grazed_hits += 1                               # float32 Moller-Trumbore accepts the grazed triangle itself
        h0 = prod.trace(org, d, mode=0)                    # literal brute force over every triangle (RS:243-266)
        for o in (prod, own):
            h1 = o.trace(org, d, mode=1)
            same = bits(h0["distance"]) == bits(h1["distance"]) and h0["kind"] == h1["kind"] and np.array_equal(bits(h0["normal"]), bits(h1["normal"]))
            mismatches += 0 if same else 1
    print(f"grazing rays: {n_rays}, of which the grazed triangle itself is a Moller-Trumbore hit: {grazed_hits}; "
          f"BVH-culled vs brute-force disagreements (two BVHs): {mismatches}")
    assert grazed_hits > n_rays // 4                       # the construction really produces near-threshold hits
    assert mismatches == 0                                 # DESIGN.md §4 quotes this count


def test_oracle_refuses_bvh_mode_without_a_triangle_bvh():
    """mode 1 walks the triangle BVH handed over with set_blas; asking for it on a mesh scene without one is an argument error."""
    sc = scenes.mixed_test_scene(16, 8)
    o = pyoracle.Oracle(sc)
    with pytest.raises(ValueError):
        o.render(mode=1)
    assert np.isfinite(o.render(mode=0)).all()


def test_object_level_cull_moves_no_pixel_and_skips_work():
    """The product skips a MeshObject whose verified heap-leaf box the ray passes, or that lies behind the origin or beyond the ground-plane
    hit, by a margin (include/urt_math.h tlas_cull) — the reference intersects it (RS:294-326: every popped leaf once `tests` is set).
    Counted, not assumed: the culled BVH mode == the unculled BVH mode == the LITERAL brute force (mode 0, which never culls) on 40 random
    multi-mesh scenes (rotated, non-uniformly scaled objects, quads, several bounces) and on reduced C4 / C5 / the reference's Scene1;
    and the cull does skip triangle tests."""
    import json
    import os
    from tests.test_gpu_fuzz import random_scene
    from unityraytracer_amd import debug_build_blas, scenes
    cases = []
    for seed in range(3000, 3080):
        sc, _, _ = random_scene(seed)
        if len(sc.mesh_objects) >= 2:
            cases.append(sc)
        if len(cases) == 40:
            break
    cases += [scenes.CONFIGS["C4"](160, 90, slices=24, stacks=19), scenes.CONFIGS["C5"](160, 90, level=2)]
    fx = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scene_Scene1.json")
    if os.path.exists(fx):
        cases.append(scenes.from_unity_fixture(json.load(open(fx)), 96, 54))
    saved = culled_scenes = 0
    for sc in cases:
        o = pyoracle.Oracle(sc)
        o.build_own_blas()
        if o.s.n_blas_nodes <= 0:                            # (only single-leaf MeshObjects: the oracle's BVH mode wants at least one interior node)
            continue
        flags = o.cull_flags()
        assert flags is not None and len(flags) == len(sc.mesh_objects)
        a, ca = o.render(mode=1, threads=8, counters=True)
        o.set_cull(False)
        b, cb = o.render(mode=1, threads=8, counters=True)
        lit = o.render(mode=0, threads=8)
        assert np.array_equal(a.view(np.uint32), lit.view(np.uint32)), f"{sc.name}: the cull changed pixels against the literal brute force"
        assert np.array_equal(b.view(np.uint32), lit.view(np.uint32)), sc.name
        assert ca["rays"] == cb["rays"] and ca["tlas_nodes"] == cb["tlas_nodes"] and ca["tri_tests"] <= cb["tri_tests"] and ca["blas_nodes"] <= cb["blas_nodes"]
        saved += cb["tri_tests"] + cb["blas_nodes"] - ca["tri_tests"] - ca["blas_nodes"]
        culled_scenes += int(flags.any())
    assert culled_scenes >= 30 and saved > 0, (culled_scenes, saved)


def test_object_level_cull_needs_a_box_that_contains_the_object():
    """A heap leaf whose box does NOT contain its MeshObject's triangles (the scene's data may say anything: `_MeshBVH` is an input) is
    never culled — the verification rule clears its flag (oracle restatement of csrc/cullflags.hip; the GPU pass is tested in
    tests/test_gpu_parity.py) — and a MeshObject that two leaves name is not eligible either."""
    from unityraytracer_amd import scenes
    sc = scenes.CONFIGS["C4"](64, 36, slices=12, stacks=9)
    o = pyoracle.Oracle(sc)
    assert o.cull_flags().all()
    bad = sc.mesh_bvh.copy()
    leaves = [i for i in range(len(bad)) if bad[i]["index"] >= 0]
    i0 = leaves[0]
    bad[i0]["vmax"] = bad[i0]["vmin"] + (bad[i0]["vmax"] - bad[i0]["vmin"]) * 0.5      # shrunk: half of the object pokes out
    sc2 = scenes.Scene(sc.name, sc.width, sc.height, sc.num_bounces, sc.num_rays, mesh_objects=sc.mesh_objects, vertices=sc.vertices, indices=sc.indices,
                       normals=sc.normals, mesh_bvh=bad, spheres=sc.spheres, sphere_bvh=sc.sphere_bvh, sky=sc.sky)
    sc2.camera_to_world, sc2.camera_inverse_projection = sc.camera_to_world, sc.camera_inverse_projection
    f2 = pyoracle.Oracle(sc2).cull_flags()
    assert f2[bad[i0]["index"]] == 0 and f2.sum() == len(f2) - 1
    dup = sc.mesh_bvh.copy()
    dup[leaves[1]]["index"] = dup[leaves[0]]["index"]                                   # two leaves name one MeshObject, none names the other
    sc3 = scenes.Scene(sc.name, sc.width, sc.height, sc.num_bounces, sc.num_rays, mesh_objects=sc.mesh_objects, vertices=sc.vertices, indices=sc.indices,
                       normals=sc.normals, mesh_bvh=dup, spheres=sc.spheres, sphere_bvh=sc.sphere_bvh, sky=sc.sky)
    f3 = pyoracle.Oracle(sc3).cull_flags()
    assert f3[dup[leaves[0]]["index"]] == 0 and f3[sc.mesh_bvh[leaves[1]]["index"]] == 0
