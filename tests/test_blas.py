"""The product's triangle BVH (unityraytracer_amd/csrc/blas_builder.cpp, reached through the C ABI's host-only
introspection entry points) is (a) structurally valid and (b) result-neutral: the oracle's BVH-culled mode returns
exactly the pixels of the literal brute-force loop RS:243-266 — also with the oracle's own independent BVH."""
import numpy as np
import pytest

from oracle import pyoracle
from unityraytracer_amd import debug_build_blas, scenes


def decode_leaf(code):
    c = (~np.int64(code)) & 0xFFFFFFFF
    return int(c >> 3), int(c & 7) + 1


def tri_world(scene, slot):
    mo = next(m for m in scene.mesh_objects if m["indices_offset"] <= slot < m["indices_offset"] + m["indices_count"])
    m = np.asarray(mo["localToWorldMatrix"], np.float64).reshape(4, 4).T
    v = scene.vertices[scene.indices[slot:slot + 3]].astype(np.float64)
    return v @ m[:3, :3].T + m[:3, 3]


def validate(scene, nodes, tri_index, mesh_root):
    seen = np.zeros(len(tri_index), bool)
    max_depth = 0
    for m, root in enumerate(mesh_root):
        if root == 0x7FFFFFFF:
            continue
        stack = [(int(root), None, 1)]
        while stack:
            code, box, depth = stack.pop()
            max_depth = max(max_depth, depth)
            if code < 0:
                first, cnt = decode_leaf(code)
                assert 1 <= cnt <= 8
                for k in range(first, first + cnt):
                    assert not seen[k]
                    seen[k] = True
                    w = tri_world(scene, int(tri_index[k]))
                    if box is not None:
                        assert np.all(w >= box[0] - 1e-7) and np.all(w <= box[1] + 1e-7), "triangle outside its leaf box"
            else:
                n = nodes[code]
                c = n[12:14].view(np.int32)
                for j in range(2):
                    lo, hi = n[6 * j:6 * j + 3].astype(np.float64), n[6 * j + 3:6 * j + 6].astype(np.float64)
                    assert np.all(lo <= hi)
                    if box is not None:
                        assert np.all(lo >= box[0] - 1e-4) and np.all(hi <= box[1] + 1e-4), "child box escapes its parent"
                    stack.append((int(c[j]), (lo, hi), depth + 1))
    assert seen.all(), "a triangle is missing from the BVH"
    return max_depth


def test_parallel_build_is_deterministic(built_library):
    """MeshObjects are built on several host threads and concatenated in MeshObject order: the BVH must not depend on the
    thread count (URT_BLAS_THREADS overrides it for this test)."""
    import hashlib, os, subprocess, sys
    code = ("import sys, hashlib; sys.path.insert(0, %r)\n"
            "from unityraytracer_amd import scenes, debug_build_blas\n"
            "sc = scenes.many_meshes_scene(n=40, level=2)\n"
            "r = debug_build_blas(sc.mesh_objects, sc.vertices, sc.indices)\n"
            "print(hashlib.sha256(b''.join(a.tobytes() for a in r[:4])).hexdigest(), r[4])\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for threads in ("1", "2", "7"):
        env = dict(os.environ, URT_BLAS_THREADS=threads)
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip())
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 64


def test_top_of_forest_is_breadth_first(built_library):
    """Node indices [0, 256) are the top of the forest in breadth-first order (roots of all meshes first): the phase
    scheduler copies a prefix of the node array to LDS and relies on every parent of a prefix node being in the prefix."""
    sc = scenes.mixed_test_scene(64, 48, blob=(40, 31))
    nodes, tri, root, first, depth = debug_build_blas(sc.mesh_objects, sc.vertices, sc.indices)
    roots = [int(r) for r in root if 0 <= r != 0x7FFFFFFF]
    assert roots == list(range(len(roots))), "interior roots must be nodes 0..k-1 in MeshObject order"
    queue, order = list(roots), []
    while queue and len(order) < 256:
        n = queue.pop(0)
        order.append(n)
        queue += [int(c) for c in nodes[n][12:14].view(np.int32) if c >= 0]
    assert order == list(range(len(order))), "breadth-first walk must meet nodes 0, 1, 2, ... in order"
    parent = {}
    for i, n in enumerate(nodes):
        for c in n[12:14].view(np.int32):
            if c >= 0:
                assert int(c) not in parent, "a node has two parents"
                parent[int(c)] = i
    for t in (8, 16, 64, min(256, len(nodes))):
        assert all(parent[i] < t for i in range(len(roots), min(t, len(nodes)))), f"prefix {t} is not closed under parent"


@pytest.mark.parametrize("scene_fn", [lambda: scenes.mixed_test_scene(64, 48), lambda: scenes.config3(64, 36, slices=40, stacks=31, sky=scenes.make_sky(64, 32))])
def test_product_blas_is_valid(built_library, scene_fn):
    sc = scene_fn()
    nodes, tri, root, first, depth = debug_build_blas(sc.mesh_objects, sc.vertices, sc.indices)
    assert len(tri) == sc.n_triangles and sorted(tri.tolist()) == list(range(0, 3 * sc.n_triangles, 3))
    assert validate(sc, nodes, tri, root) <= depth


def test_culled_mode_equals_brute_force(built_library):
    sc = scenes.mixed_test_scene(120, 80)
    sc.num_bounces = 6
    o = pyoracle.Oracle(sc)
    brute, cb = o.render(mode=0, threads=8, counters=True)
    nodes, tri, root, _, _ = debug_build_blas(sc.mesh_objects, sc.vertices, sc.indices)
    o.set_blas(nodes, tri, root)
    culled, cc = o.render(mode=1, threads=8, counters=True)
    assert np.array_equal(brute.view(np.uint32), culled.view(np.uint32))
    assert cc["rays"] == cb["rays"] and cc["hit_tri"] == cb["hit_tri"] and cc["tri_tests"] < cb["tri_tests"]
    o.build_own_blas()                                     # the oracle's independent median-split builder
    own = o.render(mode=1, threads=8)
    assert np.array_equal(brute.view(np.uint32), own.view(np.uint32))


def test_culled_mode_equals_brute_force_on_a_chain_shaped_bvh(built_library):
    """scenes.deep_chain_scene: coordinates from 1 to 4e18 in one MeshObject (the build-time pad is 2^-16 of that: every near box swallows the
    camera) and a BVH that is one long chain — brute force and the culled walk must still agree, and the product BVH must be valid."""
    sc = scenes.deep_chain_scene()
    o = pyoracle.Oracle(sc)
    brute, cb = o.render(mode=0, threads=8, counters=True)
    nodes, tri, root, _, depth = debug_build_blas(sc.mesh_objects, sc.vertices, sc.indices)
    assert validate(sc, nodes, tri, root) <= depth and depth >= 10
    o.set_blas(nodes, tri, root)
    culled, cc = o.render(mode=1, threads=8, counters=True)
    assert np.array_equal(brute.view(np.uint32), culled.view(np.uint32))
    assert cc["hit_tri"] == cb["hit_tri"] > 0 and np.isfinite(culled).all()


def test_culled_mode_equals_brute_force_on_dense_mesh_crop(built_library):
    sc = scenes.config3(160, 90, slices=60, stacks=47, sky=scenes.make_sky(64, 32))      # 5,520 triangles
    o = pyoracle.Oracle(sc)
    rect = (56, 30, 104, 70)
    brute = o.render(rect=rect, mode=0, threads=8)
    nodes, tri, root, _, _ = debug_build_blas(sc.mesh_objects, sc.vertices, sc.indices)
    o.set_blas(nodes, tri, root)
    assert np.array_equal(brute.view(np.uint32), o.render(rect=rect, mode=1, threads=8).view(np.uint32))


def test_equal_t_tie_goes_to_lowest_index_slot(built_library):
    """A.4: two coincident triangles — brute force keeps the first (strict '<'); BVH order must not change that."""
    b = scenes.MeshSceneBuilder()
    v = np.array([[-1, 1, 2], [1, 1, 2], [0, 3, 2]], np.float32)
    t = np.array([[0, 1, 2]], np.int32)
    if not scenes.front_facing(v[0].astype(float), v[1].astype(float), v[2].astype(float), np.array([0.0, 0.0, 1.0])):
        t = t[:, [0, 2, 1]]
    red = scenes._params((0, 0, 0), (0, 0, 0), (1, 0, 0), 0)
    green = scenes._params((0, 0, 0), (0, 0, 0), (0, 1, 0), 0)
    # one MeshObject holding the same triangle many times, then a second object with it again
    many = np.concatenate([t] * 9)
    b.add(v, many, scenes.trs(), red)
    b.add(v, t, scenes.trs(), green)
    mo, vv, ii, nn, bvh = b.finish()
    sc = scenes.Scene("tie", 32, 32, 1, 1, mesh_objects=mo, vertices=vv, indices=ii, normals=nn, mesh_bvh=bvh, sky=np.zeros((4, 8, 4), np.float32))
    sc.camera_to_world, sc.camera_inverse_projection = scenes.camera_matrices(32, 32, position=(0, 2, -3), fov_deg=40)
    o = pyoracle.Oracle(sc)
    brute = o.render(mode=0)
    nodes, tri, root, _, _ = debug_build_blas(sc.mesh_objects, sc.vertices, sc.indices)
    o.set_blas(nodes, tri, root)
    culled = o.render(mode=1)
    assert np.array_equal(brute.view(np.uint32), culled.view(np.uint32))
    centre = brute[16, 16, :3].tolist()
    assert centre in ([1.0, 0.0, 0.0], [0.0, 1.0, 0.0])      # whichever object the heap visits first wins, identically in both modes


def test_axis_parallel_rays_match_brute_force(built_library):
    """Rays with direction components that are EXACTLY zero (a diffuse bounce whose rand() returned 0 leaves exactly
    tangent to the surface, e.g. d.y == 0 off the ground plane).  1/0 = inf must not turn the slab test into
    inf - inf = NaN (which both mis-culls and walks a whole slab of the mesh): blas_rcp() in urt_math.h."""
    sc = scenes.config3(64, 36, slices=48, stacks=37, sky=scenes.make_sky(64, 32))
    o = pyoracle.Oracle(sc)
    nodes, tri, root, _, _ = debug_build_blas(sc.mesh_objects, sc.vertices, sc.indices)
    rng = np.random.default_rng(5)
    rays = []
    for _ in range(300):
        p = np.array([rng.uniform(-3, 3), rng.uniform(0.001, 5), rng.uniform(-7, -1)])
        ax = rng.integers(0, 3)
        d = rng.normal(size=3)
        d[ax] = 0.0                                   # one exact zero
        if rng.random() < 0.3:
            d[(ax + 1) % 3] = 0.0                     # two exact zeros: axis-aligned
        d /= np.linalg.norm(d)
        rays.append((p - 6 * d, d))
    rays.append((np.array([0.9, 0.001, -8.2]), np.array([-0.104122, 0.0, 0.994565])))     # a measured pathological ray
    o.set_blas(nodes, tri, root)
    worst = 0
    for p, d in rays:
        a = o.trace(p, d, mode=0)
        c0 = pyoracle.OracleCounters()
        b = o.trace(p, d, mode=1)
        assert a["kind"] == b["kind"] and (a["distance"] == b["distance"] or (np.isinf(a["distance"]) and np.isinf(b["distance"])))
        assert np.array_equal(a["normal"], b["normal"], equal_nan=True)
    # and the traversal stays local: a horizontal ray skimming the ground under the mesh visits few nodes
    sc2 = scenes.config3(32, 18, sky=scenes.make_sky(64, 32))
    o2 = pyoracle.Oracle(sc2)
    n2, t2, r2, _, _ = debug_build_blas(sc2.mesh_objects, sc2.vertices, sc2.indices)
    o2.set_blas(n2, t2, r2)
    sc2.num_bounces = 8
    _, c = o2.render(mode=1, threads=8, counters=True)
    assert c["max_ray_steps"] < 600
