"""GPU: randomised scenes against the oracle (bit for bit).  Each case draws object counts, shapes, transforms, materials (incl.
the degenerate ones: black non-emissive = NaN roulette chances, pure mirrors, emissive-only), camera pose, resolution (ragged),
numBounces / numRays, the triangle-BVH builder (host SAH / GPU LBVH) and the number of accumulated frames.  The oracle walks its
OWN independently built BVH, so nothing of the product's acceleration structures is shared with the checker."""
import numpy as np
import pytest

from oracle import pyoracle
from unityraytracer_amd import RayTraceMaster, scenes

pytestmark = pytest.mark.gpu


def random_scene(seed):
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(17, 150)), int(rng.integers(9, 100))
    def material():
        kind = rng.integers(0, 6)
        if kind == 0:
            return scenes._params((0, 0, 0), (0, 0, 0), (0, 0, 0), 0.5)                       # absorbs everything: 0/0 chances (A.6)
        if kind == 1:
            return scenes._params((0, 0, 0), tuple(rng.uniform(0.6, 1.0, 3)), (0, 0, 0), float(rng.uniform(0.5, 1.0)))   # mirror-like
        if kind == 2:
            return scenes._params((0, 0, 0), (0, 0, 0), tuple(rng.uniform(0.5, 4.0, 3)), 0.0)  # emitter only
        return scenes._params(tuple(rng.uniform(0, 1, 3)), tuple(rng.uniform(0, 0.6, 3)), tuple(rng.uniform(0, 0.3, 3) * (rng.random() < 0.3)), float(rng.uniform(0, 1)))
    n_sph = int(rng.integers(0, 10))
    sp = np.zeros(n_sph, dtype=scenes.SPHERE_DT)
    for k in range(n_sph):
        r = float(rng.uniform(0.2, 1.2))
        sp[k]["position"] = (float(rng.uniform(-5, 5)), r + float(rng.uniform(-0.3, 1.5)), float(rng.uniform(-5, 5)))
        sp[k]["radius"] = r
        sp[k]["lighting"] = material()
    b = scenes.MeshSceneBuilder()
    for _ in range(int(rng.integers(0, 5))):
        shape = rng.integers(0, 3)
        if shape == 0:
            v, t = scenes.uv_blob(int(rng.integers(6, 24)), int(rng.integers(5, 16)), bumps=float(rng.uniform(0, 0.3)), phase=float(rng.uniform(0, 6)))
        elif shape == 1:
            v, t = scenes.icosphere(int(rng.integers(0, 3)), bumps=float(rng.uniform(0, 0.2)))
        else:
            s = float(rng.uniform(0.5, 3))
            v, t = scenes.quad((-s, 0, -s), (s, 0, -s), (s, 0, s), (-s, 0, s)) if rng.random() < 0.5 else scenes.quad((-s, 0, 0), (s, 0, 0), (s, 2 * s, 0), (-s, 2 * s, 0))
        sc3 = tuple(rng.uniform(0.4, 1.8, 3)) if rng.random() < 0.5 else float(rng.uniform(0.4, 1.8))
        b.add(v, t, scenes.trs(translate=(float(rng.uniform(-4, 4)), float(rng.uniform(0.0, 2.5)), float(rng.uniform(-4, 4))), scale=sc3,
                               yaw_deg=float(rng.uniform(0, 360))), material())
    mo, vv, ii, nn, bvh = b.finish()
    sc = scenes.Scene(f"fuzz{seed}", w, h, int(rng.integers(1, 7)), int(rng.integers(1, 4)), mesh_objects=mo, vertices=vv, indices=ii, normals=nn, mesh_bvh=bvh,
                      spheres=sp, sphere_bvh=scenes.build_object_bvh(*scenes.sphere_bounds(sp)) if n_sph else np.zeros(0, scenes.BVHNODE_DT),
                      sky=scenes.make_sky(int(rng.integers(8, 96)), int(rng.integers(4, 48))))
    pos = (float(rng.uniform(-6, 6)), float(rng.uniform(0.3, 5)), float(rng.uniform(-12, -6)))
    sc = sc.resized(w, h, position=pos, fov_deg=float(rng.uniform(40, 100)))
    sc.name = f"fuzz{seed}"
    return sc, int(rng.integers(0, 2)), int(rng.integers(1, 4))


@pytest.mark.parametrize("seed", range(16))
def test_random_scene_bit_exact(gpu_ctx, seed):
    sc, builder, frames = random_scene(1000 + seed)
    if builder == 1:
        builder = (1, 2, 3)[seed % 3]                          # the three GPU builders take turns: Karras tree, depth-budgeted tree, binned SAH
    gpu_ctx.set_option("kernel_mode", 3)
    gpu_ctx.set_option("blas_builder", builder)
    try:
        m = RayTraceMaster(gpu_ctx, sc)
        for _ in range(frames):
            m.OnRenderImage()
        got_t, got_c = m._target.GetPixels(), m._converged.GetPixels()
        m.OnDisable()
    finally:
        gpu_ctx.set_option("blas_builder", -1)
    o = pyoracle.Oracle(sc)
    if len(sc.mesh_objects):
        o.build_own_blas()
    acc = np.zeros((sc.height, sc.width, 4), np.float32)
    for i in range(frames):
        ox, oy, sd = scenes.frame_uniforms(i)
        o.set_frame((ox, oy), sd)
        img = o.render(mode=1 if len(sc.mesh_objects) else 0, threads=8)
        acc = pyoracle.accumulate(img, acc, i)
    bad = int((got_t.view(np.uint32) != img.view(np.uint32)).any(axis=2).sum())
    assert bad == 0, f"{sc.name}: {bad} pixels differ ({len(sc.spheres)} spheres, {len(sc.mesh_objects)} meshes, {sc.n_triangles} triangles, builder {builder})"
    assert np.array_equal(got_c.view(np.uint32), acc.view(np.uint32))
