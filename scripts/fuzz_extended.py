"""Extended randomised parity run (GPU vs the oracle, bit for bit): the generator of tests/test_gpu_fuzz.py over many more seeds, every
kernel mode and all three triangle-BVH builders.  Even scenes: the oracle walks its OWN BVH (with the product's object-level cull
restated); odd scenes: the oracle's LITERAL brute force (every triangle of every object the reference would intersect, no cull at
all) — so the product's cull and BVH are checked against the reference's semantics directly.
usage: python scripts/fuzz_extended.py [first_seed] [count]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from oracle import pyoracle
from unityraytracer_amd import Context, RayTraceMaster, scenes
from test_gpu_fuzz import random_scene

first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ctx = Context(0)
bad_total, t0 = 0, time.time()
pixels = tris = with_mesh = 0
lum = 0.0
for k in range(count):
    seed = first + k
    sc, builder, frames = random_scene(seed)
    mode = (3, 3, 3, 0, 2, 4, 5, 1)[k % 8]
    if builder == 1: builder = (1, 2, 3)[(k // 8) % 3]        # the three GPU builders take turns (Karras tree, depth-budgeted tree, binned SAH)
    ctx.set_option("kernel_mode", mode); ctx.set_option("blas_builder", builder)
    m = RayTraceMaster(ctx, sc)
    for _ in range(frames): m.OnRenderImage()
    got_t, got_c = m._target.GetPixels(), m._converged.GetPixels()
    wd = ctx.counters()["watchdog_trips"]
    m.OnDisable()
    o = pyoracle.Oracle(sc)
    omode = 0
    if len(sc.mesh_objects):
        nodes, _, _ = o.build_own_blas()
        omode = 1 if len(nodes) and k % 2 == 0 else 0       # odd scenes (and scenes of single-leaf MeshObjects only): the literal brute force
    acc = np.zeros((sc.height, sc.width, 4), np.float32)
    for i in range(frames):
        ox, oy, sd = scenes.frame_uniforms(i)
        o.set_frame((ox, oy), sd)
        img = o.render(mode=omode, threads=8)
        acc = pyoracle.accumulate(img, acc, i)
    bad = int((got_t.view(np.uint32) != img.view(np.uint32)).any(axis=2).sum()) + int((got_c.view(np.uint32) != acc.view(np.uint32)).any(axis=2).sum())
    pixels += sc.width * sc.height * frames; tris += sc.n_triangles; with_mesh += 1 if len(sc.mesh_objects) else 0
    lum += float(np.nan_to_num(acc[..., :3]).mean())
    if bad or wd:
        bad_total += 1
        print(f"seed {seed} mode {mode} builder {builder}: {bad} pixels differ, watchdog {wd} ({len(sc.spheres)} spheres, {len(sc.mesh_objects)} meshes, {sc.n_triangles} triangles)", flush=True)
    if (k + 1) % 25 == 0:
        print(f"{k + 1} scenes, {bad_total} failing, {time.time() - t0:.0f} s", flush=True)
ctx.set_option("blas_builder", -1); ctx.set_option("kernel_mode", 3)
print(f"done: {count} scenes from seed {first}, {bad_total} failing; {pixels} pixel-frames compared bit for bit, {with_mesh} scenes with MeshObjects ({tris} triangles in all), "
      f"mean image level {lum / count:.3f}, {time.time() - t0:.1f} s")
sys.exit(1 if bad_total else 0)
