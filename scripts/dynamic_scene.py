# time from SetData to the first frame when ONE MeshObject of a big scene moves (per-MeshObject BVH cache)
import sys, time
sys.path.insert(0, '.')
import numpy as np
from unityraytracer_amd import Context, RayTraceMaster, scenes
ctx = Context(0)
for name in (sys.argv[1:] or ["C4", "C5"]):
    sc = scenes.CONFIGS[name]()
    m = RayTraceMaster(ctx, sc)
    t = time.perf_counter(); m.OnRenderImage(); ctx.synchronize(); t_first = time.perf_counter() - t
    t = time.perf_counter(); m.OnRenderImage(); ctx.synchronize(); t_steady = time.perf_counter() - t
    mo = sc.mesh_objects.copy()
    k = len(mo) - 1
    mat = np.asarray(mo[k]["localToWorldMatrix"], np.float32).copy(); mat[12] += 0.1; mo[k]["localToWorldMatrix"] = mat
    lo, hi = scenes.mesh_bounds(mo, sc.vertices, sc.indices)
    bvh = scenes.build_object_bvh(lo, hi)
    r0, b0 = ctx.blas_cache_stats()
    t = time.perf_counter()
    for buf, data in ((m._meshObjectBuffer, mo), (m._vertexBuffer, sc.vertices), (m._indexBuffer, sc.indices), (m._normalBuffer, sc.normals), (m._meshObjectBVHBuffer, bvh)):
        buf.SetData(data)
    m.OnRenderImage(); ctx.synchronize(); t_moved = time.perf_counter() - t
    r1, b1 = ctx.blas_cache_stats()
    print(f"{name}: first frame (all {len(mo)} MeshObjects built) {t_first*1e3:.1f} ms; steady frame {t_steady*1e3:.2f} ms; "
          f"frame after moving one MeshObject and re-uploading every buffer {t_moved*1e3:.1f} ms ({b1-b0} built, {r1-r0} reused)", flush=True)
    m.OnDisable()
