# Scene-preparation time (SetData -> ready device scene, host wall clock reported by the library) for the reference's dynamic-scene
# protocol (RM:215-230: any move re-uploads EVERY buffer): nothing moved / one MeshObject moved / all moved, with the host SAH builder
# (per-MeshObject BVH cache) and with the three GPU builders (1 Karras radix tree, 2 depth-budgeted radix tree, 3 binned SAH).   python scripts/dynamic_scene.py [C4 C5]
import sys, time
sys.path.insert(0, '.')
import numpy as np
from unityraytracer_amd import Context, RayTraceMaster, scenes
ctx = Context(0)
for name in (sys.argv[1:] or ["C4", "C5"]):
    sc = scenes.CONFIGS[name](640, 360)
    for builder in (0, 1, 2, 3):
        ctx.set_option("blas_builder", builder)
        m = RayTraceMaster(ctx, sc)
        m.OnRenderImage(); first = ctx.scene_info()["prepare_ms"]
        tag = []
        def reupload(mo):
            lo, hi = scenes.mesh_bounds(mo, sc.vertices, sc.indices)
            bvh = scenes.build_object_bvh(lo, hi)
            p0 = ctx.refit_stats()[1]
            t0 = time.perf_counter()
            for buf, data in ((m._meshObjectBuffer, mo), (m._vertexBuffer, sc.vertices), (m._indexBuffer, sc.indices), (m._normalBuffer, sc.normals), (m._meshObjectBVHBuffer, bvh)):
                buf.SetData(data)
            t_set = (time.perf_counter() - t0) * 1e3
            ms = ctx.scene_info()["prepare_ms"]
            ctx.synchronize()                                                     # the refit kernels are stream-ordered: include them in the wall time
            wall = (time.perf_counter() - t0) * 1e3
            tag.append(f"SetData x5 {t_set:.1f} ms, SetData-to-GPU-done {wall:.1f} ms" + (", in place" if ctx.refit_stats()[1] > p0 else ""))
            return ms
        same = min(reupload(sc.mesh_objects) for _ in range(3))
        one = sc.mesh_objects.copy()
        mat = np.asarray(one[-1]["localToWorldMatrix"], np.float32).copy(); mat[12] += 0.1; one[-1]["localToWorldMatrix"] = mat
        r0, b0 = ctx.blas_cache_stats()
        t_one = reupload(one)
        r1, b1 = ctx.blas_cache_stats()
        allm = sc.mesh_objects.copy()
        for k in range(len(allm)):
            mat = np.asarray(allm[k]["localToWorldMatrix"], np.float32).copy(); mat[12] += 0.05 * (k + 1); allm[k]["localToWorldMatrix"] = mat
        t_all = reupload(allm)
        print(f"{name} {sc.n_triangles} triangles, {len(sc.mesh_objects)} MeshObjects, builder {('host SAH', 'GPU Karras tree', 'GPU depth-budgeted tree', 'GPU binned SAH')[builder]}: first {first:.1f} ms; "
              f"re-upload unchanged: not stale ({tag[0]}); one moved {t_one:.2f} ms [{tag[3]}]" + (f" ({b1 - b0} built, {r1 - r0} reused)" if not builder else "") + f"; all moved {t_all:.2f} ms [{tag[4]}]", flush=True)
        m.OnDisable()
ctx.set_option("blas_builder", -1)
