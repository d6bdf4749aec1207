"""Quality of the triangle BVHs of the three builders (0 host SAH, 1 GPU Morton tree, 2 GPU Morton leaves + PLOC upper tree): surface-area
cost of the tree (interior: sum of A(node) / A(root); leaves: sum of A(leaf) x triangles / A(root)), depth, and what one frame really
visits (count_stats: triangle-BVH nodes and triangle tests per ray).   python scripts/bvh_quality.py [C3 C3D C4 C5]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from unityraytracer_amd import Context, RayTraceMaster, scenes

def area(lo, hi):
    d = np.maximum(hi - lo, 0)
    return d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0]

ctx = Context(0)
for cfg in sys.argv[1:] or ["C3", "C3D", "C4", "C5"]:
    sc = scenes.CONFIGS[cfg]()
    for builder in (0, 1, 2, 3):
        ctx.set_option("blas_builder", builder)
        ctx.set_option("count_stats", 1)
        m = RayTraceMaster(ctx, sc)
        ctx.reset_counters()
        m.OnRenderImage()
        c = ctx.counters()
        nodes, tri, root, info = ctx.read_scene_blas(len(sc.mesh_objects))
        m.OnDisable()
        ctx.set_option("count_stats", 0)
        lo0, hi0, lo1, hi1 = nodes[:, 0:3], nodes[:, 3:6], nodes[:, 6:9], nodes[:, 9:12]
        code = nodes[:, 12:14].view(np.int32)
        a0, a1 = area(lo0, hi0), area(lo1, hi1)
        # per tree: normalise by the root's area
        cost_i = cost_l = 0.0
        for r in root:
            if r < 0 or r == 0x7FFFFFFF:
                continue
            ra = area(np.minimum(lo0[r], lo1[r]), np.maximum(hi0[r], hi1[r]))
            # walk the tree
            stack = [int(r)]
            ci = cl = 0.0
            while stack:
                n = stack.pop()
                for k, a in ((0, a0[n]), (1, a1[n])):
                    ch = int(code[n, k])
                    if ch >= 0:
                        ci += a; stack.append(ch)
                    else:
                        cl += a * (((~ch) & 7) + 1)
            cost_i += 1.0 + ci / ra; cost_l += cl / ra
        print(f"{cfg:4s} builder {builder}: nodes {info['n_nodes']:7d} depth {info['max_depth']:3d}  SA cost interior {cost_i:8.2f} leaves {cost_l:8.2f}  |  per ray: "
              f"BVH nodes {c['blas_nodes'] / c['rays']:6.2f} triangle tests {c['tri_tests'] / c['rays']:5.2f}  prepare {info['prepare_ms']:.1f} ms", flush=True)
ctx.set_option("blas_builder", -1)
