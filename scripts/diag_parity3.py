import sys
sys.path.insert(0, '.')
import numpy as np
from oracle import pyoracle
from unityraytracer_amd import Context, RayTraceMaster, debug_build_blas, scenes
ctx = Context(0)
def run(sc, label, mode=0):
    o = pyoracle.Oracle(sc)
    if len(sc.mesh_objects):
        nodes, tri, root, _, _ = debug_build_blas(sc.mesh_objects, sc.vertices, sc.indices)
        o.set_blas(nodes, tri, root)
    ref, oc = o.render(mode=1, threads=8, counters=True)
    ctx.set_option('kernel_mode', mode); ctx.set_option('count_stats', 1); ctx.reset_counters()
    m = RayTraceMaster(ctx, sc); m.OnRenderImage(); gpu = m._target.GetPixels(); c = ctx.counters(); m.OnDisable()
    d = (gpu.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
    ys, xs = np.nonzero(d)
    print(label, 'mode', mode, 'diff pixels', int(d.sum()), 'rays', c['rays'], oc['rays'], 'hit_tri', c['hit_tri'], oc['hit_tri'], 'tri_tests', c['tri_tests'], oc['tri_tests'], 'blas_nodes', c['blas_nodes'], oc['blas_nodes'], flush=True)
    print('    first diffs', [(int(x), int(y), gpu[y, x, :3].tolist(), ref[y, x, :3].tolist()) for y, x in list(zip(ys, xs))[:4]], flush=True)
for nb in (1, 4):
    sc = scenes.mixed_test_scene(200, 120); sc.num_bounces = nb
    run(sc, f'mixed b={nb}')
