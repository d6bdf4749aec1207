import sys
sys.path.insert(0, '.')
import numpy as np
from oracle import pyoracle
from unityraytracer_amd import Context, RayTraceMaster, debug_build_blas, scenes
modes = [int(a) for a in sys.argv[1:]] or [1, 0]
ctx = Context(0)
sc = scenes.mixed_test_scene(200, 120)
o = pyoracle.Oracle(sc)
nodes, tri, root, _, _ = debug_build_blas(sc.mesh_objects, sc.vertices, sc.indices)
o.set_blas(nodes, tri, root)
ref, oc = o.render(mode=1, threads=8, counters=True)
print('oracle', oc, flush=True)
for mode in modes:
    ctx.set_option('kernel_mode', mode); ctx.set_option('count_stats', 1); ctx.reset_counters()
    m = RayTraceMaster(ctx, sc); m.OnRenderImage(); gpu = m._target.GetPixels(); c = ctx.counters(); m.OnDisable()
    d = (gpu.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
    print('mode', mode, 'diff pixels', int(d.sum()), {k: c[k] for k in oc}, flush=True)
    ys, xs = np.nonzero(d)
    for y, x in list(zip(ys, xs))[:6]:
        print('   px', x, y, 'gpu', gpu[y, x, :3], 'ref', ref[y, x, :3], flush=True)
