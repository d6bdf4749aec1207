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
    print(label, 'mode', mode, 'diff pixels', int(d.sum()), 'rays', c['rays'], oc['rays'], flush=True)
for nb in (1, 2, 3):
    sc = scenes.mixed_test_scene(200, 120); sc.num_bounces = nb
    run(sc, f'mixed b={nb}')
sc = scenes.config1(); run(sc, 'C1 (spheres only, b=1)')
sc = scenes.config1(); sc.num_bounces = 4; run(sc, 'C1 b=4')
sc = scenes.mixed_test_scene(200, 120); sc.spheres = sc.spheres[:0]; sc.sphere_bvh = sc.sphere_bvh[:0]; run(sc, 'mixed meshes only b=4')
sc = scenes.mixed_test_scene(200, 120); sc.mesh_objects = sc.mesh_objects[:0]; sc.mesh_bvh = sc.mesh_bvh[:0]; run(sc, 'mixed spheres only b=4')
