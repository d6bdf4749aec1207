import sys
sys.path.insert(0, '.')
import numpy as np
from unityraytracer_amd import Context, RayTraceMaster, scenes
ctx = Context(0)
sc = scenes.mixed_test_scene(200, 120); sc.num_bounces = 1
out = {}
for mode in (1, 0):
    ctx.set_option('kernel_mode', mode); ctx.set_option('count_stats', 0)
    m = RayTraceMaster(ctx, sc); m.OnRenderImage(); out[mode] = m._target.GetPixels(); m.OnDisable()
a, b = out[1], out[0]
print('mode1 (wavefront) debug at (92,73):', a[73, 92], ' mode0 (mega):', b[73, 92])
d = (a[..., :3] != b[..., :3]).any(axis=2)
print('pixels where (kind,id,mesh) differ:', int(d.sum()))
ys, xs = np.nonzero(d)
for y, x in list(zip(ys, xs))[:8]:
    print('  ', x, y, 'wf', a[y, x], 'mega', b[y, x])
k3 = a[..., 0] == 3
print('tri pixels', int(k3.sum()), 'wf mesh ids', np.unique(a[k3][:, 2]), 'mega mesh ids', np.unique(b[b[..., 0] == 3][:, 2]))
