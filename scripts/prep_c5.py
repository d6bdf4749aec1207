import sys
sys.path.insert(0, '.')
import numpy as np
from unityraytracer_amd import Context, RayTraceMaster, scenes
b = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sc = scenes.CONFIGS["C5"](640, 360)
ctx = Context(0)
ctx.set_option("blas_builder", b)
m = RayTraceMaster(ctx, sc)
m.OnRenderImage()
for k in range(2):
    v = np.ascontiguousarray(sc.vertices, np.float32).copy()
    v[0, 0] = np.nextafter(v[0, 0], np.float32(np.inf if k % 2 == 0 else -np.inf))
    m._vertexBuffer.SetData(v)
    print("prepare ms", ctx.scene_info()["prepare_ms"], flush=True)
m.OnDisable()
ctx.close()
