import sys, time, itertools
sys.path.insert(0, '.')
import numpy as np
from unityraytracer_amd import Context, RayTraceMaster, scenes
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
ctx = Context(0)
sc = scenes.CONFIGS[cfg]()
ref = None
for mode, bt, run in itertools.product((0, 1), (64, 128, 256), (1, 4, 16, 64, 1024)):
    ctx.set_option("kernel_mode", mode); ctx.set_option("count_stats", 0); ctx.set_option("time_dispatch", 1)
    ctx.set_option("block_threads", bt); ctx.set_option("xcd_run", run)
    m = RayTraceMaster(ctx, sc)
    m.OnRenderImage(); ctx.synchronize()
    img = m._target.GetPixels()
    if ref is None: ref = img
    same = np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    for _ in range(3): m.OnRenderImage()
    ctx.synchronize(); ctx.reset_counters()
    t = time.time()
    for _ in range(10): m.OnRenderImage()
    ctx.synchronize(); dt = (time.time() - t) / 10
    c = ctx.counters()
    print(f"{cfg} mode {mode} block {bt:3d} xcd_run {run:4d}: trace {c['trace_ms']/10:7.3f} ms  wall {dt*1e3:7.3f} ms  {c['rays']/c['trace_ms']/1e3:8.1f} Mrays/s  same_pixels={same}", flush=True)
    m.OnDisable()
