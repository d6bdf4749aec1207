import sys, time, itertools
sys.path.insert(0, '.')
import numpy as np
from unityraytracer_amd import Context, RayTraceMaster, scenes
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
ctx = Context(0)
sc = scenes.CONFIGS[cfg]()
ref = None
def run(label, **opts):
    global ref
    for k, v in opts.items(): ctx.set_option(k, v)
    ctx.set_option("count_stats", 0); ctx.set_option("time_dispatch", 1)
    m = RayTraceMaster(ctx, sc)
    m.OnRenderImage(); ctx.synchronize()
    img = m._target.GetPixels()
    if ref is None: ref = img
    same = np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    for _ in range(3): m.OnRenderImage()
    ctx.synchronize(); ctx.reset_counters()
    for _ in range(10): m.OnRenderImage()
    c = ctx.counters()
    print(f"{cfg} {label}: trace {c['trace_ms']/10:7.3f} ms  {c['rays']/c['trace_ms']/1e3:8.1f} Mrays/s  same_pixels={same}", flush=True)
    m.OnDisable()
run("persist(2) b64", kernel_mode=2, block_threads=64, waves_per_cu=16, refill_min=16)
for bmin, bexit, rf in itertools.product((16, 32, 40, 48, 56, 64), (1, 8, 16, 24), (8, 16, 32)):
    if bexit > bmin: continue
    run(f"sched(3) blas_min {bmin:2d} blas_exit {bexit:2d} refill_min {rf:2d}", kernel_mode=3, block_threads=64, waves_per_cu=16, blas_min=bmin, blas_exit=bexit, refill_min=rf)
