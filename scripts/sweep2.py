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
run("mega b64", kernel_mode=0, block_threads=64)
for bt, wpc, rf in itertools.product((64, 256), (8, 12, 16, 20, 24, 32), (1, 8, 16, 32, 48, 64)):
    run(f"persist block {bt:3d} waves/cu {wpc:2d} refill_min {rf:2d}", kernel_mode=2, block_threads=bt, waves_per_cu=wpc, refill_min=rf)
