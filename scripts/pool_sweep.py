# kernel_mode 4 (path pool) vs 3 on the BASELINE configurations, plus a knob sweep
import sys, itertools
sys.path.insert(0, '.')
from unityraytracer_amd import Context, RayTraceMaster, scenes
ctx = Context(0)
def run(sc, opts, frames=8):
    for k, v in opts.items(): ctx.set_option(k, v)
    ctx.set_option("count_stats", 0); ctx.set_option("time_dispatch", 1)
    m = RayTraceMaster(ctx, sc)
    for _ in range(3): m.OnRenderImage()
    ctx.synchronize(); ctx.reset_counters()
    for _ in range(frames): m.OnRenderImage()
    c = ctx.counters(); m.OnDisable()
    return c['trace_ms'] / frames, c['watchdog_trips']
cfgs = sys.argv[1:] or ["C3"]
for cfg in cfgs:
    sc = scenes.CONFIGS[cfg]()
    ms, wd = run(sc, {"kernel_mode": 3, "waves_per_cu": 0})
    print(f"{cfg} mode 3 default: {ms:7.3f} ms wd {wd}", flush=True)
    for k, rf, bmin, omin, bex, inl in itertools.product((1, 2, 3), (32, 64), (48, 64, 96), (16, 32), (8, 16), (16,)):
        if bmin > 64 * k: continue
        ms, wd = run(sc, {"kernel_mode": 4, "pool_k": k, "waves_per_cu": 0, "pool_refill": rf, "pool_blas_min": bmin, "pool_other_min": omin, "pool_blas_exit": bex, "pool_inloop": inl})
        print(f"{cfg} mode 4 k {k} refill {rf} blas_min {bmin} other_min {omin} exit {bex} inloop {inl}: {ms:7.3f} ms wd {wd}", flush=True)
