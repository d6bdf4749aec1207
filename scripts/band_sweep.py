# tile-run length of the work shards (xcd_run) x kernel mode, C3 and others
import sys
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
for cfg in (sys.argv[1:] or ["C3"]):
    sc = scenes.CONFIGS[cfg]()
    for mode, extra in ((3, {"waves_per_cu": 16, "refill_min": 32, "blas_min": 24, "blas_exit": 4}), (3, {"waves_per_cu": 0, "refill_min": 48, "blas_min": 16, "blas_exit": 8}), (4, {"pool_k": 1, "waves_per_cu": 0, "pool_refill": 32, "pool_blas_min": 64, "pool_inloop": 8})):
        for G in (1, 4, 16, 60, 240, 480, 2000):
            o = {"kernel_mode": mode, "xcd_run": G}; o.update(extra)
            ms, wd = run(sc, o)
            print(f"{cfg} mode {mode} {extra.get('waves_per_cu')} run {G}: {ms:7.3f} ms wd {wd}", flush=True)
