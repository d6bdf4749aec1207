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
wpcs = [int(a) for a in sys.argv[1].split(",")]
for name in sys.argv[2:]:
    sc = scenes.config3(3840, 2160) if name == "C3@4K" else scenes.CONFIGS[name]()
    for wpc in wpcs:
        for rf, bmin, bex in ((32, 32, 4), (32, 24, 8), (32, 28, 6), (24, 32, 4), (40, 32, 4), (32, 40, 4)):
            ms, wd = run(sc, {"kernel_mode": 3, "waves_per_cu": wpc, "refill_min": rf, "blas_min": bmin, "blas_exit": bex}, frames=4 if name in ("C4", "C5") else 8)
            print(f"{name:6s} wpc {wpc} refill {rf} blas_min {bmin} exit {bex}: {ms:8.3f} ms wd {wd}", flush=True)
