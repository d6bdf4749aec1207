# knob sweep after the work counter was sharded (fetches are cheap now): refill_min x blas_min x blas_exit x waves/CU on C3
import sys, itertools
sys.path.insert(0, '.')
from unityraytracer_amd import Context, RayTraceMaster, scenes
ctx = Context(0)
cfgs = sys.argv[1:] or ["C3"]
for cfg in cfgs:
    sc = scenes.CONFIGS[cfg]()
    for wpc, rf, bmin, bex in itertools.product((10, 12, 16), (8, 16, 32, 48), (8, 16, 24), (4, 8)):
        if bex > bmin: continue
        for k, v in (("kernel_mode", 3), ("waves_per_cu", wpc), ("refill_min", rf), ("blas_min", bmin), ("blas_exit", bex), ("count_stats", 0), ("time_dispatch", 1)):
            ctx.set_option(k, v)
        m = RayTraceMaster(ctx, sc)
        for _ in range(3): m.OnRenderImage()
        ctx.synchronize(); ctx.reset_counters()
        for _ in range(8): m.OnRenderImage()
        c = ctx.counters()
        print(f"{cfg} waves/cu {wpc} refill {rf} blas_min {bmin} blas_exit {bex}: trace {c['trace_ms']/8:7.3f} ms wd {c['watchdog_trips']}", flush=True)
        m.OnDisable()
