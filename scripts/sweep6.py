import sys, time, itertools
sys.path.insert(0, '.')
from unityraytracer_amd import Context, RayTraceMaster, scenes
ctx = Context(0)
for cfg in ("C3", "C5", "C2"):
    sc = scenes.CONFIGS[cfg]()
    for wpc, rf, bmin in itertools.product((12, 16, 20, 24), (32, 48), (16, 24)):
        ctx.set_option("kernel_mode", 3); ctx.set_option("waves_per_cu", wpc); ctx.set_option("refill_min", rf); ctx.set_option("blas_min", bmin)
        ctx.set_option("count_stats", 0); ctx.set_option("time_dispatch", 1)
        m = RayTraceMaster(ctx, sc)
        for _ in range(3): m.OnRenderImage()
        ctx.synchronize(); ctx.reset_counters()
        for _ in range(8): m.OnRenderImage()
        c = ctx.counters()
        print(f"{cfg} waves/cu {wpc} refill {rf} blas_min {bmin}: trace {c['trace_ms']/8:7.3f} ms {c['rays']/c['trace_ms']/1e3:8.1f} Mrays/s wd {c['watchdog_trips']}", flush=True)
        m.OnDisable()
