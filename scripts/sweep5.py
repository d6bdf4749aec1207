import sys, time, itertools
sys.path.insert(0, '.')
import numpy as np
from unityraytracer_amd import Context, RayTraceMaster, scenes
ctx = Context(0)
for (w, h) in ((1920, 1080), (3840, 2160)):
    sc = scenes.config3(w, h)
    ref = None
    def run(label, **opts):
        global ref
        for k, v in opts.items(): ctx.set_option(k, v)
        ctx.set_option("count_stats", 0); ctx.set_option("time_dispatch", 1)
        m = RayTraceMaster(ctx, sc)
        for _ in range(4): m.OnRenderImage()
        ctx.synchronize(); ctx.reset_counters()
        t = time.perf_counter()
        for _ in range(10): m.OnRenderImage()
        ctx.synchronize(); wall = (time.perf_counter() - t) / 10
        c = ctx.counters()
        print(f"{w}x{h} {label}: trace {c['trace_ms']/10:7.3f} ms wall {wall*1e3:7.3f} ms {c['rays']/c['trace_ms']/1e3:8.1f} Mrays/s wd {c['watchdog_trips']}", flush=True)
        m.OnDisable()
    for to, rf, bmin in itertools.product((0,), (32, 48, 56, 64), (8, 16, 24)):
        run(f"sched(3) tile_order {to} refill {rf} blas_min {bmin}", kernel_mode=3, refill_min=rf, blas_min=bmin, blas_exit=8, waves_per_cu=16)
