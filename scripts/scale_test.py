import sys, time
sys.path.insert(0, '.')
import numpy as np
from unityraytracer_amd import Context, RayTraceMaster, scenes
ctx = Context(0)
for (w, h) in ((960, 540), (1920, 1080), (3840, 2160)):
    for b in (1, 2, 8):
        sc = scenes.config3(w, h); sc.num_bounces = b
        for mode in (2, 3):
            ctx.set_option("kernel_mode", mode); ctx.set_option("count_stats", 0); ctx.set_option("time_dispatch", 1)
            m = RayTraceMaster(ctx, sc)
            for _ in range(3): m.OnRenderImage()
            ctx.synchronize(); ctx.reset_counters()
            for _ in range(8): m.OnRenderImage()
            c = ctx.counters()
            print(f"{w}x{h} bounces {b} mode {mode}: trace {c['trace_ms']/8:7.3f} ms  rays/frame {c['rays']//8}  {c['rays']/c['trace_ms']/1e3:8.1f} Mrays/s watchdog {c['watchdog_trips']}", flush=True)
            m.OnDisable()
