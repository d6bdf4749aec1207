import sys, time, itertools
sys.path.insert(0, '.')
import numpy as np
from unityraytracer_amd import Context, RayTraceMaster, scenes
ctx = Context(0)
for (w, h) in ((1920, 1080), (3840, 2160)):
    sc = scenes.config3(w, h)
    def run(label, **opts):
        for k, v in opts.items(): ctx.set_option(k, v)
        ctx.set_option("count_stats", 0); ctx.set_option("time_dispatch", 1)
        m = RayTraceMaster(ctx, sc)
        for _ in range(3): m.OnRenderImage()
        ctx.synchronize(); ctx.reset_counters()
        for _ in range(8): m.OnRenderImage()
        c = ctx.counters()
        print(f"{w}x{h} {label}: trace {c['trace_ms']/8:7.3f} ms  {c['rays']/c['trace_ms']/1e3:8.1f} Mrays/s", flush=True)
        m.OnDisable()
    run("mega(0) b64", kernel_mode=0, block_threads=64)
    for wpc, rf in itertools.product((12, 16, 20), (8, 16, 32)):
        run(f"persist(2) waves/cu {wpc} refill {rf}", kernel_mode=2, block_threads=64, waves_per_cu=wpc, refill_min=rf)
    for wpc, bmin, bexit, rf in itertools.product((12, 16), (8, 16, 24, 32, 48), (1, 8), (8, 16, 32)):
        if bexit > bmin: continue
        run(f"sched(3) waves/cu {wpc} blas_min {bmin:2d} blas_exit {bexit:2d} refill {rf:2d}", kernel_mode=3, block_threads=64, waves_per_cu=wpc, blas_min=bmin, blas_exit=bexit, refill_min=rf)
