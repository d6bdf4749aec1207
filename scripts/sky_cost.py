"""What the 32 MB RGBA32F sky costs the trace kernel: the BASELINE configs with the 2048x1024 sky and with the same sky at 256x128
(fits any cache).  Usage: python scripts/sky_cost.py [C3 C4 C5 ...]"""
import sys, time
sys.path.insert(0, '.')
from unityraytracer_amd import Context, RayTraceMaster, scenes

ctx = Context(0)
for name in (sys.argv[1:] or ["C3"]):
    for (w, h) in ((2048, 1024), (1024, 512), (256, 128)):
        sc = scenes.CONFIGS[name]()
        sc.sky = scenes.make_sky(w, h)
        m = RayTraceMaster(ctx, sc)
        for _ in range(3): m.OnRenderImage()
        ctx.synchronize(); ctx.set_option("time_dispatch", 1); ctx.reset_counters()
        t0 = time.perf_counter()
        for _ in range(32): m.OnRenderImage()
        ctx.synchronize()
        wall = (time.perf_counter() - t0) / 32 * 1e3
        c = ctx.counters()
        m.OnDisable()
        print(f"{name} sky {w}x{h}: kernel {c['trace_ms'] / 32:.3f} ms/frame, wall {wall:.3f}", flush=True)
