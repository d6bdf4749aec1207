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
for rep in range(2):
    for name in (sys.argv[1:] or ["C3", "C3@4K", "C4", "C5", "C2"]):
        sc = scenes.config3(3840, 2160) if name == "C3@4K" else scenes.CONFIGS[name]()
        for order in (1, 8, 16, 24, 32, 48):
            ms, wd = run(sc, {"kernel_mode": 3, "shade_min": order}, frames=4 if name in ("C4", "C5") else 8)
            print(f"{name:6s} shade_min {order:2d}: {ms:8.3f} ms wd {wd}", flush=True)
