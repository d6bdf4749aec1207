# A/B of ONE 0/1 option in one process: python scripts/onoff.py OPTION [CFG ...]
import sys
sys.path.insert(0, '.')
from unityraytracer_amd import Context, RayTraceMaster, scenes
ctx = Context(0)
opt = sys.argv[1]
def run(sc, opts, frames=8):
    for k, v in opts.items(): ctx.set_option(k, v)
    ctx.set_option("count_stats", 0); ctx.set_option("time_dispatch", 1)
    m = RayTraceMaster(ctx, sc)
    for _ in range(3): m.OnRenderImage()
    ctx.synchronize(); ctx.reset_counters()
    for _ in range(frames): m.OnRenderImage()
    c = ctx.counters(); m.OnDisable()
    return c['trace_ms'] / frames
for rep in range(3):
    for name in (sys.argv[2:] or ["C3", "C3@4K", "C2", "C4", "C5"]):
        sc = scenes.config3(3840, 2160) if name == "C3@4K" else scenes.CONFIGS[name]()
        for v in (0, 1):
            print(f"{name:6s} {opt} {v}: {run(sc, {'kernel_mode': 3, opt: v}, frames=4 if name in ('C4', 'C5') else 8):8.3f} ms", flush=True)
