# benefit of a larger LDS top at EQUAL occupancy (12 waves/CU leaves room for 256 nodes): is a smaller stack footprint worth it?
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
    return c['trace_ms'] / frames
for rep in range(2):
    for name in ("C3", "C3@4K", "C5"):
        sc = scenes.config3(3840, 2160) if name == "C3@4K" else scenes.CONFIGS[name]()
        for wpc in (12,):
            for t in (0, 16, 64, 128, 256):
                ms = run(sc, {"kernel_mode": 3, "waves_per_cu": wpc, "top_nodes": t}, frames=4 if name == "C5" else 8)
                print(f"{name:6s} waves/cu {wpc} top_nodes {t:3d}: {ms:8.3f} ms", flush=True)
