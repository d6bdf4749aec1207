import sys
sys.path.insert(0, '.')
from unityraytracer_amd import Context, RayTraceMaster, scenes
ctx = Context(0)
def run(sc, opts, frames=6):
    for k, v in opts.items(): ctx.set_option(k, v)
    ctx.set_option("count_stats", 0); ctx.set_option("time_dispatch", 1)
    m = RayTraceMaster(ctx, sc)
    for _ in range(2): m.OnRenderImage()
    ctx.synchronize(); ctx.reset_counters()
    for _ in range(frames): m.OnRenderImage()
    c = ctx.counters(); m.OnDisable()
    return c['trace_ms'] / frames
for name in ("C3@4K", "C5", "C2"):
    sc = scenes.config3(3840, 2160) if name == "C3@4K" else scenes.CONFIGS[name]()
    print(f"{name:6s} mode 3: {run(sc, {'kernel_mode': 3}):8.3f} ms", flush=True)
    for k, bmin, bex in ((1, 64, 8), (2, 64, 8), (2, 64, 32), (2, 96, 32), (3, 96, 32), (3, 128, 48)):
        ms = run(sc, {"kernel_mode": 4, "pool_k": k, "pool_blas_min": bmin, "pool_blas_exit": bex, "pool_refill": 64 if k > 1 else 32})
        print(f"{name:6s} mode 4 k {k} blas_min {bmin} exit {bex}: {ms:8.3f} ms", flush=True)
