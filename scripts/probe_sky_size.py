# Probe: how much of the frame is the sky texture's footprint in L2?  The same scene with the 2048x1024 sky (33.5 MB, the reference's import size),
# a 256x128 one (0.5 MB) and a 16x8 one: same rays and traversal (the sky only colours the misses — later bounces see other energies, not other geometry).
import sys, time
sys.path.insert(0, '.')
from unityraytracer_amd import Context, RayTraceMaster, scenes
ctx = Context(0)
ctx.set_option("time_dispatch", 1)
for cfg in sys.argv[1:] or ["C3", "C3D", "C5"]:
    for sw, sh in ((2048, 1024), (256, 128), (16, 8), (2048, 1024)):
        sc = scenes.CONFIGS[cfg](sky=scenes.make_sky(sw, sh))
        m = RayTraceMaster(ctx, sc)
        for _ in range(8): m.OnRenderImage()
        ctx.synchronize(); ctx.reset_counters()
        n = 128
        t = time.perf_counter()
        for _ in range(n): m.OnRenderImage()
        ctx.synchronize(); dt = (time.perf_counter() - t) / n
        c = ctx.counters()
        print(f"{cfg} sky {sw}x{sh}: {dt*1e3:.4f} ms per frame, kernel {c['trace_ms']/n:.4f} ms per frame, {c['rays']/n:.0f} rays, sky lookups {c['hit_sky']/n:.0f}", flush=True)
        m.OnDisable()
