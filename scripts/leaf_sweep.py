import sys, time
sys.path.insert(0, '.')
from unityraytracer_amd import Context, RayTraceMaster, scenes
ctx = Context(0)
for cfg in ("C3", "C4", "C5", "C3"):
    sc = scenes.CONFIGS[cfg]()
    for leaf in (2, 3, 4, 5):
        ctx.set_option("blas_leaf_max", leaf); ctx.set_option("count_stats", 0); ctx.set_option("time_dispatch", 1)
        m = RayTraceMaster(ctx, sc)
        for _ in range(3): m.OnRenderImage()
        ctx.synchronize(); ctx.reset_counters()
        for _ in range(8): m.OnRenderImage()
        c = ctx.counters()
        ctx.set_option("count_stats", 1); ctx.reset_counters(); m.OnRenderImage(); cc = ctx.counters(); ctx.set_option("count_stats", 0)
        print(f"{cfg} leaf_max {leaf}: trace {c['trace_ms']/8:7.3f} ms {c['rays']/c['trace_ms']/1e3:8.1f} Mrays/s | nodes/ray {cc['blas_nodes']/cc['rays']:.2f} tris/ray {cc['tri_tests']/cc['rays']:.2f}", flush=True)
        m.OnDisable()
