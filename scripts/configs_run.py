import sys, time
sys.path.insert(0, '.')
import numpy as np
from unityraytracer_amd import Context, RayTraceMaster, scenes
ctx = Context(0)
for cfg in sys.argv[1:] or ["C2", "C3", "C4", "C5"]:
    t = time.time(); sc = scenes.CONFIGS[cfg](); tb = time.time() - t
    ctx.set_option("count_stats", 0); ctx.set_option("time_dispatch", 1)
    m = RayTraceMaster(ctx, sc)
    t = time.time(); m.OnRenderImage(); ctx.synchronize(); first = time.time() - t
    for _ in range(2): m.OnRenderImage()
    ctx.synchronize(); ctx.reset_counters()
    n = 5
    for _ in range(n): m.OnRenderImage()
    c = ctx.counters()
    print(f"{cfg} {sc.name} {sc.width}x{sc.height} b={sc.num_bounces}: scene gen {tb:.1f}s, first frame (upload+BVH build) {first:.2f}s, trace {c['trace_ms']/n:8.3f} ms, rays/frame {c['rays']//n}, {c['rays']/c['trace_ms']/1e3:8.1f} Mrays/s, wd {c['watchdog_trips']}", flush=True)
    ctx.set_option("count_stats", 1); ctx.reset_counters(); m.OnRenderImage(); cc = ctx.counters()
    print("    per-ray: blas_nodes %.1f tri_tests %.1f tlas_nodes %.1f sphere_tests %.1f" % tuple(cc[k] / cc['rays'] for k in ('blas_nodes', 'tri_tests', 'tlas_nodes', 'sphere_tests')), flush=True)
    m.OnDisable()
