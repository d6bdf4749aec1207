import sys, time
sys.path.insert(0, '.')
import numpy as np
from unityraytracer_amd import Context, RayTraceMaster, scenes
ctx = Context(0)
for name, frames in (("C5", 1024), ("C4", 100), ("C3", 1000), ("C2", 1000)):   # C5: BASELINE config 5 is a 1024-spp progressive accumulation
    sc = scenes.CONFIGS[name]()
    ctx.set_option("count_stats", 0); ctx.set_option("time_dispatch", 1); ctx.reset_counters()
    m = RayTraceMaster(ctx, sc)
    t = time.perf_counter()
    for _ in range(frames): m.OnRenderImage()
    ctx.synchronize(); dt = time.perf_counter() - t
    c = ctx.counters()
    img = m._converged.GetPixels()
    print(f"{name}: {frames} frames in {dt:.2f} s wall ({dt/frames*1e3:.3f} ms/frame incl. host), {c['launches']} launches, trace {c['trace_ms']/frames:.3f} ms/frame, {c['rays']/c['trace_ms']/1e3:.0f} Mrays/s, watchdog {c['watchdog_trips']}, finite {bool(np.isfinite(img).all())}, mean {img[...,:3].mean():.5f}", flush=True)
    m.OnDisable()
