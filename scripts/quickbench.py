import sys, time, json
sys.path.insert(0, '.')
import numpy as np
from unityraytracer_amd import Context, RayTraceMaster, scenes
ctx = Context(0)
for cfg in ("C3", "C2"):
    sc = scenes.CONFIGS[cfg]()
    for mode in (1, 0):
        ctx.set_option("kernel_mode", mode); ctx.set_option("count_stats", 0); ctx.set_option("time_dispatch", 1)
        m = RayTraceMaster(ctx, sc)
        t=time.time(); m.OnRenderImage(); ctx.synchronize(); first=time.time()-t
        ctx.reset_counters()
        t=time.time()
        for _ in range(10): m.OnRenderImage()
        ctx.synchronize(); dt=(time.time()-t)/10
        c = ctx.counters()
        print(cfg, 'mode', mode, 'first %.3fs'%first, 'frame %.3f ms'%(dt*1e3), 'trace_ms/frame %.3f'%(c['trace_ms']/10), 'rays/frame', c['rays']//10, 'Mrays/s %.1f'%(c['rays']/c['trace_ms']/1e3), flush=True)
        ctx.set_option("count_stats", 1); ctx.reset_counters(); m.OnRenderImage(); print('   counters', ctx.counters(), flush=True)
        m.OnDisable()
