# PCIe-inclusive note of DESIGN.md: time of reading a 1080p RGBA32F frame back per frame (urt_texture_get_pixels, pageable host memory)
import sys, time
sys.path.insert(0, '.')
from unityraytracer_amd import Context, RayTraceMaster, scenes
ctx = Context(0)
sc = scenes.config3()
m = RayTraceMaster(ctx, sc)
for _ in range(5): m.OnRenderImage()
ctx.synchronize()
n = 50
t = time.perf_counter()
for _ in range(n): m.OnRenderImage()
ctx.synchronize(); t_trace = (time.perf_counter() - t) / n
t = time.perf_counter()
for _ in range(n):
    m.OnRenderImage(); img = m._converged.GetPixels()
t_rb = (time.perf_counter() - t) / n
ctx.reset_counters(); m.OnRenderImage(); ctx.synchronize(); rays = ctx.counters()["rays"]
print(f"frame without readback {t_trace*1e3:.3f} ms ({rays/t_trace/1e6:.0f} Mrays/s); with a full-frame readback every frame {t_rb*1e3:.3f} ms ({rays/t_rb/1e6:.0f} Mrays/s); readback alone {1e3*(t_rb-t_trace):.3f} ms for {img.nbytes/1e6:.1f} MB")
