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
# pipelined: frame i is read while frames i+1, i+2 render (urt_texture_read_begin / _end, two in flight)
tickets = []
t = time.perf_counter()
for _ in range(n):
    m.OnRenderImage()
    tickets.append(m._converged.ReadBegin())
    if len(tickets) > 2:
        img2 = m._converged.ReadEnd(tickets.pop(0), copy=False)
while tickets:
    img2 = m._converged.ReadEnd(tickets.pop(0), copy=False)
t_pipe = (time.perf_counter() - t) / n
# ... in the format of the host's destination, converted on the GPU (urt_texture_read_begin_format)
t_fmt = {}
for fmt in ("RGBA16F", "RGBA8_SRGB"):
    tickets = []
    t = time.perf_counter()
    for _ in range(n):
        m.OnRenderImage()
        tickets.append(m._converged.ReadBegin(fmt))
        if len(tickets) > 2:
            img3 = m._converged.ReadEnd(tickets.pop(0), copy=False)
    while tickets:
        img3 = m._converged.ReadEnd(tickets.pop(0), copy=False)
    t_fmt[fmt] = ((time.perf_counter() - t) / n, img3.nbytes)
# what one launch per frame costs without any readback (the floor of a host that looks at every frame)
ctx.set_option("frames_per_launch", 1)
for _ in range(3): m.OnRenderImage()
ctx.synchronize()
t = time.perf_counter()
for _ in range(n): m.OnRenderImage()
ctx.synchronize(); t_one = (time.perf_counter() - t) / n
ctx.set_option("frames_per_launch", 0)
print(f"frame without readback {t_trace*1e3:.3f} ms ({rays/t_trace/1e6:.0f} Mrays/s); with a full-frame readback every frame {t_rb*1e3:.3f} ms ({rays/t_rb/1e6:.0f} Mrays/s); readback alone {1e3*(t_rb-t_trace):.3f} ms for {img.nbytes/1e6:.1f} MB; "
      f"PIPELINED readback of every frame (urt_texture_read_begin / _end, two frames in flight, the pinned image handed out) {t_pipe*1e3:.3f} ms per frame ({rays/t_pipe/1e6:.0f} Mrays/s)")
print("pipelined, converted on the GPU to the destination's format: " + "; ".join(f"{k} {v[0]*1e3:.3f} ms per frame ({rays/v[0]/1e6:.0f} Mrays/s, {v[1]/1e6:.1f} MB per frame)" for k, v in t_fmt.items())
      + f"; one launch per frame without any readback {t_one*1e3:.3f} ms")
