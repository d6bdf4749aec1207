# A/B of option "overlap_launches": a host that submits (and optionally reads back) EVERY frame.  ms per frame, C3 (or argv[1]).
import sys, time
sys.path.insert(0, '.')
from unityraytracer_amd import Context, RayTraceMaster, RenderTexture, scenes
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
ctx = Context(0)
sc = scenes.CONFIGS[cfg]()
m = RayTraceMaster(ctx, sc)
dest = RenderTexture(ctx, sc.width, sc.height)
for _ in range(5): m.OnRenderImage(dest)
ctx.synchronize()
n = 100
for rep in range(2):
    for overlap in (0, 1, 2):
        ctx.set_option("overlap_launches", overlap)
        for mode in ("flush", "RGBA8_SRGB"):
            for _ in range(4):
                m.OnRenderImage(dest); ctx.flush()
            ctx.synchronize()
            tickets = []
            t = time.perf_counter()
            for _ in range(n):
                m.OnRenderImage(dest)
                if mode == "flush":
                    ctx.flush()
                else:
                    tickets.append(dest.ReadBegin(mode))
                    if len(tickets) > 2:
                        dest.ReadEnd(tickets.pop(0), copy=False)
            while tickets:
                dest.ReadEnd(tickets.pop(0), copy=False)
            ctx.synchronize()
            dt = (time.perf_counter() - t) / n
            info = ctx.launch_info()
            print(f"{cfg} overlap_launches={overlap} {mode:11s} {dt*1e3:.3f} ms per frame   (last launch: stream {info['trace_stream']}, overlapped {info['overlapped']}, total overlapped {info['overlapped_launches']})", flush=True)
