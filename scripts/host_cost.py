"""Host time of one frame of RM's protocol through the Python mirror (uniforms -> Dispatch -> Blit, all deferred by the library):
what the GPU waits for before a batch is submitted.  Usage: python scripts/host_cost.py"""
import sys, time
sys.path.insert(0, '.')
from unityraytracer_amd import Context, RayTraceMaster, scenes

ctx = Context(0)
sc = scenes.mixed_test_scene(64, 40)                 # tiny frames: the GPU side is negligible, the host loop is what is timed
m = RayTraceMaster(ctx, sc)
for _ in range(70): m.OnRenderImage()
ctx.synchronize()
for n in (20, 64, 640):
    t0 = time.perf_counter()
    for _ in range(n): m.OnRenderImage()
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    print(f"{n} frames: host {1e6 * (t1 - t0) / n:.1f} us per frame, + {1e3 * (t2 - t1):.2f} ms to drain", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(640): m.OnRenderImage()
pr.disable(); ctx.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
