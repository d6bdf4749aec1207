import sys
sys.path.insert(0, '.')
from unityraytracer_amd import Context, RayTraceMaster, scenes
cfg, mode, bt = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 5
bounces = int(sys.argv[5]) if len(sys.argv) > 5 else None
ctx = Context(0)
sc = scenes.CONFIGS[cfg]()
if bounces: sc.num_bounces = bounces
ctx.set_option("kernel_mode", mode); ctx.set_option("block_threads", bt)
m = RayTraceMaster(ctx, sc)
for _ in range(frames): m.OnRenderImage()
ctx.synchronize()
print(ctx.counters())
