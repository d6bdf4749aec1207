# candidate default knob sets for kernel_mode 3 (and mode 4, K=1) across the BASELINE configurations, same process / same box
import sys
sys.path.insert(0, '.')
from unityraytracer_amd import Context, RayTraceMaster, scenes
ctx = Context(0)
def run(sc, opts, frames=8):
    for k, v in opts.items(): ctx.set_option(k, v)
    ctx.set_option("count_stats", 0); ctx.set_option("time_dispatch", 1)
    m = RayTraceMaster(ctx, sc)
    for _ in range(3): m.OnRenderImage()
    ctx.synchronize(); ctx.reset_counters()
    for _ in range(frames): m.OnRenderImage()
    c = ctx.counters(); m.OnDisable()
    return c['trace_ms'] / frames, c['watchdog_trips']
cands = [
    ("m3 auto/48/16/8 (current)", {"kernel_mode": 3, "waves_per_cu": 0, "refill_min": 48, "blas_min": 16, "blas_exit": 8}),
    ("m3 16/32/24/4", {"kernel_mode": 3, "waves_per_cu": 16, "refill_min": 32, "blas_min": 24, "blas_exit": 4}),
    ("m3 auto/32/24/4", {"kernel_mode": 3, "waves_per_cu": 0, "refill_min": 32, "blas_min": 24, "blas_exit": 4}),
    ("m3 20/32/24/4", {"kernel_mode": 3, "waves_per_cu": 20, "refill_min": 32, "blas_min": 24, "blas_exit": 4}),
    ("m3 20/32/24/8", {"kernel_mode": 3, "waves_per_cu": 20, "refill_min": 32, "blas_min": 24, "blas_exit": 8}),
    ("m3 20/32/32/4", {"kernel_mode": 3, "waves_per_cu": 20, "refill_min": 32, "blas_min": 32, "blas_exit": 4}),
    ("m4 k1 fit/32/64/8", {"kernel_mode": 4, "pool_k": 1, "waves_per_cu": 0, "pool_refill": 32, "pool_blas_min": 64, "pool_blas_exit": 8, "pool_other_min": 32}),
]
names = sys.argv[1:] or ["C3", "C3@4K", "C2", "C5", "C4"]
for rep in range(2):
    for name in names:
        sc = scenes.config3(3840, 2160) if name == "C3@4K" else scenes.CONFIGS[name]()
        for label, o in cands:
            ms, wd = run(sc, o, frames=4 if name in ("C4", "C5") else 8)
            print(f"{name:6s} {label:28s}: {ms:8.3f} ms wd {wd}", flush=True)
