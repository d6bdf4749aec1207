"""A/B of two kernel modes on one box: pixels of mode B must equal mode A's bit for bit (frames accumulate), then the trace
kernels' time per frame of each.  Usage: python scripts/mode_ab.py [A=3] [B=5] [--configs C3 C4 ...] [--frames 32] [k=v ...] (k=v: options for mode B only)"""
import sys, time
sys.path.insert(0, '.')
import os; os.environ.setdefault("URT_ALLOW_EXPERIMENT", "1")   # a measurement tool: may load an A/B / diagnostic build (csrc/experiments.h)
import numpy as np
from unityraytracer_amd import Context, RayTraceMaster, scenes

args = [a for a in sys.argv[1:]]
cfgs, frames, extra, modes = [], 32, [], []
i = 0
while i < len(args):
    a = args[i]
    if a == "--configs":
        i += 1
        while i < len(args) and not args[i].startswith("--") and "=" not in args[i]:
            cfgs.append(args[i]); i += 1
        continue
    if a == "--frames": frames = int(args[i + 1]); i += 2; continue
    if "=" in a: extra.append(a.split("="))
    else: modes.append(int(a))
    i += 1
A, B = (modes + [3, 5])[:2] if len(modes) < 2 else modes[:2]
cfgs = cfgs or ["mixed", "C3"]
ctx = Context(0)


def scene_of(name):
    if name == "mixed":
        sc = scenes.mixed_test_scene(200, 120); return sc
    if name == "mixed3":
        sc = scenes.mixed_test_scene(200, 120); sc.num_rays = 3; return sc
    return scenes.CONFIGS[name]()


DEFAULTS = {"blas_min": 0, "blas_exit": 0, "serve_refill": 16, "refill_min": 16, "shade_min": 32, "sky_min": 32, "waves_per_cu": 0}


def run(sc, mode, opts, n):
    ctx.set_option("kernel_mode", mode)
    for k, _ in extra: ctx.set_option(k, DEFAULTS[k])          # options given for mode B do not leak into mode A's runs
    for k, v in opts: ctx.set_option(k, int(v))
    ctx.set_option("time_dispatch", 1)
    m = RayTraceMaster(ctx, sc)
    for _ in range(3): m.OnRenderImage()
    ctx.synchronize(); ctx.reset_counters()
    t0 = time.perf_counter()
    for _ in range(n): m.OnRenderImage()
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    c = ctx.counters()
    ker = c['trace_ms'] / n
    img = m._converged.GetPixels()
    m.OnDisable()
    return img, ker, wall, c


for name in cfgs:
    sc = scene_of(name)
    a_img, a_ker, a_wall, ca = run(sc, A, [], frames)
    print(f"{name}: mode {A}: kernel {a_ker:.3f} ms/frame, wall {a_wall:.3f} ms/frame, watchdog {ca['watchdog_trips']}", flush=True)
    b_img, b_ker, b_wall, cb = run(sc, B, extra, frames)
    same = np.array_equal(a_img.view(np.uint32), b_img.view(np.uint32))
    if B == 5:                                                # what the traversal service did (counting build of the kernel: not timed)
        ctx.set_option("kernel_mode", 5); ctx.set_option("count_stats", 1)
        for k, v in extra: ctx.set_option(k, int(v))
        m = RayTraceMaster(ctx, sc)
        for _ in range(3): m.OnRenderImage()
        ctx.synchronize(); ctx.reset_counters()
        for _ in range(16): m.OnRenderImage()
        ctx.synchronize()
        sv = ctx.serve_stats(); m.OnDisable(); ctx.set_option("count_stats", 0)
        print(f"   service per frame: {sv['visits'] / 16:.0f} visits, {sv['trips'] / 16:.0f} trips at {sv['lane_trips'] / max(1, sv['trips']):.1f} lanes, "
              f"{sv['claim_rounds'] / 16:.0f} claim rounds x {sv['claimed'] / max(1, sv['claim_rounds']):.1f} rays, {sv['suspended'] / 16:.0f} suspended", flush=True)
    print(f"{name}: mode {B} {dict(extra)}: kernel {b_ker:.3f} ms/frame, wall {b_wall:.3f} ms/frame, watchdog {cb['watchdog_trips']}, pixels {'IDENTICAL' if same else 'DIFFER: ' + str(int((a_img != b_img).sum()))}", flush=True)
