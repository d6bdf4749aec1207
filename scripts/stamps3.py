import sys, ctypes as C
sys.path.insert(0, '.')
import os; os.environ.setdefault("URT_ALLOW_EXPERIMENT", "1")   # a measurement tool: may load an A/B / diagnostic build (csrc/experiments.h)
import numpy as np
from unityraytracer_amd import Context, RayTraceMaster, scenes, _lib
ctx = Context(0)
lib = _lib.load()
lib.urt_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
import itertools
cfg = sys.argv[1] if len(sys.argv) > 1 else None
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 16          # frames of the ONE launch the stamps describe (frame batching)
for (w, h, b), wpc in itertools.product(((1920, 1080, 8),), (20,)):
    sc = scenes.CONFIGS[cfg]() if cfg else scenes.config3(w, h)
    if not cfg: sc.num_bounces = b
    w, h, b = sc.width, sc.height, sc.num_bounces
    ctx.set_option("kernel_mode", 3); ctx.set_option("waves_per_cu", wpc)
    for kv in sys.argv[3:]:                                        # extra options: name=value ...
        k, v = kv.split("="); ctx.set_option(k, int(v))
    print(f"--- waves/CU {wpc}")
    m = RayTraceMaster(ctx, sc)
    nw = 8192
    for _ in range(3): m.OnRenderImage()
    ctx.synchronize()
    junk = np.zeros((nw, 32), np.uint64); lib.urt_debug_read_stamps(ctx._h, junk.ctypes.data_as(C.c_void_p), nw * 32)   # clears
    for _ in range(frames): m.OnRenderImage()
    ctx.synchronize()
    st = np.zeros((nw, 32), np.uint64)
    lib.urt_debug_read_stamps(ctx._h, st.ctypes.data_as(C.c_void_p), nw * 32)
    t = st[:, 0:4].astype(np.float64) / 100.0; lanes = st[:, 4:8].astype(np.float64); trips = st[:, 8:12].astype(np.float64)
    life = (st[:, 13] - st[:, 12]).astype(np.float64) / 100.0
    ok = st[:, 13] > 0
    end = (st[:, 13].astype(np.float64) - float(st[ok, 12].min())) / 100.0
    st, t, lanes, trips, life, end = st[ok], t[ok], lanes[ok], trips[ok], life[ok], end[ok]
    t0 = st[:, 12].min()
    dry = (st[:, 14].astype(np.float64) - float(t0)) / 100.0
    start = (st[:, 12] - t0).astype(np.float64) / 100.0
    print(f"   waves {ok.sum()}: start p50 {np.median(start):.0f} max {start.max():.0f} us; work ran dry (per wave) p10 {np.percentile(dry, 10):.0f} p50 {np.median(dry):.0f} max {dry.max():.0f} us; "
          f"end p10 {np.percentile(end, 10):.0f} p50 {np.median(end):.0f} p90 {np.percentile(end, 90):.0f} p99 {np.percentile(end, 99):.0f} max {end.max():.0f} us")
    names = ["FRONT", "BLAS", "SHADE"]
    print(f"{w}x{h} b={b}, {frames} frames in the launch: wave lifetime mean {life.mean():.0f} us, end p50 {np.median(end):.0f} p90 {np.percentile(end, 90):.0f} max {end.max():.0f} us")
    for q in range(3):
        print(f"   {names[q]:5s}: {t[:, q].sum() / life.sum() * 100:5.1f} % of wave time, {trips[:, q].mean():7.1f} trips/wave, {lanes[:, q].sum() / max(1, trips[:, q].sum()):5.1f} lanes/trip, {t[:, q].sum() / max(1, trips[:, q].sum()):7.2f} us/trip")
    print(f"   BLAS inner: {trips[:, 3].mean():8.1f} steps/wave, {lanes[:, 3].sum() / max(1, trips[:, 3].sum()):5.1f} active lanes/step, {t[:, 1].sum() / max(1, trips[:, 3].sum()) * 1000:7.1f} ns/step")
    fs = st[:, 25:32].astype(np.float64)
    if fs[:, 3].sum() == 0 and fs[:, 2].sum() > 0:               # single-mesh instantiation: slots 25-27 hold the refill's split
        print(f"   REFILL: {fs[:, 2].mean():7.1f} refills/wave; work-counter hand-out {fs[:, 0].sum() / 100 / life.sum() * 100:5.1f} % of wave time ({fs[:, 0].sum() / 100 / fs[:, 2].sum():5.2f} us each), "
              f"camera rays {fs[:, 1].sum() / 100 / life.sum() * 100:5.1f} % ({fs[:, 1].sum() / 100 / fs[:, 2].sum():5.2f} us each); "
              f"unaccounted {(1 - (t[:, 0:3].sum() + (fs[:, 0].sum() + fs[:, 1].sum()) / 100) / life.sum()) * 100:5.1f} %")
        steps = trips[:, 3].sum()
        print(f"   BLAS vote: {fs[:, 4].sum() / max(1, steps):5.1f} lanes take part per step of {lanes[:, 3].sum() / max(1, steps):5.1f} active; node trips {fs[:, 5].sum() / max(1, steps) * 100:5.1f} % of steps, "
              f"{fs[:, 6].sum() / max(1, fs[:, 5].sum()):5.1f} lanes wait at a leaf per node trip")
    if fs[:, 3].sum() > 0:                                       # listed FRONT: where its time goes
        ft = t[:, 0].sum()
        print(f"   FRONT split: heap walk {fs[:, 0].sum() / 100 / ft * 100:5.1f} % ({fs[:, 0].sum() / 100 / fs[:, 3].sum():6.2f} us per walk, {fs[:, 6].sum() / fs[:, 3].sum():5.1f} fresh lanes), "
              f"single-leaf tests {fs[:, 1].sum() / 100 / ft * 100:5.1f} % ({fs[:, 4].sum() / max(1, trips[:, 0].sum()):4.2f} rounds/trip), "
              f"BVH-top walks {fs[:, 2].sum() / 100 / ft * 100:5.1f} % ({fs[:, 5].sum() / max(1, trips[:, 0].sum()):4.2f} per trip)")
    dr = st[:, 16:25].astype(np.float64)
    drain = end - dry
    order = np.argsort(drain)
    for label, sel in (("all waves", order), ("slowest 5 %", order[-len(order) // 20:])):
        d = dr[sel]
        print(f"   drain [{label}]: {drain[sel].mean():.0f} us after dry with {d[:, 7].mean():.1f} live paths; trips FRONT {d[:, 0].mean():.1f} BLAS {d[:, 1].mean():.1f} SHADE {d[:, 2].mean():.1f}; "
              f"time FRONT {d[:, 4].mean() / 100:.0f} BLAS {d[:, 5].mean() / 100:.0f} SHADE {d[:, 6].mean() / 100:.0f} us; BLAS steps {d[:, 3].mean():.0f} at {d[:, 8].sum() / max(1, d[:, 3].sum()):.1f} lanes")
    m.OnDisable()
