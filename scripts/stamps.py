import sys, ctypes as C
sys.path.insert(0, '.')
import os; os.environ.setdefault("URT_ALLOW_EXPERIMENT", "1")   # a measurement tool: may load an A/B / diagnostic build (csrc/experiments.h)
import numpy as np
from unityraytracer_amd import Context, RayTraceMaster, scenes, _lib
ctx = Context(0)
lib = _lib.load()
lib.urt_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
for (w, h, b) in ((1920, 1080, 8), (1920, 1080, 1), (960, 540, 8)):
    sc = scenes.config3(w, h); sc.num_bounces = b
    ctx.set_option("kernel_mode", 2); ctx.set_option("block_threads", 64); ctx.set_option("waves_per_cu", 16); ctx.set_option("refill_min", 16)
    m = RayTraceMaster(ctx, sc)
    for _ in range(3): m.OnRenderImage()
    ctx.synchronize()
    nw = 4096
    st = np.zeros((nw, 4), np.uint64)
    lib.urt_debug_read_stamps(ctx._h, st.ctypes.data_as(C.c_void_p), nw)
    t0 = st[:, 0].min()
    start = (st[:, 0] - t0).astype(np.float64) / 100.0      # us (100 MHz)
    exh = (st[:, 1].astype(np.int64) - np.int64(t0)).astype(np.float64) / 100.0
    end = (st[:, 2] - t0).astype(np.float64) / 100.0
    iters = (st[:, 3] >> np.uint64(32)).astype(np.int64); fetch = (st[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
    print(f"{w}x{h} b={b}: start p50 {np.median(start):.1f} max {start.max():.1f} us | pool exhausted first seen at {exh[exh>0].min():.1f} .. {exh[exh>0].max():.1f} us | end p10 {np.percentile(end,10):.1f} p50 {np.median(end):.1f} p90 {np.percentile(end,90):.1f} p99 {np.percentile(end,99):.1f} max {end.max():.1f} us | iters mean {iters.mean():.1f} max {iters.max()} | fetches mean {fetch.mean():.1f} total {fetch.sum()}", flush=True)
    m.OnDisable()
