# per-phase statistics of kernel_mode 4 (needs a -DURT_STAMPS build: URT_LIB_PATH=...)
import sys, ctypes as C, itertools
sys.path.insert(0, '.')
import os; os.environ.setdefault("URT_ALLOW_EXPERIMENT", "1")   # a measurement tool: may load an A/B / diagnostic build (csrc/experiments.h)
import numpy as np
from unityraytracer_amd import Context, RayTraceMaster, scenes, _lib
ctx = Context(0)
lib = _lib.load()
lib.urt_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
for k, rf, bmin, bex, omin in ((1, 32, 64, 8, 24), (2, 64, 64, 8, 24), (2, 64, 64, 32, 24), (2, 64, 64, 48, 24), (2, 64, 96, 32, 16), (3, 64, 128, 32, 24), (3, 64, 96, 48, 24)):
    sc = scenes.config3(1920, 1080)
    ctx.set_option("kernel_mode", 4); ctx.set_option("pool_k", k); ctx.set_option("pool_refill", rf); ctx.set_option("pool_blas_min", bmin); ctx.set_option("pool_blas_exit", bex); ctx.set_option("pool_other_min", omin)
    m = RayTraceMaster(ctx, sc)
    nw = 8192
    for _ in range(3): m.OnRenderImage()
    ctx.synchronize()
    junk = np.zeros((nw, 32), np.uint64); lib.urt_debug_read_stamps(ctx._h, junk.ctypes.data_as(C.c_void_p), nw * 32)
    m.OnRenderImage(); ctx.synchronize()
    st = np.zeros((nw, 32), np.uint64)
    lib.urt_debug_read_stamps(ctx._h, st.ctypes.data_as(C.c_void_p), nw * 32)
    st = st[st[:, 13] > 0]
    t = st[:, 0:4].astype(np.float64) / 100.0; lanes = st[:, 4:8].astype(np.float64); trips = st[:, 8:12].astype(np.float64)
    life = (st[:, 13] - st[:, 12]).astype(np.float64) / 100.0
    t0 = st[:, 12].min()
    end = (st[:, 13] - t0).astype(np.float64) / 100.0
    dry = (st[:, 14].astype(np.float64) - float(t0)) / 100.0
    print(f"--- pool_k {k} refill {rf} blas_min {bmin} exit {bex} other_min {omin}: {len(st)} waves, lifetime mean {life.mean():.0f} us, work dry p50 {np.median(dry):.0f}, end p50 {np.median(end):.0f} max {end.max():.0f} us; loop trips/wave {st[:,18].mean():.0f}; refills/wave {st[:,16].mean():.1f} x {st[:,17].sum()/max(1,st[:,16].sum()):.1f} slots")
    for q, nm in enumerate(["FRONT", "BLAS", "SHADE"]):
        print(f"   {nm:5s}: {t[:, q].sum() / life.sum() * 100:5.1f} % of wave time, {trips[:, q].mean():7.1f} trips/wave, {lanes[:, q].sum() / max(1, trips[:, q].sum()):5.1f} lanes/trip, {t[:, q].sum() / max(1, trips[:, q].sum()):7.2f} us/trip")
    print(f"   BLAS inner: {trips[:, 3].mean():8.1f} steps/wave, {lanes[:, 3].sum() / max(1, trips[:, 3].sum()):5.1f} active lanes/step, {t[:, 1].sum() / max(1, trips[:, 3].sum()) * 1000:7.1f} ns/step")
    print(f"   unaccounted (census, refill, select): {(1 - t[:, 0:3].sum() / life.sum()) * 100:.1f} %")
    m.OnDisable()
