"""sha256 of the accumulated image after N frames of each configuration (reduced or full size), through whatever library URT_LIB_PATH
names: `URT_LIB_PATH=a.so python scripts/lib_pixels.py C3 C4` vs the same with b.so — equal hashes = bit-identical pixels of two BUILDS."""
import hashlib, sys
sys.path.insert(0, '.')
import os; os.environ.setdefault("URT_ALLOW_EXPERIMENT", "1")   # a measurement tool: may load an A/B / diagnostic build (csrc/experiments.h)
import numpy as np
from unityraytracer_amd import Context, RayTraceMaster, scenes
ctx = Context(0)
frames = 4
for cfg in sys.argv[1:] or ["C2", "C3", "C3D", "C4", "C5"]:
    sc = scenes.CONFIGS[cfg]()
    m = RayTraceMaster(ctx, sc)
    for _ in range(frames):
        m.OnRenderImage()
    img = m._converged.GetPixels()
    c = ctx.counters()
    print(cfg, hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest()[:24], "wd", c["watchdog_trips"], flush=True)
    m.OnDisable()
