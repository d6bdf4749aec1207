"""GPU: the reference's own two scenes (tests/golden/scene_*.json, see tests/test_reference_scenes.py) through the HIP path against
the oracle, bit for bit: Scene1 (6 spheres + 4 meshes, numBounces 2) and SampleScene with ITS settings — numBounces 10 x numRays 25,
camera pitched 10 degrees — at reduced resolution, several kernel modes, accumulated frames, and the frame protocol with a present."""
import json
import os

import numpy as np
import pytest

from oracle import pyoracle
from unityraytracer_amd import RayTraceMaster, RenderTexture, scenes

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name,w,h,frames", [("Scene1", 348, 182, 3), ("SampleScene", 200, 112, 2)])
def test_reference_scene_bit_exact(gpu_ctx, name, w, h, frames):
    sc = scenes.from_unity_fixture(json.load(open(os.path.join(GOLD, f"scene_{name}.json"))), w, h)
    o = pyoracle.Oracle(sc)
    o.build_own_blas()                                         # the oracle's own BVH: nothing shared with the product's
    acc = np.zeros((h, w, 4), np.float32)
    for i in range(frames):
        ox, oy, sd = scenes.frame_uniforms(i)
        o.set_frame((ox, oy), sd)
        acc = pyoracle.accumulate(o.render(mode=1, threads=8), acc, i)
    for mode in (3, 0, 2):
        gpu_ctx.set_option("kernel_mode", mode)
        gpu_ctx.reset_counters()
        m = RayTraceMaster(gpu_ctx, sc)
        dest = RenderTexture(gpu_ctx, w, h)
        for _ in range(frames):
            m.OnRenderImage(dest)                              # RM:848 with a destination: dispatch, accumulate, present
        got = dest.GetPixels()
        c = gpu_ctx.counters()
        dest.Release(); m.OnDisable()
        bad = int((got.view(np.uint32) != acc.view(np.uint32)).any(axis=2).sum())
        assert bad == 0, f"{name} mode {mode}: {bad} pixels differ"
        assert c["watchdog_trips"] == 0 and c["rays"] > w * h * sc.num_rays * frames * 0.9
    gpu_ctx.set_option("kernel_mode", 3)
