"""GPU: the HIP path against the committed golden fixtures (no oracle call in this file)."""
import importlib.util
import os

import numpy as np
import pytest

from unityraytracer_amd import RayTraceMaster

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "make_golden.py"))
make_golden = importlib.util.module_from_spec(spec)


@pytest.mark.parametrize("name", ["c1_crop", "mixed_frame", "multi_ray_accum", "c3_crop"])
@pytest.mark.parametrize("mode", [5, 4, 3, 2, 1, 0])
def test_hip_matches_golden(gpu_ctx, name, mode):
    spec.loader.exec_module(make_golden)
    scene, rect, _, frames = make_golden.cases()[name]
    want = np.load(os.path.join(HERE, name + ".npz"))["image"]
    gpu_ctx.set_option("kernel_mode", mode)
    gpu_ctx.set_option("count_stats", 0)
    m = RayTraceMaster(gpu_ctx, scene)
    for _ in range(frames):
        m.OnRenderImage()
    img = (m._converged if frames > 1 else m._target).GetPixels()
    m.OnDisable()
    if rect is not None:
        x0, y0, x1, y1 = rect
        img = img[y0:y1, x0:x1]
    diff = np.abs(img.astype(np.float64) - want.astype(np.float64))
    assert np.array_equal(img.view(np.uint32), want.view(np.uint32)), f"max |d| = {diff.max():.3e} (north_star tolerance 1e-4)"
