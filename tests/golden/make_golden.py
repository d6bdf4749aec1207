#!/usr/bin/env python3
"""Regenerate the golden fixtures under tests/golden/ from the oracle (oracle/oracle.cpp).

The reference holds no golden vectors or fixtures for this path and cannot be executed in this image
("parity unpinned"), so these fixtures are outputs of the repo's own normative restatement on
the deterministic synthetic inputs of unityraytracer_amd.scenes.  They pin (a) the oracle against regressions
(tests/test_golden.py, CPU) and (b) the HIP path against a committed artefact (tests/test_gpu_golden.py, GPU).
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from unityraytracer_amd import scenes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def cases():
    """name -> (scene, rect or None, oracle mode, frames)"""
    c1 = scenes.config1()
    mixed = scenes.mixed_test_scene(96, 64)
    multi = scenes.mixed_test_scene(64, 40)
    multi.num_rays, multi.num_bounces = 3, 5
    c3 = scenes.config3(480, 270, sky=scenes.make_sky(512, 256))
    return {
        "c1_crop": (c1, (96, 100, 160, 164), 0, 1),          # 16 spheres around the horizon, 1 bounce, literal traversal
        "mixed_frame": (mixed, None, 0, 1),                   # every primitive kind, brute-force triangles
        "multi_ray_accum": (multi, None, 0, 3),               # numRays 3, numBounces 5, 3-frame running mean
        "c3_crop": (c3, (216, 120, 264, 152), 1, 1),          # 69,600-triangle mesh through the oracle's own BVH
    }


def render_case(scene, rect, mode, frames):
    o = pyoracle.Oracle(scene)
    if mode == 1:
        o.build_own_blas()
    conv = None
    for f in range(frames):
        ox, oy, seed = scenes.frame_uniforms(f)
        o.set_frame((ox, oy), seed)
        img = o.render(rect=rect, mode=mode, threads=8)
        conv = img if conv is None else pyoracle.accumulate(img, conv, f)
        if f == 0 and frames > 1:
            conv = pyoracle.accumulate(img, np.zeros_like(img), 0)
    return conv


def math_vectors():
    rng = np.random.default_rng(20261004)
    x = rng.uniform(-1.6e4, 1.6e4, 4096).astype(np.float32)
    u = rng.uniform(0, 1, 4096).astype(np.float32)
    y = rng.normal(size=4096).astype(np.float32)
    px = rng.integers(0, 3840, 4096).astype(np.float32)
    py = rng.integers(0, 2160, 4096).astype(np.float32)
    seed = (rng.integers(0, 40, 4096) * 0.5 + 0.37).astype(np.float32)
    return {"x": x, "u": u, "y": y, "px": px, "py": py, "seed": seed,
            "sin": pyoracle.math_probe("sin", x), "cos": pyoracle.math_probe("cos", x),
            "pow": pyoracle.math_probe("pow", u, (1.0 / (1.0 + 100.0 * u[::-1])).astype(np.float32)),
            "acos": pyoracle.math_probe("acos", (2 * u - 1).astype(np.float32)),
            "atan2": pyoracle.math_probe("atan2", y, y[::-1].copy()),
            "rand": pyoracle.math_probe("rand", seed, px, py)}


def main():
    for name, (scene, rect, mode, frames) in cases().items():
        img = render_case(scene, rect, mode, frames)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), image=img)
        print(name, img.shape, float(np.nanmean(img[..., :3])))
    np.savez_compressed(os.path.join(HERE, "math_vectors.npz"), **math_vectors())


if __name__ == "__main__":
    main()
