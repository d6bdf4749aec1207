"""The oracle reproduces the committed golden fixtures bit-for-bit (tests/golden/*.npz, made by
tests/golden/make_golden.py).  These are the repo's own vectors: the reference ships none."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import pyoracle

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "make_golden.py"))
make_golden = importlib.util.module_from_spec(spec)
spec.loader.exec_module(make_golden)


@pytest.mark.parametrize("name", ["c1_crop", "mixed_frame", "multi_ray_accum", "c3_crop"])
def test_oracle_matches_golden_image(name):
    scene, rect, mode, frames = make_golden.cases()[name]
    want = np.load(os.path.join(HERE, name + ".npz"))["image"]
    got = make_golden.render_case(scene, rect, mode, frames)
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.isfinite(want[..., :3]).all() and want[..., :3].max() > 0.1          # a real picture, not zeros


def test_math_vectors_bit_exact():
    g = np.load(os.path.join(HERE, "math_vectors.npz"))
    v = make_golden.math_vectors()
    for k in ("sin", "cos", "pow", "acos", "atan2", "rand"):
        assert np.array_equal(v[k].view(np.uint32), g[k].view(np.uint32)), k
