"""Known-answer and accuracy tests of the normative float32 arithmetic (include/urt_math.h) as compiled into
the oracle.  The reference has no tests for this path, so these pin the restatement:
accuracy against float64 libm, exact identities, and the structural properties of rand()."""
import numpy as np

from oracle import pyoracle


def ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    ulp = np.maximum(ulp, np.float64(np.finfo(np.float32).tiny))
    return np.abs(got.astype(np.float64) - ref64) / ulp


def test_sin_cos_accuracy_over_rand_argument_range():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-2e4, 2e4, 200000), rng.uniform(-8, 8, 50000), [0.0, np.pi / 2, -np.pi, 1e-20]]).astype(np.float32)
    s = pyoracle.math_probe("sin", x)
    c = pyoracle.math_probe("cos", x)
    x64 = x.astype(np.float64)
    # absolute error stays within ~1 ulp of 1.0 even where sin crosses zero (relative ulp is meaningless there)
    assert np.max(np.abs(s - np.sin(x64))) < 1.5e-7
    assert np.max(np.abs(c - np.cos(x64))) < 1.5e-7
    big = np.abs(np.sin(x64)) > 0.1
    assert np.max(ulp_err(s[big], np.sin(x64[big]))) <= 2.0
    assert pyoracle.math_probe("sin", np.float32([0.0]))[0] == 0.0
    assert pyoracle.math_probe("cos", np.float32([0.0]))[0] == 1.0


def test_log2_exp2_pow():
    rng = np.random.default_rng(2)
    x = np.exp(rng.uniform(-80, 80, 100000)).astype(np.float32)
    l = pyoracle.math_probe("log2", x)
    ref = np.log2(x.astype(np.float64))
    assert np.max(np.abs(l - ref) / np.maximum(1.0, np.abs(ref))) < 2.5e-7
    assert pyoracle.math_probe("log2", np.float32([1.0, 2.0, 0.5, 1024.0])).tolist() == [0.0, 1.0, -1.0, 10.0]
    assert pyoracle.math_probe("log2", np.float32([0.0]))[0] == -np.inf
    e = rng.uniform(-120, 120, 100000).astype(np.float32)
    p = pyoracle.math_probe("exp2", e)
    assert np.max(ulp_err(p, np.exp2(e.astype(np.float64)))) <= 2.0
    assert pyoracle.math_probe("exp2", np.float32([0, 1, -1, 10, -np.inf])).tolist() == [1.0, 2.0, 0.5, 1024.0, 0.0]
    # HLSL pow semantics used at RS:104 and RS:401 (A.10)
    assert pyoracle.math_probe("pow", np.float32([0.0]), np.float32([0.5]))[0] == 0.0          # pow(0, y>0) = 0
    base = rng.uniform(0.0, 1.0, 50000).astype(np.float32)
    expo = (1.0 / (rng.uniform(0.0, 1000.0, 50000) + 1.0)).astype(np.float32)
    got = pyoracle.math_probe("pow", base, expo)
    ref = np.power(base.astype(np.float64), expo.astype(np.float64))
    assert np.max(np.abs(got - ref)) < 4e-7
    sq = (np.linspace(0, 1, 1001) ** 2).astype(np.float32)
    a = pyoracle.math_probe("pow", np.full(1001, 1000.0, np.float32), sq)
    assert np.max(np.abs(a / np.power(1000.0, sq.astype(np.float64)) - 1.0)) < 2e-6


def test_acos_atan2():
    x = np.linspace(-1, 1, 200001).astype(np.float32)
    assert np.max(np.abs(pyoracle.math_probe("acos", x) - np.arccos(x.astype(np.float64)))) < 5e-7
    # normalize() may return 1 + 1ulp: the normative acos clamps instead of returning NaN
    assert pyoracle.math_probe("acos", np.float32([1.0000001, -1.0000001])).tolist() == [0.0, float(np.float32(3.14159274))]
    rng = np.random.default_rng(3)
    y, xx = rng.normal(size=200000).astype(np.float32), rng.normal(size=200000).astype(np.float32)
    assert np.max(np.abs(pyoracle.math_probe("atan2", y, xx) - np.arctan2(y.astype(np.float64), xx.astype(np.float64)))) < 6e-7
    assert pyoracle.math_probe("atan2", np.float32([0.0]), np.float32([0.0]))[0] == 0.0
    q = pyoracle.math_probe("atan2", np.float32([0, 1, 0, -1]), np.float32([1, 0, -1, 0]))
    assert np.allclose(q, [0, np.pi / 2, np.pi, -np.pi / 2], atol=3e-7)


def test_rand_structure():
    """RS:77-81 / A.1: pixel (0,0) always yields 0; values in [0,1); the hash is frac(sin(...) * 43758.5453)
    evaluated in float32 (quantised to the float ulp at that magnitude)."""
    assert pyoracle.math_probe("rand", np.float32([0.37, 4.37]), np.float32([0, 0]), np.float32([0, 0])).tolist() == [0.0, 0.0]
    rng = np.random.default_rng(4)
    px = rng.integers(0, 1920, 100000).astype(np.float32)
    py = rng.integers(0, 1080, 100000).astype(np.float32)
    seed = rng.uniform(0, 13, 100000).astype(np.float32)
    r = pyoracle.math_probe("rand", seed, px, py)
    assert r.min() >= 0.0 and r.max() < 1.0
    assert 0.45 < r.mean() < 0.55
    # independent evaluation of the same formula (float64 sin of the float32 argument): agrees wherever the
    # 43758.5x amplification of a last-place difference in sin() stays small (SURVEY §8c measured ~13 % otherwise)
    a = ((seed + seed / np.float32(17)) / np.float32(100)).astype(np.float32)
    d = np.float32(px.astype(np.float64) * np.float64(np.float32(12.9898)))
    d = (py.astype(np.float64) * np.float64(np.float32(78.233)) + d.astype(np.float64)).astype(np.float32)   # fma(py, c, px*c)
    arg = (a * d).astype(np.float32)
    v = (np.sin(arg.astype(np.float64)).astype(np.float32) * np.float32(43758.5453)).astype(np.float32)
    ref = v - np.floor(v)
    wrap = np.minimum(np.abs(r - ref), 1.0 - np.abs(r - ref))
    assert (wrap < 0.0079).mean() > 0.99        # within 2 quanta (2^-7) of the independent evaluation almost always
    assert (wrap == 0).mean() > 0.80            # and mostly identical


def test_division_by_the_rand_constants_is_the_ieee_quotient_for_every_float():
    """rand() divides by 17 and by 100 (RS:78), the sky lookup by -PI (RS:424-425).  include/urt_math.h evaluates both as a reciprocal multiplication plus one
    fma correction step (f_div_const; the divider only for zeros, the bottom of the range, infinities and NaN) — ten GPU
    instructions fewer per division.  EXHAUSTIVE: all 2^32 bit patterns of x give the bits of x / c, for both constants."""
    assert pyoracle.check_div_const(17.0, threads=8) == 0
    assert pyoracle.check_div_const(100.0, threads=8) == 0
    assert pyoracle.check_div_const(-3.14159265, threads=8) == 0     # the sky lookup's "/ -PI" (RS:424-425; HIP side only)
    # the golden values of rand() itself are unchanged: the same formula with true divisions, evaluated in numpy float32
    seed = np.float32([0.5, 1.0, 7.25, 63.5, 1e-3])
    a = ((seed + seed / np.float32(17)) / np.float32(100)).astype(np.float32)
    px, py = np.float32([3, 100, 1919, 7, 640]), np.float32([5, 200, 1079, 0, 360])
    d = (py.astype(np.float64) * np.float64(np.float32(78.233)) + np.float64(np.float32(px.astype(np.float64) * np.float64(np.float32(12.9898))))).astype(np.float32)
    s_ = pyoracle.math_probe("sin", (a * d).astype(np.float32))
    v = (s_ * np.float32(43758.5453)).astype(np.float32)
    assert np.array_equal(pyoracle.math_probe("rand", seed, px, py), (v - np.floor(v)).astype(np.float32))
