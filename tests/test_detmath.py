"""include/rt_detmath.h: the polynomial transcendentals stay within a few ulp of libm, and the HLSL float<->int
conversion rules hold.  (They define the arithmetic of both the oracle and the HIP kernels, so their accuracy is
what ties the build to a real HLSL implementation's values.)"""
import math

import numpy as np
import pytest

import harness as h

lib = h.oracle_lib()


def ulp_err(got, want):
    want32 = np.float32(want)
    if want32 == 0:
        return abs(got) / np.finfo(np.float32).tiny
    return abs(float(got) - want) / float(np.spacing(np.abs(want32)))


@pytest.mark.parametrize("which,fn,lo,hi,tol_ulp", [
    (0, math.sin, 0.0, 6.4, None), (1, math.cos, 0.0, 6.4, None), (2, math.log, 1e-10, 1.0, 4),
    (3, math.exp, -20.0, 5.0, 4), (4, math.asin, -1.0, 1.0, 4), (5, math.acos, -1.0, 1.0, 4),
    (6, math.atan, -50.0, 50.0, 4)])
def test_unary_against_libm(which, fn, lo, hi, tol_ulp):
    rng = np.random.default_rng(1234 + which)
    xs = rng.uniform(lo, hi, 20000).astype(np.float32)
    worst = 0.0
    for x in xs:
        got = lib.orc_detmath1(which, float(x))
        want = fn(float(x))
        if tol_ulp is None:  # sin/cos: absolute error (results near zero have large relative error by nature)
            worst = max(worst, abs(got - want))
        else:
            worst = max(worst, ulp_err(got, want))
    if tol_ulp is None:
        assert worst < 2.5e-7
    else:
        assert worst <= tol_ulp


def test_pow5_and_atan2():
    rng = np.random.default_rng(7)
    for x in rng.uniform(0.0, 1.0, 5000).astype(np.float32):
        got = lib.orc_detmath2(0, float(x), 5.0)
        want = float(x) ** 5
        assert abs(got - want) <= 1e-5 * want + 1e-30
    for _ in range(5000):
        y, x = rng.uniform(-3, 3, 2).astype(np.float32)
        assert abs(lib.orc_detmath2(1, float(y), float(x)) - math.atan2(float(y), float(x))) < 1e-6
    assert lib.orc_detmath2(0, 0.0, 5.0) == 0.0 and lib.orc_detmath2(0, 1.0, 5.0) == 1.0


def test_special_values():
    assert lib.orc_detmath1(2, 0.0) == -math.inf          # log(0) = -inf, as HLSL
    assert math.isnan(lib.orc_detmath1(4, 1.5))           # asin out of range
    assert lib.orc_detmath1(2, 1.0) == 0.0
    assert lib.orc_detmath1(0, 0.0) == 0.0 and lib.orc_detmath1(1, 0.0) == 1.0
