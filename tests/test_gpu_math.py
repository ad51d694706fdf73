"""Accuracy of the kernel's domain-specialised fp64 math (csrc/mcr_math.h), measured on the device
against long-double / exact references.  These are the claims the path tolerance rests on."""

from __future__ import annotations

import numpy as np
import pytest

from monte_carlo_retirement_amd import _native as N
from monte_carlo_retirement_amd import engine as E

pytestmark = pytest.mark.gpu


def _ulps(got, exact_ld):
    got = np.asarray(got, dtype=np.float64)
    ref = exact_ld.astype(np.float64)
    ulp = np.spacing(np.abs(ref))
    return np.abs((got.astype(np.longdouble) - exact_ld) / ulp.astype(np.longdouble)).astype(np.float64)


def test_division_is_correctly_rounded_in_the_balance_range():
    """fdiv == IEEE a/b bit-for-bit for operands anywhere in the path's dynamic range."""
    rng = np.random.default_rng(1)
    a = rng.uniform(-1, 1, 400_000) * 10.0 ** rng.uniform(-8, 16, 400_000)
    b = rng.uniform(0.1, 1, 400_000) * 10.0 ** rng.uniform(-7, 16, 400_000) * rng.choice([-1.0, 1.0], 400_000)
    got = E.eval_helper_host(N.MCR_HELPER_MATH_DIV, None, np.column_stack((a, b)))[:, 0]
    assert np.array_equal(got, a / b)


def test_path_division_one_newton_step_is_still_correctly_rounded():
    """Inside the path the reciprocal takes ONE Newton step (fdiv<false>): the residual correction squares its error, so
    the quotient equals IEEE a/b except within ~2^-100 of a rounding boundary.  4 million random operand pairs over the
    path's dynamic range, plus the near-1 and power-of-two neighbourhoods where reciprocal seeds are worst: no mismatch."""
    rng = np.random.default_rng(11)
    n = 4_000_000
    a = rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-8, 16, n)
    b = rng.uniform(0.1, 1, n) * 10.0 ** rng.uniform(-7, 16, n) * rng.choice([-1.0, 1.0], n)
    a[:200_000] = 1.0 + rng.uniform(-1e-9, 1e-9, 200_000)
    b[:200_000] = 1.0 + rng.uniform(-1e-9, 1e-9, 200_000)
    b[200_000:400_000] = 2.0 ** rng.integers(-20, 40, 200_000) * (1.0 + rng.choice([0.0, 2.0 ** -52, -2.0 ** -53], 200_000))
    got = E.eval_helper_host(N.MCR_HELPER_MATH_DIV_PATH, None, np.column_stack((a, b)))[:, 0]
    assert np.array_equal(got, a / b), int((got != a / b).sum())


def test_exp_error_bound():
    """<= 1.5 ulp for the arguments the path produces (monthly log-returns, |x| < 2) and + 0.5 ulp per unit of |x| beyond:
    the argument reduction uses ONE constant (ln 2 / 512 rounded), whose representation error scales with |x|."""
    x = np.concatenate([np.random.default_rng(2).uniform(-3, 3, 300_000), np.linspace(-20, 20, 50_001), [0.0, -0.0, 1e-300]])
    got = E.eval_helper_host(N.MCR_HELPER_MATH_EXP, None, x.reshape(-1, 1))[:, 0]
    u = _ulps(got, np.exp(x.astype(np.longdouble)))
    assert np.all(u <= 1.5 + 0.5 * np.abs(x)), float((u / (1.5 + 0.5 * np.abs(x))).max())
    small = np.abs(x) <= 2.0
    assert u[small].max() <= 2.0, u[small].max()
    assert np.mean(u[small] <= 0.5) > 0.60   # mostly correctly rounded


def test_sqrt_relative_error():
    """v_rsq_f64 seed (~2^-23.7 on gfx950) + ONE coupled Goldschmidt step: relative error 1.5 e^2 <= 1e-14 (measured
    7.9e-15 = 36 ulp).  The radius of a Box-Muller pair needs no more: the normal's error is that times |z| <= 6.7,
    and sigma/sqrt(12) of it reaches a monthly growth factor — below the error of the exp() that follows."""
    w = np.concatenate([np.random.default_rng(3).uniform(0, 46, 300_000), 10.0 ** np.random.default_rng(4).uniform(-10, 2, 100_000)])
    got = E.eval_helper_host(N.MCR_HELPER_MATH_SQRT, None, w.reshape(-1, 1))[:, 0]
    exact = np.sqrt(w.astype(np.longdouble))
    rel = np.abs((got.astype(np.longdouble) - exact) / exact).astype(np.float64)
    assert rel.max() <= 1.0e-14, rel.max()


def test_neg2log_absolute_error():
    """-2 ln u of the Box-Muller radius: what matters is the ABSOLUTE error (the radius^2 feeds sigma*z)."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.integers(0, 2**32, 400_000), [0, 1, 2, 2**31, 2**32 - 1, 2**32 - 2]]).astype(np.float64)
    got = E.eval_helper_host(N.MCR_HELPER_MATH_NEG2LOG, None, x.reshape(-1, 1))[:, 0]
    u = (x.astype(np.longdouble) + np.longdouble(0.5)) * np.longdouble(2.0) ** -32
    exact = -2 * np.log(u)
    err = np.abs(got.astype(np.longdouble) - exact).astype(np.float64)
    bound = np.maximum(2.0 * np.spacing(got), 3e-16)  # <= 2 ulp of the result (measured 1.6), 3e-16 near u -> 1
    assert np.all(err <= bound), float((err / bound).max())
    assert np.all(got > 0)


def test_sincos_absolute_error():
    rng = np.random.default_rng(6)
    x = np.concatenate([rng.integers(0, 2**32, 400_000), [0, 2**24 - 1, 2**24, 2**31 - 1, 2**31, 2**32 - 1]]).astype(np.float64)
    got = E.eval_helper_host(N.MCR_HELPER_MATH_SINCOS, None, x.reshape(-1, 1))
    # long-double pi is only ~1e-19 accurate: fine against a 3e-16 bound
    two_pi = np.longdouble("6.283185307179586476925286766559005768")
    ang = (x.astype(np.longdouble) + np.longdouble(0.5)) * two_pi / np.longdouble(2.0) ** 32
    es = np.abs(got[:, 0].astype(np.longdouble) - np.sin(ang)).astype(np.float64)
    ec = np.abs(got[:, 1].astype(np.longdouble) - np.cos(ang)).astype(np.float64)
    assert max(es.max(), ec.max()) < 3e-16, (es.max(), ec.max())
    assert np.abs(got[:, 0] ** 2 + got[:, 1] ** 2 - 1.0).max() < 5e-16


def test_path_forms_of_the_series_stay_inside_their_stated_bounds():
    """The month loop runs shorter series (template parameter PATH of csrc/mcr_math.h): their truncation errors are sized to
    the 1e-9 PATH tolerance — what a growth factor may be off by is ~1e-9 / 833 months per month in the worst, one-signed
    case — not to the last ulp.  Stated bounds, measured here on the device:
      exp         relative error <= 3.5e-15 with ZERO MEAN (r^2/24 replaced by its mean-square fit a^2/40, a = ln 2 / 1024: the
                  error r^4/24 - a^2 r^2/40 changes sign inside the reduction interval; a one-signed truncation would add
                  up over 2040 months) on top of the full form's 2 ulp;
      -2 ln u     absolute error <= 3.6e-13 (the dropped 2 r^5/5 - r^6/3, |r| <= 2^-8) on top of 2 ulp;
      cos         absolute error <= 4.7e-15 (the dropped dl^6/720, |dl| <= pi/256) on top of 3e-16; sin as the full form.
    Path-level effect (tools/k1_accuracy.py, 2e5 paths x 3 scenarios against the oracle): worst error 7e-11 of the 1e-9
    allowed, no Success flag flipped."""
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(-2, 2, 300_000), [0.0, 1e-300, 0.6931471805599453 / 1024, -0.6931471805599453 / 1024]])
    got = E.eval_helper_host(N.MCR_HELPER_MATH_EXP_PATH, None, x.reshape(-1, 1))[:, 0]
    exact = np.exp(x.astype(np.longdouble))
    rel = np.abs((got.astype(np.longdouble) - exact) / exact).astype(np.float64)
    assert rel.max() <= 3.5e-15 + 2 * 2.2e-16, rel.max()
    assert rel.max() > 1.5e-15                       # (the form under test really is the shorter one)
    signed = ((got.astype(np.longdouble) - exact) / exact).astype(np.float64)[:300_000]
    assert abs(signed.mean()) < 1.5e-16, signed.mean()   # zero mean: sqrt(months), not months, times the per-month error

    u32 = np.concatenate([rng.integers(0, 2**32, 400_000), [0, 1, 2, 2**31, 2**32 - 1, 2**32 - 2]]).astype(np.float64)
    got = E.eval_helper_host(N.MCR_HELPER_MATH_NEG2LOG_PATH, None, u32.reshape(-1, 1))[:, 0]
    u = (u32.astype(np.longdouble) + np.longdouble(0.5)) * np.longdouble(2.0) ** -32
    err = np.abs(got.astype(np.longdouble) + 2 * np.log(u)).astype(np.float64)
    assert np.all(err <= 3.6e-13 + 2.0 * np.spacing(got)), float(err.max())
    assert err.max() > 1e-14 and np.all(got > 0)

    got = E.eval_helper_host(N.MCR_HELPER_MATH_SINCOS_PATH, None, u32.reshape(-1, 1))
    two_pi = np.longdouble("6.283185307179586476925286766559005768")
    ang = (u32.astype(np.longdouble) + np.longdouble(0.5)) * two_pi / np.longdouble(2.0) ** 32
    es = np.abs(got[:, 0].astype(np.longdouble) - np.sin(ang)).astype(np.float64)
    ec = np.abs(got[:, 1].astype(np.longdouble) - np.cos(ang)).astype(np.float64)
    assert es.max() < 3e-16 + 4.7e-15 and ec.max() < 3e-16 + 4.7e-15, (es.max(), ec.max())   # (the rotation mixes cos(dl) into both)
    assert ec.max() > 1e-15
