"""The PATH FORMS of the state-machine helpers (csrc/mcr_device.h: withdraw2, net_liquidation_values2, rebalance_path,
annual_gain_taxes<false, ...>) — the code the path kernel actually runs — against the oracle's restatement of the
reference's helpers (simulation.py:201-450), BIT FOR BIT, on random reachable states.

The path forms drop clamps that are provable no-ops when 0 <= balance, 0 <= cost basis, 0 <= rate <= 1 and
"amount sold <= amount held", replace selections by masked moves, share reciprocals and skip dead tax arithmetic at
compile time.  The claim is that none of this changes a single bit; whole paths can only be compared to 1e-9 (exp), so
this is where the claim is checked directly — including magnitudes up to 1e14, dust, zero balances, cost bases above
the balance (losses), 0 % / 100 % rates and the 0 / 1 allocations."""

from __future__ import annotations

import numpy as np
import pytest

from conftest import load_golden
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import _native as N
from monte_carlo_retirement_amd import engine as E

pytestmark = pytest.mark.gpu


def _params(rng):
    pick = lambda *xs: xs[int(rng.integers(len(xs)))]  # noqa: E731
    base = load_golden("helpers.json")["tax_cfgs"][0]
    cfg = dict(base,
               allocation_inv1_pct=float(pick(0.0, 1.0, 0.5, rng.uniform(0, 1))),
               inv1_annual_tax_on_gains_rate=float(pick(0.0, 1.0, rng.uniform(0, 1))),
               inv1_realized_gains_tax_rate=float(pick(0.0, 1.0, rng.uniform(0, 1))),
               inv1_use_realized_gains_tax_system=bool(rng.integers(2)),
               inv2_annual_tax_on_gains_rate=float(pick(0.0, 1.0, rng.uniform(0, 1))),
               inv2_realized_gains_tax_rate=float(pick(0.0, 1.0, rng.uniform(0, 1))),
               inv2_use_realized_gains_tax_system=bool(rng.integers(2)))
    return cfg, params_from_config(Config(**cfg))


def _money(rng, n):
    """Amounts as the path sees them: zero, dust around the reference's epsilon, ordinary, enormous."""
    kind = rng.integers(0, 6, n)
    x = 10.0 ** rng.uniform(-1, 14, n) * rng.uniform(0.1, 1.0, n)
    x = np.where(kind == 0, 0.0, x)
    x = np.where(kind == 1, rng.uniform(0, 3e-6, n), x)
    return x


def _states(rng, n):
    b1, b2 = _money(rng, n), _money(rng, n)
    # cost bases: equal to the balance, below (gains), above (losses), zero
    def basis(b):
        f = rng.choice([0.0, 0.3, 1.0, 1.0, 1.7], n) * rng.uniform(0.5, 1.0, n)
        f = np.where(rng.integers(0, 4, n) == 0, 1.0, f)
        return b * f
    return b1, basis(b1), b2, basis(b2)


def _same(got, exp, what):
    got, exp = np.asarray(got, dtype=np.float64), np.asarray(exp, dtype=np.float64)
    bad = np.nonzero(got.view(np.uint64) != exp.view(np.uint64))[0]
    # (+0.0 / -0.0 never differ here: every value is a sum, product or clamp of non-negative amounts)
    assert bad.size == 0, (what, bad[:5].tolist(), got[bad[:5]].tolist(), exp[bad[:5]].tolist())


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_path_forms_are_bit_identical_to_the_reference_arithmetic(oracle, seed):
    rng = np.random.default_rng(seed)
    n = 4000
    for _ in range(12):
        cfg, p = _params(rng)
        use1, r1 = cfg["inv1_use_realized_gains_tax_system"], cfg["inv1_realized_gains_tax_rate"]
        use2, r2 = cfg["inv2_use_realized_gains_tax_system"], cfg["inv2_realized_gains_tax_rate"]
        b1, c1, b2, c2 = _states(rng, n)

        # withdrawals: targets from zero to more than the asset holds
        t1 = np.where(rng.integers(0, 5, n) == 0, 0.0, b1 * rng.uniform(0, 1.4, n))
        t2 = np.where(rng.integers(0, 5, n) == 0, 0.0, b2 * rng.uniform(0, 1.4, n) + rng.uniform(0, 2e-6, n))
        got = E.eval_helper_host(N.MCR_HELPER_WITHDRAW2_PATH, p, np.column_stack((b1, c1, t1, b2, c2, t2)))
        exp = np.array([oracle.withdraw(b1[i], c1[i], t1[i], use1, r1) + oracle.withdraw(b2[i], c2[i], t2[i], use2, r2) for i in range(n)])
        _same(got.ravel(), exp.ravel(), ("withdraw2", cfg))

        got = E.eval_helper_host(N.MCR_HELPER_NLV2_PATH, p, np.column_stack((b1, c1, b2, c2)))
        exp = np.array([(oracle.nlv(b1[i], c1[i], use1, r1), oracle.nlv(b2[i], c2[i], use2, r2)) for i in range(n)])
        _same(got.ravel(), exp.ravel(), ("nlv2", cfg))

        got = E.eval_helper_host(N.MCR_HELPER_REBALANCE_PATH, p, np.column_stack((b1, c1, b2, c2)))
        exp = np.array([oracle.rebalance(p, b1[i], c1[i], b2[i], c2[i]) for i in range(n)])
        _same(got.ravel(), exp.ravel(), ("rebalance", cfg))

        # annual settlement: gains from losses to more than the portfolio is worth
        g1 = (b1 + 1.0) * rng.uniform(-0.5, 2.5, n)
        g2 = (b2 + 1.0) * rng.uniform(-0.5, 2.5, n)
        got = E.eval_helper_host(N.MCR_HELPER_ANNUAL_TAX_PATH, p, np.column_stack((b1, c1, b2, c2, g1, g2)))
        exp = np.array([[*o[:4], 1.0 if o[4] else 0.0] for o in (oracle.annual_tax(p, b1[i], c1[i], b2[i], c2[i], g1[i], g2[i]) for i in range(n))])
        _same(got.ravel(), exp.ravel(), ("annual_tax", cfg))


def _close(got, exp, scale, what, rel=2e-14, abs_tol=4e-6):
    """|got - exp| <= rel x the state's money scale (a few hundred roundings' worth), or <= abs_tol for the DUST states this
    test feeds on purpose (balances of 0 .. 3e-6 dollars around the reference's 1e-6 threshold, `_money`): with a total
    liquidation value below 1e-6 the reference splits the target by allocation weight instead of capacity share
    (simulation.py:750-755), so whether such a balance is sold out or kept differs — by the balance itself."""
    got, exp = np.asarray(got, dtype=np.float64), np.asarray(exp, dtype=np.float64)
    err = np.abs(got - exp)
    ok = (err <= abs_tol) | (err <= rel * scale)
    bad = np.nonzero(~ok)[0]
    assert bad.size == 0, (what, bad[:5].tolist(), got[bad[:5]].tolist(), exp[bad[:5]].tolist(), scale[bad[:5]].tolist())


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_month_forms_the_kernel_runs_match_the_reference_arithmetic(oracle, seed):
    """The month AS THE PATH KERNEL RUNS IT (MCR_HELPER_WITHDRAW_MONTH / MCR_HELPER_REBALANCE_MONTH: the closed form of
    csrc/mcr_device.h — both assets sell target / capacity, one-quotient rebalance, FMAs, uncorrected reciprocals; the exact path
    forms for parameter blocks with a realized-gains rate above 1 - 1e-6) against the oracle's restatement of the reference's
    sequence (simulation.py:726-796: two liquidation values, the proportional split, two withdrawals; then the rebalance) on
    random reachable states: within 2e-14 of the state's money scale, or within the reference's own 1e-6 dust threshold."""
    rng = np.random.default_rng(seed)
    n = 4000
    kinds = set()
    for _ in range(16):
        cfg, p = _params(rng)
        use1, r1 = cfg["inv1_use_realized_gains_tax_system"], cfg["inv1_realized_gains_tax_rate"]
        use2, r2 = cfg["inv2_use_realized_gains_tax_system"], cfg["inv2_realized_gains_tax_rate"]
        a1 = cfg["allocation_inv1_pct"]
        kinds.add(bool((use1 and r1 > 1 - 1e-6) or (use2 and r2 > 1 - 1e-6)))
        b1, c1, b2, c2 = _states(rng, n)
        total = b1 + b2
        need = np.where(rng.integers(0, 6, n) == 0, 0.0, total * rng.uniform(0, 1.3, n) + rng.uniform(0, 3e-6, n))
        got = E.eval_helper_host(N.MCR_HELPER_WITHDRAW_MONTH, p, np.column_stack((b1, c1, b2, c2, need)))
        exp = np.empty((n, 6))
        for i in range(n):
            cap1, cap2 = oracle.nlv(b1[i], c1[i], use1, r1), oracle.nlv(b2[i], c2[i], use2, r2)
            cap = cap1 + cap2
            target = max(0.0, min(need[i], cap))
            prop1 = cap1 / cap if cap > 1e-6 else a1
            w1 = oracle.withdraw(b1[i], c1[i], target * prop1, use1, r1)
            w2 = oracle.withdraw(b2[i], c2[i], target * (1.0 - prop1), use2, r2)
            exp[i] = (w1[0], w1[1], w2[0], w2[1], w1[2] + w2[2], w1[3] + w2[3])
        scale = np.maximum(np.maximum(total, c1 + c2), 1.0)
        for k, name in enumerate(("b1", "c1", "b2", "c2", "gross", "net")):
            _close(got[:, k], exp[:, k], scale, ("withdraw month", name, cfg))
        got = E.eval_helper_host(N.MCR_HELPER_REBALANCE_MONTH, p, np.column_stack((b1, c1, b2, c2)))
        exp = np.array([oracle.rebalance(p, b1[i], c1[i], b2[i], c2[i]) for i in range(n)])
        for k, name in enumerate(("b1", "c1", "b2", "c2")):
            _close(got[:, k], exp[:, k], scale, ("rebalance month", name, cfg))
    assert kinds == {True, False}        # both the closed form and the exact-month parameter blocks were exercised
