"""Host-side derivation of the kernel parameter block from a ``Config``.

Restates reference backend/simulation.py:14-29 (``arithmetic_to_log_params``) and the
part of ``RetirementMonteCarloSimulator.__init__`` that derives the lognormal parameters
(:156-170).  The result is the POD ``mcr_params`` of include/mcr.h.
"""

from __future__ import annotations

import math
from typing import Tuple

from ._native import McrParams
from .config import Config


def arithmetic_to_log_params(mean: float, vol: float) -> Tuple[float, float]:
    """Arithmetic annual mean/vol -> lognormal (mu, sigma) with E[gross] = 1 + mean.

    simulation.py:14-29: sigma = sqrt(ln(1 + vol^2/(1+mean)^2)), mu = ln(1+mean) - sigma^2/2;
    vol == 0 degenerates to (ln(1+mean), 0); ``ValueError`` for mean <= -1 or vol < 0.
    """
    if mean <= -1.0:
        raise ValueError("Arithmetic mean must be greater than -100%.")
    if vol < 0:
        raise ValueError("Volatility cannot be negative.")
    if vol == 0:
        return math.log(1.0 + mean), 0.0
    one_plus_mean = 1.0 + mean
    sigma_log = math.sqrt(math.log(1.0 + (vol**2) / (one_plus_mean**2)))
    mu_log = math.log(one_plus_mean) - 0.5 * sigma_log**2
    return mu_log, sigma_log


def params_from_config(cfg: Config) -> McrParams:
    """Fill the ``mcr_params`` block (raw Config scalars + derived log parameters)."""
    p = McrParams()
    p.initial_balance = cfg.initial_balance
    p.monthly_contribution = cfg.monthly_contribution
    p.contribution_growth_rate_annual = cfg.contribution_growth_rate_annual
    p.monthly_expenses = cfg.monthly_expenses
    p.current_age = cfg.current_age
    p.allocation_inv1_pct = cfg.allocation_inv1_pct
    p.inv1_annual_tax_on_gains_rate = cfg.inv1_annual_tax_on_gains_rate
    p.inv1_realized_gains_tax_rate = cfg.inv1_realized_gains_tax_rate
    p.inv2_annual_tax_on_gains_rate = cfg.inv2_annual_tax_on_gains_rate
    p.inv2_realized_gains_tax_rate = cfg.inv2_realized_gains_tax_rate
    p.inv1_mu_log, p.inv1_sigma_log = arithmetic_to_log_params(
        cfg.inv1_returns_mean, cfg.inv1_returns_volatility
    )
    p.inf_mu_log, p.inf_sigma_log = arithmetic_to_log_params(
        cfg.inflation_rate_mean, cfg.inflation_rate_volatility
    )
    p.prem_mu_log, p.prem_sigma_log = arithmetic_to_log_params(
        cfg.inv2_premium_over_inflation_mean, cfg.inv2_premium_over_inflation_volatility
    )
    p.equity_inflation_rho = cfg.equity_inflation_correlation
    p.retirement_years = cfg.retirement_years
    p.inv1_use_realized_gains_tax_system = int(cfg.inv1_use_realized_gains_tax_system)
    p.inv2_use_realized_gains_tax_system = int(cfg.inv2_use_realized_gains_tax_system)
    # other_income_streams: any length, list order preserved (config.py:99; simulation.py:602-621 loops over all of them)
    p.set_streams(
        (s.monthly_amount_today, s.start_at_age, s.tax_rate, s.duration_years, s.inflation_indexed)
        for s in cfg.other_income_streams
    )
    return p
