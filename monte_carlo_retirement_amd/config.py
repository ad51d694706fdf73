"""Configuration surface of the engine.

Schema-compatible with the reference's pydantic models (backend/config.py:12-126): the same
field names, alias (``scenario`` -> ``Nickname``), bounds, defaults, ``validate_assignment`` and
``ValueError`` (pydantic ``ValidationError``) behaviour, so ``Config(**json_dict)`` written for the
reference validates identically here (pinned by tests/test_simulator_cpu.py).  Bounds are expressed
once as constrained types; the two soft validators only warn (backend/config.py:103-122).
"""

from __future__ import annotations

import json
import os
from typing import Annotated, Any, Dict, List, Optional

from pydantic import BaseModel, Field, ValidationInfo, field_validator

from ._logging import logger

# ---- constrained scalar types (the bounds of backend/config.py, stated once) ---------------------
Money = Annotated[float, Field(ge=0)]                  # a non-negative amount / rate of growth
Fraction = Annotated[float, Field(ge=0.0, le=1.0)]     # tax rates, allocation weight
GrossMean = Annotated[float, Field(gt=-1.0)]           # arithmetic mean return: 1 + mean must stay positive
Volatility = Annotated[float, Field(ge=0.0)]
Age = Annotated[float, Field(ge=0, le=120)]
PositiveInt = Annotated[int, Field(gt=0)]
Count = Annotated[int, Field(ge=0)]


class ConfigurationError(Exception):
    """The configuration file is missing or is not valid JSON (backend/config.py:8-9)."""


class OtherIncomeStreamConfig(BaseModel):
    """One extra income stream paid during retirement (backend/config.py:12-47).

    ``monthly_amount_today`` is in T=0 money; payments start at max(retirement age, ``start_at_age``)
    and last ``duration_years`` (None = for life).  ``inflation_indexed`` = True tracks the price level
    from T=0; False freezes the nominal amount at the first payment."""

    name: str
    monthly_amount_today: Money
    start_at_age: Age
    duration_years: Optional[Count] = None
    inflation_indexed: bool = True
    tax_rate: Fraction


class Config(BaseModel):
    """Scenario definition (backend/config.py:48-126)."""

    model_config = {"validate_by_name": True, "validate_assignment": True}

    Nickname: str = Field("DefaultScenario", alias="scenario")

    # household
    initial_balance: Money
    monthly_contribution: Money
    contribution_growth_rate_annual: Money = 0.0
    monthly_expenses: Money                      # in T=0 money
    current_age: Age
    retirement_years: PositiveInt

    # investment 1 (equity-like): arithmetic annual mean / volatility, tax treatment
    allocation_inv1_pct: Fraction
    inv1_returns_mean: GrossMean
    inv1_returns_volatility: Volatility
    inv1_annual_tax_on_gains_rate: Fraction
    inv1_realized_gains_tax_rate: Fraction = 0.0
    inv1_use_realized_gains_tax_system: bool = False

    # investment 2 (inflation + premium)
    inv2_premium_over_inflation_mean: GrossMean
    inv2_premium_over_inflation_volatility: Volatility
    inv2_annual_tax_on_gains_rate: Fraction
    inv2_realized_gains_tax_rate: Fraction = 0.0
    inv2_use_realized_gains_tax_system: bool = True

    # inflation
    inflation_rate_mean: GrossMean
    inflation_rate_volatility: Volatility
    equity_inflation_correlation: Annotated[float, Field(ge=-1.0, le=1.0)] = 0.0

    # simulation control
    num_simulations_main: PositiveInt
    num_simulations_search: PositiveInt
    target_probability: Annotated[float, Field(ge=0.0, le=100.0)]
    starting_working_months_search: Count
    seed: Optional[Count] = None
    #: accepted for drop-in compatibility; the HIP engine has no host worker pool
    num_processes: Optional[Annotated[int, Field(ge=1)]] = 1

    other_income_streams: List[OtherIncomeStreamConfig] = Field(default_factory=list)

    @field_validator("inflation_rate_volatility", "inv1_returns_volatility")
    @classmethod
    def _soft_volatility_checks(cls, v: float, info: ValidationInfo) -> float:
        scenario = info.data.get("Nickname", "N/A")
        if info.field_name == "inflation_rate_volatility" and v > 0.05:
            logger.warning(f"Inflation volatility ({v * 100:.1f}%) is relatively high for scenario '{scenario}'.")
        if info.field_name == "inv1_returns_volatility" and v < 0.05:
            logger.warning(f"Equity (Inv1) volatility ({v * 100:.1f}%) is unusually low for scenario '{scenario}'.")
        return v

    @property
    def allocation_inv2_pct(self) -> float:
        return 1.0 - self.allocation_inv1_pct


def load_config_from_json(file_path: str) -> Dict[str, Any]:
    """Read a scenario JSON into a dict (backend/config.py:129-144)."""
    if not os.path.exists(file_path):
        raise ConfigurationError(f"Configuration file not found at: {file_path}")
    try:
        with open(file_path, "r", encoding="utf-8") as fh:
            return json.load(fh)
    except json.JSONDecodeError as exc:
        raise ConfigurationError(f"Error parsing JSON file '{file_path}': {exc}") from exc
    except OSError as exc:  # pragma: no cover - I/O errors
        raise ConfigurationError(f"Unexpected error reading config file '{file_path}': {exc}") from exc
