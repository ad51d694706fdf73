"""Configuration surface of the engine.

Field-for-field mirror of the reference's pydantic models (backend/config.py:12-126):
same names, alias (`scenario` -> `Nickname`), bounds, defaults and `ValueError`
(pydantic ``ValidationError``) behaviour, so a ``Config(**json_dict)`` written for the
reference validates identically here.  The two soft validators only warn
(backend/config.py:103-122).
"""

from __future__ import annotations

import json
import os
from typing import Any, Dict, List, Optional

from pydantic import BaseModel, Field, ValidationInfo, field_validator

from ._logging import logger


class ConfigurationError(Exception):
    """The configuration file is missing or is not valid JSON (backend/config.py:8-9)."""


class OtherIncomeStreamConfig(BaseModel):
    """One extra income stream paid during retirement (backend/config.py:12-47)."""

    name: str = Field(..., description="Label of the stream, e.g. 'Pension'.")
    monthly_amount_today: float = Field(
        ..., ge=0, description="Monthly amount in T=0 (today's) money."
    )
    start_at_age: float = Field(
        ...,
        ge=0,
        le=120,
        description="Eligibility age; paid from max(retirement age, this age).",
    )
    duration_years: Optional[int] = Field(
        None, ge=0, description="Years of payments once started; None = for life."
    )
    inflation_indexed: bool = Field(
        True,
        description="True: tracks the price level from T=0. False: nominal amount "
        "is frozen at the first payment.",
    )
    tax_rate: float = Field(..., ge=0.0, le=1.0, description="Tax on this income.")


class Config(BaseModel):
    """Scenario definition (backend/config.py:48-126)."""

    Nickname: str = Field("DefaultScenario", alias="scenario")
    initial_balance: float = Field(..., ge=0)
    monthly_contribution: float = Field(..., ge=0)
    contribution_growth_rate_annual: float = Field(0.0, ge=0)
    monthly_expenses: float = Field(..., ge=0, description="In T=0 money.")
    current_age: float = Field(..., ge=0, le=120)
    retirement_years: int = Field(..., gt=0)

    allocation_inv1_pct: float = Field(..., ge=0.0, le=1.0)
    inv1_returns_mean: float = Field(..., gt=-1.0)
    inv1_returns_volatility: float = Field(..., ge=0.0)
    inv1_annual_tax_on_gains_rate: float = Field(..., ge=0.0, le=1.0)
    inv1_realized_gains_tax_rate: float = Field(0.0, ge=0.0, le=1.0)
    inv1_use_realized_gains_tax_system: bool = Field(False)

    inv2_premium_over_inflation_mean: float = Field(..., gt=-1.0)
    inv2_premium_over_inflation_volatility: float = Field(..., ge=0.0)
    inv2_annual_tax_on_gains_rate: float = Field(..., ge=0.0, le=1.0)
    inv2_realized_gains_tax_rate: float = Field(0.0, ge=0.0, le=1.0)
    inv2_use_realized_gains_tax_system: bool = Field(True)

    inflation_rate_mean: float = Field(..., gt=-1.0)
    inflation_rate_volatility: float = Field(..., ge=0.0)
    equity_inflation_correlation: float = Field(0.0, ge=-1.0, le=1.0)

    num_simulations_main: int = Field(..., gt=0)
    num_simulations_search: int = Field(..., gt=0)
    target_probability: float = Field(..., ge=0.0, le=100.0)
    starting_working_months_search: int = Field(..., ge=0)
    seed: Optional[int] = Field(None, ge=0)
    #: Kept for drop-in compatibility; the HIP engine does not use host worker processes.
    num_processes: Optional[int] = Field(1, ge=1)

    other_income_streams: List[OtherIncomeStreamConfig] = Field(default_factory=list)

    model_config = {"validate_by_name": True, "validate_assignment": True}

    @field_validator("inflation_rate_volatility")
    @classmethod
    def _warn_high_inflation_vol(cls, v: float, info: ValidationInfo) -> float:
        if v > 0.05:
            logger.warning(
                "Inflation volatility (%.1f%%) is relatively high for scenario '%s'."
                % (v * 100, info.data.get("Nickname", "N/A"))
            )
        return v

    @field_validator("inv1_returns_volatility")
    @classmethod
    def _warn_low_equity_vol(cls, v: float, info: ValidationInfo) -> float:
        if v < 0.05:
            logger.warning(
                "Equity (Inv1) volatility (%.1f%%) is unusually low for scenario '%s'."
                % (v * 100, info.data.get("Nickname", "N/A"))
            )
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
    except Exception as exc:  # pragma: no cover - I/O errors
        raise ConfigurationError(
            f"Unexpected error reading config file '{file_path}': {exc}"
        ) from exc
