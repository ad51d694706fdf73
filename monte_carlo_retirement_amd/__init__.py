"""MI355X-native Monte Carlo retirement path engine.

Drop-in for the hot path of rflamino/monte_carlo_retirement (backend/simulation.py's
per-path loop and its batch driver) behind the reference's own Python surface:
``Config`` + ``RetirementMonteCarloSimulator.run_monte_carlo_simulations``.
The paths run in hand-written HIP kernels for gfx950 (csrc/), reached through the C ABI
of include/mcr.h.
"""

from .config import Config, ConfigurationError, OtherIncomeStreamConfig, load_config_from_json
from .constants import MONTHS_PER_YEAR, SMALL_EPSILON
from .params import arithmetic_to_log_params, params_from_config

__all__ = [
    "Config",
    "ConfigurationError",
    "OtherIncomeStreamConfig",
    "load_config_from_json",
    "MONTHS_PER_YEAR",
    "SMALL_EPSILON",
    "arithmetic_to_log_params",
    "params_from_config",
]
