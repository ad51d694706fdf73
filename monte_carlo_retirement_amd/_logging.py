"""Logger shim: use loguru when it is installed (the reference does, backend/config.py:5),
otherwise fall back to the standard library so the engine has no hard dependency on it."""

from __future__ import annotations

import logging

try:  # pragma: no cover - depends on the environment
    from loguru import logger  # type: ignore
except Exception:  # loguru is not installed in the build image
    logger = logging.getLogger("monte_carlo_retirement_amd")

__all__ = ["logger"]
