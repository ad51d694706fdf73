"""Constants shared by the host code and (hard-coded identically) by the HIP kernels.

Mirrors reference backend/constants.py:1-3.
"""

MONTHS_PER_YEAR: int = 12
#: Absolute dollar threshold used by every balance comparison on the path.
SMALL_EPSILON: float = 1e-6
