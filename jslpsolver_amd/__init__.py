"""jslpsolver_amd -- MI355X-native dense-tableau simplex engine behind jsLPSolver's `Solve(model)` API.

Only the hot path lives here: csrc/ (HIP kernels for gfx950 + the C ABI of
include/jslp_engine.h) and the thin host mirror of the reference interface for that path.
"""
from .engine import Tableau, pivot_digest  # noqa: F401
from .model import Model, UnsupportedModel  # noqa: F401
from .solver import Solve  # noqa: F401

__all__ = ["Tableau", "Model", "Solve", "UnsupportedModel", "pivot_digest"]
