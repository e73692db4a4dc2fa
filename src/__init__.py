"""Drop-in for the reference's `src` package (src/__init__.py:4-7)."""
from .metrics import MetricsCalculator
from .pipeline import FastEditor

__all__ = ["FastEditor", "MetricsCalculator"]
