"""``models.ema.ExponentialMovingAverage`` of the reference (models/ema.py:14-140) - same import path; the implementation (one flat
shadow buffer, HIP update kernel) lives in ``conditional_score_diffusion_amd.optim``."""
from ..optim import ExponentialMovingAverage  # noqa: F401
