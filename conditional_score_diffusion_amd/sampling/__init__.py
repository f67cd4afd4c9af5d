from . import predictors, correctors  # noqa: F401  (populate the registries)
from .conditional import get_conditional_sampling_fn  # noqa: F401
from .unconditional import get_sampling_fn  # noqa: F401
