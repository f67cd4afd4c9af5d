from . import utils  # noqa: F401
from . import ddpm  # noqa: F401  (registers ddpm / ddpm_paired / ddpm_paired_SR3)
from . import ncsnpp  # noqa: F401  (registers ncsnpp / ncsnpp_paired)
