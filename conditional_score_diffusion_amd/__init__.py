"""conditional_score_diffusion_amd - MI355X (gfx950)-native hot path of
GBATZOLIS/conditional_score_diffusion: score-network forward + reverse-SDE predictor-corrector
sampling, as hand-written HIP kernels behind a C ABI (include/csd.h), mirrored on the host by the
reference's own plugin surface (models.utils registry, sde_lib, sampling.*).
"""
__version__ = '0.1.0'
