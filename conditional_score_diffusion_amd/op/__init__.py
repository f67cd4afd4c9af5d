"""``op`` package of the reference (op/__init__.py): the same three names on the HIP kernels of libcsd_hip.so instead of the JIT-built
CUDA extensions op/upfirdn2d_kernel.cu and op/fused_bias_act_kernel.cu."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu  # noqa: F401
from .upfirdn2d import upfirdn2d  # noqa: F401
