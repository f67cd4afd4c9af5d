"""op/upfirdn2d.py of the reference: ``upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0))`` with the reference's autograd structure
(backward = the same FIR kernel flipped, up/down swapped, pads g_pad; double backward), on csd_upfirdn2d."""
from ..ops import upfirdn2d  # noqa: F401
