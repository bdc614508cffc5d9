"""cplxmodule_amd -- MI355X-native complex-valued layers and variational dropout.

Drop-in for the hot path of ivannz/cplxmodule (same nn.Module API and state-dict layout):
`CplxLinear`, `CplxConv2d`, `CplxBatchNorm{1,2,3}d`, `nn.relevance.{LinearVD, LinearARD,
CplxLinearVD, CplxLinearARD, Conv2dVD/ARD, CplxConv2dVD/ARD}` and the penalty / relevance
walkers, all running on hand-written HIP kernels for gfx950 (libcplxamd.so).
"""
__version__ = "0.1.0"

from .cplx import Cplx, from_real, to_real  # noqa: F401
from . import nn  # noqa: F401
from .x3 import fp32_mode, get_fp32_mode, set_fp32_mode  # noqa: F401  (float32 products: 'auto' | 'x3' | 'exact')
