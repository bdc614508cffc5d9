from .base import CplxToCplx, CplxParameter  # noqa: F401
from .linear import CplxLinear  # noqa: F401
from .conv import CplxConv2d  # noqa: F401
from .batchnorm import CplxBatchNorm1d, CplxBatchNorm2d, CplxBatchNorm3d  # noqa: F401
