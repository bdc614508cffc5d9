from .base import CplxToCplx, CplxParameter  # noqa: F401
from .linear import CplxLinear, CplxBilinear  # noqa: F401
from .linear import CplxReal, CplxImag, CplxIdentity, CplxPhaseShift  # noqa: F401
from .container import CplxSequential  # noqa: F401
from .conv import CplxConv1d, CplxConv2d, CplxConv3d  # noqa: F401
from .conv import CplxConvTranspose1d, CplxConvTranspose2d  # noqa: F401
from .batchnorm import CplxBatchNorm1d, CplxBatchNorm2d, CplxBatchNorm3d  # noqa: F401
from .casting import AsTypeCplx, TensorToCplx, CplxToTensor  # noqa: F401
from .casting import InterleavedRealToCplx, ConcatenatedRealToCplx  # noqa: F401
from .casting import CplxToInterleavedReal, CplxToConcatenatedReal  # noqa: F401
from .casting import CplxToInterleavedReal as CplxToReal  # noqa: F401
from .casting import InterleavedRealToCplx as RealToCplx  # noqa: F401
from .activation import CplxModReLU, CplxAdaptiveModReLU  # noqa: F401
from .casting import CplxModulus, CplxAngle  # noqa: F401
from .extra import CplxDropout  # noqa: F401
from .pooling import CplxMaxPool1d, CplxMaxPool2d, CplxMaxPool3d  # noqa: F401
