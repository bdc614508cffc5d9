"""Complex max pooling layers: in each window the element of largest modulus
survives with both of its parts.  API counterpart of cplxmodule/nn/modules/pooling.py; the work is
one kernel (csrc/pool.hip) with a deterministic gather backward."""
from ... import cplx
from .base import CplxToCplx

_FIELDS = ("kernel_size", "stride", "padding", "dilation", "ceil_mode")


class _CplxMaxPool(CplxToCplx):
    """Holds torch's max-pool hyper-parameters; subclasses name the functional that applies them."""
    _pool = None

    def __init__(self, kernel_size, stride=None, padding=0, dilation=1, return_indices=False,
                 ceil_mode=False):
        super().__init__()
        self.kernel_size, self.padding, self.dilation = kernel_size, padding, dilation
        self.stride = kernel_size if stride is None else stride
        self.return_indices, self.ceil_mode = return_indices, ceil_mode

    def forward(self, input):
        return type(self)._pool(input, *(getattr(self, f) for f in _FIELDS))

    def extra_repr(self):
        return ", ".join(f"{f}={getattr(self, f)}" for f in _FIELDS)


class CplxMaxPool1d(_CplxMaxPool):
    _pool = staticmethod(cplx.max_pool1d)


class CplxMaxPool2d(_CplxMaxPool):
    _pool = staticmethod(cplx.max_pool2d)


class CplxMaxPool3d(_CplxMaxPool):
    _pool = staticmethod(cplx.max_pool3d)
