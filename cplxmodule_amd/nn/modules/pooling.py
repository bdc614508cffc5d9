"""Complex max pooling layers (reference: cplxmodule/nn/modules/pooling.py)."""
from ... import cplx
from .base import CplxToCplx


class CplxMaxPoolNd(CplxToCplx):
    def __init__(self, kernel_size, stride=None, padding=0, dilation=1, return_indices=False,
                 ceil_mode=False):
        super().__init__()
        self.kernel_size = kernel_size
        self.stride = stride if (stride is not None) else kernel_size
        self.padding, self.dilation = padding, dilation
        self.return_indices, self.ceil_mode = return_indices, ceil_mode

    def extra_repr(self):
        return ("kernel_size={kernel_size}, stride={stride}, padding={padding}"
                ", dilation={dilation}, ceil_mode={ceil_mode}".format(**self.__dict__))


class CplxMaxPool1d(CplxMaxPoolNd):
    def forward(self, input):
        return cplx.max_pool1d(input, self.kernel_size, self.stride, self.padding, self.dilation,
                               self.ceil_mode)


class CplxMaxPool2d(CplxMaxPoolNd):
    def forward(self, input):
        return cplx.max_pool2d(input, self.kernel_size, self.stride, self.padding, self.dilation,
                               self.ceil_mode)
