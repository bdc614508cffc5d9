"""CplxConv1d / CplxConv2d / CplxConv3d: complex cross-correlation layers
(cplxmodule/nn/modules/conv.py:11-247)."""
import math

from torch.nn.modules.utils import _pair, _single, _triple

from .base import CplxToCplx, CplxParameter
from .. import init
from ... import cplx


class CplxConv2d(CplxToCplx):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, padding_mode="zeros"):
        super().__init__()
        if in_channels % groups != 0:
            raise ValueError("in_channels must be divisible by groups")
        if out_channels % groups != 0:
            raise ValueError("out_channels must be divisible by groups")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.transposed, self.output_padding = False, _pair(0)
        self.groups, self.padding_mode = groups, padding_mode
        self.weight = CplxParameter(
            cplx.Cplx.empty(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = CplxParameter(cplx.Cplx.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        weight = self.weight
        init.cplx_kaiming_uniform_(weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init.get_fans(weight)
            bound = 1 / math.sqrt(fan_in)
            init.cplx_uniform_independent_(self.bias, -bound, bound)

    def forward(self, input):
        from ... import conv
        # (cplx.conv2d with the layer's training state: an evaluation-mode layer never runs the conv -> batch-norm
        #  moments epilogue, armed or not)
        return conv.cplx_conv2d(input, self.weight, self.bias, self.stride, self.padding,
                                self.dilation, self.groups, self.padding_mode, training=self.training)

    def extra_repr(self):
        s = (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, "
             f"stride={self.stride}")
        if any(self.padding):
            s += f", padding={self.padding}"
        if any(d != 1 for d in self.dilation):
            s += f", dilation={self.dilation}"
        if self.groups != 1:
            s += f", groups={self.groups}"
        if self.bias is None:
            s += ", bias=False"
        if self.padding_mode != "zeros":
            s += f", padding_mode='{self.padding_mode}'"
        return s


class CplxConv1d(CplxConv2d):
    """[B, C, L] complex convolution; weight [out, in / groups, k].  Runs the 2-d kernels on a
    height-1 image (cplx.conv1d)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, padding_mode="zeros"):
        CplxToCplx.__init__(self)
        if in_channels % groups != 0:
            raise ValueError("in_channels must be divisible by groups")
        if out_channels % groups != 0:
            raise ValueError("out_channels must be divisible by groups")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _single(kernel_size), _single(stride)
        self.padding, self.dilation = _single(padding), _single(dilation)
        self.transposed, self.output_padding = False, _single(0)
        self.groups, self.padding_mode = groups, padding_mode
        self.weight = CplxParameter(
            cplx.Cplx.empty(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = CplxParameter(cplx.Cplx.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def forward(self, input):
        return cplx.conv1d(input, self.weight, self.bias, self.stride, self.padding,
                           self.dilation, self.groups, self.padding_mode)


class CplxConv3d(CplxConv2d):
    """[B, C, D, H, W] complex convolution; weight [out, in / groups, kd, kh, kw] (cplx.conv3d)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, padding_mode="zeros"):
        CplxToCplx.__init__(self)
        if in_channels % groups != 0:
            raise ValueError("in_channels must be divisible by groups")
        if out_channels % groups != 0:
            raise ValueError("out_channels must be divisible by groups")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _triple(kernel_size), _triple(stride)
        self.padding, self.dilation = _triple(padding), _triple(dilation)
        self.transposed, self.output_padding = False, _triple(0)
        self.groups, self.padding_mode = groups, padding_mode
        self.weight = CplxParameter(
            cplx.Cplx.empty(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = CplxParameter(cplx.Cplx.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def forward(self, input):
        return cplx.conv3d(input, self.weight, self.bias, self.stride, self.padding,
                           self.dilation, self.groups, self.padding_mode)


class CplxConvTranspose2d(CplxConv2d):
    """Complex transposed convolution; weight [in, out / groups, kh, kw].  Constructor as the
    reference's (conv.py:348-380; note its default `bias=None`, i.e. no bias unless asked for).
    `forward(input, output_size=None)` resolves the output padding like torch's ConvTranspose2d
    (the reference borrows torch's private helper, whose signature changed in torch 2)."""
    _nd = 2
    _fn = staticmethod(cplx.conv_transpose2d)

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 output_padding=0, groups=1, bias=None, padding_mode="zeros"):
        CplxToCplx.__init__(self)
        if padding_mode not in ("zeros", "circular"):
            raise ValueError(f'Only "zeros" or "circular" padding mode are supported by `{type(self).__name__}`')
        if in_channels % groups != 0:
            raise ValueError("in_channels must be divisible by groups")
        if out_channels % groups != 0:
            raise ValueError("out_channels must be divisible by groups")
        tup = _pair if self._nd == 2 else _single
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = tup(kernel_size), tup(stride)
        self.padding, self.dilation = tup(padding), tup(dilation)
        self.transposed, self.output_padding = True, tup(output_padding)
        self.groups, self.padding_mode = groups, padding_mode
        self.weight = CplxParameter(
            cplx.Cplx.empty(in_channels, out_channels // groups, *self.kernel_size))
        if bias:
            self.bias = CplxParameter(cplx.Cplx.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def _resolve_output_padding(self, input, output_size):
        if output_size is None:
            return self.output_padding
        want = list(output_size)[-self._nd:]
        pads = []
        for n, size in enumerate(want):
            L = input.shape[2 + n]
            lo = (L - 1) * self.stride[n] - 2 * self.padding[n] + self.dilation[n] * (self.kernel_size[n] - 1) + 1
            hi = lo + max(self.stride[n], self.dilation[n]) - 1
            if not lo <= size <= hi:
                raise ValueError(f"requested an output size of {tuple(want)}, but valid sizes of dim {n} "
                                 f"range from {lo} to {hi}")
            pads.append(size - lo)
        return tuple(pads)

    def forward(self, input, output_size=None):
        return type(self)._fn(input, self.weight, self.bias, self.stride, self.padding,
                              self._resolve_output_padding(input, output_size), self.groups,
                              self.dilation, self.padding_mode)

    def extra_repr(self):
        s = super().extra_repr()
        if any(self.output_padding):
            s += f", output_padding={self.output_padding}"
        return s


class CplxConvTranspose1d(CplxConvTranspose2d):
    _nd = 1
    _fn = staticmethod(cplx.conv_transpose1d)
