"""CplxLinear: y = x W^T + b in C, on the complex MFMA GEMM (cplxmodule/nn/modules/linear.py:24-64);
CplxBilinear: y = x1^H W x2 + b (linear.py:67-117)."""
import math

import torch

from .base import CplxToCplx, CplxParameter, BaseCplxToReal
from .. import init
from ... import cplx


class CplxLinear(CplxToCplx):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = CplxParameter(cplx.Cplx.empty(out_features, in_features))
        if bias:
            self.bias = CplxParameter(cplx.Cplx.empty(out_features))
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
        # late-bound module attribute, exactly like the reference's operator seam (cplx.py:698)
        return cplx.linear(input, self.weight, self.bias)

    def extra_repr(self):
        return (f"in_features={self.in_features}, out_features={self.out_features}, "
                f"bias={self.bias is not None}")


class CplxBilinear(CplxToCplx):
    """(u, v) -> (u^H A_o v + b_o)_o, or u^T A_o v with `conjugate=False`; weight [out, in1, in2]."""

    def __init__(self, in1_features, in2_features, out_features, bias=True, conjugate=True):
        super().__init__()
        self.in1_features, self.in2_features = in1_features, in2_features
        self.out_features, self.conjugate = out_features, conjugate
        self.weight = CplxParameter(cplx.Cplx.empty(out_features, in1_features, in2_features))
        if bias:
            self.bias = CplxParameter(cplx.Cplx.empty(out_features))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    reset_parameters = CplxLinear.reset_parameters

    def forward(self, input1, input2):
        return cplx.bilinear(input1, input2, self.weight, self.bias, self.conjugate)

    def extra_repr(self):
        return (f"in1_features={self.in1_features}, in2_features={self.in2_features}, "
                f"out_features={self.out_features}, bias={self.bias is not None}, "
                f"conjugate={self.conjugate}")


class CplxIdentity(torch.nn.Identity, CplxToCplx):
    pass


class CplxReal(BaseCplxToReal):
    def forward(self, input):
        return input.real


class CplxImag(BaseCplxToReal):
    def forward(self, input):
        return input.imag


class CplxPhaseShift(CplxToCplx):
    """z -> z exp(i phi) with a learnable phase of shape `dim`, broadcast against the input
    (linear.py:120-143); initial phases are N(0, 0.02^2)."""

    def __init__(self, *dim):
        super().__init__()
        self.phi = torch.nn.Parameter(torch.randn(*dim) * 0.02)

    def forward(self, input):
        return cplx.phaseshift(input, self.phi)
