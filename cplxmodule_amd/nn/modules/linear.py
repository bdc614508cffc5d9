"""CplxLinear: y = x W^T + b in C, on the complex MFMA GEMM (cplxmodule/nn/modules/linear.py:24-64)."""
import math

from .base import CplxToCplx, CplxParameter
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
