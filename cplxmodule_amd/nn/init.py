"""Parameter initialisers used by the default state of the complex layers.

Restates the observable behaviour of cplxmodule/nn/init.py:12-30, 50-56, 126-130, including the
reference's fan quirk: for 2-d weights `get_fans` reports fan_in = shape[0] (the OUTPUT size),
which is what the bias bound of CplxLinear ends up using.
"""
import math

import torch

from ..cplx import Cplx


def get_fans(cplxtensor):
    shape = tuple(cplxtensor.shape)
    if len(shape) < 2:
        raise ValueError("Fan in and fan out can not be computed for tensor with "
                         "fewer than 2 dimensions.")
    if len(shape) == 2:
        return shape[0], shape[1]
    field = 1
    for s in shape[2:]:
        field *= s
    return shape[1] * field, shape[0] * field


def cplx_kaiming_uniform_(tensor, a=0.0, mode="fan_in", nonlinearity="leaky_relu"):
    """Independent Kaiming-uniform planes with the slope widened to sqrt(1 + 2 a^2), i.e. each
    plane carries half of the complex variance."""
    assert isinstance(tensor, Cplx)
    slope = math.sqrt(1 + 2 * a * a)
    for plane in (tensor.real, tensor.imag):
        torch.nn.init.kaiming_uniform_(plane, a=slope, mode=mode, nonlinearity=nonlinearity)
    return tensor


def cplx_uniform_independent_(tensor, a=0.0, b=1.0):
    for plane in (tensor.real, tensor.imag):
        torch.nn.init.uniform_(plane, a, b)
    return tensor
