"""CplxSequential (cplxmodule/nn/modules/container.py:9-34): torch.nn.Sequential restricted to
layers that take and return `Cplx`."""
from collections import OrderedDict

import torch

from .base import CplxToCplx, BaseCplxToReal, BaseRealToCplx


def is_from_cplx(module):
    """accepts a Cplx input (an instance, a class, or a Sequential judged by its first layer)"""
    if isinstance(module, type):
        return issubclass(module, (CplxToCplx, BaseCplxToReal))
    if isinstance(module, torch.nn.Sequential) and not isinstance(module, CplxToCplx):
        return len(module) > 0 and is_from_cplx(module[0])
    return isinstance(module, (CplxToCplx, BaseCplxToReal))


def is_to_cplx(module):
    """returns a Cplx output (a Sequential is judged by its last layer)"""
    if isinstance(module, type):
        return issubclass(module, (CplxToCplx, BaseRealToCplx))
    if isinstance(module, torch.nn.Sequential) and not isinstance(module, CplxToCplx):
        return len(module) > 0 and is_to_cplx(module[-1])
    return isinstance(module, (CplxToCplx, BaseRealToCplx))


def is_cplx_to_cplx(module):
    return is_from_cplx(module) and is_to_cplx(module)


class CplxSequential(torch.nn.Sequential, CplxToCplx):
    def __init__(self, *args):
        named = args[0].items() if len(args) == 1 and isinstance(args[0], OrderedDict) else enumerate(args)
        rejected = [str(name) for name, layer in named if not is_cplx_to_cplx(layer)]
        if rejected:
            raise TypeError(f"Only complex-to-complex modules can be used in {type(self).__name__}. "
                            f"The following modules failed: {rejected}.")
        super().__init__(*args)
