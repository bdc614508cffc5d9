"""Complex parameters and the base class of complex-to-complex layers.

Contract kept from cplxmodule/nn/modules/base.py:8-130, 181-208:
 * a complex parameter is a ParameterDict {real, imag}: state-dict keys `<name>.real`, `<name>.imag`;
 * reading `layer.<name>` yields a fresh `Cplx(real, imag)` view of the two Parameters;
 * loading a state dict that holds a REAL tensor under `<name>` promotes it (imag = 0);
 * `CplxToCplx[torch.nn.ReLU]` builds the split (per-plane) version of a real layer / callable.
"""
import functools

import torch

from ...cplx import Cplx


class CplxParameter(torch.nn.ParameterDict):
    def __init__(self, cplx):
        if not isinstance(cplx, Cplx):
            raise TypeError(f"`{type(self).__name__}` accepts only Cplx tensors.")
        super().__init__({"real": torch.nn.Parameter(cplx.real),
                          "imag": torch.nn.Parameter(cplx.imag)})

    @property
    def data(self):
        return Cplx(self["real"].data, self["imag"].data)

    def extra_repr(self):
        return ", ".join(map(str, self["real"].shape))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        have = [part for part in ("real", "imag") if prefix + part in state_dict]
        missing, unexpected = [], []
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing,
                                      unexpected, error_msgs)
        if not have:
            missing = [prefix[:-1]]  # the parameter as a whole is absent, not just one part
        elif len(have) == 1:
            error_msgs.append("Complex parameter requires both `.real` and `.imag` parts. "
                              f"Missing `{missing[0] if missing else prefix}`.")
            missing = []
        if strict and unexpected:
            error_msgs.append("Complex parameter disallows redundant key(s) in "
                              f"state_dict: {unexpected}.")
        missing_keys.extend(missing)
        unexpected_keys.extend(unexpected)


class CplxParameterAccessor:
    """Attribute lookup that turns a stored CplxParameter into a `Cplx` pair on access, plus
    real -> complex promotion when a state dict holds a plain tensor under the parameter's name
    (done here, in the owning module: torch hands child modules a pre-filtered state dict)."""

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for name, child in self._modules.items():
            key = prefix + name
            if isinstance(child, CplxParameter) and key in state_dict \
                    and key + ".real" not in state_dict and key + ".imag" not in state_dict:
                value = state_dict.pop(key)
                state_dict[key + ".real"] = value
                state_dict[key + ".imag"] = torch.zeros_like(value)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def __getattr__(self, name):
        value = super().__getattr__(name)
        if isinstance(value, CplxParameter):
            return Cplx(value["real"], value["imag"])
        return value


class BaseRealToCplx(torch.nn.Module):
    pass


class BaseCplxToReal(torch.nn.Module):
    pass


def _split_from_callable(fn):
    class SplitFunc(CplxToCplx):
        def __init__(self, *args, **kwargs):
            super().__init__()
            self.args, self.kwargs = args, kwargs

        def forward(self, input):
            return input.apply(fn, *self.args, **self.kwargs)

        def extra_repr(self):
            parts = [repr(a) for a in self.args] + [f"{k}={v!r}" for k, v in self.kwargs.items()]
            return ", ".join(parts)

    SplitFunc.__name__ = f"CplxSplitFunc{fn.__name__.title()}"
    return SplitFunc


def _split_from_module(Module):
    class SplitLayer(Module, CplxToCplx):
        def forward(self, input):
            return input.apply(super().forward)

    if Module is torch.nn.ReLU:
        # the split activation of the Deep-Complex-Net style models: both planes in ONE launch each way (4 -> 2 launches
        # per layer and step); in-place modules, CPU tensors and other dtypes keep torch's kernels
        def forward(self, input):
            re, im = input.real, input.imag
            if (self.inplace or not re.is_cuda or re.dtype not in (torch.float32, torch.bfloat16) or re.dtype != im.dtype
                    or re.shape != im.shape):
                return input.apply(torch.nn.ReLU.forward.__get__(self))
            from ... import ops
            return Cplx(*ops.split_relu(re, im))
        SplitLayer.forward = forward

    SplitLayer.__name__ = f"CplxSplitLayer{Module.__name__}"
    return SplitLayer


class _SplitPromotion(type):
    @functools.lru_cache(maxsize=None)
    def __getitem__(cls, base):
        if isinstance(base, type) and issubclass(base, torch.nn.Module):
            if issubclass(base, (CplxToCplx, BaseRealToCplx)):
                return base
            return CplxToCplx if base is torch.nn.Module else _split_from_module(base)
        if callable(base):
            return _split_from_callable(base)
        raise TypeError("Expecting either a torch.nn.Module subclass, or a callable for "
                        f"promotion. Got `{type(base)}`.")


class CplxToCplx(CplxParameterAccessor, torch.nn.Module, metaclass=_SplitPromotion):
    pass
