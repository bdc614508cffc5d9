"""Fixed-sparsity ("masked") layers for fine-tuning after sparsification, and the mask plumbing
between the ARD phase and the masked phase.

Behaviour of cplxmodule/nn/masked/{base,real,complex}.py: a `mask` buffer (None = dense) that
follows the weight's device / dtype / shape, settable by attribute, `deploy_masks`, or
`load_state_dict`; `weight_masked = weight * mask` feeds the same GEMM / conv kernels as the
dense layers (the mask multiply is a weight-sized elementwise op).
"""
import torch

from .modules.linear import CplxLinear, CplxBilinear
from .modules.conv import CplxConv1d, CplxConv2d, CplxConv3d
from .utils.sparsity import SparsityStats
from .. import cplx, ops


class BaseMasked(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("mask", None)

    @property
    def is_sparse(self):
        return isinstance(self.mask, torch.Tensor)

    def mask_(self, mask):
        """Set (tensor) or drop (None) the sparsity mask."""
        if mask is not None and not isinstance(mask, torch.Tensor):
            raise TypeError(f"`mask` must be either a Tensor or `None`. Got {type(mask).__name__}.")
        if mask is None:
            if self.is_sparse:
                del self.mask
                self.register_buffer("mask", None)
            return self
        w = self.weight
        mask = mask.detach().to(w.device, w.dtype).expand(w.shape).contiguous()
        self.register_buffer("mask", mask)
        return self

    def __setattr__(self, name, value):
        if name == "mask":
            self.mask_(value)
        else:
            super().__setattr__(name, value)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        key = prefix + "mask"
        rest = {k: v for k, v in state_dict.items() if k != key}
        super()._load_from_state_dict(rest, prefix, local_metadata, strict, missing_keys,
                                      unexpected_keys, error_msgs)
        listed = key in missing_keys        # torch lists it iff the buffer currently exists
        if key in state_dict:
            if listed:
                missing_keys.remove(key)
            self.mask_(state_dict[key])
        elif strict:
            if not listed:
                missing_keys.append(key)   # an absent mask is always reported in strict mode
        elif listed:
            missing_keys.remove(key)

    @property
    def weight_masked(self):
        if not self.is_sparse:
            raise RuntimeError(f"`{type(self).__name__}` has no sparsity mask. Please, either set "
                               "a mask attribute, or call `deploy_masks()`.")
        w = self.weight
        probe = w.real if isinstance(w, cplx.Cplx) else w
        if not probe.is_cuda:
            # host-side inspection of a module that has not been moved to the GPU yet (the layers'
            # forward itself has no CPU path): the reference's expression, cplxmodule/nn/masked/base.py:137-147
            return w * self.mask
        if isinstance(w, cplx.Cplx):
            return cplx.Cplx(*ops.MaskMulFn.apply(w.real, w.imag, self.mask))
        return ops.MaskMulFn.apply(w, None, self.mask)

    def _require_mask(self):
        if not self.is_sparse:
            raise RuntimeError(f"`{type(self).__name__}` has no sparsity mask. Please, either set "
                               "a mask attribute, or call `deploy_masks()`.")
        return self.mask


class _MaskedStats(BaseMasked, SparsityStats):
    __sparsity_ignore__ = ("mask",)

    def _n_dropped(self, numel, hard):
        if not self.is_sparse:
            return 0.0
        mask = torch.gt(self.mask, 0) if hard else self.mask
        return float(numel) - float(mask.sum().item())


class CplxLinearMasked(CplxLinear, _MaskedStats):
    def forward(self, input):
        # the mask rides in the GEMM operand preparation / weight-gradient epilogue (ops.CplxLinearFn)
        w, b = self.weight, self.bias
        br, bi = (None, None) if b is None else (b.real, b.imag)
        yr, yi = ops.CplxLinearFn.apply(input.real, input.imag, w.real, w.imag, br, bi, 0, self._require_mask())
        return cplx.Cplx(yr, yi)

    def sparsity(self, *, hard=True, **kwargs):
        w = self.weight
        n = self._n_dropped(w.real.numel(), hard)
        return [(id(w.real), n), (id(w.imag), n)]


class CplxConv2dMasked(CplxConv2d, _MaskedStats):
    def forward(self, input):
        return cplx.conv2d(input, self.weight_masked, self.bias, self.stride, self.padding,
                           self.dilation, self.groups, self.padding_mode)

    def sparsity(self, *, hard=True, **kwargs):
        w = self.weight
        n = self._n_dropped(w.real.numel(), hard)
        return [(id(w.real), n), (id(w.imag), n)]


class _CplxMaskedStats(_MaskedStats):
    def sparsity(self, *, hard=True, **kwargs):
        w = self.weight
        n = self._n_dropped(w.real.numel(), hard)
        return [(id(w.real), n), (id(w.imag), n)]


class _RealMaskedStats(_MaskedStats):
    def sparsity(self, *, hard=True, **kwargs):
        return [(id(self.weight), self._n_dropped(self.weight.numel(), hard))]


class CplxConv1dMasked(CplxConv1d, _CplxMaskedStats):
    def forward(self, input):
        return cplx.conv1d(input, self.weight_masked, self.bias, self.stride, self.padding,
                           self.dilation, self.groups, self.padding_mode)


class CplxBilinearMasked(CplxBilinear, _CplxMaskedStats):
    def forward(self, input1, input2):
        return cplx.bilinear(input1, input2, self.weight_masked, self.bias, self.conjugate)


class BilinearMasked(torch.nn.Bilinear, _RealMaskedStats):
    def forward(self, input1, input2):
        return ops.RealBilinearFn.apply(input1, input2, self.weight_masked, self.bias, None, None, 0, 0)


class Conv1dMasked(torch.nn.Conv1d, _RealMaskedStats):
    def forward(self, input):
        from .. import conv
        if self.padding_mode != "zeros":
            raise ValueError("Conv1dMasked supports `zeros` padding only")
        y = conv.RealConv2dFn.apply(input.unsqueeze(2), self.weight_masked.unsqueeze(2), self.bias,
                                    (1, self.stride[0]), (0, self.padding[0]), (1, self.dilation[0]),
                                    self.groups)
        return y.squeeze(2)


class CplxConv3dMasked(CplxConv3d, _CplxMaskedStats):
    def forward(self, input):
        return cplx.conv3d(input, self.weight_masked, self.bias, self.stride, self.padding,
                           self.dilation, self.groups, self.padding_mode)


class Conv3dMasked(torch.nn.Conv3d, _RealMaskedStats):
    def forward(self, input):
        from .. import conv3d
        if self.padding_mode != "zeros":
            raise ValueError("Conv3dMasked supports `zeros` padding only")
        return conv3d.real_conv3d(input, self.weight_masked, self.bias, self.stride, self.padding,
                                  self.dilation, self.groups)


class LinearMasked(torch.nn.Linear, _MaskedStats):
    def forward(self, input):
        return ops.RealLinearFn.apply(input, self.weight, self.bias, self._require_mask())

    def sparsity(self, *, hard=True, **kwargs):
        return [(id(self.weight), self._n_dropped(self.weight.numel(), hard))]


class Conv2dMasked(torch.nn.Conv2d, _MaskedStats):
    def forward(self, input):
        from .. import conv
        if self.padding_mode != "zeros":
            raise ValueError("Conv2dMasked supports `zeros` padding only")
        return conv.RealConv2dFn.apply(input, self.weight_masked, self.bias, self.stride,
                                       self.padding, self.dilation, self.groups)

    def sparsity(self, *, hard=True, **kwargs):
        return [(id(self.weight), self._n_dropped(self.weight.numel(), hard))]


def is_sparse(module):
    return isinstance(module, BaseMasked) and module.is_sparse


def named_masks(module, prefix=""):
    for name, mod in module.named_modules(prefix=prefix):
        if isinstance(mod, BaseMasked):
            yield name, mod.mask


def deploy_masks(module, *, state_dict=None, prefix="", reset=False):
    """Set the masks named "<module>.mask" in `state_dict`; with `reset`, drop all others."""
    if not isinstance(state_dict, dict) or not isinstance(module, torch.nn.Module):
        return module
    for name, mod in module.named_modules(prefix=prefix):
        if isinstance(mod, BaseMasked):
            key = (name + "." if name else "") + "mask"
            if key in state_dict:
                mod.mask = state_dict[key]
            elif reset:
                mod.mask = None
    return module


def binarize_masks(state_dict, masks):
    """Fold soft masks into the weights and return 0/1 masks (negative zeros cleaned up)."""
    with torch.no_grad():
        out = {}
        for name, par in state_dict.items():
            if "weight" in name:
                key = name.rsplit("weight", 1)[0] + "mask"
                if key in masks:
                    par = par * masks[key].to(par)
                    par[par == 0] = 0          # -0.0 -> +0.0
            out[name] = par
        hard = {name: torch.ne(mask, 0).to(mask) for name, mask in masks.items()}
    return out, hard
