"""The sparsification interface and the walkers that collect KL terms and relevance masks.

Same observable behaviour as cplxmodule/nn/relevance/base.py:4-216.  One addition: layers may
offer `_penalty_reduced(reduction)`, a fused elementwise + reduction kernel, which the walkers
prefer over materialising the penalty tensor and reducing it in a second pass.
"""
import torch


class BaseARD(torch.nn.Module):
    """Layers exposing a differentiable `.penalty` and a `.relevance(**kw)` mask."""

    @property
    def penalty(self):
        raise NotImplementedError("Derived classes must compute their own penalty.")

    def relevance(self, **kwargs):
        raise NotImplementedError(
            "Derived classes must implement a float mask of relevant coefficients.")


def named_penalties(module, reduction="sum", prefix=""):
    if reduction is not None and reduction not in ("mean", "sum"):
        raise ValueError(f"`reduction` must be either `None`, `sum` or `mean`. Got {reduction}.")
    # named_modules() visits shared submodules once, so a reused layer is penalised once
    for name, mod in module.named_modules(prefix=prefix):
        if not isinstance(mod, BaseARD):
            continue
        fused = getattr(mod, "_penalty_reduced", None)
        if reduction is not None and fused is not None:
            yield name, fused(reduction)
            continue
        value = mod.penalty
        if reduction == "sum":
            value = value.sum()
        elif reduction == "mean":
            value = value.mean()
        yield name, value


def penalties(module, reduction="sum"):
    for _, value in named_penalties(module, reduction=reduction):
        yield value


def named_relevance(module, prefix="", **kwargs):
    for name, mod in module.named_modules(prefix=prefix):
        if isinstance(mod, BaseARD):
            yield name, mod.relevance(**kwargs).detach()


def compute_ard_masks(module, *, prefix="", **kwargs):
    """{"<module>.mask": mask} ("mask" for the root), ready for nn.masked-style deployment."""
    if not isinstance(module, torch.nn.Module):
        return {}
    return {(name + "." if name else "") + "mask": mask
            for name, mask in named_relevance(module, prefix=prefix, **kwargs)}
