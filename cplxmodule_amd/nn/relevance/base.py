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


class KLFusion:
    """Lets a layer's training forward carry its own KL term (ops.*LRTFn with `kl_kind`): one fused
    kernel then prepares the GEMM operands AND evaluates the KL sum + gradients, and the backward adds
    the KL gradients inside the weight-gradient GEMM epilogues.  Opt-in by use: the first
    `penalties(model)` after a training forward finds no cached term, computes it stand-alone and
    arms the fusion for the following steps; a cached term is valid only while the parameters it was
    computed from are unchanged (tensor identity + in-place version counters).  Purely an
    optimisation: values and gradients are those of the stand-alone `PenaltySumFn`."""

    _kl_fuse = False        # armed: the next training forward computes the KL
    _kl_cache = None        # (kl, ((param, version), ...), consumed flag holder)
    _kl_misses = 0

    def _kl_kind_for_forward(self):
        if not (self._kl_fuse and torch.is_grad_enabled()):
            return None
        c = self._kl_cache
        if c is not None and not c[2][0]:
            # the previous fused term was never asked for (e.g. penalties() runs BEFORE forward in this
            # training loop): stop paying for it after two such steps
            self._kl_misses += 1
            if self._kl_misses >= 2:
                self._kl_fuse = False
                self._kl_cache = None
                return None
        return self._kl_kind

    def _kl_put(self, kl, params):
        self._kl_cache = (kl, tuple((p, p._version, p.data_ptr()) for p in params), [False])

    def _kl_get(self, params):
        """The cached fused KL sum if it belongs to the current parameter values, else None (and arm)."""
        c = self._kl_cache
        if c is not None and torch.is_grad_enabled() and len(c[1]) == len(params) and all(
                p is q and p._version == v and p.data_ptr() == d for p, (q, v, d) in zip(params, c[1])):
            c[2][0] = True
            self._kl_misses = 0
            return c[0]
        if self.training and self._kl_misses < 2:
            self._kl_fuse = True
        return None


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
