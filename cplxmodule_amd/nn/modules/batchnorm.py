"""Complex batch normalisation (Trabelsi et al.) on the 2-pass moment / apply kernels.

Layer contract of cplxmodule/nn/modules/batchnorm.py:281-407: affine weight [2,2,F] (init I2),
bias [2,F], running_mean [2,F], running_var [2,2,F] (init I2), num_batches_tracked; momentum=None
selects the cumulative average; statistics are those of the local batch unless `process_group` is set
(cplxmodule_amd.dp.convert_sync_batchnorm: SURVEY 8(e)'s optional cross-rank statistics).
"""
import torch

from .base import CplxToCplx
from ... import cplx


def cplx_batch_norm(input, running_mean, running_var, weight=None, bias=None, training=True,
                    momentum=0.1, eps=1e-5, *, process_group=None, _tracked=None):
    """Functional form (batchnorm.py:189-278).  Running statistics are updated in place.
    process_group (not in the reference; None = its behaviour): share the training-mode batch statistics between
    the ranks of that torch.distributed group (True = the default group), see bn.CplxBatchNormFn.
    _tracked (module-internal): an int64 device scalar the forward adds 1 to in its finalize launch."""
    assert (running_mean is None) == (running_var is None)
    assert (weight is None) == (bias is None)
    from ... import bn
    yr, yi = bn.CplxBatchNormFn.apply(input.real, input.imag, weight, bias, running_mean,
                                      running_var, bool(training), float(momentum), float(eps), process_group, _tracked)
    return cplx.Cplx(yr, yi)


class _CplxBatchNorm(CplxToCplx):
    _dims = ()
    # None: statistics of the local batch (the reference); a process group / True: shared between its ranks in
    # training mode (set by cplxmodule_amd.dp.convert_sync_batchnorm; not part of the state dict)
    process_group = None

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True,
                 track_running_stats=True):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.affine, self.track_running_stats = affine, track_running_stats
        if affine:
            self.weight = torch.nn.Parameter(torch.empty(2, 2, num_features))
            self.bias = torch.nn.Parameter(torch.empty(2, num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        if track_running_stats:
            self.register_buffer("running_mean", torch.empty(2, num_features))
            self.register_buffer("running_var", torch.empty(2, 2, num_features))
            self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        else:
            self.register_parameter("running_mean", None)
            self.register_parameter("running_var", None)
            self.register_parameter("num_batches_tracked", None)
        self.reset_running_stats()
        self.reset_parameters()

    def reset_running_stats(self):
        if self.track_running_stats:
            self.num_batches_tracked.zero_()
            self.running_mean.zero_()
            self.running_var.copy_(torch.eye(2).unsqueeze(-1))

    def reset_parameters(self):
        if self.affine:
            with torch.no_grad():
                self.weight.copy_(torch.eye(2).unsqueeze(-1))
                self.bias.zero_()

    def _check_input_dim(self, input):
        if input.dim() not in self._dims:
            want = " or ".join(f"{d}D" for d in self._dims)
            raise ValueError(f"expected {want} input (got {input.dim()}D input)")

    def forward(self, input):
        self._check_input_dim(input)
        factor = 0.0 if self.momentum is None else self.momentum
        tracked = None
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            if (self.momentum is not None and self.process_group is None and self.num_batches_tracked.is_cuda
                    and self.num_batches_tracked.device == input.real.device):
                tracked = self.num_batches_tracked       # the counter is bumped by the forward's finalize launch
            else:
                self.num_batches_tracked += 1
            if self.momentum is None:
                factor = 1.0 / float(self.num_batches_tracked)
        return cplx_batch_norm(input, self.running_mean, self.running_var, self.weight, self.bias,
                               self.training or not self.track_running_stats, factor, self.eps,
                               process_group=self.process_group, _tracked=tracked)

    def extra_repr(self):
        return (f"{self.num_features}, eps={self.eps}, momentum={self.momentum}, "
                f"affine={self.affine}, track_running_stats={self.track_running_stats}")


class CplxBatchNorm1d(_CplxBatchNorm):
    _dims = (2, 3)


class CplxBatchNorm2d(_CplxBatchNorm):
    _dims = (4,)


class CplxBatchNorm3d(_CplxBatchNorm):
    _dims = (5,)
