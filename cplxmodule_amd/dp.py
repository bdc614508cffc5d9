"""Data parallelism for the hot path: one process per GPU, batch rows sharded, parameters
replicated, ONE flat gradient bucket all-reduced per step over RCCL/xGMI (torch.distributed
backend "nccl"), plus the scalar KL term.

The reference has no distributed code; this is the exchange step of section 8(e).
Bucket layout = the reference's parameter order (named_parameters(): log_sigma2, weight.imag,
weight.real, bias.imag, bias.real per layer), so a bucket can be compared across implementations.
The collective is the only cross-rank traffic: forward / backward kernels never communicate.
"""
import torch
import torch.distributed as dist


# Testing aid: run every collective even in a world of one (exercises the RCCL calls -- AVG,
# async work handles, broadcast -- on a single-GPU box; tests/dp_gpu_check.py --rccl1).
FORCE_COLLECTIVES = False


def is_initialized():
    return dist.is_available() and dist.is_initialized()


def _exchanging():
    return is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def world():
    return (dist.get_rank(), dist.get_world_size()) if is_initialized() else (0, 1)


def shard_rows(n_rows, rank=None, world_size=None):
    """[lo, hi) slice of `n_rows` owned by `rank` (contiguous, sizes differ by at most 1)."""
    if rank is None:
        rank, world_size = world()
    base, rem = divmod(n_rows, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradBucket:
    """Flat float32 bucket over the gradients of `module`'s parameters."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.names = [n for n, p in module.named_parameters() if p.requires_grad]
        self.sizes = [p.numel() for p in self.params]
        total = sum(self.sizes)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=torch.float32, device=ref.device)
        self.views, off = [], 0
        for p, n in zip(self.params, self.sizes):
            self.views.append(self.flat[off:off + n].view_as(p))
            off += n

    def nbytes(self):
        return self.flat.numel() * 4

    def adopt(self):
        """Make every .grad a view into the flat bucket: backward then writes (accumulates)
        straight into it and no gather / scatter copy is needed around the collective."""
        if not _exchanging():
            for p in self.params:          # single process: nothing to exchange, let autograd
                p.grad = None              # adopt the kernels' output buffers (no extra pass)
            return
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v

    def all_reduce_mean(self, async_op=False):
        """In-place mean over ranks of the whole bucket (a no-op for a single process)."""
        if not _exchanging():
            return None
        for p, v in zip(self.params, self.views):     # tolerate grads that were re-pointed
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v
        if dist.get_backend() == "nccl":
            return dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, async_op=async_op)
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=False)
        self.flat.div_(dist.get_world_size())
        return work


class OverlapHook:
    """Installed as `ops.dp_hook` by DataParallel: averages a layer's data-term parameter
    gradients across ranks asynchronously (RCCL stream) while the layer's backward continues
    with the input-gradient GEMMs.  Parameters handled here are skipped by the bucket."""

    def __init__(self):
        self.done = set()

    def reduce(self, flat, param_ptrs):
        self.done.update(param_ptrs)
        if dist.get_backend() == "nccl":
            return (dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=True), None)
        return (dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat)

    def finish(self, handle):
        work, flat = handle
        work.wait()
        if flat is not None:
            flat.div_(dist.get_world_size())


def all_reduce_scalar_mean(value):
    """Mean over ranks of a 0-d tensor (the KL term / the loss, for logging and for the sharded
    KL option); returns a new tensor, leaves autograd alone."""
    out = value.detach().clone()
    if _exchanging():
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
        out /= dist.get_world_size()
    return out


class DataParallel(torch.nn.Module):
    """Thin wrapper: forward = module forward on the local shard; call `sync_gradients()` after
    backward.  Gradients are the MEAN over ranks (torch DDP convention).  The KL penalty is a
    function of the replicated weights only, so its gradient is identical on every rank and the
    mean leaves it unchanged."""

    def __init__(self, module, overlap=True):
        super().__init__()
        self.module = module
        self.bucket = GradBucket(module)
        self.hook = None
        if _exchanging():
            for p in module.parameters():              # replicate rank 0's parameters
                dist.broadcast(p.data, src=0)
            for b in module.buffers():
                dist.broadcast(b.data, src=0)
            if overlap:
                from . import ops
                self.hook = ops.dp_hook = OverlapHook()

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def zero_grad(self, set_to_none=False):
        if self.hook is not None:
            self.hook.done.clear()
            for p in self.bucket.params:               # gradients arrive already averaged (hook)
                p.grad = None                          # or are averaged below
            return
        self.bucket.adopt()

    def sync_gradients(self):
        """Average whatever the overlap hook did not already average.  (The KL term is a function
        of the replicated weights: its gradient is identical on every rank and needs no exchange.)"""
        if self.hook is None:
            return self.bucket.all_reduce_mean()
        rest = [p for p in self.bucket.params if p.grad is not None and p.data_ptr() not in self.hook.done]
        if rest:
            flat = torch.cat([p.grad.reshape(-1) for p in rest])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.div_(dist.get_world_size())
            off = 0
            for p in rest:
                n = p.numel()
                p.grad.copy_(flat[off:off + n].view_as(p))
                off += n
        return None
