"""Data parallelism for the hot path: one process per GPU, batch rows sharded, parameters
replicated, gradients averaged over RCCL/xGMI (torch.distributed backend "nccl") in flat float32
buckets, plus the scalar KL term.

The reference has no distributed code; this is the exchange step of section 8(e).

Bucket layout.  The parameters in the reference's order (named_parameters(): per layer log_sigma2,
weight.imag, weight.real, bias.imag, bias.real) are cut into buckets of about `bucket_mb` MiB, filled in
REVERSE order (the backward pass produces the last layer's gradients first).  Each bucket is one flat
float32 buffer; `GradBuckets.names` lists (bucket, offset, name) so that a bucket can be compared
across implementations.

Exchange.  Every parameter carries a post-accumulate-grad hook.  When its gradient exists the hook makes
`.grad` a view of its bucket slice (no copy if the producing kernel already wrote there: the linear
layers' backward asks `ops.grad_buffer(param)` for its output storage; one copy otherwise -- conv / BN
parameters are small) and counts it ready; the bucket's `all_reduce(AVG)` is launched asynchronously as
soon as its last parameter is ready, so it overlaps the rest of the backward pass (earlier layers).
A collective is launched ONLY from a parameter's post-accumulate-grad hook, i.e. when autograd has summed
every path into that leaf (an in-loss weight regulariser, the stand-alone KL node, a shared use): a
gradient that is still being accumulated is never reduced.  The LRT linear layers additionally ANNOUNCE
their parameter gradients before their own input-gradient GEMMs (`ops.dp_hook.early_ready`): the
announcement records a HIP event behind the weight-gradient kernels, and the bucket's all-reduce -- issued
from a side stream -- waits for that event only, so it still overlaps the layer's own input-gradient GEMMs
(the host runs ahead of the device), which gives the single-layer headline config its overlap.  A slice that
was not announced (copied in, or accumulated into after the announcement) gets an event at the moment its
hook runs; the collective waits for the slices' events, never for what the host queues afterwards, so a
copied bias or BatchNorm parameter in the same bucket does not cost the in-place weights their early start.
Nothing waits inside backward: `sync_gradients()` waits for all buckets,
reduces what never completed (parameters without a gradient contribute zeros) and hands out the
averaged views.

Sharing the chip.  While a bucket's all-reduce is in flight its RCCL kernels hold CUs next to the rest of the backward
pass.  The persistent GEMM launches (one workgroup per CU, tiles dealt out statically) would wait for those CUs with
their last workgroups -- until the first ones END -- so from the first announcement / asynchronous launch until
`sync_gradients()` has waited for all of them every GEMM and channels-last convolution launch carries
`CPLXAMD_LAUNCH_SHARED` (one workgroup per tile; bit-identical results, 1-2 % slower alone).  The flag is an ARGUMENT of
each call (`_lib.launch_flags()`, ABI 19): the library holds no launch state, so other threads / streams of the process
are affected only through this host-side window, which they can override (`_lib.launch_policy`).

The collective is the only cross-rank traffic: forward / backward kernels never communicate.  The KL
penalty is a function of the replicated weights only, so its gradient is identical on every rank and the
mean leaves it unchanged; `all_reduce_scalar_mean` is the north star's "scalar KL term".
Several backward passes per exchange (gradient accumulation over micro-batches, or the reference's two-call pattern
`nll.backward(); (c * kl).backward()`): wrap every pass but the LAST in `DataParallel.no_sync()` -- no collective is
launched inside it, gradients accumulate in `.grad` as autograd would without the wrapper, and the last pass exchanges
the totals.  A backward pass that reaches a parameter whose bucket's all-reduce is already in flight (i.e. a second
pass WITHOUT no_sync) raises: the collective may be reading the storage autograd has just accumulated into, and no
ordering applied afterwards can undo that.  A parameter used by several layers is fine: bucket storage is
handed to at most one producer per pass (the others get private tensors and autograd sums), and the exchange starts only
from the post-accumulate hook, i.e. after every use has been summed.  Call `zero_grad()` -- which sets every `.grad` to
None -- before each step.
"""
import torch
import torch.distributed as dist


# Testing aid: run every collective even in a world of one (exercises the RCCL calls -- AVG,
# async work handles, broadcast -- on a single-GPU box; tests/dp_gpu_check.py --rccl1).
FORCE_COLLECTIVES = False


def is_initialized():
    return dist.is_available() and dist.is_initialized()


def init_process_group(backend="nccl", device=None, max_channels=None, **kwargs):
    """torch.distributed.init_process_group with what the overlap of this module needs: RCCL's stream on a
    high-priority hardware queue (TORCH_NCCL_HIGH_PRIORITY=1, read when the process group is created; on the compute
    stream's queue its kernels run in queue order, i.e. not next to the GEMMs they are meant to overlap) and the
    loop-back rendezvous address of a single node as the default.

    max_channels (or env CPLXAMD_RCCL_MAX_CHANNELS): the CU split between the collectives and the GEMMs they overlap.
    An RCCL ring / tree kernel holds one CU per channel for as long as it runs; while a bucket is in flight the GEMM
    kernels launch one workgroup per tile (`CPLXAMD_LAUNCH_SHARED`) and simply get the remaining CUs, so the
    price of C channels is C / 256 of the matrix throughput for the duration of the exchange.  Eight ranks on xGMI are
    link-bound long before they need 32+ CUs (7 links x ~153 GB/s per GPU against ~25 GB/s a CU can copy), so a cap of
    8-16 is the value to try first when an 8-GPU run shows the input-gradient GEMMs starving; None (default) leaves
    RCCL's own choice.  Sets NCCL_MAX_NCHANNELS before the communicator exists (it is read once, at creation)."""
    import os
    os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
    if max_channels is None and os.environ.get("CPLXAMD_RCCL_MAX_CHANNELS"):
        max_channels = int(os.environ["CPLXAMD_RCCL_MAX_CHANNELS"])
    if max_channels:
        prev = os.environ.get("NCCL_MAX_NCHANNELS")
        if prev is not None and prev != str(int(max_channels)):
            import warnings
            warnings.warn(f"NCCL_MAX_NCHANNELS={prev} in the environment is replaced by max_channels={int(max_channels)}")
        os.environ["NCCL_MAX_NCHANNELS"] = str(int(max_channels))      # an explicit argument wins
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl" and device is not None:
        kwargs.setdefault("device_id", torch.device(device))
    return dist.init_process_group(backend, **kwargs)


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


class _Bucket:
    def __init__(self, index):
        self.index = index
        self.entries = []          # (name, param, offset, numel)
        self.numel = 0
        self.flat = None
        self.reset()

    def reset(self):
        self.ready = set()         # id() of the parameters whose FINAL gradient is in its slice
        self.work = None           # async handle once launched
        self.launched = False
        self.events = []           # HIP events behind the kernels that wrote the slices in place
        self.early = False         # launched on the side stream, behind the recorded stream positions only
        self.in_order = False      # no stream position could be recorded for a slice (host tensors): plain stream order
        self.copy_bumps = 0        # version-counter increments of `flat` made by on_grad's own copies (see on_grad)


class GradBuckets:
    """Flat float32 gradient buckets over the parameters of `module` (see the module docstring)."""

    def __init__(self, module, bucket_mb=32.0):
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        self.params = [p for _, p in named]
        self.param_names = [n for n, _ in named]
        cap = max(1, int(bucket_mb * (1 << 20) / 4))
        self.buckets, cur = [], _Bucket(0)
        for n, p in reversed(named):
            if cur.entries and cur.numel + p.numel() > cap:
                self.buckets.append(cur)
                cur = _Bucket(len(self.buckets))
            cur.entries.append((n, p, cur.numel, p.numel()))
            cur.numel += (p.numel() + 3) & ~3          # 16-byte aligned slices (the kernels move 16 B per lane)
        if cur.entries:
            self.buckets.append(cur)
        self.where = {}
        for b in self.buckets:
            dev = b.entries[0][1].device
            b.flat = torch.zeros(b.numel, dtype=torch.float32, device=dev)
            for n, p, off, numel in b.entries:
                # keyed by the parameter OBJECT: `.to()` / `.data = ...` re-allocations keep the registration
                self.where[id(p)] = (b, off, numel, tuple(p.shape))

    @property
    def names(self):
        return [(b.index, off, n) for b in self.buckets for n, _, off, _ in b.entries]

    def nbytes(self):
        return sum(b.numel for b in self.buckets) * 4

    def view(self, param):
        """A FRESH view of the bucket slice of `param` (None: not registered)."""
        e = self.where.get(id(param))
        if e is None:
            return None
        b, off, numel, shape = e
        return b.flat[off:off + numel].view(shape)


def _all_reduce(flat, async_op):
    """In-place cross-rank SUM (gloo) or AVG (RCCL); returns (work or None, needs_division)."""
    if dist.get_backend() == "nccl":
        return dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=async_op), False
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op), True


class BucketHook:
    """Registered for its wrapper's parameters (`ops.register_dp_hook`) by DataParallel."""

    def __init__(self, buckets, overlap=True):
        self.buckets = buckets
        self.overlap = overlap
        self.pending = []          # (bucket, work, needs_division)
        self._shared_chip = False  # collectives in flight next to compute kernels (see _share_chip)
        self._announced = {}       # id(param) -> HIP event recorded behind the kernels that wrote its slice
        self._side = None          # stream the early collectives are issued from
        self.accumulate = False    # inside DataParallel.no_sync(): no exchange, gradients pile up in .grad
        self._handed = set()       # id(param) whose bucket slice a producer holds in this pass (view_for)

    # -- storage for gradients (zero-copy path of the linear layers) ---------------------------
    def view_for(self, param):
        """Bucket storage for the gradient `param` is about to receive, or None = use a private tensor: when a gradient
        has been accumulated already (an earlier pass under no_sync: writing the slice would destroy it -- autograd adds
        the private tensor instead) and when another producer holds the slice in this pass (a parameter shared by two
        layers: both writing the same storage would make autograd sum one of them twice)."""
        if param.grad is not None or id(param) in self._handed:
            return None
        v = self.buckets.view(param)
        if v is not None:
            self._handed.add(id(param))
        return v

    # -- sharing the chip with RCCL -----------------------------------------------------------------
    def _share_chip(self):
        if not self._shared_chip and dist.get_backend() == "nccl":
            # RCCL's kernels are about to hold CUs next to the rest of the backward pass: a persistent launch
            # (one workgroup per CU for its whole duration) would wait for them with its last workgroups and take
            # twice as long, one workgroup per tile just runs on the CUs that are left (csrc/gemm.h)
            from . import _lib
            _lib.shared_chip_enter(self)       # every launch from here on carries CPLXAMD_LAUNCH_SHARED (per call: _lib)
            self._shared_chip = True

    def _chip_is_ours(self):
        if self._shared_chip:
            from . import _lib
            _lib.shared_chip_leave(self)
            self._shared_chip = False

    # -- readiness ------------------------------------------------------------------------------
    def _ready(self, param, b):
        b.ready.add(id(param))
        if self.overlap and not b.launched and len(b.ready) == len(b.entries):
            self._launch(b, async_op=True)

    def _launch(self, b, async_op):
        b.launched = True
        side = None
        if async_op and b.flat.is_cuda:
            self._share_chip()
            if b.events and not b.in_order and dist.get_backend() == "nccl":
                # every slice was written in place by kernels that precede these events: the collective need not wait
                # for what the host has queued since (the layer's own input-gradient GEMMs)
                if self._side is None:
                    # HIGH priority: ROCm multiplexes HIP streams over a few hardware queues, and a normal-priority side
                    # stream may share the compute stream's queue -- its event record then sits BEHIND the input-gradient
                    # GEMMs already queued there and the collective starts after them (profiles/r03_dp_timeline.txt);
                    # high-priority streams get a queue of their own.  The process group's stream must be high-priority
                    # too (TORCH_NCCL_HIGH_PRIORITY=1 before init_process_group: dp.init_process_group does it).
                    self._side = torch.cuda.Stream(device=b.flat.device, priority=-1)
                side = self._side
                for ev in b.events:
                    side.wait_event(ev)
        b.early = side is not None     # (introspection: the collective was ordered behind the announced position only)
        if side is not None:
            with torch.cuda.stream(side):
                work, div = _all_reduce(b.flat, async_op)
            # join the side stream again (it holds no work of its own: the collective runs on the backend's stream, which
            # `work.wait()` joins in sync()): a step captured into a hipGraph must not end with an unjoined fork
            torch.cuda.current_stream(b.flat.device).wait_stream(side)
        else:
            work, div = _all_reduce(b.flat, async_op)
        self.pending.append((b, work if async_op else None, div))

    def early_ready(self, *params):
        """Called from inside a layer's backward: the data gradients of `params` have been written into their
        `view_for` storage by kernels already queued.  Nothing is launched here -- other autograd paths may still add
        to these leaves -- but the stream position is remembered: if the gradient autograd finally delivers IS that
        storage, its bucket's all-reduce waits for this point only."""
        if not self.overlap or self.accumulate:
            return
        ev = None
        for p in params:
            if p is None or id(p) not in self.buckets.where:
                continue
            if ev is None and p.is_cuda:
                ev = torch.cuda.Event()
                ev.record()
                self._share_chip()       # the kernels queued from here on may run next to a collective
            # (the version counter is the bucket's: views share it.  Autograd summing another path INTO the announced
            #  storage after this point bumps it, and the shortcut below is then not taken; the bumps of on_grad's own
            #  copies of OTHER parameters into the same bucket are discounted -- ADVICE r3: a bias or BN parameter that
            #  shares the bucket must not push the in-place weights off the early path)
            b = self.buckets.where[id(p)][0]
            self._announced[id(p)] = (ev, b.flat._version - b.copy_bumps)

    def on_grad(self, param):
        """post-accumulate-grad hook of every registered parameter: `param.grad` is final for this backward."""
        g = param.grad
        if g is None:
            return
        e = self.buckets.where.get(id(param))
        if e is None:
            return
        b, off, numel, shape = e
        self._handed.discard(id(param))
        if self.accumulate:
            return                 # no_sync(): .grad keeps accumulating (in the bucket slice or in a private tensor)
        if b.launched:
            # a second backward pass into an exchanged bucket: autograd has ALREADY accumulated into storage the in-flight
            # all-reduce may be reading or writing -- nothing done here can order the two any more.  Fail loudly instead of
            # exchanging a gradient that may be torn; the pattern has a supported spelling.
            name = next((n for n, p, _, _ in b.entries if p is param), "?")
            raise RuntimeError(
                f"cplxmodule_amd.dp: a backward pass reached parameter '{name}' after the all-reduce of its bucket was "
                "launched in this step.  Wrap every backward pass of a step except the last one in "
                "`DataParallel.no_sync()` (gradient accumulation; `nll.backward(); (c * kl).backward()`), and call "
                "`zero_grad()` + `sync_gradients()` once per step.")
        view = b.flat[off:off + numel].view(shape)
        ev, version = self._announced.pop(id(param), (None, None))
        if g.data_ptr() != view.data_ptr() or g.stride() != view.stride():
            before = b.flat._version
            view.copy_(g)
            b.copy_bumps += b.flat._version - before
            param.grad = view
            ev = None
        elif ev is not None and version != b.flat._version - b.copy_bumps:
            ev = None              # in place, but not (only) by the kernels in front of the announced stream position
        if ev is None:
            # the slice is final HERE: everything that wrote it is queued on this stream by now.  The bucket's collective
            # waits for this position (and the other slices' positions), not for what the host queues afterwards
            if b.flat.is_cuda:
                ev = torch.cuda.Event()
                ev.record()
            else:
                b.in_order = True
        if ev is not None:
            b.events.append(ev)
        self._ready(param, b)

    # -- step boundary --------------------------------------------------------------------------
    def reset(self):
        for _, work, _ in self.pending:      # (a step abandoned mid-exchange: never drop a collective that is in flight)
            if work is not None:
                work.wait()
        self._chip_is_ours()
        self.pending = []
        self._announced = {}
        self._handed = set()
        for b in self.buckets.buckets:
            b.reset()

    def sync(self):
        bk = self.buckets
        for b, work, div in self.pending:
            if work is not None:
                work.wait()
            if div:
                b.flat.div_(dist.get_world_size())
        self.pending = []
        for b in bk.buckets:
            if not b.launched:
                # parameters that received no gradient contribute zeros; a gradient that was produced but never
                # delivered through the hook (hooks bypassed, or accumulated after the exchange) is taken from .grad
                for n, p, off, numel in b.entries:
                    sl = b.flat[off:off + numel]
                    if p.grad is None:
                        if id(p) not in b.ready:
                            sl.zero_()
                    elif p.grad.data_ptr() != sl.data_ptr():
                        sl.view_as(p).copy_(p.grad)
                self._launch(b, async_op=False)
        for b, work, div in self.pending:
            if div:
                b.flat.div_(dist.get_world_size())
        self._chip_is_ours()
        for b in bk.buckets:
            for n, p, off, numel in b.entries:
                if p.grad is not None or id(p) in b.ready:
                    v = b.flat[off:off + numel].view_as(p)
                    if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                        p.grad = v
        self.pending = []


def all_reduce_scalar_mean(value):
    """Mean over ranks of a 0-d tensor (the KL term / the loss, for logging and for the sharded
    KL option); returns a new tensor, leaves autograd alone."""
    out = value.detach().clone()
    if _exchanging():
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
        out /= dist.get_world_size()
    return out


def convert_sync_batchnorm(module, process_group=True):
    """Make every complex batch-norm layer of `module` share its training-mode batch statistics between the ranks
    of `process_group` (True = the default group, None = back to the reference's local statistics): SURVEY 8(e)'s
    optional SyncBN.  Per layer and pass ONE all-reduce of 5 F + 1 (forward) / 6 F (backward) float64 values -- 2.6 kB
    for 64 features; nothing changes in evaluation mode or outside an initialised process group.  Returns `module`."""
    from .nn.modules.batchnorm import _CplxBatchNorm
    for m in module.modules():
        if isinstance(m, _CplxBatchNorm):
            m.process_group = process_group
    return module


class _NoSync:
    def __init__(self, hook):
        self.hook = hook

    def __enter__(self):
        if self.hook is not None:
            self.prev = self.hook.accumulate
            self.hook.accumulate = True
        return self

    def __exit__(self, *exc):
        if self.hook is not None:
            self.hook.accumulate = self.prev
        return False


class DataParallel(torch.nn.Module):
    """Thin wrapper: forward = module forward on the local shard; call `zero_grad()` before and
    `sync_gradients()` after backward.  Gradients are the MEAN over ranks (torch DDP convention).
    overlap=True: bucket all-reduces start during backward as buckets fill; overlap=False: every bucket is
    reduced in `sync_gradients()` (A/B and debugging)."""

    def __init__(self, module, overlap=True, bucket_mb=32.0):
        super().__init__()
        self.module = module
        self.buckets = GradBuckets(module, bucket_mb)
        self.bucket = self.buckets                     # (round-1 attribute name)
        self.hook = None
        self._handles = []
        if _exchanging():
            from . import ops
            from .nn.relevance.noise import noise
            for p in module.parameters():              # replicate rank 0's parameters
                dist.broadcast(p.data, src=0)
            for b in module.buffers():
                dist.broadcast(b.data, src=0)
            # found per parameter (ops.hook_of): several wrapped models can live in one process; the first one is also the
            # process-wide fallback `ops.dp_hook`
            self.hook = BucketHook(self.buckets, overlap=overlap)
            ops.register_dp_hook(self.hook, self.buckets.params)
            if ops.dp_hook is None:
                ops.dp_hook = self.hook
            for p in self.buckets.params:
                self._handles.append(p.register_post_accumulate_grad_hook(self.hook.on_grad))
            # decorrelate the local-reparameterization / dropout noise of the ranks (same torch seed on
            # every rank is the usual set-up): fold the rank into the Philox key
            noise.fold_rank(dist.get_rank())

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def zero_grad(self, set_to_none=True):
        for p in self.buckets.params:
            p.grad = None
        if self.hook is not None:
            self.hook.reset()

    def sync_gradients(self):
        """Wait for / finish the gradient exchange; afterwards every `.grad` is the mean over ranks."""
        if self.hook is not None:
            self.hook.sync()

    def no_sync(self):
        """Context manager for every backward pass of a step EXCEPT the last (torch DDP's name and meaning): gradients
        accumulate in `.grad`, nothing is exchanged; the first backward pass outside it exchanges the totals.

            dpm.zero_grad()
            with dpm.no_sync():
                for xb in micro_batches[:-1]:
                    loss(dpm(xb)).backward()
            loss(dpm(micro_batches[-1])).backward()
            dpm.sync_gradients()

        In a world of one (no hook) it does nothing."""
        return _NoSync(self.hook)

    def remove(self):
        """Detach the hooks (tests that wrap the same module more than once)."""
        from . import ops
        for h in self._handles:
            h.remove()
        self._handles = []
        if self.hook is not None:
            ops.unregister_dp_hook(self.hook)
        if ops.dp_hook is self.hook:
            ops.dp_hook = None
        if self.hook is not None:
            self.hook._chip_is_ours()
            from .nn.relevance.noise import noise
            noise.fold_rank(0)                         # single-process runs afterwards draw the unfolded stream again
        self.hook = None
