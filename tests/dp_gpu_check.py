"""Functional check of the data-parallel overlap path on ONE GPU shared by 2 gloo ranks:
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tests/dp_gpu_check.py
hook-averaged gradients == mean over ranks of the locally computed gradients.

    python tests/dp_gpu_check.py --rccl1
runs the same check in a world of ONE rank on the RCCL backend with the collectives forced on: the
values are trivially the local ones, but every RCCL call of the data-parallel path (communicator
set-up bound to the device, ReduceOp.AVG, async work handles, broadcast, scalar all-reduce) is
issued for real -- the multi-GPU bench launches exactly these calls."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from cplxmodule_amd import Cplx, dp, ops
from cplxmodule_amd.nn import relevance as rel
from cplxmodule_amd.nn.relevance import noise


def check_sync_batchnorm(rank, world, dev):
    """Batch statistics shared between the ranks (dp.convert_sync_batchnorm) == the same layer, local statistics,
    on the concatenation of all ranks' batches: outputs, input gradients and running statistics of this rank's rows;
    weight / bias gradients summed over the ranks == the full-batch ones.  Ranks hold batches of different sizes."""
    from cplxmodule_amd import nn
    worst = 0.0
    cases = [("2d fp32", torch.float32, (6, 24), None, nn.CplxBatchNorm1d, 1e-5),
             ("planes fp32", torch.float32, (3, 12, 9, 7), None, nn.CplxBatchNorm2d, 1e-5),
             ("rows bf16 channels-last", torch.bfloat16, (2, 64, 48, 64), torch.channels_last, nn.CplxBatchNorm2d, 2e-2),
             ("rows fp32 channels-last", torch.float32, (2, 32, 48, 64), torch.channels_last, nn.CplxBatchNorm2d, 1e-5)]
    for name, dt, shape, fmt, cls, tol in cases:
        sizes = [shape[0] + r for r in range(world)]                      # rank r holds shape[0] + r samples
        g = torch.Generator(device="cpu").manual_seed(len(name))
        mk = lambda n: (torch.randn(n, *shape[1:], generator=g) * 1.5 + 0.3).to(dt).to(dev)  # noqa: E731
        full = [mk(sum(sizes)) for _ in range(4)]                         # xr, xi, gr, gi of the whole batch
        if fmt is not None:
            full = [t.contiguous(memory_format=fmt) for t in full]
        lo = sum(sizes[:rank])
        mine = [t[lo:lo + sizes[rank]] for t in full]
        if fmt is not None:
            mine = [t.contiguous(memory_format=fmt) for t in mine]
        torch.manual_seed(3)
        F = shape[1]
        outs = []
        for sync, (xr, xi, gr, gi) in ((False, full), (True, mine)):
            bn = cls(F).to(dev)
            with torch.no_grad():
                bn.weight.copy_(torch.eye(2, device=dev).unsqueeze(-1) + 0.1 * torch.arange(4 * F, device=dev).view(2, 2, F) / (4 * F))
                bn.bias.copy_(torch.linspace(-1, 1, 2 * F, device=dev).view(2, F))
            if sync:
                dp.convert_sync_batchnorm(bn)
            bn.train()
            xr, xi = xr.clone().requires_grad_(True), xi.clone().requires_grad_(True)
            y = bn(Cplx(xr, xi))
            torch.autograd.backward((y.real, y.imag), (gr, gi))
            outs.append(dict(yr=y.real.detach().float(), yi=y.imag.detach().float(), dxr=xr.grad.float(), dxi=xi.grad.float(),
                             dw=bn.weight.grad.clone(), db=bn.bias.grad.clone(), rm=bn.running_mean.clone(),
                             rv=bn.running_var.clone()))
        ref, got = outs
        for k in ("dw", "db"):
            dist.all_reduce(got[k])
        for k in ("yr", "yi", "dxr", "dxi"):
            ref[k] = ref[k][lo:lo + sizes[rank]]
        for k, want in ref.items():
            t = tol if k in ("yr", "yi", "dxr", "dxi") else max(1e-5, tol * 0.05)
            err = float((got[k] - want).abs().max() / (want.abs().max() + 1e-12))
            worst = max(worst, err if dt == torch.float32 else 0.0)
            assert err < t, (name, k, err)
    return worst


def main():
    rccl1 = "--rccl1" in sys.argv
    torch.cuda.set_device(0)
    if rccl1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        dp.FORCE_COLLECTIVES = True
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = "cuda"
    noise.fold_rank(rank)      # what DataParallel does: set it up front so that every pass below draws the same noise
    torch.manual_seed(0)
    layer = rel.CplxLinearVD(64, 96).to(dev)
    with torch.no_grad():
        layer.log_sigma2.uniform_(-8, 0)
    torch.manual_seed(10 + rank)
    x = Cplx(torch.randn(128, 64, device=dev).bfloat16(), torch.randn(128, 64, device=dev).bfloat16())
    klw = 1e-2

    def run():
        noise.manual_seed(77 + rank)
        y = layer(x)
        kl = sum(rel.penalties(layer))
        (y.real.float().square().sum() + y.imag.float().square().sum() + klw * kl).backward()

    layer.train()

    def hand_average(module, run_fn):
        """plain local gradients (no wrapper), averaged with explicit all-reduces"""
        ops.dp_hook = None
        noise.fold_rank(rank)                                # (DataParallel.remove() undoes the fold: same noise as under the wrapper)
        module.zero_grad(set_to_none=True)
        run_fn()
        out = {}
        for n, p in module.named_parameters():
            if p.grad is None:
                continue
            t = p.grad.detach().clone()
            dist.all_reduce(t)
            out[n] = t / world
        module.zero_grad(set_to_none=True)
        return out

    def check(module, run_fn, ref, tol, **kw):
        from cplxmodule_amd import _lib
        lib = _lib.load()
        model = dp.DataParallel(module, **kw)
        worst = 0.0
        for _ in range(2):                                   # two steps: the buckets are reused
            model.zero_grad()
            run_fn()
            # persistent GEMM launches are off exactly while RCCL collectives are in flight (dp.BucketHook._launch)
            in_flight = dist.get_backend() == "nccl" and kw.get("overlap", True) and any(b.launched for b in model.buckets.buckets)
            # ... as a per-call flag (ABI 19): the hook moves no process state of the library
            assert _lib.launch_flags() == (_lib.LAUNCH_SHARED if in_flight else _lib.LAUNCH_DEFAULT)
            assert lib.cplxamd_gemm_set_persistent(1) == 1
            model.sync_gradients()
            assert _lib.launch_flags() == _lib.LAUNCH_DEFAULT
            for n, p in module.named_parameters():
                if n not in ref:
                    assert p.grad is None, n
                    continue
                err = float((p.grad - ref[n]).abs().max() / (ref[n].abs().max() + 1e-12))
                worst = max(worst, err)
                assert err < tol, (n, err, kw)
                assert p.grad.data_ptr() == model.buckets.view(p).data_ptr(), n
        launched = sum(1 for b in model.buckets.buckets if b.launched)
        model.remove()
        return worst, launched

    run()                                                    # arms the fused KL: every later step is alike
    ref = hand_average(layer, run)
    worst, _ = check(layer, run, ref, 2e-3, overlap=True)    # KL part replicated, data part bf16 GEMMs
    check(layer, run, ref, 2e-3, overlap=False)
    check(layer, run, ref, 2e-3, overlap=True, bucket_mb=0.01)

    # ---- several backward passes per exchange (round 5: DataParallel.no_sync) on the real kernels ----------------------
    # (a) the reference's two-call pattern nll.backward(); (c * kl).backward(): the first pass under no_sync, the exchange
    #     happens once, with the totals == the single combined backward (fused KL epilogues, bucket storage and all)
    def run_two_calls(model):
        noise.manual_seed(77 + rank)
        y = layer(x)
        kl = sum(rel.penalties(layer))
        with model.no_sync():
            (y.real.float().square().sum() + y.imag.float().square().sum()).backward()
            assert not any(b.launched for b in model.buckets.buckets), "nothing is exchanged inside no_sync"
        (klw * kl).backward()

    # (b) gradient accumulation over two micro-batches (rows 0:64 under no_sync, rows 64:128 outside) == one pass over
    #     both with the same noise per row is not expressible (the noise stream is per launch), so compare with the
    #     hand-made sum of the two micro-batch gradients instead
    halves = [Cplx(x.real[i:i + 64].contiguous(), x.imag[i:i + 64].contiguous()) for i in (0, 64)]

    def micro(i):
        noise.manual_seed(500 + 10 * rank + i)
        y = layer(halves[i])
        kl = sum(rel.penalties(layer))
        (y.real.float().square().sum() + y.imag.float().square().sum() + 0.5 * klw * kl).backward()

    def run_micro_plain():
        micro(0)
        micro(1)                          # plain autograd accumulation into .grad

    ref_micro = hand_average(layer, run_micro_plain)
    model = dp.DataParallel(layer)
    for _ in range(2):
        model.zero_grad()
        run_two_calls(model)
        model.sync_gradients()
        for n, p in layer.named_parameters():
            err = float((p.grad - ref[n]).abs().max() / (ref[n].abs().max() + 1e-12))
            assert err < 2e-3, ("two-call pattern under no_sync", n, err)
            assert p.grad.data_ptr() == model.buckets.view(p).data_ptr(), n
        model.zero_grad()
        with model.no_sync():
            micro(0)
        micro(1)
        model.sync_gradients()
        for n, p in layer.named_parameters():
            err = float((p.grad - ref_micro[n]).abs().max() / (ref_micro[n].abs().max() + 1e-12))
            assert err < 2e-3, ("micro-batches under no_sync", n, err)
    # (c) WITHOUT no_sync a second pass into exchanged buckets is an error, not a torn gradient
    model.zero_grad()
    micro(0)
    if any(b.launched for b in model.buckets.buckets):
        try:
            micro(1)
            raise AssertionError("a second backward pass into an exchanged bucket must raise")
        except RuntimeError as e:
            assert "no_sync" in str(e), e
    model.zero_grad()
    model.remove()

    # cfg5-shaped model: 6 x (CplxConv2d + CplxBatchNorm2d + split-ReLU) + CplxLinearARD head, several
    # buckets; every parameter's averaged gradient == the hand-averaged one
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "train_sparsify", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples",
                                       "train_sparsify.py"))
    ts = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ts)
    torch.manual_seed(0)
    net = ts.Net(rel.CplxLinearARD, width=4).to(dev)
    xs, ys = ts.synthetic_complex_mnist(32, dev, seed=3 + rank)

    def run_net():
        noise.manual_seed(5)
        loss = torch.nn.functional.cross_entropy(net(xs), ys)
        kl = sum(rel.penalties(net), torch.zeros((), device=dev))
        (loss + 1e-3 * kl).backward()

    net.train()
    run_net()
    net.zero_grad(set_to_none=True)
    bn_state = {k: v.clone() for k, v in net.state_dict().items() if "running" in k or "num_batches" in k}

    def run_net_same_stats():
        net.load_state_dict(bn_state, strict=False)          # identical running statistics for every pass
        run_net()

    ref5 = hand_average(net, run_net_same_stats)
    w5, nb = check(net, run_net_same_stats, ref5, 5e-4, overlap=True, bucket_mb=0.05)
    assert nb >= 3, nb
    check(net, run_net_same_stats, ref5, 5e-4, overlap=False)
    worst = max(worst, w5)
    worst = max(worst, check_sync_batchnorm(rank, world, dev))
    kl_mean = dp.all_reduce_scalar_mean(sum(rel.penalties(layer)))
    assert torch.isfinite(kl_mean)
    if rank == 0:
        print(f"dp_gpu_check OK: backend={dist.get_backend()} world={world}, worst relative deviation {worst:.2e}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
