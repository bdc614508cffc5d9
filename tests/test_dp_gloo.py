"""Data-parallel exchange on CPU: world_size 2, gloo.  Checks the bucket layout (reference
parameter order), the mean all-reduce through the adopted .grad views, parameter broadcast,
row sharding and the scalar reduction -- the N > 1 logic that bench.py runs over RCCL."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cplxmodule_amd import dp, ops
    from cplxmodule_amd.nn import relevance as rel
    from cplxmodule_amd.nn.relevance.noise import noise
    torch.manual_seed(100 + rank)              # different init per rank: broadcast must fix it
    # our VD layers (parameters only: their kernels need the GPU) next to plain torch layers that DO run here
    model = torch.nn.Sequential()
    model.add_module("vd", rel.CplxLinearVD(6, 5))
    model.add_module("ard", rel.LinearARD(5, 3))
    model.add_module("fc1", torch.nn.Linear(7, 9))
    model.add_module("fc2", torch.nn.Linear(9, 4))
    wrapped = dp.DataParallel(model, overlap=True, bucket_mb=200 * 4 / (1 << 20))    # ~200 floats per bucket
    layout = wrapped.buckets.names
    first = model.vd.weight.real.detach().clone()
    assert ops.dp_hook is wrapped.hook and noise._rank == rank
    seeds = noise.seed

    def step(w):
        w.zero_grad()
        assert all(p.grad is None for p in model.parameters())
        torch.manual_seed(7 + rank)            # rank-dependent data
        x = torch.randn(11, 7)
        y = model.fc2(torch.tanh(model.fc1(x)))
        (y ** 2).sum().backward()
        local = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        w.sync_gradients()
        return local

    local = step(wrapped)
    # hand-averaged reference
    ok = True
    for n, p in model.named_parameters():
        if n.startswith("fc"):
            ref = local[n].clone()
            dist.all_reduce(ref)
            ref /= world
            ok &= bool(torch.allclose(p.grad, ref, rtol=1e-6, atol=1e-7))
            ok &= p.grad.data_ptr() == wrapped.buckets.view(p).data_ptr()      # .grad IS the bucket slice
        else:
            ok &= p.grad is None                                                           # no gradient -> stays None
    launched_async = sum(1 for b in wrapped.buckets.buckets if b.launched)
    # a second step reuses the buckets
    step(wrapped)
    # the linear layers' zero-copy path: write into grad_buffer, announce early, autograd-style adoption
    wrapped.zero_grad()
    w = model.vd.log_sigma2
    buf = ops.grad_buffer(w)
    assert buf.data_ptr() == wrapped.buckets.view(w).data_ptr()
    buf.fill_(float(rank + 1))
    ops._announce(w)
    w.grad = buf
    wrapped.hook.on_grad(w)                                          # what the post-accumulate hook does
    wrapped.sync_gradients()
    early = float(w.grad.mean())                                     # mean of (1, 2) = 1.5
    others_none = all(p.grad is None or n.startswith("fc") or n == "vd.log_sigma2"
                      for n, p in model.named_parameters())
    # a layer that announces early + ANOTHER autograd path into the same parameter (in-loss regulariser, the
    # stand-alone KL node): the exchanged gradient must be the sum of both paths (round-2 advisor finding: the
    # early launch reduced the data term only)
    class EarlyLinear(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return x @ w.t()

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            dw = ops.grad_buffer(w)                   # the bucket slice, as the linear layers do
            torch.mm(g.t(), x, out=dw)
            ops._announce(w)
            return g @ w, dw

    two_path = True
    for extra_first in (False, True):
        wrapped.zero_grad()
        torch.manual_seed(21 + rank)
        x = torch.randn(11, 7)
        w1 = model.fc1.weight
        reg = (w1 ** 3).sum() * (rank + 1.0)
        data = (EarlyLinear.apply(x, w1) ** 2).sum()
        loss = reg + data if extra_first else data + reg
        expect = torch.autograd.grad(loss, w1, retain_graph=True)[0]
        loss.backward()
        wrapped.sync_gradients()
        dist.all_reduce(expect)
        two_path &= bool(torch.allclose(w1.grad, expect / world, rtol=1e-5, atol=1e-6))
        two_path &= w1.grad.data_ptr() == wrapped.buckets.view(w1).data_ptr()
    # a parameter re-allocated after wrapping stays registered (buckets are keyed by the parameter object)
    model.fc2.bias.data = model.fc2.bias.data.clone()
    moved = wrapped.buckets.view(model.fc2.bias) is not None
    wrapped.remove()
    unfolded = noise._rank == 0
    # overlap=False: nothing is launched before sync_gradients
    plain = dp.DataParallel(model, overlap=False, bucket_mb=1.0)
    plain.zero_grad()
    torch.manual_seed(7 + rank)
    (model.fc2(torch.tanh(model.fc1(torch.randn(11, 7)))) ** 2).sum().backward()
    none_launched = not any(b.launched for b in plain.buckets.buckets)
    plain.sync_gradients()
    ok2 = True
    for n, p in model.named_parameters():
        if n.startswith("fc"):
            ref = local[n].clone()
            dist.all_reduce(ref)
            ok2 &= bool(torch.allclose(p.grad, ref / world, rtol=1e-6, atol=1e-7))
    plain.remove()
    lo, hi = dp.shard_rows(11)
    kl = dp.all_reduce_scalar_mean(torch.tensor(float(rank)))
    out.put((rank, layout, ok, ok2, first.numpy(), (lo, hi), float(kl), launched_async, early, others_none,
             none_launched, seeds, ops.dp_hook is None, two_path, moved, unfolded))
    dist.destroy_process_group()


def test_dp_bucket_allreduce_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0, r1 = res
    # bucket layout: reference parameter order (named_parameters()) walked BACKWARDS in ~200-float buckets
    assert r0[1] == r1[1]
    names = [n for _, _, n in r0[1]]
    assert names == ["fc2.bias", "fc2.weight", "fc1.bias", "fc1.weight", "ard.log_sigma2", "ard.bias", "ard.weight",
                     "vd.bias.real", "vd.bias.imag", "vd.weight.real", "vd.weight.imag", "vd.log_sigma2"]
    assert [b for b, _, _ in r0[1]] == [0] * 10 + [1, 1]
    assert len({b for b, _, _ in r0[1]}) >= 2 and all(off % 4 == 0 for _, off, _ in r0[1])
    for r in (r0, r1):
        assert r[2] and r[3], "averaged gradients == hand-averaged gradients (overlap and plain)"
        assert r[7] >= 1, "at least one bucket was all-reduced asynchronously during backward"
        assert abs(r[8] - 1.5) < 1e-6 and r[9] and r[10] and r[12]
        assert r[13], "two autograd paths into an early-announced parameter: both are in the exchanged gradient"
        assert r[14] and r[15]
    np.testing.assert_array_equal(r0[4], r1[4])                    # broadcast from rank 0
    assert r0[5] == (0, 6) and r1[5] == (6, 11)
    assert r0[6] == r1[6] == 0.5
    assert r0[11] != r1[11]                                        # the ranks' noise keys differ


def test_shard_rows_cover_everything():
    from cplxmodule_amd import dp
    for n in (1, 7, 8, 1 << 20):
        for w in (1, 2, 3, 8):
            spans = [dp.shard_rows(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
