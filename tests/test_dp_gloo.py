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


# ---- the same logic at the world size the driver will launch (8 ranks, CPU, gloo) ---------------
def _worker8(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from cplxmodule_amd import dp, ops
    from cplxmodule_amd.nn import relevance as rel
    from cplxmodule_amd.nn.relevance.noise import noise
    torch.manual_seed(1000 + rank)
    model = torch.nn.Sequential()
    model.add_module("vd", rel.CplxLinearVD(6, 5))             # parameters only (bucket layout); its kernels need the GPU
    model.add_module("fc1", torch.nn.Linear(7, 16))
    model.add_module("fc2", torch.nn.Linear(16, 16))
    model.add_module("fc3", torch.nn.Linear(16, 3))
    wrapped = dp.DataParallel(model, overlap=True, bucket_mb=150 * 4 / (1 << 20))     # ~150 floats per bucket
    nb = len(wrapped.buckets.buckets)
    key = noise.seed
    net = lambda x: model.fc3(torch.tanh(model.fc2(torch.tanh(model.fc1(x)))))  # noqa: E731

    # one global data set, rows sharded unevenly (1003 = 8 * 125 + 3); per-row losses weighted so that the MEAN over
    # ranks of the local gradients is the gradient of the global mean loss
    N = 1003
    g = torch.Generator().manual_seed(5)
    X, Y = torch.randn(N, 7, generator=g), torch.randn(N, 3, generator=g)
    lo, hi = dp.shard_rows(N)

    def local_loss(x, y):
        return ((net(x) - y) ** 2).sum() * (world / N)

    wrapped.zero_grad()
    local_loss(X[lo:hi], Y[lo:hi]).backward()
    async_launched = sum(1 for b in wrapped.buckets.buckets if b.launched)
    wrapped.sync_gradients()
    got = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    wrapped.zero_grad()
    model.zero_grad(set_to_none=True)
    full = torch.autograd.grad(((net(X) - Y) ** 2).sum() / N, [p for n, p in model.named_parameters() if n.startswith("fc")])
    names = [n for n, _ in model.named_parameters() if n.startswith("fc")]
    err = max(float((got[n] - f).abs().max() / (f.abs().max() + 1e-12)) for n, f in zip(names, full))
    vd_none = all(p.grad is None for n, p in model.named_parameters() if n.startswith("vd"))

    # gradient accumulation: two micro-batches under no_sync + a third outside == one pass over the three
    mid1, mid2 = lo + (hi - lo) // 3, lo + 2 * (hi - lo) // 3
    wrapped.zero_grad()
    with wrapped.no_sync():
        local_loss(X[lo:mid1], Y[lo:mid1]).backward()
        none_in_flight = not any(b.launched for b in wrapped.buckets.buckets)
        local_loss(X[mid1:mid2], Y[mid1:mid2]).backward()
        none_in_flight &= not any(b.launched for b in wrapped.buckets.buckets)
    local_loss(X[mid2:hi], Y[mid2:hi]).backward()
    wrapped.sync_gradients()
    acc_err = max(float((model.get_parameter(n).grad - got[n]).abs().max() / (got[n].abs().max() + 1e-12)) for n in names)
    in_bucket = all(model.get_parameter(n).grad.data_ptr() == wrapped.buckets.view(model.get_parameter(n)).data_ptr()
                    for n in names)

    # a second backward pass WITHOUT no_sync reaches an exchanged bucket: loud error, not a torn gradient
    wrapped.zero_grad()
    local_loss(X[lo:mid1], Y[lo:mid1]).backward()
    raised = False
    try:
        local_loss(X[mid1:hi], Y[mid1:hi]).backward()
    except RuntimeError as e:
        raised = "no_sync" in str(e)
    wrapped.zero_grad()                 # (waits for what is in flight)

    # a parameter used by two zero-copy producers in one pass (tied weights): the second one must not get the same
    # bucket storage -- autograd would add one of the two gradients twice
    class Producer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return x @ w.t()

        @staticmethod
        def backward(ctx, gy):
            x, w = ctx.saved_tensors
            dw = ops.grad_buffer(w)
            torch.mm(gy.t(), x, out=dw)
            ops._announce(w)
            return gy @ w, dw

    w2 = model.fc2.weight
    h = torch.randn(9, 16, generator=torch.Generator().manual_seed(11 + rank))
    tied = lambda f: (f(torch.tanh(f(h, w2)), w2) ** 2).sum()  # noqa: E731
    expect = torch.autograd.grad(tied(lambda a, w: a @ w.t()), w2)[0]
    dist.all_reduce(expect)
    wrapped.zero_grad()
    tied(Producer.apply).backward()
    wrapped.sync_gradients()
    tied_err = float((w2.grad - expect / world).abs().max() / expect.abs().max())

    kl = float(dp.all_reduce_scalar_mean(torch.tensor(float(rank))))
    wrapped.remove()
    out.put((rank, nb, key, (lo, hi), err, vd_none, async_launched, acc_err, none_in_flight, in_bucket, raised, tied_err, kl))
    dist.destroy_process_group()


def test_dp_world8_gloo():
    world = 8
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(world))
    assert all(r[1] >= 3 for r in res), "at least three buckets per rank"
    assert len({r[2] for r in res}) == world, "eight distinct rank-folded Philox keys"
    spans = [r[3] for r in res]
    assert spans[0][0] == 0 and spans[-1][1] == 1003 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert sorted(b - a for a, b in spans) == [125] * 5 + [126] * 3, "uneven shards differ by one row"
    for r in res:
        assert r[4] < 2e-5, ("mean over ranks of the weighted local gradients == gradient of the global mean loss", r[4])
        assert r[5] and r[6] >= 1
        assert r[7] < 2e-5 and r[8] and r[9], "no_sync: accumulated micro-batches == one pass; nothing launched inside"
        assert r[10], "second backward without no_sync raises"
        assert r[11] < 2e-5, "tied weights with two zero-copy producers"
        assert r[12] == 3.5


def test_init_process_group_channel_cap_wins_over_the_environment(monkeypatch):
    """an explicit max_channels must not be ignored because NCCL_MAX_NCHANNELS is already set (setdefault did);
    the env form CPLXAMD_RCCL_MAX_CHANNELS feeds the same argument; no process group is created here."""
    import warnings
    from cplxmodule_amd import dp
    seen = {}
    monkeypatch.setattr(dist, "init_process_group", lambda backend, **kw: seen.update(backend=backend, **kw))
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "32")
    monkeypatch.delenv("CPLXAMD_RCCL_MAX_CHANNELS", raising=False)
    monkeypatch.delenv("TORCH_NCCL_HIGH_PRIORITY", raising=False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        dp.init_process_group("gloo", max_channels=8, rank=0, world_size=1)
    assert os.environ["NCCL_MAX_NCHANNELS"] == "8" and any("NCCL_MAX_NCHANNELS" in str(x.message) for x in w)
    assert os.environ["TORCH_NCCL_HIGH_PRIORITY"] == "1" and os.environ["MASTER_ADDR"] == "127.0.0.1"
    assert seen["backend"] == "gloo" and seen["rank"] == 0
    monkeypatch.setenv("CPLXAMD_RCCL_MAX_CHANNELS", "12")
    dp.init_process_group("gloo", rank=0, world_size=1)
    assert os.environ["NCCL_MAX_NCHANNELS"] == "12"
    monkeypatch.delenv("CPLXAMD_RCCL_MAX_CHANNELS")
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "20")
    dp.init_process_group("gloo", rank=0, world_size=1)              # no cap asked for: the environment stands
    assert os.environ["NCCL_MAX_NCHANNELS"] == "20"


def _worker_two_models(rank, world, port, out):
    """VERDICT r05 housekeeping: the hook is found per parameter -- TWO DataParallel wrappers in one process each exchange
    their own model's gradients through their own buckets (zero-copy path included), and an unwrapped third model is left
    alone."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cplxmodule_amd import dp, ops
    torch.manual_seed(3)
    a, b, c = torch.nn.Linear(5, 4), torch.nn.Linear(6, 3), torch.nn.Linear(5, 2)
    wa, wb = dp.DataParallel(a), dp.DataParallel(b)
    ok = ops.hook_of(a.weight) is wa.hook and ops.hook_of(b.weight) is wb.hook and wa.hook is not wb.hook
    ok &= ops.dp_hook is wa.hook                      # the first wrapper is the process-wide fallback
    ok &= ops.grad_buffer(b.weight).data_ptr() == wb.buckets.view(b.weight).data_ptr()
    ok &= ops.grad_buffer(c.weight).data_ptr() not in (wa.buckets.buckets[0].flat.data_ptr(), wb.buckets.buckets[0].flat.data_ptr())
    for w_ in (wa, wb):
        w_.zero_grad()
    torch.manual_seed(10 + rank)
    xa, xb = torch.randn(7, 5), torch.randn(7, 6)
    loss = (a(xa) ** 2).sum() + (b(xb) ** 2).sum() + (c(xa) ** 2).sum()
    ps = [p for m in (a, b, c) for p in m.parameters()]
    # (the local gradients by autograd.grad -- no hooks run: reading p.grad after backward() would race the bucket
    #  all-reduces the hooks have already launched)
    local = {id(p): g.clone() for p, g in zip(ps, torch.autograd.grad(loss, ps, retain_graph=True))}
    loss.backward()
    wa.sync_gradients()
    wb.sync_gradients()
    for m in (a, b):
        for p in m.parameters():
            ref = local[id(p)].clone()
            dist.all_reduce(ref)
            ok &= bool(torch.allclose(p.grad, ref / world, rtol=1e-6, atol=1e-7))
    for p in c.parameters():                          # nobody exchanged the unwrapped model's gradients
        ok &= bool(torch.equal(p.grad, local[id(p)]))
    wb.remove()
    ok &= ops.hook_of(b.weight) is wa.hook            # (fallback only; wa's buckets do not know b's parameters)
    ok &= ops.grad_buffer(b.weight).data_ptr() != wb.buckets.view(b.weight).data_ptr()
    wa.remove()
    ok &= ops.dp_hook is None and ops.hook_of(a.weight) is None and not ops._dp_hooks
    out.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_two_wrapped_models_in_one_process_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_two_models, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]
