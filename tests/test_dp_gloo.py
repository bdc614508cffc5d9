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
    from cplxmodule_amd import dp
    from cplxmodule_amd.nn import relevance as rel
    torch.manual_seed(100 + rank)              # different init per rank: broadcast must fix it
    model = torch.nn.Sequential(rel.CplxLinearVD(6, 5), rel.LinearARD(5, 3))
    wrapped = dp.DataParallel(model, overlap=False)
    names = wrapped.bucket.names
    first = model[0].weight.real.detach().clone()
    wrapped.zero_grad()
    # emulate a backward: rank-dependent gradients written through .grad (views of the bucket)
    for i, p in enumerate(model.parameters()):
        p.grad.add_(float(rank + 1) * (i + 1))
    wrapped.sync_gradients()
    grads = [float(p.grad.mean()) for p in model.parameters()]
    lo, hi = dp.shard_rows(11)
    kl = dp.all_reduce_scalar_mean(torch.tensor(float(rank)))
    alias = (model[0].weight.real.grad.data_ptr() ==
             wrapped.bucket.views[names.index("0.weight.real")].data_ptr())
    # overlap path: the hook averages a layer's flat gradient buffer asynchronously, the wrapper
    # then averages only what the hook has not handled
    over = dp.DataParallel(model, overlap=True)
    over.zero_grad()
    assert all(p.grad is None for p in model.parameters())
    flat = torch.full((7,), float(rank + 1))
    h = over.hook.reduce(flat, (model[0].log_sigma2.data_ptr(),))
    over.hook.finish(h)
    assert torch.allclose(flat, torch.full((7,), 1.5))
    model[0].log_sigma2.grad = torch.full_like(model[0].log_sigma2, 9.0 + rank)   # "already averaged"
    model[1].weight.grad = torch.full_like(model[1].weight, float(rank))          # left to the wrapper
    over.sync_gradients()
    assert float(model[0].log_sigma2.grad.mean()) == 9.0 + rank                    # untouched
    assert abs(float(model[1].weight.grad.mean()) - 0.5) < 1e-6                    # averaged
    from cplxmodule_amd import ops
    ops.dp_hook = None
    out.put((rank, names, grads, first.numpy(), (lo, hi), float(kl),
             wrapped.bucket.flat.numel(), alias))
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
    (r0, names0, g0, w0, s0, kl0, n0, alias0), (r1, names1, g1, w1, s1, kl1, n1, alias1) = res
    assert names0 == names1 == ["0.log_sigma2", "0.weight.imag", "0.weight.real", "0.bias.imag",
                                "0.bias.real", "1.weight", "1.bias", "1.log_sigma2"]
    np.testing.assert_array_equal(w0, w1)                    # broadcast from rank 0
    want = [1.5 * (i + 1) for i in range(len(g0))]           # mean of (1, 2) * (i + 1)
    np.testing.assert_allclose(g0, want)
    np.testing.assert_allclose(g1, want)
    assert s0 == (0, 6) and s1 == (6, 11)
    assert kl0 == kl1 == 0.5
    assert n0 == 30 + 30 + 30 + 5 + 5 + 15 + 3 + 15 and alias0 and alias1


def test_shard_rows_cover_everything():
    from cplxmodule_amd import dp
    for n in (1, 7, 8, 1 << 20):
        for w in (1, 2, 3, 8):
            spans = [dp.shard_rows(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
