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
    # reference: plain local gradients, averaged by hand
    ops.dp_hook = None
    layer.zero_grad(set_to_none=True)
    run()
    names = [n for n, _ in layer.named_parameters()]
    local = [p.grad.detach().clone() for p in layer.parameters()]
    ref = []
    for g in local:
        t = g.clone()
        dist.all_reduce(t)
        ref.append(t / world)
    # overlap path
    model = dp.DataParallel(layer, overlap=True)
    model.zero_grad()
    run()
    model.sync_gradients()
    worst = 0.0
    for n, p, r in zip(names, layer.parameters(), ref):
        err = float((p.grad - r).abs().max() / (r.abs().max() + 1e-12))
        worst = max(worst, err)
        assert err < 2e-3, (n, err)          # KL part is replicated, data part bf16 GEMMs
    # and without overlap (flat bucket through .grad views)
    ops.dp_hook = None
    model2 = dp.DataParallel(layer, overlap=False)
    model2.zero_grad()
    run()
    model2.sync_gradients()
    for n, p, r in zip(names, layer.parameters(), ref):
        err = float((p.grad - r).abs().max() / (r.abs().max() + 1e-12))
        assert err < 2e-3, (n, err)
    kl_mean = dp.all_reduce_scalar_mean(sum(rel.penalties(layer)))
    assert torch.isfinite(kl_mean)
    if rank == 0:
        print(f"dp_gpu_check OK: backend={dist.get_backend()} world={world}, worst relative deviation {worst:.2e}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
