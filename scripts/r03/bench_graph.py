"""The headline step of bench.py (CplxLinearVD 4096 -> 4096, bf16, batch 8192) eager vs replayed as one hipGraph
(cplxmodule_amd.utils.graphs.GraphedStep, noise position on the device): ms per step, same box, interleaved rounds.
  python scripts/r03/bench_graph.py [--rccl1]     (--rccl1: RCCL world of one with the collectives forced, captured too)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
from cplxmodule_amd import Cplx, dp
from cplxmodule_amd.nn import relevance as rel
from cplxmodule_amd.nn.relevance import noise
from cplxmodule_amd.utils.graphs import GraphedStep

rccl1 = "--rccl1" in sys.argv
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if rccl1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    dp.FORCE_COLLECTIVES = True
torch.manual_seed(0)
layer = rel.CplxLinearVD(4096, 4096).to(dev)
with torch.no_grad():
    layer.log_sigma2.uniform_(-12, 4)
model = dp.DataParallel(layer)
noise.manual_seed(1234)
B = 8192
x = Cplx(torch.randn(B, 4096, device=dev).bfloat16().requires_grad_(True), torch.randn(B, 4096, device=dev).bfloat16().requires_grad_(True))
klw = torch.tensor(1e-3, device=dev)
layer.train()

def step():
    model.zero_grad()
    x.real.grad = x.imag.grad = None
    y = model(x)
    kl = sum(rel.penalties(layer, reduction="sum"))
    gy_r, gy_i = y.real.detach() * 2, y.imag.detach() * 2
    torch.autograd.backward((y.real, y.imag, kl), (gy_r, gy_i, klw))
    model.sync_gradients()
    return dp.all_reduce_scalar_mean(kl) if rccl1 else kl

def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for _ in range(5):
    step()
eager0 = timed(step, 30)
noise.set_mode("philox-device")
eager_dev = timed(step, 30)
g = GraphedStep(step, modules=[layer])
res = {"eager": [eager0], "eager_devnoise": [eager_dev], "graph": []}
for _ in range(3):
    res["graph"].append(timed(g.replay, 30))
    res["eager_devnoise"].append(timed(step, 30))
kl = float(g.outputs)
print("rccl1" if rccl1 else "plain", {k: [round(v, 4) for v in vs] for k, vs in res.items()}, "kl", round(kl, 2),
      "grad finite", bool(torch.isfinite(layer.log_sigma2.grad).all()))
if rccl1:
    dist.destroy_process_group()
