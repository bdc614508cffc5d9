"""BASELINE configs[4] (6 x (CplxConv2d + CplxBatchNorm2d + split-ReLU) + CplxLinearARD head on synthetic complex MNIST, one
train step = forward + loss + KL + backward [+ gradient exchange] + Adam): eager vs one hipGraph replay per step, without
and with dp.DataParallel (RCCL world of one, collectives forced: the all-reduce kernels are captured in the graph).
  python scripts/r03/cfg5_graph.py [--rccl1] [--batch 256] [--width 8]"""
import argparse, importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from cplxmodule_amd import dp
from cplxmodule_amd.nn import relevance as rel
from cplxmodule_amd.utils.graphs import GraphedStep
spec = importlib.util.spec_from_file_location("train_sparsify", os.path.join(ROOT, "examples", "train_sparsify.py"))
ts = importlib.util.module_from_spec(spec); spec.loader.exec_module(ts)

ap = argparse.ArgumentParser()
ap.add_argument("--rccl1", action="store_true"); ap.add_argument("--batch", type=int, default=256); ap.add_argument("--width", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
if a.rccl1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29575")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    dp.FORCE_COLLECTIVES = True
torch.manual_seed(0)
net = ts.Net(rel.CplxLinearARD, a.width).to(dev)
x, y = ts.synthetic_complex_mnist(a.batch, dev, seed=100)
par = dp.DataParallel(net, bucket_mb=1.0) if a.rccl1 else None
rel.noise.set_mode("philox-device")
opt = torch.optim.Adam(net.parameters(), lr=2e-3, capturable=True, fused=True)
net.train()

def step():
    if par is not None:
        par.zero_grad()
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(net(x), y)
    kl = sum(rel.penalties(net), torch.zeros((), device=dev))
    (loss + 2e-3 * kl).backward()
    if par is not None:
        par.sync_gradients()
    opt.step()
    return loss.detach(), kl.detach()

def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for _ in range(5):
    step()
eager = [timed(step, 50)]
g = GraphedStep(step, modules=[net])
l0 = float(g.outputs[0])
graph = []
for _ in range(3):
    graph.append(timed(g.replay, 50))
    eager.append(timed(step, 50))
print(("rccl1 " if a.rccl1 else "plain ") + f"batch {a.batch} width {a.width}: eager ms/step {[round(v, 3) for v in eager]}  graph replay {[round(v, 3) for v in graph]}"
      f"  buckets {len(par.buckets.buckets) if par else 0}  loss {l0:.3f} -> {float(g.outputs[0]):.3f}", flush=True)
if a.rccl1:
    dist.destroy_process_group()
