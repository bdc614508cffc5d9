"""BASELINE configs[4] step (scripts/r03/cfg5_graph.py's model, batch 256) for a kernel trace: 5 eager warm-up steps, then N
eager steps (default) or N hipGraph replays (--graph) and nothing else, so that calls / N = launches per step.
  rocprofv3 --kernel-trace --stats -- python scripts/r03/cfg5_steps_prof.py [--graph] [--steps 100]"""
import argparse, importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cplxmodule_amd.nn import relevance as rel
from cplxmodule_amd.utils.graphs import GraphedStep
spec = importlib.util.spec_from_file_location("train_sparsify", os.path.join(ROOT, "examples", "train_sparsify.py"))
ts = importlib.util.module_from_spec(spec); spec.loader.exec_module(ts)
ap = argparse.ArgumentParser()
ap.add_argument("--graph", action="store_true"); ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--batch", type=int, default=256); ap.add_argument("--width", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
torch.manual_seed(0)
net = ts.Net(rel.CplxLinearARD, a.width).to(dev)
x, y = ts.synthetic_complex_mnist(a.batch, dev, seed=100)
rel.noise.set_mode("philox-device")
opt = torch.optim.Adam(net.parameters(), lr=2e-3, capturable=True, fused=True)
net.train()


def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(net(x), y)
    kl = sum(rel.penalties(net), torch.zeros((), device=dev))
    (loss + 2e-3 * kl).backward()
    opt.step()
    return loss.detach(), kl.detach()


fn = step
if a.graph:
    g = GraphedStep(step, modules=[net], warmup=5)
    fn = g.replay
else:
    for _ in range(5):
        step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    fn()
torch.cuda.synchronize()
print(f"{'graph' if a.graph else 'eager'} batch {a.batch} width {a.width}: {(time.perf_counter() - t0) / a.steps * 1e3:.3f} ms / step over {a.steps} steps")
