"""Host-side profile (cProfile) of eager configs[4] steps: where the Python time of a launch-bound step goes.
  python scripts/r03/cfg5_host_profile.py [steps=60]"""
import cProfile, importlib.util, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cplxmodule_amd.nn import relevance as rel
spec = importlib.util.spec_from_file_location("train_sparsify", os.path.join(ROOT, "examples", "train_sparsify.py"))
ts = importlib.util.module_from_spec(spec); spec.loader.exec_module(ts)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
torch.manual_seed(0)
net = ts.Net(rel.CplxLinearARD, 8).to(dev)
x, y = ts.synthetic_complex_mnist(256, dev, seed=100)
opt = torch.optim.Adam(net.parameters(), lr=2e-3, capturable=True, fused=True)
net.train()


def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(net(x), y)
    kl = sum(rel.penalties(net), torch.zeros((), device=dev))
    (loss + 2e-3 * kl).backward()
    opt.step()


for _ in range(10):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms / step")
pr = cProfile.Profile(); pr.enable()
for _ in range(steps):
    step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime"); st.print_stats(28)
