import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
from cplxmodule_amd import Cplx, dp
from cplxmodule_amd.nn import relevance as rel
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29573")
os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dp.FORCE_COLLECTIVES = True
torch.manual_seed(0)
layer = rel.CplxLinearVD(4096, 4096).to(dev)
model = dp.DataParallel(layer)
x = Cplx(torch.randn(8192, 4096, device=dev).bfloat16().requires_grad_(True), torch.randn(8192, 4096, device=dev).bfloat16().requires_grad_(True))
klw = torch.tensor(1e-3, device=dev)
layer.train()
for it in range(3):
    model.zero_grad(); x.real.grad = x.imag.grad = None
    y = model(x)
    kl = sum(rel.penalties(layer))
    torch.autograd.backward((y.real, y.imag, kl), (y.real.detach() * 2, y.imag.detach() * 2, klw))
    for b in model.buckets.buckets:
        print(it, b.index, [n for n, *_ in b.entries], "launched", b.launched, "early", b.early, "in_order", b.in_order, "events", len(b.events), flush=True)
    model.sync_gradients()
torch.cuda.synchronize()
dist.destroy_process_group()
