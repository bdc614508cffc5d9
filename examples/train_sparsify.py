"""cfg5: Deep-Complex-Net style model -- 6 x (CplxConv2d + CplxBatchNorm2d + split-ReLU) and a
CplxLinearARD head -- on synthetic complex "MNIST" (2-d FFT of seeded 28x28 prototype digits +
noise; no dataset is available offline), trained in the reference's three phases
(tests/test_relevance.py:98-253, tests/test_mnist.py:199-245 of the reference):

    dense (CplxLinear head)  ->  ARD (CplxLinearARD, loss + klw * sum(penalties))
                             ->  masked fine-tune (CplxLinearMasked with the ARD masks)

    python examples/train_sparsify.py --steps 60
    python -m torch.distributed.run --nproc-per-node 8 examples/train_sparsify.py   (data parallel)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from cplxmodule_amd import Cplx, dp, nn  # noqa: E402
from cplxmodule_amd.nn import masked, relevance as rel  # noqa: E402
from cplxmodule_amd.nn.utils import sparsity  # noqa: E402


def synthetic_complex_mnist(n, device, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    protos = torch.rand(10, 28, 28, generator=g)
    labels = torch.randint(0, 10, (n,), generator=g)
    imgs = protos[labels] + 0.3 * torch.rand(n, 28, 28, generator=g)
    z = torch.fft.fft2(imgs) / 28.0
    x = Cplx(z.real.unsqueeze(1).to(device, dtype).contiguous(), z.imag.unsqueeze(1).to(device, dtype).contiguous())
    return x, labels.to(device)


class Net(torch.nn.Module):
    def __init__(self, head, width=8):
        super().__init__()
        chans = [1, width, width, 2 * width, 2 * width, 4 * width, 4 * width]
        strides = [1, 2, 1, 2, 1, 1]
        layers = []
        for i in range(6):
            layers += [nn.CplxConv2d(chans[i], chans[i + 1], 3, stride=strides[i], padding=1),
                       nn.CplxBatchNorm2d(chans[i + 1]), nn.CplxToCplx[torch.nn.ReLU]()]
        self.features = torch.nn.Sequential(*layers)
        self.head = head(chans[-1] * 7 * 7, 10)

    def forward(self, x):
        z = self.features(x)
        z = self.head(z.flatten(1))
        return abs(z)                       # logits = modulus of the 10 complex outputs


REPLAY_MS = []          # per phase: ms per replayed step (train_graph)


def train_graph(model, x, y, steps, klw, lr, wrap=False):
    """The whole step -- forward, loss + KL, backward [, gradient exchange], Adam -- captured once in a hipGraph
    (cplxmodule_amd.utils.graphs.GraphedStep) and replayed: the model is launch-bound (tens of small kernels per
    layer), so this is where a HIP graph replaces what a tracing compiler would be used for.  The LRT noise position
    lives on the device ("philox-device"), so every replay draws fresh noise.  With `wrap` the step runs under
    dp.DataParallel and the bucket all-reduces (RCCL kernels) are part of the captured graph."""
    from cplxmodule_amd.utils.graphs import GraphedStep
    rel.noise.set_mode("philox-device")
    par = dp.DataParallel(model) if wrap else None
    opt = torch.optim.Adam(model.parameters(), lr=lr, capturable=True, fused=True)   # one multi-tensor kernel per step, not 80 tiny ones
    model.train()
    hist = []

    def step():
        if par is not None:
            par.zero_grad()
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(model(x), y)
        kl = sum(rel.penalties(model), torch.zeros((), device=y.device))
        (loss + klw * kl).backward()
        if par is not None:
            par.sync_gradients()
        opt.step()
        return loss.detach(), kl.detach()

    g = GraphedStep(step, modules=[model], warmup=3)       # warm-up: allocator, caches, Adam state (3 real steps)
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps - 3):
        out = g.replay()
        hist.append((out[0].clone(), out[1].clone()))       # no host sync inside the loop
    torch.cuda.synchronize()
    REPLAY_MS.append((time.perf_counter() - t0) / max(steps - 3, 1) * 1e3)   # steady state, without warm-up / capture
    rel.noise.set_mode("philox")
    if par is not None:
        par.remove()
    return [(float(a), float(b)) for a, b in hist]


def train(model, x, y, steps, klw, lr, wrap, graph=False):
    if graph:
        return train_graph(model, x, y, steps, klw, lr, wrap)
    par = dp.DataParallel(model) if wrap else None
    opt = torch.optim.Adam(model.parameters(), lr=lr, fused=True)
    model.train()
    hist = []
    for _ in range(steps):
        (par or opt).zero_grad()
        if par is None:
            opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(model(x), y)
        kl = sum(rel.penalties(model), torch.zeros((), device=y.device))
        (loss + klw * kl).backward()
        if par is not None:
            par.sync_gradients()
        opt.step()
        hist.append((float(loss), float(kl)))
    return hist


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--width", type=int, default=8)
    ap.add_argument("--klw", type=float, default=2e-3)
    ap.add_argument("--threshold", type=float, default=1.0)
    ap.add_argument("--graph", action="store_true", help="capture each phase's training step in a hipGraph")
    ap.add_argument("--time", action="store_true", help="print the wall time of the three phases")
    a = ap.parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dp.init_process_group("nccl", device=dev)           # (RCCL's stream on its own hardware queue: see dp.py)
    rank = dist.get_rank() if world > 1 else 0
    torch.manual_seed(0)
    x, y = synthetic_complex_mnist(a.batch, dev, seed=100 + rank)

    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dense = Net(nn.CplxLinear, a.width).to(dev)
    h1 = train(dense, x, y, a.steps, 0.0, 2e-3, world > 1, a.graph)

    ard = Net(rel.CplxLinearARD, a.width).to(dev)
    ard.load_state_dict(dense.state_dict(), strict=False)
    h2 = train(ard, x, y, 2 * a.steps, a.klw, 5e-3, world > 1, a.graph)
    sp = sparsity(ard, hard=True, threshold=a.threshold)

    masks = rel.compute_ard_masks(ard, hard=False, threshold=a.threshold)
    state, masks = masked.binarize_masks(ard.state_dict(), masks)
    fine = Net(masked.CplxLinearMasked, a.width).to(dev)
    fine.load_state_dict(state, strict=False)
    masked.deploy_masks(fine, state_dict=masks)
    h3 = train(fine, x, y, a.steps, 0.0, 1e-3, world > 1, a.graph)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    fine.eval()
    with torch.no_grad():
        acc = float((fine(x).argmax(1) == y).float().mean())
    if rank == 0:
        print(f"dense  loss {h1[0][0]:.3f} -> {h1[-1][0]:.3f}")
        print(f"ard    loss {h2[0][0]:.3f} -> {h2[-1][0]:.3f}   kl {h2[0][1]:.1f} -> {h2[-1][1]:.1f}   "
              f"sparsity@{a.threshold} {sp:.3f}")
        print(f"masked loss {h3[0][0]:.3f} -> {h3[-1][0]:.3f}   train acc {acc:.3f}   "
              f"kept {int(masks['head.mask'].sum())}/{masks['head.mask'].numel()} head weights")
        if a.time:
            print(f"{4 * a.steps} steps in {elapsed:.2f} s = {1e3 * elapsed / (4 * a.steps):.2f} ms/step "
                  f"({'hipGraph replay, incl. three warm-ups + captures' if a.graph else 'eager'})")
            if a.graph and REPLAY_MS:
                print("replayed steps alone: " + " / ".join(f"{v:.3f}" for v in REPLAY_MS[-3:]) + " ms per step (dense / ARD / masked)")
    if world > 1:
        dist.destroy_process_group()
    return dict(dense=h1, ard=h2, masked=h3, sparsity=sp, acc=acc, masks=masks)


if __name__ == "__main__":
    main()
