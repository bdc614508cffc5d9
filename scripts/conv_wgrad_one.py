"""bf16 complex conv weight-gradient timing (ablation: CPLXAMD_CONV_DBG)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cplxmodule_amd import conv

B, C, Co, H, W, K = [int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (64, 64, 64, 256, 256, 3))]
dev = "cuda"
torch.manual_seed(0)
xr, xi = [torch.randn(B, C, H, W, device=dev).bfloat16() for _ in range(2)]
geom, oshape = conv._geom(xr.shape, (Co, C, K, K), 1, 0, 1, 1)
gr, gi = [torch.randn(oshape, device=dev).bfloat16() for _ in range(2)]


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


flop = 8.0 * B * Co * oshape[2] * oshape[3] * C * K * K
t = timeit(lambda: conv.conv_wgrad(gr, gi, xr, xi, geom, (Co, C, K, K)))
tp = timeit(lambda: (conv.input_grid(xr, xi, geom), conv.grad_grid(gr, gi, geom)))
print(f"dbg={os.environ.get('CPLXAMD_CONV_DBG', '0')} wgrad {t:.3f} ms (4 pad passes ~{tp:.3f} ms) -> kernel ~{t - tp:.3f} ms = "
      f"{flop / (t - tp) / 1e9:.0f} TF/s")
