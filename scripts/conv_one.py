"""One bf16 complex conv forward shape, timed with HIP events (ablation runs: CPLXAMD_CONV_DBG)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cplxmodule_amd import conv

B, C, Co, H, W, K = [int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (64, 64, 64, 256, 256, 3))]
dev = "cuda"
torch.manual_seed(0)
xr, xi = [torch.randn(B, C, H, W, device=dev).bfloat16() for _ in range(2)]
wr, wi = [(torch.randn(Co, C, K, K, device=dev) * 0.05).bfloat16() for _ in range(2)]
geom, oshape = conv._geom(xr.shape, wr.shape, 1, 0, 1, 1)


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
t = timeit(lambda: conv.conv_fwd(xr, xi, wr, wi, None, None, geom, oshape))
tp = timeit(lambda: conv.input_grid(xr, xi, geom))
print(f"dbg={os.environ.get('CPLXAMD_CONV_DBG', '0')} fwd {t:.3f} ms (pad passes {tp:.3f} ms) -> kernel ~{t - tp:.3f} ms = "
      f"{flop / (t - tp) / 1e9:.0f} TF/s")
