"""The six convolutions of BASELINE configs[4] (fp32, batch 256, the generic NCHW kernels of csrc/conv.hip), forward / data
gradient / weight gradient each timed alone with HIP events over 50 launches: us per launch.
  python scripts/r03/cfg5_conv_layers.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from cplxmodule_amd import conv  # noqa: E402

dev = "cuda"
B, width = 256, 8
chans = [1, width, width, 2 * width, 2 * width, 4 * width, 4 * width]
strides = [1, 2, 1, 2, 1, 1]
H = 28


def timed(fn, n=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


tot = [0.0, 0.0, 0.0]
for i in range(6):
    ci, co, s = chans[i], chans[i + 1], strides[i]
    xr, xi = torch.randn(B, ci, H, H, device=dev), torch.randn(B, ci, H, H, device=dev)
    wr, wi = torch.randn(co, ci, 3, 3, device=dev), torch.randn(co, ci, 3, 3, device=dev)
    geom, oshape = conv._geom(xr.shape, wr.shape, (s, s), (1, 1), (1, 1), 1)
    gr, gi = torch.randn(oshape, device=dev), torch.randn(oshape, device=dev)
    t = [timed(lambda: conv.conv_fwd(xr, xi, wr, wi, None, None, geom, oshape)),
         timed(lambda: conv.conv_dgrad(gr, gi, wr, wi, geom, xr.shape)),
         timed(lambda: conv.conv_wgrad(gr, gi, xr, xi, geom, wr.shape))]
    tot = [a + b for a, b in zip(tot, t)]
    print(f"L{i + 1} {ci:>2}->{co:<2} {H}x{H} s{s}: fwd {t[0]:6.1f}  dgrad {t[1]:6.1f}  wgrad(+slab sum) {t[2]:6.1f} us")
    H = oshape[2]
print(f"sum: fwd {tot[0]:.1f} dgrad {tot[1]:.1f} (incl. layer 1) wgrad {tot[2]:.1f} us")
