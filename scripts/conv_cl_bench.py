"""Time the channels-last conv kernels (csrc/conv_cl.hip, conv_cl_wgrad.hip) on the cfg3 layer: CplxConv2d(64, 64, 3)
on 256 x 256 (PAD=0, the layer's default, or PAD=1), against the round-1 path (pad passes + conv_nhwc*.hip).
   python scripts/conv_cl_bench.py [B]      env: PAD (0), ONLY (prefix filter)"""
import os
import sys
import statistics

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cplxmodule_amd import conv, _lib  # noqa: E402
from cplxmodule_amd._lib import call, ptr, stream_ptr  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
PAD = int(os.environ.get("PAD", "0"))
C = Co = 64
H = W = 256
Ho, Wo = H + 2 * PAD - 2, W + 2 * PAD - 2
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
xr, xi = (torch.randn(B, C, H, W, device=dev).to(bf) for _ in range(2))
wr, wi = (torch.randn(Co, C, 3, 3, device=dev).mul(0.05).to(bf) for _ in range(2))
br, bi = torch.randn(Co, device=dev), torch.randn(Co, device=dev)
geom, oshape = conv._geom(xr.shape, wr.shape, 1, PAD, 1, 1)
xr_cl, xi_cl = conv.to_channels_last(xr), conv.to_channels_last(xi)
gr_n, gi_n = (torch.randn(B, Co, Ho, Wo, device=dev).to(bf) for _ in range(2))
gr_cl, gi_cl = conv.to_channels_last(gr_n), conv.to_channels_last(gi_n)
flop = 8.0 * B * Ho * Wo * C * Co * 9


def timeit(fn, n=10, rounds=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return statistics.median(ts), min(ts)


wp, wpd = conv._cl_pack(wr, wi, False), conv._cl_pack(wr, wi, True)
yr = torch.empty((B, Co, Ho, Wo), dtype=bf, device=dev, memory_format=torch.channels_last)
yi = torch.empty_like(yr)
dxr = torch.empty((B, C, H, W), dtype=bf, device=dev, memory_format=torch.channels_last)
dxi = torch.empty_like(dxr)
ENTRY = os.environ.get("ENTRY", "cplxamd_conv2d_cl")
ws = torch.empty(int(_lib.load().cplxamd_conv2d_cl_ws_bytes(Co)), dtype=torch.uint8, device=dev)


def k_fwd():
    call(ENTRY, ptr(xr_cl), ptr(xi_cl), ptr(wp), ptr(br), ptr(bi), ptr(yr), ptr(yi), B, H, W, C, Co, 3, 3,
         1, 1, PAD, PAD, 0, ptr(ws), ws.numel(), stream_ptr())


def k_dgrad():
    call(ENTRY, ptr(gr_cl), ptr(gi_cl), ptr(wpd), None, None, ptr(dxr), ptr(dxi), B, H, W, Co, C, 3, 3,
         1, 1, PAD, PAD, 1, ptr(ws), ws.numel(), stream_ptr())


a_cl = (xr_cl.float() ** 2).to(bf).contiguous(memory_format=torch.channels_last)
S_w = torch.rand(Co, C, 3, 3, device=dev).mul(0.01).to(bf)
RF = 0.25                                            # real flops = complex / 4
rows = [("cl kernel fwd (+bias)", k_fwd), ("cl kernel dgrad", k_dgrad),
        ("cl REAL fwd (x0.25 flop)", lambda: conv.cl_conv_real(a_cl, S_w, None, geom)),
        ("cl REAL dgrad (x0.25 flop)", lambda: conv.cl_conv_real(gr_cl, S_w, None, geom, dgrad=True)),
        ("cl REAL wgrad (x0.25 flop)", lambda: conv.cl_wgrad_real(gr_cl, a_cl, geom, wr.shape)),
        ("cl wgrad (+ slab reduce)", lambda: conv.cl_wgrad(gr_cl, gi_cl, xr_cl, xi_cl, geom, wr.shape)),
        ("cl fwd incl. weight pack", lambda: conv.cl_conv(xr_cl, xi_cl, wr, wi, br, bi, geom)),
        ("r01 fwd (2 pads + conv_nhwc)", lambda: conv.conv_fwd(xr, xi, wr, wi, br, bi, geom, oshape)),
        ("r01 dgrad (2 pads + conv_nhwc)", lambda: conv.conv_dgrad(gr_n, gi_n, wr, wi, geom, xr.shape)),
        ("r01 wgrad (4 pads + kernel)", lambda: conv.conv_wgrad(gr_n, gi_n, xr, xi, geom, wr.shape)),
        ("NCHW -> channels-last copy x2", lambda: (conv.to_channels_last(xr), conv.to_channels_last(xi)))]
if os.environ.get("ONLY"):
    rows = [r for r in rows if r[0].startswith(tuple(os.environ["ONLY"].split(",")))]
print(f"# B={B} C={C} Co={Co} {H}x{W} 3x3 pad {PAD}: {flop / 1e12:.3f} TFLOP per launch; median ms (min) [TF/s, frac of 2.5 PF/s]")
for name, fn in rows:
    med, mn = timeit(fn)
    fl = flop * (RF if "REAL" in name else 1.0)
    print(f"{name:34s} {med:8.4f} ({mn:.4f})  [{fl / med / 1e9:7.1f}  {fl / med / 1e9 / 2500:.3f}]")
