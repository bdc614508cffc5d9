"""Stress of the folded weight-gradient launch (csrc/conv_cl_wgrad.hip FOLD) at configs[2]'s full size: the kernel counts
its own memory operations (s_waitcnt vmcnt(5) per stage), so a miscount would show as a value that depends on timing.
Repeated launches must be BIT-identical to each other -- alone, under both launch policies, and while another stream
saturates the HBM -- and dX must stay within one bf16 step of the separate apply pass."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cplxmodule_amd import _lib, bn as bnmod  # noqa: E402
from cplxmodule_amd._lib import call, ptr, stream_ptr  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
B, Ci, Co, H, W, pad = int(os.environ.get("BATCH", "256")), 64, 64, 256, 256, 0
Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
P = B * Ho * Wo
cl = lambda b, c, h, w: torch.randn(b, c, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)  # noqa: E731
xr, xi = cl(B, Ci, H, W), cl(B, Ci, H, W)
zr, zi, gr, gi = (cl(B, Co, Ho, Wo) for _ in range(4))
w = torch.tensor([[1.2, 0.1], [0.1, 0.8]], device=dev).reshape(2, 2, 1).repeat(1, 1, Co).contiguous()
saved = torch.empty(8, Co, device=dev)
ws = bnmod._ws(torch.device(dev, 0), Co)
yr, yi = torch.empty_like(zr), torch.empty_like(zi)
rm, rv, b = torch.zeros(2, Co, device=dev), torch.ones(2, 2, Co, device=dev), torch.zeros(2, Co, device=dev)
call("cplxamd_bn_fwd_ex", ptr(zr), ptr(zi), ptr(yr), ptr(yi), P, Co, 1, ptr(w), ptr(b), ptr(rm), ptr(rv), ptr(saved), 1,
     _lib.BF16, 0.1, 1e-5, None, ptr(ws), ws.numel(), stream_ptr())
del yr, yi
dw1, db1 = torch.empty(2, 2, Co, device=dev), torch.empty(2, Co, device=dev)
dxr, dxi, s1 = torch.empty_like(zr), torch.empty_like(zi), torch.empty(2, Co, device=dev)
call("cplxamd_bn_bwd_sums", ptr(gr), ptr(gi), ptr(zr), ptr(zi), ptr(dxr), ptr(dxi), P, Co, 1, ptr(w), ptr(saved), ptr(dw1),
     ptr(db1), 1, _lib.BF16, ptr(s1), ptr(ws), ws.numel(), stream_ptr())
coef, s2 = torch.empty(Co, 12, device=dev), torch.empty(2, Co, device=dev)
call("cplxamd_bn_bwd_coef", ptr(gr), ptr(gi), ptr(zr), ptr(zi), P, Co, 1, ptr(w), ptr(saved), ptr(dw1), ptr(db1), 1,
     _lib.BF16, ptr(coef), ptr(s2), ptr(ws), ws.numel(), stream_ptr())
wws = torch.empty(int(_lib.load().cplxamd_conv2d_cl_wgrad_ws_bytes(B, H, W, Ci, Co)), dtype=torch.uint8, device=dev)
hog_src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
hog_dst = torch.empty_like(hog_src)
side = torch.cuda.Stream()


def launch(flags):
    dyr, dyi = torch.empty_like(zr), torch.empty_like(zi)
    dwr, dwi = torch.empty(Co, Ci, 3, 3, device=dev), torch.empty(Co, Ci, 3, 3, device=dev)
    call("cplxamd_conv2d_cl_wgrad_bn_fl", ptr(gr), ptr(gi), ptr(zr), ptr(zi), ptr(coef), ptr(xr), ptr(xi), ptr(dyr), ptr(dyi),
         ptr(dwr), ptr(dwi), B, H, W, Ci, Co, 3, 3, 1, 1, pad, pad, ptr(wws), wws.numel(), flags, stream_ptr())
    return dyr, dyi, dwr, dwi


bad = 0
for flags, name in ((0, "default"), (_lib.LAUNCH_SHARED if hasattr(_lib, "LAUNCH_SHARED") else 1, "shared")):
    ref = launch(flags)
    torch.cuda.synchronize()
    d = (dxr.float() - ref[0].float()).abs()
    ulp = 2.0 ** -7 * dxr.float().abs().clamp_min(1e-30)
    over = d > ulp * 1.01 + 1e-30
    print(f"{name}: dX vs the separate apply: entries that differ {float((d > 0).float().mean()):.4%}, beyond one bf16 step of "
          f"their own value {float(over.float().mean()):.2e} (largest |dX| among those {float(dxr.float().abs()[over].max()) if bool(over.any()) else 0.0:.2e}), "
          f"max |diff| / max |dX| {float(d.max()) / float(dxr.float().abs().max()):.2e}", flush=True)
    del over
    del d, ulp
    for rep in range(int(os.environ.get("REPS", "12"))):
        if rep % 2:                                   # every other launch shares the HBM with a copy stream
            with torch.cuda.stream(side):
                for _ in range(6):
                    hog_dst.copy_(hog_src, non_blocking=True)
        out = launch(flags)
        torch.cuda.synchronize()
        same = [torch.equal(a, c) for a, c in zip(ref, out)]
        if not all(same):
            bad += 1
            print(f"  {name} launch {rep}: NOT bit-identical {same}", flush=True)
        del out
    print(f"{name}: {int(os.environ.get('REPS', '12'))} repeated launches compared bit for bit", flush=True)
print("FOLD STRESS", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
