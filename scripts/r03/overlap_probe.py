"""Does a kernel on a second stream, ordered behind an EVENT in the middle of the compute stream, start while later GEMMs of the
compute stream run?  (rocprofv3 --kernel-trace timeline; mode: persist|onetile, side stream priority: high|normal)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cplxmodule_amd import ops, _lib
mode, prio = sys.argv[1], sys.argv[2]
dev = "cuda"; bf = torch.bfloat16
B, I, O = 8192, 4096, 4096
xr, xi = (torch.randn(B, I, device=dev).to(bf) for _ in range(2))
wr, wi = (torch.randn(O, I, device=dev).mul(0.01).to(bf) for _ in range(2))
buf = torch.zeros(16 << 20, device=dev)          # 64 MiB, what a bucket is
_lib.load().cplxamd_gemm_set_persistent(1 if mode == "persist" else 0)
side = torch.cuda.Stream(priority=-1 if prio == "high" else 0)
def gemm():
    return ops.cgemm(xr, xi, (I, 1), wr, wi, (I, 1), B, O, I, out_dtype=bf)
for it in range(3):
    gemm(); gemm()
    ev = torch.cuda.Event(); ev.record()
    gemm(); gemm(); gemm()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        buf.mul_(1.0001)                         # X: 128 MiB of traffic, ~30 us alone
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
