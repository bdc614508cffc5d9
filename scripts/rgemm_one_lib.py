"""A few launches of the bench step's REAL (variance) forward GEMM from ONE build of the library (ctypes): for rocprofv3
--pmc passes over A/B builds.  usage: rgemm_one_lib.py <lib.so> [iters]"""
import ctypes
import os
import sys
from ctypes import c_int, c_void_p

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cplxmodule_amd import _lib as L  # noqa: E402

lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
lib.cplxamd_rgemm.argtypes = L.SIGNATURES["cplxamd_rgemm"]
lib.cplxamd_rgemm.restype = c_int
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B, I, O = 8192, 4096, 4096
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
a = torch.randn(B, I, device=dev).square().to(bf)
s = torch.empty(O, I, device=dev).uniform_(-12, 4).exp().to(bf)
y = torch.empty(B, O, device=dev, dtype=bf)
p = lambda t: c_void_p(t.data_ptr())  # noqa: E731
st = c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(n):
    rc = lib.cplxamd_rgemm(p(a), I, 1, p(s), I, 1, None, None, p(y), O, B, O, I, L.BF16, L.BF16, 0, None, 0, st)
    assert rc == 0
torch.cuda.synchronize()
