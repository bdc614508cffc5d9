"""A few launches of the bench step's complex forward GEMM from ONE build of the library (ctypes, no
package import): for rocprofv3 --pmc passes over A/B builds.  usage: gemm_one_lib.py <lib.so> [iters]"""
import ctypes
import os
import sys
from ctypes import c_int, c_void_p

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cplxmodule_amd import _lib as L  # noqa: E402

lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
lib.cplxamd_cgemm.argtypes = L.SIGNATURES["cplxamd_cgemm"]
lib.cplxamd_cgemm.restype = c_int
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B, I, O = 8192, 4096, 4096
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
bound = (1.0 / (2 * I)) ** 0.5
xr, xi = (torch.randn(B, I, device=dev).to(bf) for _ in range(2))
wr, wi = (torch.empty(O, I, device=dev).uniform_(-bound, bound).to(bf) for _ in range(2))
br, bi = (torch.zeros(O, device=dev) for _ in range(2))
yr, yi = (torch.empty(B, O, device=dev, dtype=bf) for _ in range(2))
p = lambda t: c_void_p(t.data_ptr())  # noqa: E731
st = c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(n):
    rc = lib.cplxamd_cgemm(p(xr), p(xi), I, 1, p(wr), p(wi), I, 1, p(br), p(bi), p(yr), p(yi), O, B, O, I, 0,
                           L.BF16, L.BF16, 0, 0, None, 0, st)
    assert rc == 0
torch.cuda.synchronize()
