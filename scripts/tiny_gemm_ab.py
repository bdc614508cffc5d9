"""Launch-latency-bound GEMMs of cfg1 (fp32, 64 x 128 x 128) for several builds of the library, interleaved."""
import ctypes, os, statistics, sys
from ctypes import c_int, c_void_p
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cplxmodule_amd import _lib as L
libs = []
for a in sys.argv[1:]:
    n, path = a.split("=")
    lib = ctypes.CDLL(os.path.abspath(path))
    for f in ("cplxamd_cgemm", "cplxamd_rgemm"):
        getattr(lib, f).argtypes = L.SIGNATURES[f]; getattr(lib, f).restype = c_int
    libs.append((n, lib))
dev = "cuda"
B, I, O = 64, 128, 128
x = [torch.randn(B, I, device=dev) for _ in range(2)]
w = [torch.randn(O, I, device=dev) for _ in range(2)]
y = [torch.empty(B, O, device=dev) for _ in range(2)]
p = lambda t: c_void_p(t.data_ptr())
st = c_void_p(torch.cuda.current_stream().cuda_stream)
def run(lib, cplx):
    if cplx:
        lib.cplxamd_cgemm(p(x[0]), p(x[1]), I, 1, p(w[0]), p(w[1]), I, 1, None, None, p(y[0]), p(y[1]), O, B, O, I, 0, L.F32, L.F32, 0, 0, None, 0, st)
    else:
        lib.cplxamd_rgemm(p(x[0]), I, 1, p(w[0]), I, 1, None, None, p(y[0]), O, B, O, I, L.F32, L.F32, 0, None, 0, st)
res = {}
for r in range(9):
    for n, lib in (libs if r % 2 == 0 else libs[::-1]):
        for cplx in (0, 1):
            for _ in range(5): run(lib, cplx)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200): run(lib, cplx)
            e1.record(); torch.cuda.synchronize()
            res.setdefault((n, cplx), []).append(e0.elapsed_time(e1) / 200 * 1e3)
for (n, c), v in sorted(res.items()):
    print(f"{n:10s} {'complex' if c else 'real   '} {statistics.median(v):7.2f} us per launch (min {min(v):.2f})")
