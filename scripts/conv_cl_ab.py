"""In-process interleaved A/B of builds of libcplxamd.so on the channels-last conv kernel (cfg3 layer).
    python scripts/conv_cl_ab.py base=cplxmodule_amd/libcplxamd.so nt=cplxmodule_amd/libcplxamd_nt.so ...
env: B (64), ROUNDS (7), PER (8)"""
import ctypes
import os
import statistics
import sys
from ctypes import c_int, c_void_p

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cplxmodule_amd import _lib as L  # noqa: E402

B = int(os.environ.get("B", "64"))
ROUNDS, PER = int(os.environ.get("ROUNDS", "7")), int(os.environ.get("PER", "8"))
C = Co = 64
H = W = 256


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name in ("cplxamd_conv2d_cl", "cplxamd_conv2d_cl_pack"):
        fn = getattr(lib, name)
        fn.argtypes = L.SIGNATURES[name]
        fn.restype = c_int
    lib.cplxamd_conv2d_cl_pack_bytes.restype = ctypes.c_int64
    lib.cplxamd_conv2d_cl_ws_bytes.restype = ctypes.c_int64
    return lib


def main():
    libs = [(a.split("=")[0], load(a.split("=")[1])) for a in sys.argv[1:]]
    dev, bf = "cuda", torch.bfloat16
    torch.manual_seed(0)
    xr, xi = (torch.randn(B, H, W, C, device=dev).to(bf) for _ in range(2))
    wr, wi = (torch.randn(Co, C, 3, 3, device=dev).mul(0.05).to(bf) for _ in range(2))
    br, bi = torch.randn(Co, device=dev), torch.randn(Co, device=dev)
    yr, yi = torch.empty(B, H, W, Co, device=dev, dtype=bf), torch.empty(B, H, W, Co, device=dev, dtype=bf)
    p = lambda t: c_void_p(t.data_ptr())  # noqa: E731
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    l0 = libs[0][1]
    wp = torch.empty(int(l0.cplxamd_conv2d_cl_pack_bytes(Co, C, 3, 3)), dtype=torch.uint8, device=dev)
    assert l0.cplxamd_conv2d_cl_pack(p(wr), p(wi), p(wp), Co, C, 3, 3, 0, st) == 0
    ws = torch.empty(int(l0.cplxamd_conv2d_cl_ws_bytes(Co)), dtype=torch.uint8, device=dev)

    def run(lib):
        rc = lib.cplxamd_conv2d_cl(p(xr), p(xi), p(wp), p(br), p(bi), p(yr), p(yi), B, H, W, C, Co, 3, 3, 1, 1, 1, 1, 0,
                                   p(ws), ws.numel(), st)
        assert rc == 0, rc

    times = {n: [] for n, _ in libs}
    outs = {}
    for n, lib in libs:
        for _ in range(3):
            run(lib)
        torch.cuda.synchronize()
        outs[n] = (yr.clone(), yi.clone())
    for r in range(ROUNDS):
        for n, lib in (libs if r % 2 == 0 else libs[::-1]):
            run(lib)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(PER):
                run(lib)
            e1.record()
            torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) / PER)
    flop = 8.0 * B * H * W * C * Co * 9
    print(f"# B={B}: {ROUNDS} interleaved rounds x {PER} launches; median ms (min) [frac of 2.5 PF/s]   same result as first build")
    base = outs[libs[0][0]]
    for n, _ in libs:
        med, mn = statistics.median(times[n]), min(times[n])
        same = torch.equal(outs[n][0], base[0]) and torch.equal(outs[n][1], base[1])
        print(f"{n:14s} {med:.4f} ({mn:.4f}) [{flop / med / 1e9 / 2500:.3f}]   {same}")


if __name__ == "__main__":
    main()
