"""N launches of ONE GEMM of scripts/r05/w4p_ab.py on one kernel form, for rocprofv3 passes.
usage: w4p_one.py <form: w8|w4|w4p> <shape: c_fwd0|c_dgrad|r_fwd|r_dgrad> [iters] [B I O]     (w4p complex: CPLXAMD_W4P_CPLX=1)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
import w4p_ab  # noqa: E402

form, shape = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
B, I, O = (int(v) for v in sys.argv[4:7]) if len(sys.argv) > 6 else (8192, 4096, 4096)
lib = w4p_ab.L.load()
fn = w4p_ab.make(lib, B, I, O)[0][shape][0]
fl = dict(w4p_ab.FAMS)[form]
for _ in range(n):
    fn(fl)
torch.cuda.synchronize()
