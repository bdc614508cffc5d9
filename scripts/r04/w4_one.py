"""N launches of ONE bench-step GEMM from one build and one kernel family, for rocprofv3 passes.
usage: w4_one.py <lib.so> <family 0|1> <shape name of w4_ab.make> [iters] [B I O]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
import w4_ab  # noqa: E402

lib = w4_ab.load(sys.argv[1])
fam, shape = int(sys.argv[2]), sys.argv[3]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 8
B, I, O = (int(v) for v in sys.argv[5:8]) if len(sys.argv) > 7 else (8192, 4096, 4096)
fn = w4_ab.make(lib, B, I, O)[shape][0]
lib.cplxamd_gemm_set_family(-1 if fam else 0)
for _ in range(n):
    fn()
torch.cuda.synchronize()
