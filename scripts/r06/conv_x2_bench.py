"""BASELINE configs[2] in float32 (CplxConv2d(64, 64, 3) @ 256 x 256 + CplxBatchNorm2d, batch 256, fwd + bwd): the half
split products (fp32 mode 'x2' / 'auto') against the float32-MFMA kernels ('exact').  MODES=x2,exact python ..."""
import importlib.util
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
from cplxmodule_amd import fp32_mode  # noqa: E402

for mode in os.environ.get("MODES", "x2,exact").split(","):
    with fp32_mode(mode):
        out = bench.conv_point(torch.device("cuda", 0), batch=int(os.environ.get("BATCH", "256")), dtype=torch.float32)
    print(mode, json.dumps(out), flush=True)
    torch.cuda.empty_cache()
