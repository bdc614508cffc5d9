"""overlap_probe.py with the side-stream kernel replaced by a 1-rank RCCL all_reduce (oneRankReduce): is it the process
group that serialises behind the compute stream?  argv[1]: 'side' (all_reduce issued under a side stream that waits for
the event) | 'direct' (ncclStream ordering left to the process group: issued on the compute stream at the event)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
from cplxmodule_amd import ops, _lib
how = sys.argv[1]
dev = torch.device("cuda", 0); torch.cuda.set_device(dev); bf = torch.bfloat16
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29579")
if len(sys.argv) > 2:
    os.environ["TORCH_NCCL_HIGH_PRIORITY"] = sys.argv[2]
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
B, I, O = 8192, 4096, 4096
xr, xi = (torch.randn(B, I, device=dev).to(bf) for _ in range(2))
wr, wi = (torch.randn(O, I, device=dev).mul(0.01).to(bf) for _ in range(2))
buf = torch.zeros(16 << 20, device=dev)
_lib.load().cplxamd_gemm_set_persistent(0)
side = torch.cuda.Stream(priority=-1 if (len(sys.argv) > 3 and sys.argv[3] == 'hi') else 0)
gemm = lambda: ops.cgemm(xr, xi, (I, 1), wr, wi, (I, 1), B, O, I, out_dtype=bf)
for it in range(3):
    gemm(); gemm()
    if how == "direct":
        w = dist.all_reduce(buf, op=dist.ReduceOp.AVG, async_op=True)
        gemm(); gemm(); gemm()
    else:
        ev = torch.cuda.Event(); ev.record()
        gemm(); gemm(); gemm()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            w = dist.all_reduce(buf, op=dist.ReduceOp.AVG, async_op=True)
        torch.cuda.current_stream().wait_stream(side)
    w.wait()
    torch.cuda.synchronize()
dist.destroy_process_group()
