"""Which launch of the bf16 LRT step breaks hipGraph capture?  One case per process (a bad graph segfaults)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cplxmodule_amd import Cplx, ops
from cplxmodule_amd.nn import relevance as rel
from cplxmodule_amd.nn.relevance import noise

case = sys.argv[1]
dev = "cuda"
bf = torch.bfloat16
B, I, O = (int(v) for v in (sys.argv[2:5] if len(sys.argv) > 4 else (8192, 4096, 4096)))
torch.manual_seed(0)
xr, xi = (torch.randn(B, I, device=dev).to(bf) for _ in range(2))
wr, wi = (torch.randn(O, I, device=dev).mul(0.01).to(bf) for _ in range(2))
gr, gi = (torch.randn(B, O, device=dev).to(bf) for _ in range(2))
w32r, w32i, ls2 = wr.float(), wi.float(), torch.full((O, I), -5.0, device=dev)
noise.set_mode("philox-device")

def cap(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    g.replay(); torch.cuda.synchronize()
    return out

if case == "cgemm_nn":
    cap(lambda: ops.cgemm(xr, xi, (I, 1), wr, wi, (I, 1), B, O, I, out_dtype=bf))
elif case == "cgemm_nt":
    cap(lambda: ops._cplx_linear_dx(gr, gi, wr, wi, bf))
elif case == "cgemm_tt":
    cap(lambda: ops._cplx_linear_dw(gr, gi, xr, xi))
elif case == "rgemm_nn":
    cap(lambda: ops.rgemm(xr, (I, 1), wr, (I, 1), B, O, I, out_dtype=bf))
elif case == "prep":
    cap(lambda: ops.prep_kl("cplx_vd", w32r, w32i, ls2, True))
elif case == "reparam":
    sd, of = noise.next(torch.device(dev))
    cap(lambda: ops.reparam_fwd(gr.clone(), gi.clone(), xr[:, :O].contiguous() if I >= O else gr, None, *noise.next(torch.device(dev))))
elif case in ("fwd", "fwdbwd", "fwdbwd_kl", "fwdbwd_kl_ret"):
    klw = torch.tensor(1e-3, device=dev)
    layer = rel.CplxLinearVD(I, O).to(dev)
    x = Cplx(xr.clone().requires_grad_(True), xi.clone().requires_grad_(True))
    def step():
        layer.zero_grad(set_to_none=True); x.real.grad = x.imag.grad = None
        y = layer(x)
        if case == "fwd":
            return y.real
        if case == "fwdbwd":
            torch.autograd.backward((y.real, y.imag), (y.real.detach() * 2, y.imag.detach() * 2))
        else:
            kl = sum(rel.penalties(layer))
            torch.autograd.backward((y.real, y.imag, kl), (y.real.detach() * 2, y.imag.detach() * 2, klw))
            if case == "fwdbwd_kl_ret":
                return kl
        return y.real
    cap(step)
print(case, B, I, O, "OK", flush=True)
