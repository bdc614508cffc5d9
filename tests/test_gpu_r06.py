"""Round 6: bench.py launched the way the driver launches it for N > 1 (plain `python bench.py --gpus N`), `--check`."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "TORCHELASTIC_RUN_ID", "GROUP_RANK"):
        env.pop(k, None)
    return env


def test_bench_py_plain_command_launches_itself_for_n_gt_1():
    """VERDICT r05 "what's weak" 7: `python bench.py --gpus 2 ...` with NO launcher around it (WORLD_SIZE unset) re-executes
    itself under torch.distributed.run (two gloo ranks sharing this GPU here) and prints exactly one JSON line, n_gpus 2."""
    from test_gpu_r03 import _bench_line
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device",
                        "--steps", "2", "--warmup", "1", "--batch", "512", "--no-cpu-baseline"],
                       env=_clean_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = _bench_line(r.stdout)
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 1024 and line["config"]["parallelism"] == "dp2"
    assert line["steps"] == 2 and np.isfinite(line["value"]) and line["value"] > 0


def test_bench_py_check_flag_small_batch():
    """`bench.py --check` (untimed value check of the workload as a graph replay) reports per-quantity errors in the line."""
    from test_gpu_r03 import _bench_line
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "1024",
                        "--no-cpu-baseline", "--check"], env=_clean_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = _bench_line(r.stdout)
    assert line["check"]["passed"] is True, line["check"]
    assert any(k.startswith("dW.real (+ KL accumulate)") for k in line["check"]["max_err_over_max_ref"])


# ---- second derivatives -- computed by the same kernels for the linear family / products / matmul,
# ---- a loud error (once_differentiable) everywhere else; never silently wrong numbers ----------------------------------------
def _ref_linear64(xr, xi, wr, wi, br, bi):
    """cplx.linear spelled with torch ops in float64 on the CPU (cplxmodule/cplx.py:634-648)."""
    import torch.nn.functional as F
    return F.linear(xr, wr, br) - F.linear(xi, wi), F.linear(xr, wi, bi) + F.linear(xi, wr)


@pytest.mark.parametrize("mode", ("exact", "x3"))
def test_cplx_linear_double_backward(mode):
    """Gradient penalty through CplxLinear: grad(create_graph=True) then a second backward, against float64 autograd of
    the reference's formula -- also with a NON-CONTIGUOUS input (the saved tensors must be the graph-connected ones)."""
    import torch
    from cplxmodule_amd import cplx, fp32_mode
    torch.manual_seed(3)
    B, I, O = 64, 96, 64
    base = [torch.randn(I, B), torch.randn(I, B), torch.randn(O, I) * 0.2, torch.randn(O, I) * 0.2, torch.randn(O), torch.randn(O)]

    def run(dev, dt):
        leaves = [t.to(dev, dt).requires_grad_(True) for t in base]
        xr, xi = leaves[0].t(), leaves[1].t()                      # transposed views: not contiguous
        if dev == "cpu":
            yr, yi = _ref_linear64(xr, xi, *leaves[2:])
        else:
            with fp32_mode(mode):
                y = cplx.linear(cplx.Cplx(xr, xi), cplx.Cplx(leaves[2], leaves[3]), cplx.Cplx(leaves[4], leaves[5]))
            yr, yi = y.real, y.imag
        loss = (yr ** 2).sum() + (yr * yi).sum()
        g = torch.autograd.grad(loss, leaves[:4], create_graph=True)
        pen = sum((t ** 2).sum() for t in g)
        second = torch.autograd.grad(pen, leaves)
        return [t.detach().double().cpu().numpy() for t in (*g, *second)]

    got, ref = run("cuda", torch.float32), run("cpu", torch.float64)
    for a, b in zip(got, ref):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-5 * float(np.abs(b).max()))


def test_real_linear_and_products_double_backward():
    import torch
    from cplxmodule_amd import Cplx, nn, ops
    torch.manual_seed(4)
    lin = nn.CplxLinear(32, 48).cuda()                      # masked / real variants share the Function pattern
    w = torch.randn(40, 32, device="cuda", requires_grad=True)
    x = torch.randn(16, 32, device="cuda", requires_grad=True)
    y = ops.RealLinearFn.apply(x, w, None, None)
    (gx,) = torch.autograd.grad((y ** 3).sum(), x, create_graph=True)
    (gw,) = torch.autograd.grad((gx ** 2).sum(), w)
    x64, w64 = x.detach().double().cpu().requires_grad_(True), w.detach().double().cpu().requires_grad_(True)
    (gx64,) = torch.autograd.grad(((x64 @ w64.t()) ** 3).sum(), x64, create_graph=True)
    (gw64,) = torch.autograd.grad((gx64 ** 2).sum(), w64)
    np.testing.assert_allclose(gw.cpu().double().numpy(), gw64.numpy(), rtol=0, atol=3e-5 * float(gw64.abs().max()))
    # Cplx * Cplx with a transposed (non-contiguous) operand: ADVICE r05 -- second-order terms through b must survive
    a = [torch.randn(24, 40, device="cuda", requires_grad=True) for _ in range(2)]
    bT = [torch.randn(40, 24, device="cuda", requires_grad=True) for _ in range(2)]
    z = Cplx(a[0], a[1]) * Cplx(bT[0].t(), bT[1].t())
    ga = torch.autograd.grad((z.real ** 2).sum() + (z.imag ** 3).sum(), a, create_graph=True)
    gb = torch.autograd.grad(sum((t ** 2).sum() for t in ga), bT)
    a64 = [t.detach().double().cpu().requires_grad_(True) for t in a]
    b64 = [t.detach().double().cpu().requires_grad_(True) for t in bT]
    zr = a64[0] * b64[0].t() - a64[1] * b64[1].t()
    zi = a64[0] * b64[1].t() + a64[1] * b64[0].t()
    ga64 = torch.autograd.grad((zr ** 2).sum() + (zi ** 3).sum(), a64, create_graph=True)
    gb64 = torch.autograd.grad(sum((t ** 2).sum() for t in ga64), b64)
    for got, ref in zip(gb, gb64):
        assert float(ref.abs().max()) > 0
        np.testing.assert_allclose(got.cpu().double().numpy(), ref.numpy(), rtol=0, atol=3e-5 * float(ref.abs().max()))
    # Cplx @ Cplx (2-d and batched)
    u = [torch.randn(3, 8, 12, device="cuda", requires_grad=True) for _ in range(2)]
    v = [torch.randn(3, 12, 5, device="cuda", requires_grad=True) for _ in range(2)]
    m = Cplx(u[0], u[1]) @ Cplx(v[0], v[1])
    gu = torch.autograd.grad((m.real ** 2).sum() + (m.real * m.imag).sum(), u, create_graph=True)
    gv = torch.autograd.grad(sum((t ** 2).sum() for t in gu), v)
    u64 = [t.detach().double().cpu().requires_grad_(True) for t in u]
    v64 = [t.detach().double().cpu().requires_grad_(True) for t in v]
    mr, mi = u64[0] @ v64[0] - u64[1] @ v64[1], u64[0] @ v64[1] + u64[1] @ v64[0]
    gu64 = torch.autograd.grad((mr ** 2).sum() + (mr * mi).sum(), u64, create_graph=True)
    gv64 = torch.autograd.grad(sum((t ** 2).sum() for t in gu64), v64)
    for got, ref in zip(gv, gv64):
        np.testing.assert_allclose(got.cpu().double().numpy(), ref.numpy(), rtol=0, atol=3e-5 * float(ref.abs().max()))
    del lin


@pytest.mark.parametrize("family", ("lrt_cplx", "lrt_real", "conv", "batchnorm", "penalty", "abs", "modrelu"))
def test_double_backward_raises_where_not_supported(family):
    """Every raw-kernel backward is once_differentiable: differentiating it again is an ERROR, not a wrong number."""
    import torch
    from cplxmodule_amd import Cplx, nn
    from cplxmodule_amd.nn import relevance as rel
    torch.manual_seed(5)
    dev = "cuda"
    x = Cplx(torch.randn(8, 16, device=dev, requires_grad=True), torch.randn(8, 16, device=dev, requires_grad=True))
    if family == "lrt_cplx":
        layer = rel.CplxLinearVD(16, 24).to(dev)
        y = layer(x)
        out, wrt = y.real.sum() + y.imag.sum(), [x.real, layer.weight.real]
    elif family == "lrt_real":
        layer = rel.LinearVD(16, 24).to(dev)
        out, wrt = (layer(x.real) ** 2).sum(), [x.real, layer.weight]
    elif family == "conv":
        layer = nn.CplxConv2d(4, 8, 3).to(dev)
        xi = Cplx(torch.randn(2, 4, 9, 9, device=dev, requires_grad=True), torch.randn(2, 4, 9, 9, device=dev, requires_grad=True))
        y = layer(xi)
        out, wrt = (y.real ** 2).sum() + y.imag.sum(), [xi.real, layer.weight.real]
    elif family == "batchnorm":
        layer = nn.CplxBatchNorm1d(16).to(dev)
        y = layer(x)
        out, wrt = (y.real ** 3).sum() + y.imag.sum(), [x.real]
    elif family == "penalty":
        layer = rel.CplxLinearVD(16, 24).to(dev)
        out, wrt = (layer.penalty ** 2).sum(), [layer.log_sigma2, layer.weight.real]
    elif family == "abs":
        out, wrt = (abs(x) ** 3).sum(), [x.real]
    else:
        from cplxmodule_amd import cplx
        y = cplx.modrelu(x, -0.1)
        out, wrt = (y.real ** 2).sum(), [x.real]
    g = torch.autograd.grad(out, wrt, create_graph=True)
    assert all(t.requires_grad for t in g)          # even with a constant upstream gradient (torch's own decorator: not)
    pen = sum((t ** 2).sum() for t in g)
    # a gradient penalty added to a loss: backward() reaches the error node and raises
    with pytest.raises(RuntimeError, match="once_differentiable"):
        (pen + out).backward()
