"""cfg5 end to end on the GPU: dense -> ARD -> masked phases of examples/train_sparsify.py, and the
equivalence masked-layer == dense layer with pre-multiplied weights."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_masked_layers_equal_dense_with_masked_weights():
    from gpu_util import N
    from cplxmodule_amd import Cplx, nn
    from cplxmodule_amd.nn import masked
    torch.manual_seed(0)
    lin = masked.CplxLinearMasked(24, 16).to("cuda")
    lin.mask = (torch.rand(16, 24) > 0.4).float()
    ref = nn.CplxLinear(24, 16).to("cuda")
    with torch.no_grad():
        ref.weight.real.copy_(lin.weight.real * lin.mask)
        ref.weight.imag.copy_(lin.weight.imag * lin.mask)
        ref.bias.real.copy_(lin.bias.real); ref.bias.imag.copy_(lin.bias.imag)
    x = Cplx(torch.randn(9, 24, device="cuda"), torch.randn(9, 24, device="cuda"))
    a, b = lin(x), ref(x)
    np.testing.assert_allclose(N(a.real), N(b.real), rtol=1e-6, atol=1e-6)
    (a.real.sum() + a.imag.sum()).backward()
    assert float((lin.weight.real.grad * (1 - lin.mask)).abs().max()) == 0.0   # masked grads vanish
    conv = masked.CplxConv2dMasked(4, 6, 3, padding=1).to("cuda")
    conv.mask = (torch.rand(6, 4, 3, 3) > 0.5).float()
    y = conv(Cplx(torch.randn(2, 4, 8, 8, device="cuda"), torch.randn(2, 4, 8, 8, device="cuda")))
    assert y.shape == (2, 6, 8, 8)
    rl = masked.LinearMasked(10, 7).to("cuda")
    rl.mask = torch.ones(7, 10)
    xr = torch.randn(5, 10, device="cuda")
    np.testing.assert_allclose(N(rl(xr)), N(torch.nn.functional.linear(xr, rl.weight, rl.bias)),
                               rtol=1e-5, atol=1e-5)


def test_train_sparsify_pipeline():
    import importlib.util, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples",
                        "train_sparsify.py")
    spec = importlib.util.spec_from_file_location("train_sparsify", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.main(["--steps", "25", "--batch", "128", "--width", "4"])
    assert out["dense"][-1][0] < out["dense"][0][0]          # the dense phase learns
    assert out["ard"][-1][1] < out["ard"][0][1]              # KL goes down under the penalty
    assert 0.0 <= out["sparsity"] <= 1.0
    assert set(out["masks"]["head.mask"].unique().tolist()) <= {0.0, 1.0}
    assert np.isfinite(out["masked"][-1][0])


def test_empty_batches_everywhere():
    """Empty inputs (the reference's torch ops accept them): forward shapes, zero parameter
    gradients, no kernel is asked to touch a NULL pointer."""
    import torch
    from gpu_util import DEV
    from cplxmodule_amd import Cplx, cplx, nn
    from cplxmodule_amd.nn import relevance as rel
    z = lambda *s, dt=torch.float32: Cplx(torch.randn(*s, device=DEV).to(dt), torch.randn(*s, device=DEV).to(dt))  # noqa: E731
    for dt in (torch.float32, torch.bfloat16):
        lin = nn.CplxLinear(32, 16).to(DEV)
        assert lin(z(0, 32, dt=dt)).real.shape == (0, 16)
        vd = rel.CplxLinearVD(32, 16).to(DEV)
        x = z(0, 32, dt=dt)
        x.real.requires_grad_(True)
        y = vd(x)
        (y.real.float().sum() + y.imag.float().sum() + 0 * sum(rel.penalties(vd))).backward()
        assert x.real.grad.shape == (0, 32)
        assert float(vd.weight.real.grad.abs().max()) == 0.0 and float(vd.bias.real.grad.abs().max()) == 0.0
        conv = nn.CplxConv2d(32, 64, 3, padding=1).to(DEV)
        xc = z(0, 32, 8, 8, dt=dt)
        xc.real.requires_grad_(True)
        out = conv(xc)
        assert out.real.shape == (0, 64, 8, 8)
        (out.real.float().sum() + out.imag.float().sum()).backward()
        assert float(conv.weight.real.grad.abs().max()) == 0.0
    assert cplx.modrelu(z(0, 5), 0.5).real.shape == (0, 5)
    assert cplx.from_interleaved_real(torch.randn(0, 8, device=DEV)).real.shape == (0, 4)
    assert cplx.max_pool2d(z(0, 3, 8, 8), 2).real.shape == (0, 3, 4, 4)
    assert cplx.dropout(z(0, 4), 0.5).imag.shape == (0, 4)


def test_data_parallel_paths_on_gpu():
    """tests/dp_gpu_check.py in subprocesses: two gloo ranks sharing the GPU (values: hook-averaged ==
    hand-averaged gradients) and one RCCL rank with the collectives forced on (every RCCL call of the
    multi-GPU bench path is issued for real)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(here, "dp_gpu_check.py"), "--rccl1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "dp_gpu_check OK: backend=nccl" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541",
                        os.path.join(here, "dp_gpu_check.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "dp_gpu_check OK: backend=gloo world=2" in r.stdout, r.stdout + r.stderr
