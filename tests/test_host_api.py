"""Host-side contract (no GPU): state-dict keys / shapes / parameter order, default init,
walker names, error behaviour, complex-parameter promotion, the Cplx container -- compared with
what the reference itself reports (tests/golden/api.npz)."""
import copy
import os

import numpy as np
import pytest
import torch

import cplxmodule_amd
from cplxmodule_amd import Cplx, cplx, nn
from cplxmodule_amd.nn import relevance as rel
from cplxmodule_amd._lib import CplxAmdError

LAYERS = {
    "CplxLinear": lambda: nn.CplxLinear(100, 400),
    "CplxLinearVD": lambda: rel.CplxLinearVD(12, 7),
    "CplxLinearARD": lambda: rel.CplxLinearARD(12, 7, bias=False),
    "LinearVD": lambda: rel.LinearVD(12, 7),
    "LinearARD": lambda: rel.LinearARD(12, 7),
    "CplxConv2d": lambda: nn.CplxConv2d(6, 4, (3, 2), groups=2),
    "CplxConv2dVD": lambda: rel.CplxConv2dVD(6, 4, 3),
    "Conv2dARD": lambda: rel.Conv2dARD(6, 4, 3),
    "CplxBatchNorm2d": lambda: nn.CplxBatchNorm2d(5),
}


@pytest.mark.parametrize("name", list(LAYERS))
def test_state_dict_layout_matches_reference(golden, name):
    g = golden("api")
    layer = LAYERS[name]()
    sd = layer.state_dict()
    assert list(sd) == list(g[name + "__keys"])
    assert [",".join(map(str, v.shape)) for v in sd.values()] == list(g[name + "__shapes"])
    assert [n for n, _ in layer.named_parameters()] == list(g[name + "__params"])


def test_default_init_bounds(golden):
    g = golden("api")
    torch.manual_seed(0)
    lin = nn.CplxLinear(100, 400)
    wmax = max(float(lin.weight.real.abs().max()), float(lin.weight.imag.abs().max()))
    bmax = max(float(lin.bias.real.abs().max()), float(lin.bias.imag.abs().max()))
    # weight planes ~ U(+-sqrt(1/(2 I))), bias ~ U(+-1/sqrt(O)) (the reference's fan quirk)
    assert 0.95 * (1 / 200) ** 0.5 < wmax <= (1 / 200) ** 0.5 + 1e-6
    assert 0.9 * (1 / 400) ** 0.5 < bmax <= (1 / 400) ** 0.5 + 1e-6
    assert abs(wmax - float(g["CplxLinear__wmax"])) < 2e-3
    assert abs(bmax - float(g["CplxLinear__bmax"])) < 3e-3
    cv = nn.CplxConv2d(6, 4, (3, 2), groups=2)
    assert float(cv.weight.real.abs().max()) <= float(g["CplxConv2d__wmax"]) * 1.1
    assert float(cv.bias.real.abs().max()) <= (1 / (3 * 3 * 2)) ** 0.5 + 1e-6
    vd = rel.CplxLinearVD(12, 7)
    np.testing.assert_array_equal(vd.log_sigma2.detach().numpy(), g["CplxLinearVD__ls2"])
    bn = nn.CplxBatchNorm2d(5)
    np.testing.assert_array_equal(bn.weight.detach().numpy(), np.eye(2)[:, :, None].repeat(5, 2))
    np.testing.assert_array_equal(bn.running_var.numpy(), np.eye(2)[:, :, None].repeat(5, 2))
    assert int(bn.num_batches_tracked) == 0


def test_walker_names(golden):
    g = golden("api")
    model = torch.nn.Sequential(rel.CplxLinearVD(4, 5), torch.nn.Sequential(rel.CplxLinearARD(5, 3)))
    names = [n for n, m in model.named_modules() if isinstance(m, rel.BaseARD)]
    assert names == list(g["walk__penalty_names"])
    with pytest.raises(ValueError):
        list(rel.named_penalties(model, reduction="max"))
    assert rel.compute_ard_masks(None) == {}
    shared = rel.LinearVD(3, 3)
    twice = torch.nn.Sequential(shared, shared)
    assert len([n for n, m in twice.named_modules() if isinstance(m, rel.BaseARD)]) == 1


def test_kernels_refuse_cpu_tensors_loudly():
    layer = rel.CplxLinearVD(4, 5)
    x = Cplx(torch.randn(2, 4), torch.randn(2, 4))
    with pytest.raises(CplxAmdError, match="no CPU path"):
        layer(x)
    with pytest.raises(CplxAmdError):
        list(rel.penalties(layer))
    with pytest.raises(CplxAmdError):
        layer.relevance(threshold=1.0)
    with pytest.raises(CplxAmdError):
        nn.CplxBatchNorm1d(4)(x)
    with pytest.raises(CplxAmdError):
        rel.LinearARD(4, 5)(torch.randn(2, 4))


def test_constructor_errors():
    with pytest.raises(ValueError, match="in_channels must be divisible by groups"):
        nn.CplxConv2d(3, 4, 3, groups=2)
    with pytest.raises(ValueError, match="out_channels must be divisible by groups"):
        nn.CplxConv2d(4, 3, 3, groups=2)
    with pytest.raises(ValueError, match="Only `zeros` padding mode"):
        rel.CplxConv2dVD(2, 2, 3, padding_mode="circular")
    with pytest.raises(ValueError, match="Only `zeros` padding mode"):
        rel.Conv2dVD(2, 2, 3, padding_mode="circular")
    with pytest.raises(TypeError):
        Cplx([1, 2])
    with pytest.raises(TypeError):
        Cplx(torch.zeros(2), 1.0)
    with pytest.raises(ValueError):
        Cplx(torch.zeros(2), torch.zeros(3))
    with pytest.raises(ValueError, match="expected 4D input"):
        nn.CplxBatchNorm2d(3)._check_input_dim(Cplx(torch.zeros(2, 3)))
    with pytest.raises(ValueError, match="expected 2D or 3D input"):
        nn.CplxBatchNorm1d(3)._check_input_dim(Cplx(torch.zeros(2, 3, 4, 5)))


def test_cplx_parameter_loading_and_promotion():
    lin = nn.CplxLinear(3, 2)
    sd = lin.state_dict()
    other = nn.CplxLinear(3, 2)
    other.load_state_dict(sd)
    assert torch.equal(other.weight.real, lin.weight.real)
    clone = copy.deepcopy(lin)
    assert torch.equal(clone.bias.imag, lin.bias.imag) and clone.bias.imag is not lin.bias.imag
    # a real tensor under the bare name is promoted to complex (imag = 0)
    real_sd = {"weight": torch.randn(2, 3), "bias": torch.randn(2)}
    other.load_state_dict(real_sd)
    assert torch.equal(other.weight.real, real_sd["weight"])
    assert float(other.weight.imag.abs().max()) == 0.0
    # one part missing -> error; nothing -> reported as missing
    with pytest.raises(RuntimeError):
        other.load_state_dict({"weight.real": torch.zeros(2, 3), "bias.real": torch.zeros(2),
                               "bias.imag": torch.zeros(2)})
    res = other.load_state_dict({"bias.real": torch.zeros(2), "bias.imag": torch.zeros(2)}, strict=False)
    assert res.missing_keys == ["weight"]
    w = lin.weight
    assert isinstance(w, Cplx) and w.real is lin._parameters.get("weight", lin._modules["weight"]["real"])


def test_cplx_container_arithmetic_cpu():
    rs = np.random.RandomState(0)
    a = rs.randn(4, 3) + 1j * rs.randn(4, 3)
    b = rs.randn(4, 3) + 1j * rs.randn(4, 3)
    A, B = Cplx.from_numpy(a), Cplx.from_numpy(b)
    np.testing.assert_allclose((A * B).numpy(), a * b)
    np.testing.assert_allclose((A / B).numpy(), a / b)
    np.testing.assert_allclose((A + B - 2.0).numpy(), a + b - 2.0)
    np.testing.assert_allclose((-A).conj.numpy(), -a.conj())
    np.testing.assert_allclose(abs(A).numpy(), np.abs(a))
    np.testing.assert_allclose(A.angle.numpy(), np.angle(a))
    np.testing.assert_allclose((3.0 / A).numpy(), 3.0 / a)
    assert A[1:3].shape == (2, 3) and len(A) == 4 and A.t().shape == (3, 4)
    assert A.view(2, 6).shape == (2, 6) and A.reshape(12).dim() == 1
    z = cplx.to_interleaved_real(A)
    assert z.shape == (4, 6)
    back = cplx.from_interleaved_real(z)
    np.testing.assert_allclose(back.numpy(), a)
    cat = cplx.to_concatenated_real(A)
    np.testing.assert_allclose(cplx.from_concatenated_real(cat).numpy(), a)
    assert Cplx(A) is A and Cplx(1.0 + 2j).item() == 1.0 + 2j
    e = cplx.randn(1000, 50)
    assert abs(float((e.real ** 2 + e.imag ** 2).mean()) - 1.0) < 0.05
    relu = nn.CplxToCplx[torch.nn.ReLU]()
    out = relu(A)
    np.testing.assert_allclose(out.real.numpy(), np.maximum(a.real, 0))
    assert nn.CplxToCplx[torch.nn.ReLU] is nn.CplxToCplx[torch.nn.ReLU]
    tanh = nn.CplxToCplx[torch.tanh]()
    np.testing.assert_allclose(tanh(A).imag.numpy(), np.tanh(a.imag))


def test_noise_state():
    from cplxmodule_amd.nn.relevance import noise
    noise.manual_seed(7)
    assert noise.next() == (7, 1) and noise.next() == (7, 2)
    noise.set_mode("torch")
    noise.set_mode("philox")
    with pytest.raises(ValueError):
        noise.set_mode("numpy")
    assert cplxmodule_amd.__version__


def test_convert_sync_batchnorm_is_a_flag_outside_the_state_dict():
    """dp.convert_sync_batchnorm marks the complex batch-norm layers only; parameters, buffers and the state-dict keys
    (the reference's, nn/modules/batchnorm.py:302-320) are untouched, and without a process group the layer keeps the
    local-batch statistics path (bn._sync_group -> None)."""
    from cplxmodule_amd import bn, dp
    net = torch.nn.Sequential(nn.CplxConv2d(2, 4, 3), nn.CplxBatchNorm2d(4), nn.CplxBatchNorm1d(4))
    keys = list(net.state_dict())
    assert all(m.process_group is None for m in net if isinstance(m, nn.CplxBatchNorm2d))
    assert dp.convert_sync_batchnorm(net) is net
    assert net[1].process_group is True and net[2].process_group is True and not hasattr(net[0], "process_group")
    assert list(net.state_dict()) == keys
    assert bn._sync_group(True, True) is None and bn._sync_group(True, False) is None     # no process group here
    dp.convert_sync_batchnorm(net, None)
    assert net[1].process_group is None


@pytest.mark.skipif(not os.path.isdir("/root/reference/cplxmodule"), reason="the reference tree exists in the build container only")
def test_state_dicts_travel_between_this_package_and_the_reference():
    """No CPU execution path here -- but a model built by this package loads into the REFERENCE package
    for CPU evaluation and back: same keys, shapes and dtypes, strict=True both ways.  (Runs only where /root/reference
    exists; never on the GPU box.)"""
    import subprocess
    import sys
    import tempfile
    import torch
    from cplxmodule_amd import nn
    from cplxmodule_amd.nn import relevance as rel
    torch.manual_seed(0)
    ours = torch.nn.Sequential()
    ours.add_module("lin", nn.CplxLinear(6, 5))
    ours.add_module("vd", rel.CplxLinearVD(5, 4))
    ours.add_module("conv", nn.CplxConv2d(3, 2, 3))
    ours.add_module("bn", nn.CplxBatchNorm2d(2))
    ours.add_module("ard", rel.LinearARD(4, 3))
    with tempfile.TemporaryDirectory() as d:
        src, dst = os.path.join(d, "ours.pt"), os.path.join(d, "ref.pt")
        torch.save(ours.state_dict(), src)
        # the reference in its own interpreter (its package name must not meet ours in one sys.modules)
        code = f'''
import sys, types, torch
sys.path.insert(0, "/root/reference")
m = types.ModuleType("cplxmodule.__version__"); m.__version__ = open("/root/reference/VERSION").read().strip()
sys.modules["cplxmodule.__version__"] = m
from cplxmodule import nn, cplx
from cplxmodule.nn import relevance as rel
net = torch.nn.Sequential()
net.add_module("lin", nn.CplxLinear(6, 5)); net.add_module("vd", rel.CplxLinearVD(5, 4))
net.add_module("conv", nn.CplxConv2d(3, 2, 3)); net.add_module("bn", nn.CplxBatchNorm2d(2)); net.add_module("ard", rel.LinearARD(4, 3))
missing = net.load_state_dict(torch.load("{src}"), strict=True)
net.eval()
x = cplx.Cplx(torch.ones(2, 6), torch.zeros(2, 6))
y = net.vd(net.lin(x))                       # the reference evaluates OUR parameters on the CPU
assert y.real.shape == (2, 4) and torch.isfinite(y.real).all()
torch.save(net.state_dict(), "{dst}")
print("REF_OK", float(y.real.sum()))
'''
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "REF_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
        back = torch.load(dst)
        mine = ours.state_dict()
        assert list(back.keys()) == list(mine.keys())
        for k in mine:
            assert back[k].shape == mine[k].shape and back[k].dtype == mine[k].dtype and torch.equal(back[k], mine[k]), k
        ours.load_state_dict(back, strict=True)


def test_module_level_conv_operator_names():
    """cplx.convnd / convnd_naive / _quick / _3m / symmetric_circular_padding exist with the reference's signatures
    (cplxmodule/cplx.py:701-800); the padding helper runs on CPU planes (torch's pad), values against numpy's wrap."""
    import inspect
    import numpy as np
    import torch
    from cplxmodule_amd import cplx
    assert list(inspect.signature(cplx.convnd).parameters) == ["conv", "input", "weight", "bias", "stride", "padding",
                                                                "dilation", "groups", "padding_mode"]
    assert list(inspect.signature(cplx.convnd_quick).parameters) == ["conv", "input", "weight", "stride", "padding", "dilation"]
    for name in ("convnd_naive", "convnd_3m"):
        assert list(inspect.signature(getattr(cplx, name)).parameters)[-1] == "groups"
    z = cplx.Cplx(torch.arange(24.).reshape(1, 2, 3, 4), -torch.arange(24.).reshape(1, 2, 3, 4))
    p = cplx.symmetric_circular_padding(z, (3, 2))
    want = np.pad(z.real.numpy(), ((0, 0), (0, 0), (1, 1), (2, 1)), mode="wrap")     # F.pad order: last dim first
    np.testing.assert_array_equal(p.real.numpy(), want)
    np.testing.assert_array_equal(p.imag.numpy(), -want)
    import pytest
    with pytest.raises(ValueError):
        cplx.convnd(torch.nn.functional.conv1d, z, z)
