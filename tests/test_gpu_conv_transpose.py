"""Transposed convolution (the dgrad kernels run forward) against the reference's golden vectors."""
import numpy as np
import pytest
import torch

import oracle.cplx_oracle as orc
from oracle.gen_golden_cases import CONVT_CASES
from gpu_util import DEV, T, N

pytestmark = pytest.mark.gpu


def _tol(ref, r=2e-5):
    return dict(rtol=r, atol=r * float(np.abs(ref).max()))


@pytest.fixture(autouse=True, params=["size-gated", "rows-forced"])
def kernel_choice(request):
    from cplxmodule_amd import conv
    old = conv._ROWS_FORCE
    conv._ROWS_FORCE = request.param == "rows-forced"
    yield
    conv._ROWS_FORCE = old


@pytest.mark.parametrize("case", list(CONVT_CASES))
def test_conv_transpose2d_golden(golden, case):
    from cplxmodule_amd import Cplx, cplx
    g = golden("conv_transpose")
    k = f"f32_{case}_"
    names = ("xr", "xi", "wr", "wi", "br", "bi")
    t = {n: T(g[k + n]).requires_grad_(True) for n in names}
    y = cplx.conv_transpose2d(Cplx(t["xr"], t["xi"]), Cplx(t["wr"], t["wi"]), Cplx(t["br"], t["bi"]),
                              **CONVT_CASES[case]["kw"])
    assert y.shape == g[k + "yr"].shape
    np.testing.assert_allclose(N(y.real), g[k + "yr"], **_tol(g[k + "yr"]))
    np.testing.assert_allclose(N(y.imag), g[k + "yi"], **_tol(g[k + "yi"]))
    torch.autograd.backward((y.real, y.imag), (T(g[k + "gr"]), T(g[k + "gi"])))
    for n in names:
        np.testing.assert_allclose(N(t[n].grad), g[k + "d" + n], **_tol(g[k + "d" + n], 5e-5), err_msg=n)


def test_conv_transpose1d_and_layers(golden):
    from cplxmodule_amd import Cplx, cplx
    from cplxmodule_amd.nn import CplxConvTranspose1d, CplxConvTranspose2d
    g = golden("conv_transpose")
    k = "f32_1d_"
    y = cplx.conv_transpose1d(Cplx(T(g[k + "xr"]), T(g[k + "xi"])), Cplx(T(g[k + "wr"]), T(g[k + "wi"])), None,
                              stride=2, padding=1, output_padding=1)
    np.testing.assert_allclose(N(y.real), g[k + "yr"], **_tol(g[k + "yr"]))
    np.testing.assert_allclose(N(y.imag), g[k + "yi"], **_tol(g[k + "yi"]))
    layer = CplxConvTranspose1d(3, 4, 4, stride=2, padding=1, output_padding=1).to(DEV)
    assert layer.bias is None                                  # the reference's default: bias=None
    layer.load_state_dict({"weight.real": T(g[k + "wr"]), "weight.imag": T(g[k + "wi"])})
    out = layer(Cplx(T(g[k + "xr"]), T(g[k + "xi"])))
    assert torch.equal(out.real, y.real)
    # output_size picks the output padding (what the reference's layer cannot do on torch >= 2)
    l2 = CplxConvTranspose2d(4, 6, 3, stride=2, padding=1, bias=True).to(DEV)
    x = cplx.randn(2, 4, 5, 7, device=DEV)
    assert l2(x).shape == (2, 6, 9, 13)
    assert l2(x, output_size=(10, 14)).shape == (2, 6, 10, 14)
    assert l2(x, output_size=(2, 6, 10, 13)).shape == (2, 6, 10, 13)
    with pytest.raises(ValueError):
        l2(x, output_size=(11, 13))
    w = l2.weight
    ref_r, ref_i = orc.cplx_conv_transpose2d(*(N(p).astype(np.float64) for p in (x.real, x.imag, w.real, w.imag,
                                                                              l2.bias.real, l2.bias.imag)),
                                             stride=2, padding=1, output_padding=1)
    out = l2(x, output_size=(10, 14))
    np.testing.assert_allclose(N(out.real), ref_r, **_tol(ref_r))
    np.testing.assert_allclose(N(out.imag), ref_i, **_tol(ref_i))


def test_conv_transpose_is_adjoint_of_conv_bf16():
    """<Conv_w(z), x> == <z, ConvT_conj(w)(x)> at a size where the channels-last kernels run, bf16."""
    from cplxmodule_amd import Cplx, cplx
    torch.manual_seed(0)
    z = cplx.randn(2, 32, 40, 40, device=DEV)
    w = Cplx(0.1 * torch.randn(64, 32, 3, 3, device=DEV), 0.1 * torch.randn(64, 32, 3, 3, device=DEV))
    x = cplx.randn(2, 64, 38, 38, device=DEV)
    bf = lambda c: Cplx(c.real.bfloat16(), c.imag.bfloat16())  # noqa: E731
    for cast, tol in ((lambda c: c, 1e-4), (bf, 2e-2)):
        cz = cplx.conv2d(cast(z), w)
        tx = cplx.conv_transpose2d(cast(x), Cplx(w.real, -w.imag))
        lhs = (cz.real.float() * x.real + cz.imag.float() * x.imag).sum().item()
        rhs = (z.real * tx.real.float() + z.imag * tx.imag.float()).sum().item()
        assert abs(lhs - rhs) <= tol * max(abs(lhs), abs(rhs), 1.0), (lhs, rhs)
