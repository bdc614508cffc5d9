"""Host-side pieces that need no GPU: views, containers, trivial layers."""
import pytest
import torch


def test_complex_view_and_containers():
    """utils.views.complex_view, CplxSequential's type check, the trivial Cplx -> real layers"""
    import warnings
    from collections import OrderedDict
    from cplxmodule_amd import Cplx
    from cplxmodule_amd.utils import complex_view, fix_dim
    from cplxmodule_amd import nn as cnn
    x = torch.arange(24.0).reshape(2, 3, 4).requires_grad_(True)
    re, im = complex_view(x, -1)
    assert torch.equal(re, x[..., 0::2]) and torch.equal(im, x[..., 1::2])
    assert re.data_ptr() == x.data_ptr()                       # views, not copies
    (re.sum() + 2 * im.sum()).backward()
    assert torch.equal(x.grad[..., 0::2], torch.ones(2, 3, 2)) and torch.equal(x.grad[..., 1::2], 2 * torch.ones(2, 3, 2))
    y = torch.arange(12.0).reshape(2, 2, 3)
    re, im = complex_view(y, 1)                                # size-2 axis is dropped
    assert re.shape == (2, 3) and torch.equal(im, y[:, 1])
    re, im = complex_view(y, 1, squeeze=False)
    assert re.shape == (2, 1, 3)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        re, im = complex_view(y, -1)
    assert re.shape == (2, 2, 1) and any(issubclass(m.category, RuntimeWarning) for m in w)
    assert fix_dim(-1, 3) == 2
    with pytest.raises(ValueError):
        fix_dim(3, 3)
    z = Cplx(torch.randn(4, 5), torch.randn(4, 5))
    assert cnn.CplxReal()(z) is z.real and cnn.CplxImag()(z) is z.imag and cnn.CplxIdentity()(z) is z
    seq = cnn.CplxSequential(cnn.CplxIdentity(), cnn.CplxPhaseShift(5))
    out = seq(z)
    phi = seq[1].phi.detach()
    ref = torch.complex(z.real, z.imag) * torch.exp(1j * phi)
    assert torch.allclose(out.real, ref.real, atol=1e-6) and torch.allclose(out.imag, ref.imag, atol=1e-6)
    assert len(cnn.CplxSequential(OrderedDict(a=cnn.CplxIdentity()))) == 1
    with pytest.raises(TypeError):
        cnn.CplxSequential(cnn.CplxIdentity(), torch.nn.ReLU())
    with pytest.raises(TypeError):
        cnn.CplxSequential(cnn.CplxReal())


def test_moment_hints_and_arming_bookkeeping():
    """The host side of the conv -> batch-norm moments path (no kernels): the hint on a pair of output planes is valid
    only for exactly those tensors, unmodified; the request registry holds weight planes weakly and by identity (tensors
    compare elementwise, so no set of tensors)."""
    import gc
    from cplxmodule_amd import conv, ops
    yr, yi = torch.zeros(2, 4, 3, 3), torch.zeros(2, 4, 3, 3)
    partials = torch.zeros(3 * 4 * 5, dtype=torch.float64)
    assert ops.moments_hint(yr, yi) is None
    ops.attach_moments(yr, yi, partials, 3)
    got = ops.moments_hint(yr, yi)
    assert got is not None and got[0] is partials and got[1] == 3
    assert ops.moments_hint(yr, yi.clone()) is None                      # another imaginary plane
    assert ops.moments_hint(yr.clone(), yi) is None                      # the attribute does not travel with a copy
    yi.add_(1)                                                           # either plane modified in place: stale
    assert ops.moments_hint(yr, yi) is None
    ops.attach_moments(yr, yi, partials, 3)
    yr.mul_(2)
    assert ops.moments_hint(yr, yi) is None

    conv._MOMENTS_WANTED.clear()
    w1, w2 = torch.nn.Parameter(torch.ones(4, 4, 3, 3)), torch.nn.Parameter(torch.ones(4, 4, 3, 3))
    conv.want_moments(w1)
    conv.want_moments(w1)
    assert conv.moments_wanted(w1) and not conv.moments_wanted(w2)       # equal values, different layer
    conv.want_moments(w2)
    conv.want_moments(w1, on=False)
    assert not conv.moments_wanted(w1) and conv.moments_wanted(w2)
    conv.want_moments(None)                                              # (the producer's weight is gone)
    del w2
    gc.collect()
    assert not conv._MOMENTS_WANTED                                      # a dead layer leaves nothing behind


def test_moment_requests_are_credits_that_expire():
    """a convolution whose output stops feeding a batch-norm layer must stop paying for the moments
    epilogue by itself (not only on eval()): every armed forward spends a credit, every consuming BN forward refills."""
    from cplxmodule_amd import conv
    conv._MOMENTS_WANTED.clear()
    w = torch.nn.Parameter(torch.ones(4, 4, 3, 3))
    conv.want_moments(w)                                   # a batch-norm layer consumed the output once
    spent = 0
    while conv.moments_wanted(w, spend=True):              # ... and never again: the convolution's next forwards
        spent += 1
        assert spent <= 8
    assert spent == conv._MOMENTS_CREDIT and not conv._MOMENTS_WANTED
    for _ in range(5):                                     # consumed every step: stays armed
        conv.want_moments(w)
        assert conv.moments_wanted(w, spend=True)
    assert conv.moments_wanted(w)
    # a permanent request (arm_conv_bn) is neither spent nor withdrawn by a consumer in evaluation mode
    conv.want_moments(w, permanent=True)
    for _ in range(10):
        assert conv.moments_wanted(w, spend=True)
    conv.want_moments(w, on=False)
    assert conv.moments_wanted(w)
    conv.want_moments(w)                                   # a consuming layer must not downgrade it to a credit either
    for _ in range(10):
        assert conv.moments_wanted(w, spend=True)
    conv.want_moments(w, on=False, permanent=True)
    assert not conv._MOMENTS_WANTED


def test_arm_conv_bn_finds_the_direct_pairs_only():
    from cplxmodule_amd import conv, nn
    conv._MOMENTS_WANTED.clear()
    c1, c2, c3 = nn.CplxConv2d(4, 4, 3), nn.CplxConv2d(4, 4, 3), nn.CplxConv2d(4, 4, 3)
    net = torch.nn.Sequential(c1, nn.CplxBatchNorm2d(4),
                              torch.nn.Sequential(c2, nn.CplxBatchNorm2d(4)),
                              c3, nn.CplxIdentity() if hasattr(nn, "CplxIdentity") else torch.nn.Identity(),
                              nn.CplxBatchNorm2d(4))
    assert conv.arm_conv_bn(net) == 2
    assert conv.moments_wanted(c1.weight.real) and conv.moments_wanted(c2.weight.real)
    assert not conv.moments_wanted(c3.weight.real)         # something sits between the pair
    import copy
    twin = copy.deepcopy(net)                              # requests are keyed by the parameter OBJECT: a copy starts unarmed
    assert not conv.moments_wanted(twin[0].weight.real)
    assert conv.arm_conv_bn(net, on=False) == 2 and not conv._MOMENTS_WANTED
