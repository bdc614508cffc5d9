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


def test_bench_self_launch_command(monkeypatch):
    """bench.py --gpus N (N > 1) without a launcher: the command it becomes (no GPU needed: os.execv is intercepted)."""
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_execv(path, argv):
        seen["path"], seen["argv"] = path, list(argv)
        raise SystemExit(0)

    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    import pytest
    with pytest.raises(SystemExit):
        bench.main()
    a = seen["argv"]
    assert seen["path"] == sys.executable and a[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in a and "--nproc-per-node=4" in a
    assert a[a.index("--master-addr") + 1] == "127.0.0.1" and int(a[a.index("--master-port") + 1]) > 0
    i = a.index(os.path.join(root, "bench.py"))
    assert a[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    # under a launcher (WORLD_SIZE set) with a mismatching world it refuses instead of re-launching
    monkeypatch.setenv("WORLD_SIZE", "2")
    seen.clear()
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "argv" not in seen and "WORLD_SIZE=2" in str(e.value)
    w = bench.expected_weak_8(3.2, 8192)
    assert w["bucket_bytes"] == (3 * 4096 * 4096 + 2 * 4096) * 4
    assert w["ring_unoverlapped"]["samples_per_s"] < w["direct_overlapped"]["samples_per_s"] <= 8 * 8192 / 3.2e-3 + 1


def test_x3_take_rules_host():
    from cplxmodule_amd import x3
    assert not x3.take(64, 64, 32, mode="exact") and x3.take(64, 64, 32, mode="x3") and not x3.take(64, 64, 32, mode="auto")
    assert x3.take(1024, 1024, 1024, mode="auto") and not x3.take(1024, 1020, 1024, mode="x3")
    assert not x3.take(1024, 1024, 1000, mode="x3") and not x3.take(1 << 21, 64, 64, mode="x3")
    import torch
    assert not x3.take(64, 64, 32, torch.zeros(1), mode="x3")            # CPU tensor: never
    prev = x3.set_fp32_mode("exact")
    try:
        assert x3.get_fp32_mode() == "exact"
        with x3.fp32_mode("x3"):
            assert x3.get_fp32_mode() == "x3"
        assert x3.get_fp32_mode() == "exact"
    finally:
        x3.set_fp32_mode(prev)


def test_float64_routes_resolve_and_refuse_cpu_tensors():
    """Every ops.Route names a function of f64.py taking the Function's own positional arguments (forward minus ctx);
    float64 tensors on the CPU reach the library's device check and raise -- no torch fallback behind the route."""
    import inspect
    import torch
    from cplxmodule_amd import bn, conv, cplx, f64, ops
    from cplxmodule_amd._lib import CplxAmdError
    routes = [v for m in (ops, conv, bn, cplx) for v in vars(m).values() if isinstance(v, ops.Route)]
    assert len({id(r) for r in routes}) >= 17
    for r in routes:
        target = getattr(f64, r.f64_name)
        want = list(inspect.signature(r.fn.forward).parameters)[1:]
        got = inspect.signature(target).parameters
        n_pos = sum(p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD) for p in got.values())
        var = any(p.kind == p.VAR_POSITIONAL for p in got.values())
        assert var or n_pos == len(want), (r.f64_name, want, list(got))
    x = torch.randn(4, 6, dtype=torch.float64)
    w = torch.randn(3, 6, dtype=torch.float64)
    with pytest.raises(CplxAmdError):
        cplx.linear(cplx.Cplx(x, x), cplx.Cplx(w, w))
    with pytest.raises(CplxAmdError):
        ops.ExpiFn.apply(x)
