"""Round-5 GPU tests: the per-call launch policy of ABI 19 (two models on two streams / threads, one launching as if
inside a gradient-exchange window), the conv -> batch-norm moments coupling in the situations VERDICT r04 item 5 lists,
double backward through the fused Cplx product / split ReLU, the float32 mode at BASELINE's full configs[3] batch.
Everything goes through libcplxamd.so (C ABI via ctypes)."""
import copy
import threading

import numpy as np
import pytest
import torch

from gpu_util import DEV, N

pytestmark = pytest.mark.gpu


def _bf(*shape, scale=1.0, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, device=DEV, generator=g) * scale).bfloat16()


# ---- no process-wide launch state ------------------------------------------------------------------
def _vd_step(layer, x, seed):
    """One training step of a CplxLinearVD layer (LRT forward + KL + full backward); everything it produced."""
    from cplxmodule_amd.nn import relevance as rel
    from cplxmodule_amd.nn.relevance.noise import noise
    noise.manual_seed(seed)
    layer.zero_grad(set_to_none=True)
    x.real.grad = x.imag.grad = None
    y = layer(x)
    kl = sum(rel.penalties(layer))
    torch.autograd.backward((y.real, y.imag, kl), (y.real.detach() * 2, y.imag.detach() * 2, torch.tensor(1e-3, device=DEV)))
    out = [y.real.detach().clone(), y.imag.detach().clone(), x.real.grad.clone(), x.imag.grad.clone()]
    out += [p.grad.clone() for p in layer.parameters()]
    return out


def test_two_models_two_streams_two_threads_choose_their_launch_forms_independently():
    """Model A launches as inside an exchange window (CPLXAMD_LAUNCH_SHARED: one workgroup per tile), model B as the owner
    of the chip (CPLXAMD_LAUNCH_EXCLUSIVE: persistent forms), concurrently, from two threads on two streams -- forward and
    backward (the autograd engine replays each backward on its forward's stream, which is what the policy is keyed by).
    Every result is bit-identical to the serial run under the library's defaults, and the deprecated process-wide default
    is never touched.  The shapes are ones where the two forms are DIFFERENT kernels (cplxamd_gemm_plan says which)."""
    from cplxmodule_amd import Cplx, _lib
    from cplxmodule_amd.nn import relevance as rel
    L = _lib.load()
    S, E = _lib.LAUNCH_SHARED, _lib.LAUNCH_EXCLUSIVE
    B, F = 8192, 2048                                   # K = 2048: the input gradients run on the 8-wave family
    assert L.cplxamd_gemm_plan(1, B, F, F, 0, 1, _lib.BF16, 1, S, 0) == 1      # fused input gradient: one-tile kernel ...
    assert L.cplxamd_gemm_plan(1, B, F, F, 0, 1, _lib.BF16, 1, E, 0) == 2      # ... or the persistent one
    # (the real launches of this layer are 256 tiles = one per CU: one form only; the channels-last convolutions and the
    #  real GEMMs of larger layers switch like the complex one above)
    torch.manual_seed(0)
    models = [rel.CplxLinearVD(F, F).to(DEV) for _ in range(2)]
    for m in models:
        with torch.no_grad():
            m.log_sigma2.uniform_(-10, 0)
        m.train()
    xs = [Cplx(_bf(B, F, seed=10 + i).requires_grad_(True), _bf(B, F, seed=20 + i).requires_grad_(True)) for i in range(2)]
    for i in range(2):
        _vd_step(models[i], xs[i], 1)                   # arms the fused KL: every later step is alike
    serial = [_vd_step(models[i], xs[i], 100 + i) for i in range(2)]
    torch.cuda.synchronize()
    before = L.cplxamd_gemm_set_persistent(1)
    assert before == 1
    streams = [torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)]
    got, errors = [None, None], []
    gate = threading.Barrier(2)

    def worker(i, flags):
        try:
            torch.cuda.set_device(0)
            streams[i].wait_stream(torch.cuda.default_stream())
            with torch.cuda.stream(streams[i]), _lib.launch_policy(flags):
                assert _lib.launch_flags() == flags
                gate.wait(timeout=60)
                for _ in range(3):                      # several steps each, so that the two really interleave
                    got[i] = _vd_step(models[i], xs[i], 100 + i)
            streams[i].synchronize()
        except Exception as e:  # pragma: no cover
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=worker, args=(0, S)), threading.Thread(target=worker, args=(1, E))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    torch.cuda.synchronize()
    assert not errors, errors
    assert not _lib._policy and _lib.launch_flags() == 0
    assert L.cplxamd_gemm_set_persistent(1) == 1, "no launch moved the process-wide default"
    for i in range(2):
        for a, b in zip(serial[i], got[i]):
            assert torch.equal(a, b), f"model {i}: concurrent result differs from the serial one"
    # and the other way round, serially: the flags change the launch form, never the bits
    for i, flags in ((0, E), (1, S)):
        with _lib.launch_policy(flags | _lib.LAUNCH_FAMILY(0)):          # (... nor does the kernel family)
            again = _vd_step(models[i], xs[i], 100 + i)
        for a, b in zip(serial[i], again):
            assert torch.equal(a, b)


def test_flagged_entry_points_reject_contradictory_flags_and_match_the_plain_ones():
    from cplxmodule_amd import _lib, ops
    from cplxmodule_amd._lib import ptr, stream_ptr
    L = _lib.load()
    M, K, Nn = 512, 256, 256
    ar, ai, br, bi = _bf(M, K, seed=1), _bf(M, K, seed=2), _bf(Nn, K, seed=3, scale=0.1), _bf(Nn, K, seed=4, scale=0.1)
    ref = ops.cgemm(ar, ai, (K, 1), br, bi, (K, 1), M, Nn, K, out_dtype=torch.bfloat16)
    cr, ci = torch.empty_like(ref[0]), torch.empty_like(ref[1])
    args = lambda flags: (ptr(ar), ptr(ai), K, 1, ptr(br), ptr(bi), K, 1, None, None, None, ptr(cr), ptr(ci), Nn, M, Nn, K, 0,  # noqa: E731
                          _lib.BF16, _lib.BF16, 0, None, 0, None, 0, flags, stream_ptr())
    assert L.cplxamd_cgemm_fl(*args(_lib.LAUNCH_SHARED | _lib.LAUNCH_EXCLUSIVE)) == -1
    assert L.cplxamd_cgemm_fl(*args(0x40)) == -1
    for flags in (0, _lib.LAUNCH_SHARED, _lib.LAUNCH_EXCLUSIVE, _lib.LAUNCH_FAMILY(0), _lib.LAUNCH_FAMILY(0x7f) | _lib.LAUNCH_SHARED):
        cr.zero_(); ci.zero_()
        assert L.cplxamd_cgemm_fl(*args(flags)) == 0
        assert torch.equal(cr, ref[0]) and torch.equal(ci, ref[1]), flags


# ---- the conv -> batch-norm moments coupling ---------------------------------------------------------
def _conv_bn_inputs(seed=3):
    from cplxmodule_amd import Cplx
    cl = torch.channels_last
    x = Cplx(_bf(8, 64, 48, 64, seed=seed).contiguous(memory_format=cl), _bf(8, 64, 48, 64, seed=seed + 1).contiguous(memory_format=cl))
    return x


def _bn_reference(conv_layer, bn_layers, x):
    """The same modules with the moments path switched off: what every variant must reproduce."""
    from cplxmodule_amd import conv
    keep = conv._MOMENTS
    conv._MOMENTS = False
    try:
        y = conv_layer(x)
        return [bn(y) for bn in bn_layers]
    finally:
        conv._MOMENTS = keep


def _close(a, b):
    d = float((a.float() - b.float()).abs().max())
    return d <= 2 ** -6 * float(b.float().abs().max())           # a bf16 ulp of the largest entries


def test_conv_output_consumed_by_two_batchnorm_layers_and_under_no_grad():
    """One convolution feeding TWO training-mode batch-norm layers (both read the same epilogue moments, both re-arm), then
    the same pair under torch.no_grad() in training mode: outputs and running statistics equal those of the path with the
    coupling switched off."""
    from cplxmodule_amd import conv, nn, ops
    torch.manual_seed(1)
    c = nn.CplxConv2d(64, 64, 3, padding=1).to(DEV)
    bns = [nn.CplxBatchNorm2d(64).to(DEV) for _ in range(2)]
    twins = [copy.deepcopy(b) for b in bns]
    x = _conv_bn_inputs()
    conv._MOMENTS_WANTED.clear()
    for step in range(3):
        y = c(x)
        hinted = ops.moments_hint(y.real, y.imag) is not None
        assert hinted == (step > 0), (step, hinted)            # armed by the consumers of step 0
        outs = [b(y) for b in bns]
        refs = _bn_reference(c, twins, x)
        for o, r in zip(outs, refs):
            assert _close(o.real, r.real) and _close(o.imag, r.imag)
    for b, t in zip(bns, twins):
        np.testing.assert_allclose(N(b.running_mean), N(t.running_mean), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(N(b.running_var), N(t.running_var), rtol=2e-5, atol=2e-6)
        assert int(b.num_batches_tracked) == int(t.num_batches_tracked) == 3
    with torch.no_grad():                                        # training-mode statistics without autograd
        y = c(x)
        assert ops.moments_hint(y.real, y.imag) is not None
        o = bns[0](y)
        r = _bn_reference(c, twins[:1], x)[0]
    assert _close(o.real, r.real) and _close(o.imag, r.imag)
    np.testing.assert_allclose(N(bns[0].running_var), N(twins[0].running_var), rtol=2e-5, atol=2e-6)
    conv._MOMENTS_WANTED.clear()


def test_moments_request_expires_when_the_consumer_stops_being_a_batchnorm_and_copies_start_unarmed():
    """The convolution of a Sequential whose batch-norm layer is REPLACED (model surgery, no eval() in between) pays for
    the moments epilogue at most conv._MOMENTS_CREDIT more steps; a deepcopy of an armed pair starts unarmed (requests are
    keyed by the parameter object) and arms itself; conv.arm_conv_bn arms a pair from its FIRST step."""
    from cplxmodule_amd import conv, nn, ops
    torch.manual_seed(2)
    net = torch.nn.Sequential(nn.CplxConv2d(64, 64, 3, padding=1), nn.CplxBatchNorm2d(64)).to(DEV)
    x = _conv_bn_inputs(7)
    seen = []
    hook = lambda m, a, out: seen.append(ops.moments_hint(out.real, out.imag) is not None)  # noqa: E731
    h = net[0].register_forward_hook(hook)
    conv._MOMENTS_WANTED.clear()
    net.train()
    for _ in range(3):
        net(x)
    assert seen == [False, True, True]
    twin = copy.deepcopy(net)                                   # (forward hooks are copied along)
    del seen[:]
    twin(x); twin(x)
    assert seen == [False, True], "a copied pair starts unarmed and arms itself"
    del seen[:]
    net[1] = torch.nn.Identity()                                # the consumer is gone; nobody calls eval()
    for _ in range(conv._MOMENTS_CREDIT + 3):
        net(x)
    assert seen == [True] * conv._MOMENTS_CREDIT + [False] * 3, seen
    assert not conv.moments_wanted(net[0].weight.real)
    h.remove()
    # explicit arming at build time: the very first step already runs the armed variant (deterministic from step 1)
    conv._MOMENTS_WANTED.clear()
    fresh = torch.nn.Sequential(nn.CplxConv2d(64, 64, 3, padding=1), nn.CplxBatchNorm2d(64)).to(DEV)
    assert conv.arm_conv_bn(fresh) == 1
    del seen[:]
    h = fresh[0].register_forward_hook(hook)
    fresh.train()
    a = fresh(x)
    state = {k: v.clone() for k, v in fresh[1].state_dict().items()}
    fresh[1].reset_running_stats()
    b = fresh(x)
    assert seen == [True, True]
    assert torch.equal(a.real, b.real) and torch.equal(a.imag, b.imag), "armed from step 1: step 1 and step 2 are the same kernels"
    for k, v in fresh[1].state_dict().items():
        assert torch.equal(v, state[k]), k
    fresh.eval()
    del seen[:]
    fresh(x)
    assert conv.moments_wanted(fresh[0].weight.real), "a permanent request survives evaluation passes"
    assert seen == [False], "an evaluation-mode convolution does not run the moments epilogue, armed or not (ADVICE r05)"
    h.remove()
    conv.arm_conv_bn(fresh, on=False)
    assert not conv._MOMENTS_WANTED


# ---- double backward through the fused Cplx product / quotient and the split ReLU --------------------------
@pytest.mark.parametrize("div", [False, True])
def test_cplx_product_supports_create_graph(div):
    """The reference's Cplx.__mul__ / __truediv__ are compositions of torch ops, so gradient penalties differentiate
    through their backward; the one-launch kernels' backward must too (it used to raise)."""
    from cplxmodule_amd import Cplx
    g = torch.Generator(device=DEV).manual_seed(5)
    mk = lambda: torch.randn(64, 33, device=DEV, generator=g, dtype=torch.float32).requires_grad_(True)  # noqa: E731
    ar, ai, br, bi = mk(), mk(), mk(), mk()
    with torch.no_grad():
        br.add_(3.0)                                            # (keep the divisor away from zero)

    def penalty(use_kernel):
        if use_kernel:
            z = Cplx(ar, ai) / Cplx(br, bi) if div else Cplx(ar, ai) * Cplx(br, bi)
            zr, zi = z.real, z.imag
        else:
            a, b = torch.complex(ar, ai), torch.complex(br, bi)
            z = a / b if div else a * b
            zr, zi = z.real, z.imag
        loss = (zr ** 2).sum() + (zr * zi).sum()
        grads = torch.autograd.grad(loss, [ar, ai, br, bi], create_graph=True)
        return sum((t ** 2).sum() for t in grads)

    got = torch.autograd.grad(penalty(True), [ar, ai, br, bi])
    ref = torch.autograd.grad(penalty(False), [ar, ai, br, bi])
    for a, b in zip(got, ref):
        np.testing.assert_allclose(N(a), N(b), rtol=2e-4, atol=2e-4 * float(b.abs().max()))


def test_split_relu_supports_create_graph():
    from cplxmodule_amd import ops
    g = torch.Generator(device=DEV).manual_seed(6)
    xr = torch.randn(32, 40, device=DEV, generator=g).requires_grad_(True)
    xi = torch.randn(32, 40, device=DEV, generator=g).requires_grad_(True)
    w = torch.randn(32, 40, device=DEV, generator=g)

    def penalty(fn):
        yr, yi = fn(xr, xi)
        loss = (yr ** 3 * w).sum() + (yi ** 2 * yr).sum()
        grads = torch.autograd.grad(loss, [xr, xi], create_graph=True)
        return sum((t ** 2).sum() for t in grads)

    got = torch.autograd.grad(penalty(ops.split_relu), [xr, xi])
    ref = torch.autograd.grad(penalty(lambda a, b: (torch.relu(a), torch.relu(b))), [xr, xi])
    for a, b in zip(got, ref):
        np.testing.assert_allclose(N(a), N(b), rtol=1e-5, atol=1e-5 * float(b.abs().max()))


# ---- the KL recomputation guard at every site ------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["real_vd", "cplx_ard"])
def test_stale_parameters_are_refused_wherever_the_kl_gradients_are_recomputed(kind):
    """optimizer.step() between a forward pass and a SECOND backward pass through its retained graph: the fused KL
    gradients would be recomputed from the new parameter values -- every recompute site raises (round 4 covered only
    the KL-only branch of the complex layer)."""
    from cplxmodule_amd import Cplx
    from cplxmodule_amd.nn import relevance as rel
    torch.manual_seed(3)
    cplx_ = kind.startswith("cplx")
    layer = (rel.CplxLinearARD(64, 48) if cplx_ else rel.LinearVD(64, 48)).to(DEV)
    layer.train()
    mk = lambda: _bf(32, 64, seed=9)  # noqa: E731
    x = Cplx(mk(), mk()) if cplx_ else mk()

    def losses():
        y = layer(x)
        nll = (y.real.float() ** 2).sum() + (y.imag.float() ** 2).sum() if cplx_ else (y.float() ** 2).sum()
        return nll, sum(rel.penalties(layer))

    nll, kl = losses()
    (nll + kl).backward()                      # arms the fusion
    layer.zero_grad(set_to_none=True)
    for second in ("data+kl", "kl only"):
        nll, kl = losses()
        (nll + 0.1 * kl).backward(retain_graph=True)          # consumes the forward pass's KL buffers
        with torch.no_grad():
            layer.log_sigma2.add_(0.01)                       # an optimizer step, in place
        # (the data pass unpacks its saved tensors first, so there autograd's own version check speaks; the KL-only pass
        #  needs no saved tensor -- that one is ops._kl_recompute's)
        with pytest.raises(RuntimeError, match="modified in place" if second == "kl only" else "modified (in place|by an inplace)"):
            ((nll + 0.1 * kl) if second == "data+kl" else (0.1 * kl)).backward()
        layer.zero_grad(set_to_none=True)


# ---- VERDICT r04 "parity softness": the float32 mode at configs[3]'s FULL batch --------------------------------------------
def test_cfg4_float32_forward_at_full_batch_matches_float64_rows():
    """CplxLinearVD(2048, 2048) on 2^20 float32 rows -- the 1e-5 mode at BASELINE's size (the round-4 full-size checks ran
    bf16): sampled rows of the LRT forward against float64 with the noise from the numpy Philox statement, the KL against
    the oracle.  ~70 GB of planes: skipped on a GPU with less free memory."""
    import oracle.cplx_oracle as orc
    from cplxmodule_amd import Cplx
    from cplxmodule_amd.nn import relevance as rel
    from cplxmodule_amd.nn.relevance.noise import noise
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 100 << 30:
        pytest.skip(f"{free >> 30} GiB free")
    B, F = 1 << 20, 2048
    torch.manual_seed(0)
    layer = rel.CplxLinearVD(F, F).to(DEV)
    with torch.no_grad():
        layer.log_sigma2.uniform_(-12, 2)
    layer.train()
    g = torch.Generator(device=DEV).manual_seed(1)
    xr = torch.randn(B, F, device=DEV, generator=g) * 0.7071
    xi = torch.randn(B, F, device=DEV, generator=g) * 0.7071
    noise.manual_seed(4242)                      # the first stochastic pass after it draws (seed 4242, offset 1)
    with torch.no_grad():
        y = layer(Cplx(xr, xi))
    rows = np.array([0, 1, 77, 65535, 65536, (1 << 19) + 3, B - 2, B - 1])
    tr = torch.from_numpy(rows).to(DEV)
    f = np.float64
    wr, wi, ls2 = (N(t).astype(f) for t in (layer.weight.real, layer.weight.imag, layer.log_sigma2))
    br, bi = N(layer.bias.real).astype(f), N(layer.bias.imag).astype(f)
    xs_r, xs_i = xr[tr].double().cpu().numpy(), xi[tr].double().cpu().numpy()
    mu_r = xs_r @ wr.T - xs_i @ wi.T + br
    mu_i = xs_r @ wi.T + xs_i @ wr.T + bi
    s2 = (xs_r ** 2 + xs_i ** 2) @ np.exp(ls2).T
    from test_gpu_fullsize import _cplx_noise_at
    er, ei = _cplx_noise_at((rows[:, None].astype(np.uint64) * np.uint64(F) + np.arange(F, dtype=np.uint64)[None]).ravel(), 4242, 1)
    er, ei = er.reshape(len(rows), F), ei.reshape(len(rows), F)
    sig = np.sqrt(np.maximum(s2, 1e-8))
    ref_r, ref_i = mu_r + er * sig, mu_i + ei * sig
    sc = max(np.abs(ref_r).max(), np.abs(ref_i).max())
    np.testing.assert_allclose(N(y.real[tr]), ref_r, rtol=1e-5, atol=1e-5 * sc)
    np.testing.assert_allclose(N(y.imag[tr]), ref_i, rtol=1e-5, atol=1e-5 * sc)
    kl = float(sum(rel.penalties(layer)))
    np.testing.assert_allclose(kl, orc.penalty("cplx_vd", ls2, wr, wi).sum(), rtol=2e-6)
    del y, xr, xi
    torch.cuda.empty_cache()


# ---- the ring through the tile boundaries (gemm_bf16_w4.hip: PERSIST, kernel-family bit 7) ----------------
@pytest.mark.parametrize("K", [384, 448, 512, 4096])      # K tile counts 12, 14, 16: every exit path of the six-tile loop
def test_w4_persistent_form_is_bit_identical(K):
    """One workgroup per CU walking its tiles with the K-tile ring continuing into the next output tile (the look-ahead
    loads fetch the next tile's K tiles 0 .. 3; the LDS base registers swap roles at the boundary; the epilogue stages in
    the ring slot that died with the last K tile): same MFMA sequence per accumulator as the one-tile kernels and the
    8-wave family -- identical bits -- for the complex forward / input gradient and the real ones, more tiles than CUs."""
    import os
    from cplxmodule_amd import _lib, ops
    L = _lib.load()
    W4P_CPLX = False                          # (the complex persistent form left the library in round 6)
    B, Nn = 8192, 4096                        # 32 x 32 complex tiles (32 x 16 real): four (two) per workgroup
    E = _lib.LAUNCH_EXCLUSIVE
    assert L.cplxamd_gemm_plan(0, B, Nn, K, 0, 0, _lib.BF16, 0, _lib.LAUNCH_FAMILY(0xff) | E, 0) == 6
    assert L.cplxamd_gemm_plan(0, B, Nn, K, 0, 0, _lib.BF16, 0, _lib.LAUNCH_FAMILY(0xff) | _lib.LAUNCH_SHARED, 0) == 3
    assert L.cplxamd_gemm_plan(1, B, Nn, K, 0, 0, _lib.BF16, 0, _lib.LAUNCH_FAMILY(0xff) | E, 0) == (6 if W4P_CPLX else 3)
    bf = torch.bfloat16
    xr, xi = _bf(B, K, seed=1), _bf(B, K, seed=2)
    wr, wi = _bf(Nn, K, seed=3, scale=0.05), _bf(Nn, K, seed=4, scale=0.05)       # read as B[n, k] (N,N) ...
    vr, vi = _bf(K, Nn, seed=5, scale=0.05), _bf(K, Nn, seed=6, scale=0.05)       # ... and as the stored [K, N] weight (N,T)
    a2, S = _bf(B, K, seed=7).abs(), _bf(Nn, K, seed=8).abs()

    def run():
        out = list(ops.cgemm(xr, xi, (K, 1), wr, wi, (K, 1), B, Nn, K, out_dtype=bf))
        out += list(ops.cgemm(xr, xi, (K, 1), vr, vi, (1, Nn), B, Nn, K, conj_b=True, out_dtype=bf))
        out.append(ops.rgemm(a2, (K, 1), S, (K, 1), B, Nn, K, out_dtype=bf))
        out.append(ops.rgemm(a2, (K, 1), vr, (1, Nn), B, Nn, K, out_dtype=bf))
        return [t.clone() for t in out]

    with _lib.launch_policy(_lib.LAUNCH_FAMILY(0) | E):
        ref = run()
    with _lib.launch_policy(_lib.LAUNCH_FAMILY(0xff) | E):
        got = run()
    with _lib.launch_policy(_lib.LAUNCH_FAMILY(0x7f) | E):
        one = run()
    for r, g, o in zip(ref, got, one):
        assert torch.isfinite(g.float()).all()
        assert torch.equal(r, o) and torch.equal(r, g)


def test_cfg3_float32_conv_at_full_batch_matches_float64():
    """CplxConv2d(64, 64, 3) on 256 float32 256 x 256 images, channels-last -- the 1e-5 mode at BASELINE configs[2]'s size:
    forward pixels of the last images, the data gradient of the last image and sampled weight-gradient entries against
    float64 (reference: cplx.py:717-742 and its autograd).  ~25 GB of planes."""
    import oracle.cplx_oracle as orc
    from cplxmodule_amd import Cplx, nn
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 60 << 30:
        pytest.skip(f"{free >> 30} GiB free")
    B, C, H = 256, 64, 256
    cl = torch.channels_last
    g = torch.Generator(device=DEV).manual_seed(31)
    mk = lambda *s: torch.randn(*s, device=DEV, generator=g).contiguous(memory_format=cl)  # noqa: E731
    xr, xi = mk(B, C, H, H).requires_grad_(True), mk(B, C, H, H).requires_grad_(True)
    torch.manual_seed(2)
    conv = nn.CplxConv2d(C, C, 3).to(DEV)
    y = conv(Cplx(xr, xi))
    assert y.real.shape == (B, C, H - 2, H - 2) and y.real.dtype == torch.float32
    f = np.float64
    w, b = conv.weight, conv.bias
    Wr, Wi = N(w.real).astype(f), N(w.imag).astype(f)
    sub = (slice(254, 256), slice(None), slice(100, 110), slice(200, 212))
    rr, ri = orc.cplx_conv2d(xr[sub].detach().double().cpu().numpy(), xi[sub].detach().double().cpu().numpy(), Wr, Wi,
                             N(b.real).astype(f), N(b.imag).astype(f))
    np.testing.assert_allclose(N(y.real[254:256, :, 100:108, 200:210]), rr, rtol=1e-5, atol=1e-5 * np.abs(rr).max())
    np.testing.assert_allclose(N(y.imag[254:256, :, 100:108, 200:210]), ri, rtol=1e-5, atol=1e-5 * np.abs(ri).max())
    # backward: upstream gradient = a fixed random field (kept: the references below need it)
    gr, gi = mk(B, C, H - 2, H - 2) * 0.1, mk(B, C, H - 2, H - 2) * 0.1
    torch.autograd.backward((y.real, y.imag), (gr, gi))
    del y
    # data gradient of the LAST image on a window: dX = full correlation of G with conj(W) flipped; float64 by torch on the host
    n0 = B - 1
    G = torch.complex(gr[n0:n0 + 1, :, 40:60, 50:70].double(), gi[n0:n0 + 1, :, 40:60, 50:70].double()).cpu()
    Wc = torch.complex(w.real.detach().double(), w.imag.detach().double()).cpu()
    # dx[c, p] = sum_{o, t} G[o, p - t] conj(W[o, c, t])  -> conv_transpose2d of G with conj(W)
    dx_r = torch.nn.functional.conv_transpose2d(G.real, Wc.real) + torch.nn.functional.conv_transpose2d(G.imag, Wc.imag)
    dx_i = torch.nn.functional.conv_transpose2d(G.imag, Wc.real) - torch.nn.functional.conv_transpose2d(G.real, Wc.imag)
    # interior of the window only (its border lacks the contributions of output pixels outside the window)
    got_r = xr.grad[n0, :, 42:60, 52:70].double().cpu().numpy()
    got_i = xi.grad[n0, :, 42:60, 52:70].double().cpu().numpy()
    ref_r, ref_i = dx_r[0, :, 2:20, 2:20].numpy(), dx_i[0, :, 2:20, 2:20].numpy()
    np.testing.assert_allclose(got_r, ref_r, rtol=1e-5, atol=1e-5 * np.abs(ref_r).max())
    np.testing.assert_allclose(got_i, ref_i, rtol=1e-5, atol=1e-5 * np.abs(ref_i).max())
    # weight gradient entries: dW[o, c, kh, kw] = sum_{b, p} G[b, o, p] conj(X[b, c, p + k]) over the WHOLE batch
    for (o, c, kh, kw) in ((0, 0, 0, 0), (63, 17, 2, 1), (31, 63, 1, 2)):
        Xr = xr.detach()[:, c, kh:kh + H - 2, kw:kw + H - 2].double()
        Xi = xi.detach()[:, c, kh:kh + H - 2, kw:kw + H - 2].double()
        Gr, Gi = gr[:, o].double(), gi[:, o].double()
        ref_wr = float((Gr * Xr + Gi * Xi).sum())
        ref_wi = float((Gi * Xr - Gr * Xi).sum())
        scale = float(w.real.grad.abs().max())
        # (each entry is a float32 sum of 16.5 M products in split-K slabs: 3e-5 of the largest entry, norm-wise)
        assert abs(float(w.real.grad[o, c, kh, kw]) - ref_wr) <= 3e-5 * scale, (o, c, kh, kw, float(w.real.grad[o, c, kh, kw]), ref_wr)
        assert abs(float(w.imag.grad[o, c, kh, kw]) - ref_wi) <= 3e-5 * scale, (o, c, kh, kw, float(w.imag.grad[o, c, kh, kw]), ref_wi)
    ref_b = float(gr[:, 5].double().sum())
    assert abs(float(b.real.grad[5]) - ref_b) <= 3e-5 * float(b.real.grad.abs().max())
    del xr, xi, gr, gi
    torch.cuda.empty_cache()


# ---- the persistent real kernels where workgroups run OUT of tiles at different times, and at the shortest K ------
@pytest.mark.parametrize("M,K", [(8192 + 256, 128), (8192 + 256, 192), (8192 + 256, 320), (8192 + 512, 4096), (8192 + 256, 2048)])
def test_w4_persistent_real_remainder_tiles_and_short_k(M, K):
    """tiles % CUs != 0 (some workgroups see `has_next == false` one tile earlier than others) and K of 4 / 6 / 10 K tiles
    (the look-ahead covers the next tile's whole K range at 128): family bit 7 (+ bit 6: regardless of K) against the
    one-tile family and the 8-wave kernels, bit for bit."""
    from cplxmodule_amd import _lib, ops
    L = _lib.load()
    Nn = 4096
    E = _lib.LAUNCH_EXCLUSIVE
    tiles = (M // 256) * (Nn // 256)
    assert tiles % 256 != 0
    assert L.cplxamd_gemm_plan(0, M, Nn, K, 0, 0, _lib.BF16, 0, _lib.LAUNCH_FAMILY(0xff) | E, 256) == 6
    bf = torch.bfloat16
    a2, S = _bf(M, K, seed=7).abs(), _bf(Nn, K, seed=8).abs()
    v = _bf(K, Nn, seed=5, scale=0.05)

    def run():
        return [ops.rgemm(a2, (K, 1), S, (K, 1), M, Nn, K, out_dtype=bf).clone(),
                ops.rgemm(a2, (K, 1), v, (1, Nn), M, Nn, K, out_dtype=bf).clone()]

    with _lib.launch_policy(_lib.LAUNCH_FAMILY(0) | E):
        ref = run()
    with _lib.launch_policy(_lib.LAUNCH_FAMILY(0xff) | E):
        got = run()
    with _lib.launch_policy(_lib.LAUNCH_FAMILY(0x7f) | E):
        one = run()
    for r, g_, o in zip(ref, got, one):
        assert torch.equal(r, g_) and torch.equal(r, o)
