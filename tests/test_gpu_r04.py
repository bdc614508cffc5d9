"""Round-4 GPU tests: the one-wave-per-SIMD GEMM family (csrc/gemm_bf16_w4.hip) against the 8-wave family and the
float64 oracle, and the round-3 advisor finding about the two-call backward pattern under a data-parallel hook.
Everything goes through libcplxamd.so (C ABI via ctypes)."""
import numpy as np
import pytest
import torch

import oracle.cplx_oracle as orc
from gpu_util import DEV, N

pytestmark = pytest.mark.gpu


def _bf(*shape, scale=1.0, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, device=DEV, generator=g) * scale).bfloat16()


@pytest.fixture
def family():
    """cplxamd_gemm_set_family, restored afterwards."""
    from cplxmodule_amd import _lib
    lib = _lib.load()
    prev = lib.cplxamd_gemm_set_family(-1)
    yield lib.cplxamd_gemm_set_family
    lib.cplxamd_gemm_set_family(prev)


# (B, I, O): K tile counts 4 (ring tail only), 12, 14 (6 + 6 + 2), 16, 20, 24 -- every entry / exit of the 6-tile ring loop;
# 256-row / 128-column (complex) and 256-column (real) tiles, one and several per launch
SHAPES = [(256, 128, 256), (512, 384, 512), (768, 448, 512), (1024, 640, 256), (512, 768, 768)]


@pytest.mark.parametrize("B,I,O", SHAPES)
def test_w4_family_is_bit_identical_to_w8(family, B, I, O):
    """Same MFMA sequence per accumulator, same epilogue arithmetic: forward (N,N) with bias, input gradient (N,T) plain and
    with the fused LRT term, weight gradient (T,T) float32 with the fused-KL accumulate; complex and real."""
    from cplxmodule_amd import ops
    bf = torch.bfloat16
    xr, xi, gr, gi = _bf(B, I, seed=1), _bf(B, I, seed=2), _bf(B, O, seed=3), _bf(B, O, seed=4)
    wr, wi = _bf(O, I, scale=0.05, seed=5), _bf(O, I, scale=0.05, seed=6)
    bias = (torch.randn(O, device=DEV), torch.randn(O, device=DEV))
    ga, a2, gs2 = _bf(B, I, seed=7), _bf(B, I, seed=8).abs(), _bf(B, O, seed=9)
    S = torch.empty(O, I, device=DEV).uniform_(-12, 4).exp().to(bf)
    ls2 = torch.empty(O, I, device=DEV).uniform_(-12, 4)
    kl0 = [torch.randn(O, I, device=DEV) for _ in range(2)]
    beta = torch.tensor(0.37, device=DEV)

    def launches():
        out = {}
        out["c_fwd"] = ops.cgemm(xr, xi, (I, 1), wr, wi, (I, 1), B, O, I, bias=bias, out_dtype=bf)
        out["c_dx"] = ops.cgemm(gr, gi, (O, 1), wr, wi, (1, I), B, I, O, conj_b=True, out_dtype=bf)
        out["c_dx_lrt"] = ops._cplx_lrt_dx(gr, gi, wr, wi, xr, xi, ga)
        dw = [t.clone() for t in kl0]
        ops.cgemm(gr, gi, (1, O), xr, xi, (1, I), O, I, B, conj_b=True, out=dw, accumulate=True, beta=beta)
        out["c_dw_kl"] = dw
        out["c_dw"] = ops.cgemm(gr, gi, (1, O), xr, xi, (1, I), O, I, B, conj_b=True, out_dtype=torch.float32)
        out["r_fwd"] = [ops.rgemm(a2, (I, 1), S, (I, 1), B, O, I, out_dtype=bf)]
        out["r_fwd_f32"] = [ops.rgemm(a2, (I, 1), S, (I, 1), B, O, I, bias=bias[0], out_dtype=torch.float32)]
        out["r_dx"] = [ops.rgemm(gs2, (O, 1), S, (1, I), B, I, O, out_dtype=bf)]
        out["r_dx_lrt"] = [ops._real_lrt_dx(gs2, S, xr, ga)]
        d = kl0[0].clone()
        ops.rgemm(gs2, (1, O), a2, (1, I), O, I, B, emul=ls2, emul_exp=True, out=d, accumulate=True, beta=beta)
        out["r_dw_kl"] = [d]
        return out

    family(0)
    ref = launches()
    family(-1)
    got = launches()
    for k in ref:
        for a, b in zip(ref[k], got[k]):
            assert a.dtype == b.dtype and torch.isfinite(b.float()).all(), k
            assert torch.equal(a, b), f"{k}: {(a != b).sum().item()} of {a.numel()} elements differ"
    # ... and the forward against the float64 oracle on a few rows (bf16 output rounding), so that "identical" is not
    # "identically wrong"
    rows = [0, 1, B // 2 + 3, B - 1]
    f = np.float64
    rr, ri = orc.cplx_linear(N(xr[rows].float()).astype(f), N(xi[rows].float()).astype(f), N(wr.float()).astype(f),
                             N(wi.float()).astype(f), N(bias[0]).astype(f), N(bias[1]).astype(f))
    sc = np.abs(rr).max()
    np.testing.assert_allclose(N(got["c_fwd"][0][rows].float()), rr, rtol=8e-3, atol=8e-3 * sc)
    np.testing.assert_allclose(N(got["c_fwd"][1][rows].float()), ri, rtol=8e-3, atol=8e-3 * sc)


def test_w4_family_declines_what_it_does_not_take(family):
    """Partial tiles, K not a multiple of 64, (T,N) layouts: the 8-wave / generic kernels run, the results stay right."""
    from cplxmodule_amd import ops
    bf = torch.bfloat16
    family(-1)
    for B, I, O in [(250, 128, 256), (256, 96, 256), (256, 128, 200)]:
        xr, xi = _bf(B, I, seed=1), _bf(B, I, seed=2)
        wr, wi = _bf(O, I, scale=0.05, seed=5), _bf(O, I, scale=0.05, seed=6)
        yr, yi = ops.cgemm(xr, xi, (I, 1), wr, wi, (I, 1), B, O, I, out_dtype=bf)
        f = np.float64
        rr, ri = orc.cplx_linear(N(xr.float()).astype(f), N(xi.float()).astype(f), N(wr.float()).astype(f), N(wi.float()).astype(f))
        sc = np.abs(rr).max()
        np.testing.assert_allclose(N(yr.float()), rr, rtol=8e-3, atol=8e-3 * sc)
        np.testing.assert_allclose(N(yi.float()), ri, rtol=8e-3, atol=8e-3 * sc)


# ---- nll.backward(); (c * kl).backward() under a data-parallel hook --------------------------------
class _CopyingBuckets:
    """dp.BucketHook as ops sees it -- persistent float32 storage per parameter handed out as gradient buffers, an
    announce call -- PLUS what its post-accumulate-grad hook does to that storage: the final .grad is copied into the
    bucket slice (dp.py: BucketHook.on_grad)."""

    def __init__(self, params):
        self.store = {id(p): torch.zeros(p.shape, dtype=torch.float32, device=p.device) for p in params}
        self.announced = []
        self.handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]

    def _on_grad(self, p):
        t = self.store[id(p)]
        if p.grad is not None and p.grad.data_ptr() != t.data_ptr():
            t.copy_(p.grad)

    def view_for(self, p):
        t = self.store.get(id(p))
        return None if t is None else t.view(t.shape)

    def early_ready(self, *params):
        self.announced += [p for p in params if p is not None]

    def remove(self):
        for h in self.handles:
            h.remove()


@pytest.mark.parametrize("kind", ["cplx_vd", "real_ard"])
@pytest.mark.parametrize("mode", ["nll_first", "kl_first", "one"])
def test_two_call_backward_with_fused_kl_under_dp_hook(kind, mode):
    """With the hook active the pending KL gradients live in the parameters' bucket slices -- the storage every backward
    pass's result is copied into.  The data-only pass used to leave them pending, and the KL-only pass then returned the
    (copied) data gradient scaled by c as the KL gradient.  No error, wrong numbers."""
    from cplxmodule_amd import ops
    from cplxmodule_amd.nn.relevance import penalties
    from cplxmodule_amd.nn.relevance.noise import noise
    from test_gpu_r03 import _grads, _make, _nll
    layer, x, cplx_ = _make(kind)
    c = 0.37

    def run(m):
        noise.manual_seed(11)
        layer.zero_grad(set_to_none=True)
        nll = _nll(layer(x), cplx_)
        kl = sum(penalties(layer))
        if m == "one":
            (nll + c * kl).backward()
        elif m == "nll_first":
            nll.backward()
            (c * kl).backward()
        else:
            (c * kl).backward(retain_graph=True)
            nll.backward()
        return _grads(layer)

    run("one")                      # arms the fusion
    ref = run("one")
    assert layer._kl_fuse
    hook = _CopyingBuckets(list(layer.parameters()))
    ops.dp_hook = hook
    try:
        got = run(mode)
    finally:
        ops.dp_hook = None
        hook.remove()
    assert got.keys() == ref.keys()
    for n in ref:
        r = ref[n].float().cpu().numpy()
        np.testing.assert_allclose(got[n].float().cpu().numpy(), r, rtol=2e-5, atol=2e-5 * np.abs(r).max(),
                                   err_msg=f"{kind} {mode} {n}")
        # the bucket slice holds the same total (it is what the all-reduce would average)
        p = dict(layer.named_parameters())[n]
        np.testing.assert_allclose(hook.store[id(p)].cpu().numpy(), r, rtol=2e-5, atol=2e-5 * np.abs(r).max(),
                                   err_msg=f"bucket of {n}")


# ---- bench.py with two ranks AND a graph replay (gloo ranks sharing this GPU) ------------------------
def test_bench_py_two_gloo_ranks_graph_replay():
    """The N > 1 path of bench.py with the step's compute replayed from a hipGraph: gloo collectives synchronise on the
    host and cannot be captured, so the exchange (bucket all-reduces in sync_gradients(), scalar KL all-reduce) runs
    behind each replay; same KL as the eager two-rank run from the same seeds' first step is not expected (fresh noise
    per step), finite and sane numbers are."""
    import os
    import subprocess
    import sys
    from test_gpu_r03 import _bench_line
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "3", "--warmup", "1", "--batch", "512", "--no-cpu-baseline"]
    lines = {}
    for graph, port in (("on", "29561"), ("off", "29563")):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(root, "bench.py"),
                            "--gpus", "2", "--backend", "gloo", "--share-device", "--graph", graph] + common,
                           env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        lines[graph] = _bench_line(r.stdout)
    on, off = lines["on"], lines["off"]
    assert "hipGraph replay of the compute" in on["config"]["launch"] and off["config"]["launch"] == "eager"
    for line in (on, off):
        assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 1024 and line["steps"] == 3
        assert np.isfinite(line["value"]) and line["value"] > 0 and np.isfinite(line["kl"])
    # the KL term is a function of the (identical, rank-0 broadcast) weights only: both runs report the same value
    assert abs(on["kl"] - off["kl"]) <= 1e-4 * abs(off["kl"])


# ---- ADVICE r03 (dp.py, performance note): a copied neighbour must not push in-place slices off the early path ------------
def test_mixed_bucket_is_exchanged_from_the_side_stream():
    """tests/dp_mixed_bucket_check.py: dense layer + BatchNorm + LRT layer in ONE bucket under RCCL (world of one,
    collectives forced): launched behind the per-slice stream positions, gradients equal to the run without the wrapper."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(here, "dp_mixed_bucket_check.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "mixed bucket OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


# ---- the batch-norm forward moments in the convolution epilogue ---------------------------------------
@pytest.mark.parametrize("B,C,H,W,pad,Nc", [(3, 64, 50, 70, 0, 64), (2, 64, 33, 37, 1, 64), (2, 32, 16, 32, 1, 64),
                                            (5, 64, 18, 34, 0, 64), (300, 64, 20, 40, 1, 64), (70, 32, 30, 60, 1, 128),
                                            (40, 64, 34, 66, 0, 256), (1, 32, 16, 32, 1, 128)])
def test_conv_epilogue_moments_match_a_pass_over_the_output(B, C, H, W, pad, Nc):
    """cplxamd_conv2d_cl2_mom: same output bits as cplxamd_conv2d_cl2, and per-workgroup partial rows whose column sums are
    the five batch-norm moments of the stored bf16 output (edge tiles, images smaller than a tile row, more tiles than
    CUs: the persistent loop carries the sums through several tiles; 128 / 256 output channels: every workgroup stays on
    one column tile)."""
    from cplxmodule_amd import conv, ops
    bf = torch.bfloat16
    cl = torch.channels_last
    g = torch.Generator(device=DEV).manual_seed(B * 1000 + H)
    mk = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
    xr, xi = (mk(B, C, H, W).to(bf).contiguous(memory_format=cl) for _ in range(2))
    xr, xi = xr + 0.5, xi - 0.25                                  # (non-zero means: the raw second moments carry them)
    wr, wi = (mk(Nc, C, 3, 3).mul(0.05).to(bf) for _ in range(2))
    br, bi = mk(Nc), mk(Nc)
    geom, _ = conv._geom(xr.shape, wr.shape, (1, 1), (pad, pad), (1, 1), 1)
    y0r, y0i = conv.cl_conv(xr, xi, wr, wi, br, bi, geom)
    yr, yi = conv.cl_conv(xr, xi, wr, wi, br, bi, geom, moments=True)
    hint = ops.moments_hint(yr, yi)
    assert hint is not None and ops.moments_hint(y0r, y0i) is None
    assert torch.equal(yr, y0r) and torch.equal(yi, y0i)
    partials, chunks = hint
    got = partials.view(chunks, Nc, 5).sum(0).cpu().numpy()
    r, i = yr.double().permute(1, 0, 2, 3).reshape(Nc, -1), yi.double().permute(1, 0, 2, 3).reshape(Nc, -1)
    ref = torch.stack([r.sum(1), i.sum(1), (r * r).sum(1), (i * i).sum(1), (r * i).sum(1)], 1).cpu().numpy()
    scale = np.abs(ref).max(0, keepdims=True)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6 * scale.max())
    np.testing.assert_allclose(got / scale, ref / scale, rtol=0, atol=1e-5)
    # a modified output no longer carries them
    yr.add_(1)
    assert ops.moments_hint(yr, yi) is None


def test_conv_batchnorm_pair_uses_the_epilogue_moments():
    """CplxConv2d -> CplxBatchNorm2d in training mode: the first step arms the convolution (the batch-norm layer finds the
    producer's tag on its input), from the second step on the layer's statistics come out of the convolution's epilogue;
    outputs, running statistics and gradients agree with the run in which the layer makes its own moment pass."""
    from cplxmodule_amd import Cplx, conv, ops
    from cplxmodule_amd.nn.modules.batchnorm import CplxBatchNorm2d
    from cplxmodule_amd.nn.modules.conv import CplxConv2d
    bf = torch.bfloat16
    torch.manual_seed(4)
    net = torch.nn.Sequential(CplxConv2d(64, 64, 3, padding=1), CplxBatchNorm2d(64)).to(DEV)
    x = Cplx(torch.randn(6, 64, 40, 72, device=DEV).to(bf).contiguous(memory_format=torch.channels_last),
             torch.randn(6, 64, 40, 72, device=DEV).to(bf).contiguous(memory_format=torch.channels_last))
    gy = torch.randn(6, 64, 40, 72, device=DEV).to(bf).contiguous(memory_format=torch.channels_last)
    state0 = {k: v.clone() for k, v in net.state_dict().items()}
    seen = []
    handle = net[0].register_forward_hook(lambda m, a, out: seen.append(ops.moments_hint(out.real, out.imag) is not None))

    def run(steps):
        net.load_state_dict(state0)
        net.train()
        for _ in range(steps):
            net.zero_grad(set_to_none=True)
            y = net(x)
            torch.autograd.backward((y.real, y.imag), (gy, gy))
        return (y.real.detach().float(), y.imag.detach().float(), {k: v.clone() for k, v in net.state_dict().items()},
                {n: p.grad.float().clone() for n, p in net.named_parameters()})

    try:
        conv._MOMENTS_WANTED.clear()
        fused = run(3)
        assert seen == [False, True, True], seen
        seen.clear()
        conv._MOMENTS = False
        conv._MOMENTS_WANTED.clear()
        plain = run(3)
        assert seen == [False, False, False], seen
    finally:
        conv._MOMENTS = True
        handle.remove()
    for a, b in zip(fused[:2], plain[:2]):
        assert float((a - b).abs().max()) <= 2 ** -6 * float(b.abs().max())          # a bf16 ulp of the largest entries
        assert float((a != b).float().mean()) < 1e-3
    for k in plain[2]:
        np.testing.assert_allclose(N(fused[2][k].float()), N(plain[2][k].float()), rtol=2e-5, atol=2e-6, err_msg=k)
    for k in plain[3]:
        r = N(plain[3][k])
        if k.startswith("0.bias"):
            # the bias of a convolution in front of a batch-norm layer has a zero gradient in exact arithmetic.  The plain
            # run reports the sum of the bf16 rounding of dX over 17 280 pixels (the layer's own apply pass sums what it
            # stores); the armed pair takes the round-6 path in which the apply runs inside the weight-gradient launch and
            # the sum comes from the reduce pass analytically, E sum(g) - N k = 0 up to float64 rounding
            # (tests/test_gpu_bn_fold.py): nothing to compare but the size of the noise
            assert np.abs(N(fused[3][k])).max() <= 1.5 * max(np.abs(r).max(), 1e-6), k
            continue
        np.testing.assert_allclose(N(fused[3][k]), r, rtol=0, atol=2e-3 * np.abs(r).max(), err_msg=k)
    # evaluation mode takes the request back
    net.eval()
    net(x)
    assert not conv._MOMENTS_WANTED


# ---- the boundary is a C ABI -- a C++ program with no Python / torch in it -----------------------------------
def test_cabi_from_a_plain_cpp_program(tmp_path):
    """examples/cabi_linear.cpp: includes include/cplxamd.h, links libcplxamd.so, allocates its own device buffers, runs the
    complex linear map with float32 and with bf16 operands and checks both against host loops."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "cabi_linear")
    lib = os.path.join(root, "cplxmodule_amd")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-I", os.path.join(root, "include"),
                        os.path.join(root, "examples", "cabi_linear.cpp"), "-L", lib, "-lcplxamd", f"-Wl,-rpath,{lib}", "-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    deps = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libcplxamd.so" in deps and "torch" not in deps and "python" not in deps, deps
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "cabi_linear OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_conv_batchnorm_moments_inside_a_graph_replay():
    """The conv -> batch-norm moments path captured in a hipGraph (128 output channels: two column tiles, so the
    launcher's memset of the partial rows is a graph node too): the warm-up steps arm the convolution, the capture records
    the moments variant, and replays reproduce the eager step -- outputs, gradients, running statistics advancing."""
    from cplxmodule_amd import Cplx, conv, ops
    from cplxmodule_amd.nn.modules.batchnorm import CplxBatchNorm2d
    from cplxmodule_amd.nn.modules.conv import CplxConv2d
    from cplxmodule_amd.utils.graphs import GraphedStep
    bf = torch.bfloat16
    cl = torch.channels_last
    torch.manual_seed(9)
    net = torch.nn.Sequential(CplxConv2d(64, 128, 3, padding=1), CplxBatchNorm2d(128)).to(DEV)   # (5.7 GFLOP: channels-last path)
    x = Cplx(torch.randn(8, 64, 48, 64, device=DEV).to(bf).contiguous(memory_format=cl),
             torch.randn(8, 64, 48, 64, device=DEV).to(bf).contiguous(memory_format=cl))
    gy = torch.randn(8, 128, 48, 64, device=DEV).to(bf).contiguous(memory_format=cl)
    hinted = []
    handle = net[0].register_forward_hook(lambda m, a, out: hinted.append(ops.moments_hint(out.real, out.imag) is not None))

    def step():
        net.zero_grad(set_to_none=True)
        y = net(x)
        torch.autograd.backward((y.real, y.imag), (gy, gy))
        return y.real, y.imag

    try:
        conv._MOMENTS_WANTED.clear()
        net.train()
        g = GraphedStep(step, modules=[net], warmup=2)
        assert hinted == [False, True, True], hinted          # two warm-up steps, then the capture
        state = {k: v.clone() for k, v in net.state_dict().items()}
        yr, yi = g.replay()
        torch.cuda.synchronize()
        got = (yr.clone(), yi.clone(), {n: p.grad.clone() for n, p in net.named_parameters()},
               {k: v.clone() for k, v in net.state_dict().items()})
        net.load_state_dict(state)                             # the same step, eager, from the same running statistics
        e_yr, e_yi = step()
        assert torch.equal(e_yr, got[0]) and torch.equal(e_yi, got[1])
        for n, p in net.named_parameters():
            assert torch.equal(p.grad, got[2][n]), n
        for k, v in net.state_dict().items():
            assert torch.equal(v, got[3][k]), k
        assert not torch.equal(state["1.running_mean"], got[3]["1.running_mean"])
    finally:
        handle.remove()
