"""Round-3 GPU tests: regressions for the round-2 advisor findings (two-call backward pattern with the fused KL,
gradient buffers aliasing the pending KL gradients under a data-parallel hook, log_alpha gradient at exact zeros)
and the round-3 kernels / entry points.  Everything goes through libcplxamd.so (C ABI via ctypes)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _layers():
    from cplxmodule_amd.nn import relevance as rel
    return {"cplx_vd": (rel.CplxLinearVD, True), "cplx_ard": (rel.CplxLinearARD, True),
            "real_vd": (rel.LinearVD, False), "real_ard": (rel.LinearARD, False)}


def _make(kind, I=96, O=80, dtype=torch.float32, seed=3):
    from cplxmodule_amd import Cplx
    from cplxmodule_amd.nn.relevance.noise import noise
    cls, cplx_ = _layers()[kind]
    torch.manual_seed(seed)
    layer = cls(I, O).to("cuda")
    with torch.no_grad():
        layer.log_sigma2.uniform_(-6.0, 1.0)
    noise.manual_seed(11)
    B = 64
    if cplx_:
        x = Cplx(torch.randn(B, I, device="cuda").to(dtype), torch.randn(B, I, device="cuda").to(dtype))
    else:
        x = torch.randn(B, I, device="cuda").to(dtype)
    return layer, x, cplx_


def _nll(y, cplx_):
    return ((y.real.float() ** 2).sum() + (y.imag.float() ** 2).sum()) if cplx_ else (y.float() ** 2).sum()


def _grads(layer):
    return {n: p.grad.detach().clone() for n, p in layer.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("kind", ["cplx_vd", "cplx_ard", "real_vd", "real_ard"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_two_call_backward_with_fused_kl(kind, dtype):
    """`nll.backward(); (c * sum(penalties(model))).backward()` -- the reference's two-call pattern -- keeps working
    once the KL rides in the layer's forward node ('backward through the graph a second time'), and gives
    the gradients of the single combined backward; the reverse order works with retain_graph=True on the KL pass and
    says so otherwise."""
    from cplxmodule_amd.nn.relevance import penalties
    from cplxmodule_amd.nn.relevance.noise import noise
    layer, x, cplx_ = _make(kind, dtype=dtype)
    c = 0.37

    def run(mode):
        noise.manual_seed(11)
        layer.zero_grad(set_to_none=True)
        nll = _nll(layer(x), cplx_)
        kl = sum(penalties(layer))
        if mode == "one":
            (nll + c * kl).backward()
        elif mode == "nll_first":
            nll.backward()
            (c * kl).backward()
        else:
            # the KL alone first: it runs through the layer's node, so the graph must be retained for the data term
            (c * kl).backward(retain_graph=True)
            nll.backward()
        return _grads(layer)

    run("one")                      # arms the fusion
    ref = run("one")
    assert layer._kl_fuse, "the fused path is what this test is about"
    for mode in ("nll_first", "kl_first", "one"):
        got = run(mode)
        assert got.keys() == ref.keys()
        for n in ref:
            r = ref[n].float().cpu().numpy()
            np.testing.assert_allclose(got[n].float().cpu().numpy(), r, rtol=2e-5, atol=2e-5 * np.abs(r).max(),
                                       err_msg=f"{kind} {mode} {n}")
    noise.manual_seed(11)
    layer.zero_grad(set_to_none=True)
    nll, kl = _nll(layer(x), cplx_), sum(penalties(layer))
    kl.backward()
    with pytest.raises(RuntimeError, match="retain_graph=True to the KL backward"):
        nll.backward()


class _FakeBuckets:
    """What dp.BucketHook offers ops: persistent float32 storage per parameter + an announce call."""

    def __init__(self, params):
        self.store = {id(p): torch.zeros(p.shape, dtype=torch.float32, device=p.device) for p in params}
        self.announced = []

    def view_for(self, p):
        t = self.store.get(id(p))
        return None if t is None else t.view(t.shape)

    def early_ready(self, *params):
        self.announced += [p for p in params if p is not None]


@pytest.mark.parametrize("kind", ["cplx_vd", "real_ard"])
@pytest.mark.parametrize("frozen", ["log_sigma2", "weight"])
def test_kl_gradients_do_not_alias_data_gradients_under_dp_hook(kind, frozen):
    """With a data-parallel hook the pending KL gradients and the data-gradient outputs are views of the SAME bucket
    slice.  When not every gradient is wanted (frozen log_sigma2 / frozen weight) the backward used to overwrite the
    KL gradient with the data gradient and then add it to itself scaled (ops.py:671)."""
    from cplxmodule_amd import ops
    from cplxmodule_amd.nn.relevance import penalties
    from cplxmodule_amd.nn.relevance.noise import noise
    layer, x, cplx_ = _make(kind)
    for n, p in layer.named_parameters():
        if n.startswith(frozen):
            p.requires_grad_(False)
    c = 0.81

    def run():
        noise.manual_seed(5)
        layer.zero_grad(set_to_none=True)
        (_nll(layer(x), cplx_) + c * sum(penalties(layer))).backward()
        return _grads(layer)

    run()
    ref = run()
    assert layer._kl_fuse and ref
    ops.dp_hook = _FakeBuckets([p for p in layer.parameters() if p.requires_grad])
    try:
        got = run()
        assert ops.dp_hook.announced
    finally:
        ops.dp_hook = None
    for n in ref:
        r = ref[n].cpu().numpy()
        np.testing.assert_allclose(got[n].cpu().numpy(), r, rtol=2e-5, atol=2e-5 * np.abs(r).max(), err_msg=n)


@pytest.mark.parametrize("kind", ["real_vd", "cplx_vd"])
def test_log_alpha_and_penalty_gradient_zero_at_zero_weight(kind):
    """abs() / norm subgradient: an exactly-zero weight gets gradient 0 from log_alpha and from the penalty,
    under grad mode too (+-2e12 g for the real kernels)."""
    layer, _, cplx_ = _make(kind)
    with torch.no_grad():
        ws = (layer.weight.real, layer.weight.imag) if cplx_ else (layer.weight,)
        for w in ws:
            w[::3, ::5] = 0.0
    zero = (ws[0] == 0) if not cplx_ else ((ws[0] == 0) & (ws[1] == 0))
    assert int(zero.sum()) > 50
    for f in (lambda: layer.log_alpha, lambda: layer.penalty):
        layer.zero_grad(set_to_none=True)
        out = f()
        g = torch.randn_like(out)
        finite = torch.isfinite(out)
        (out[finite] * g[finite]).sum().backward()
        for w in ws:
            assert torch.isfinite(w.grad).all()
            assert float(w.grad[zero].abs().max()) == 0.0


def _bench_line(stdout):
    import json
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_bench_py_runs_with_two_ranks_and_with_rccl():
    """bench.py itself with N > 1 (the first SCALE run must not be the first run of that code): two gloo
    ranks sharing this GPU (init, barrier, bucket exchange, scalar KL all-reduce, MAX-reduce of the elapsed time,
    the JSON line), and one RCCL rank with every collective of the N > 1 path forced on."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "2", "--warmup", "1", "--batch", "512", "--no-cpu-baseline"]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(root, "bench.py"),
                        "--gpus", "2", "--backend", "gloo", "--share-device"] + common,
                       env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = _bench_line(r.stdout)
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 1024 and line["config"]["parallelism"] == "dp2"
    assert line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert np.isfinite(line["value"]) and line["value"] > 0 and np.isfinite(line["kl"])
    assert abs(line["value"] - 2 * 512 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-3 * line["value"]
    assert "cpu_baseline" not in line and line["roofline"]["avg_launch_ms"] > 0
    env1 = dict(env, MASTER_PORT="29549")
    env1.pop("RANK", None); env1.pop("WORLD_SIZE", None); env1.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--backend", "nccl",
                        "--force-collectives"] + common, env=env1, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = _bench_line(r.stdout)
    assert line["n_gpus"] == 1 and line["config"]["global_batch"] == 512 and np.isfinite(line["value"])


@pytest.mark.parametrize("layer_kind", ["cplx", "cplx_vd"])
def test_conv_kernels_one_workgroup_per_tile_switch(layer_kind):
    """`cplxamd_gemm_set_persistent(0)` -- what the data-parallel hook selects while RCCL collectives are in flight --
    also makes the channels-last convolution kernels launch one workgroup per tile (forward, data gradient) / twice
    as many split slabs (weight gradient): forward and data gradient bit-identical, weight gradient equal up to the
    float32 order of the slab sums.  More tiles than CUs, so the two forms really differ."""
    from cplxmodule_amd import Cplx, _lib, nn
    from cplxmodule_amd.nn import relevance as rel
    from cplxmodule_amd.nn.relevance.noise import noise
    lib = _lib.load()
    torch.manual_seed(0)
    layer = (nn.CplxConv2d(64, 64, 3, padding=1) if layer_kind == "cplx" else rel.CplxConv2dVD(64, 64, 3, padding=1)).to("cuda")
    mk = lambda: (torch.randn(8, 64, 96, 128, device="cuda").bfloat16()  # noqa: E731
                  .contiguous(memory_format=torch.channels_last).requires_grad_(True))
    x = Cplx(mk(), mk())                      # 8 x 6 x 4 = 192 patch tiles... x column tiles: see the assert below
    g = (torch.randn(8, 64, 96, 128, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last),
         torch.randn(8, 64, 96, 128, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last))

    def run():
        noise.manual_seed(3)
        layer.zero_grad(set_to_none=True)
        x.real.grad = x.imag.grad = None
        y = layer(x)
        torch.autograd.backward((y.real, y.imag), g)
        grads = {n: p.grad.clone() for n, p in layer.named_parameters()}
        return y.real.detach().clone(), y.imag.detach().clone(), x.real.grad.clone(), x.imag.grad.clone(), grads

    try:
        assert lib.cplxamd_gemm_set_persistent(1) == 1
        a = run()
        assert lib.cplxamd_gemm_set_persistent(0) == 1
        b = run()
    finally:
        lib.cplxamd_gemm_set_persistent(1)
    for u, v in zip(a[:4], b[:4]):
        assert torch.equal(u, v)
    for n in a[4]:
        r = a[4][n].float().cpu().numpy()
        np.testing.assert_allclose(b[4][n].float().cpu().numpy(), r, rtol=1e-5, atol=1e-5 * np.abs(r).max(), err_msg=n)


def test_graphed_step_after_eager_steps_matches_eager():
    """utils.graphs.GraphedStep: capture AFTER eager steps on the default stream (their autograd state -- the cached fused
    KL holds the previous graph, whose AccumulateGrad nodes belong to the default stream -- used to make the capture
    crash in hipStreamEndCapture); every replay draws fresh noise and equals the eager step from the same position."""
    from cplxmodule_amd import Cplx
    from cplxmodule_amd.nn import relevance as rel
    from cplxmodule_amd.nn.relevance.noise import noise
    from cplxmodule_amd.utils.graphs import GraphedStep
    torch.manual_seed(5)
    layer = rel.CplxLinearVD(256, 192).to("cuda")
    x = Cplx(torch.randn(512, 256, device="cuda").bfloat16().requires_grad_(True),
             torch.randn(512, 256, device="cuda").bfloat16().requires_grad_(True))
    klw = torch.tensor(1e-2, device="cuda")

    def step():
        layer.zero_grad(set_to_none=True)
        x.real.grad = x.imag.grad = None
        y = layer(x)
        kl = sum(rel.penalties(layer))
        torch.autograd.backward((y.real, y.imag, kl), (y.real.detach() * 2, y.imag.detach() * 2, klw))
        return y.real, kl

    noise.manual_seed(21)
    try:
        noise.set_mode("philox-device")
        for _ in range(3):
            out = step()                     # eager, default stream; `out` and the layer's KL cache keep the graph alive
        del out
        g = GraphedStep(step, modules=[layer], warmup=2)
        state = noise.device_state(torch.device("cuda"))
        seen = []
        for _ in range(3):
            pos = int(state[1].item())
            yr, kl = g.replay()
            torch.cuda.synchronize()
            seen.append((pos, yr.clone(), float(kl), layer.log_sigma2.grad.clone(), x.real.grad.clone()))
        assert [p for p, *_ in seen] == [seen[0][0] + k for k in range(3)]
        assert not torch.equal(seen[0][1], seen[1][1])
        noise.set_mode("philox")
        for pos, yr, kl, gls2, gx in seen:
            noise.counter = pos - 1
            e_yr, e_kl = step()
            assert torch.equal(e_yr, yr) and float(e_kl) == kl
            assert torch.equal(layer.log_sigma2.grad, gls2) and torch.equal(x.real.grad, gx)
    finally:
        noise.set_mode("philox")


def test_cfg5_data_parallel_step_as_one_graph_rccl():
    """BASELINE configs[4]'s train step (conv + BN + ReLU stack, ARD head, loss + KL, backward, bucket all-reduce over
    RCCL -- world of one, collectives forced --, Adam) captured in ONE hipGraph and replayed."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29551", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "r03", "cfg5_graph.py"), "--rccl1", "--batch", "64",
                        "--width", "8"], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    m = re.search(r"rccl1 batch 64 width 8: eager ms/step \[([^\]]*)\]  graph replay \[([^\]]*)\]  buckets (\d+)  loss ([\d.]+) -> ([\d.]+)",
                  r.stdout)
    assert m, r.stdout[-2000:]
    eager = [float(v) for v in m.group(1).split(",")]
    graph = [float(v) for v in m.group(2).split(",")]
    assert int(m.group(3)) >= 1 and np.isfinite(float(m.group(5)))
    assert min(graph) < min(eager), (eager, graph)          # launch-bound model: the replay must be the faster form


@pytest.mark.parametrize("O", [4096, 4064, 4128])          # contraction over O: (O / 32) % 3 = 2, 1, 0 -> all three ring phases
def test_fused_lrt_input_gradient_is_bit_identical(O):
    """cplxamd_cgemm_lrt_dx: dX = G conj(W) + 2 X (*) ga in the epilogue of the complex (N,T) kernels (persistent, or one
    workgroup per tile for partial tiles / the data-parallel form) == cplxamd_cgemm followed by cplxamd_lrt_dx_accum, bit
    for bit; launches neither epilogue can take are declined (CPLXAMD_ESHAPE), never computed without the term."""
    from cplxmodule_amd import _lib, ops
    from cplxmodule_amd._lib import BF16, ptr, stream_ptr, try_call
    dev, bf = "cuda", torch.bfloat16
    B, I = 8192, 2048                                        # 32 x 16 = 512 tiles of 256 x 128: two rounds on 256 CUs
    torch.manual_seed(O)
    gr, gi = (torch.randn(B, O, device=dev).to(bf) for _ in range(2))
    wr, wi = (torch.randn(O, I, device=dev).mul(0.02).to(bf) for _ in range(2))
    xr, xi = (torch.randn(B, I, device=dev).to(bf) for _ in range(2))
    ga = torch.randn(B, I, device=dev).mul(0.3).to(bf)
    dxr, dxi = torch.empty(B, I, device=dev, dtype=bf), torch.empty(B, I, device=dev, dtype=bf)
    assert try_call("cplxamd_cgemm_lrt_dx", ptr(gr), ptr(gi), O, 1, ptr(wr), ptr(wi), 1, I, ptr(xr), ptr(xi), ptr(ga), I,
                    ptr(dxr), ptr(dxi), I, B, I, O, BF16, stream_ptr()), "the persistent kernel must take this shape"
    rr, ri = ops._cplx_linear_dx(gr, gi, wr, wi, bf)
    ops.lrt_dx_accum(rr, ri, xr, xi, ga)
    assert torch.equal(dxr, rr) and torch.equal(dxi, ri)
    # against float64 on sampled rows (one bf16 rounding of the GEMM result + one of the sum)
    rows = torch.randint(0, B, (8,), device=dev)
    G = (gr[rows].double() + 1j * gi[rows].double()).cpu().numpy()
    W = (wr.double() + 1j * wi.double()).cpu().numpy()
    ref = G @ W.conj() + 2 * (xr[rows].double() + 1j * xi[rows].double()).cpu().numpy() * ga[rows].double().cpu().numpy()
    got = (dxr[rows].double() + 1j * dxi[rows].double()).cpu().numpy()
    assert np.abs(got - ref).max() <= 1.2e-2 * np.abs(ref).max()
    # partial tiles and the data-parallel form (one workgroup per tile, cplxamd_gemm_set_persistent(0)): the one-tile
    # kernel carries the term in its staged epilogue -- same bits
    sr, si = torch.empty(B - 8, I, device=dev, dtype=bf), torch.empty(B - 8, I, device=dev, dtype=bf)
    assert try_call("cplxamd_cgemm_lrt_dx", ptr(gr), ptr(gi), O, 1, ptr(wr), ptr(wi), 1, I, ptr(xr), ptr(xi), ptr(ga), I,
                    ptr(sr), ptr(si), I, B - 8, I, O, BF16, stream_ptr())
    assert torch.equal(sr, rr[:B - 8]) and torch.equal(si, ri[:B - 8])
    lib = _lib.load()
    try:
        lib.cplxamd_gemm_set_persistent(0)
        dxr.zero_(); dxi.zero_()
        assert try_call("cplxamd_cgemm_lrt_dx", ptr(gr), ptr(gi), O, 1, ptr(wr), ptr(wi), 1, I, ptr(xr), ptr(xi), ptr(ga),
                        I, ptr(dxr), ptr(dxi), I, B, I, O, BF16, stream_ptr())
        assert torch.equal(dxr, rr) and torch.equal(dxi, ri)
        a, b = ops._cplx_lrt_dx(gr, gi, wr, wi, xr, xi, ga)
        assert torch.equal(a, rr) and torch.equal(b, ri)
    finally:
        lib.cplxamd_gemm_set_persistent(1)
    # declined, not mis-computed: a row pitch the 16-byte epilogue accesses cannot take
    odd = torch.empty(B, I + 4, device=dev, dtype=bf)
    assert not try_call("cplxamd_cgemm_lrt_dx", ptr(gr), ptr(gi), O, 1, ptr(wr), ptr(wi), 1, I, ptr(xr), ptr(xi), ptr(ga), I,
                        ptr(odd), ptr(odd), I + 4, B, I, O, BF16, stream_ptr())


def test_bucket_all_reduce_runs_beside_the_gemms():
    """tests/dp_overlap_check.py: an all_reduce issued as dp.BucketHook issues it (high-priority side stream behind a
    mid-stream event, RCCL's stream high-priority via dp.init_process_group) overlaps the GEMMs queued after the event;
    with equal-priority streams it ran after them (round-3 finding, profiles/r03_dp_timeline.txt)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TORCH_NCCL_HIGH_PRIORITY", None)
    r = subprocess.run([sys.executable, os.path.join(here, "dp_overlap_check.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "overlap OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 64, 64, 64, 64, 1), (3, 64, 128, 50, 70, 1), (2, 128, 64, 33, 37, 0), (5, 64, 96, 16, 32, 1)])
def test_fused_lrt_conv_input_gradient_is_bit_identical(shape):
    """cplxamd_conv2d_cl2_lrt_dx: dx = dgrad(g; w) + 2 x (*) ga in the epilogue of the 2-d-patch data-gradient kernel ==
    cplxamd_conv2d_cl2 (mode 1) followed by cplxamd_lrt_dx_accum, bit for bit -- full tiles, ragged right / bottom
    edges (pixels beyond the image go to the dump buffer), `valid` padding; and through CplxConv2dVD's backward."""
    from cplxmodule_amd import Cplx, conv, ops
    from cplxmodule_amd.nn import relevance as rel
    from cplxmodule_amd.nn.relevance.noise import noise
    B, Ci, Co, H, W, pad = shape
    dev, bf = "cuda", torch.bfloat16
    torch.manual_seed(H * W)
    cl = lambda *s, k=1.0: torch.randn(*s, device=dev).mul(k).to(bf).contiguous(memory_format=torch.channels_last)  # noqa: E731
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    gr, gi = cl(B, Co, Ho, Wo), cl(B, Co, Ho, Wo)
    xr, xi, ga = cl(B, Ci, H, W), cl(B, Ci, H, W), cl(B, Ci, H, W, k=0.3)
    wr, wi = (torch.randn(Co, Ci, 3, 3, device=dev).mul(0.05).to(bf) for _ in range(2))
    geom = (B, Ci, Co, H, W, 3, 3, 1, 1, pad, pad, 1, 1, 1)
    assert conv._LRT_DX_FUSE and conv._CL_PATCH
    dxr, dxi = conv.cl_conv_lrt_dx(gr, gi, wr, wi, geom, xr, xi, ga)
    rr, ri = conv.cl_conv(gr, gi, wr, wi, None, None, geom, dgrad=True)
    plain_r, plain_i = rr.clone(), ri.clone()
    ops.lrt_dx_accum(rr, ri, xr, xi, ga)
    assert torch.equal(dxr, rr) and torch.equal(dxi, ri)
    assert not torch.equal(dxr, plain_r)                     # (the elementwise term is there)
    # the whole layer: fused and two-launch backward give the same input gradient
    torch.manual_seed(1)
    layer = rel.CplxConv2dVD(Ci, Co, 3, padding=pad).to(dev)
    x = Cplx(xr.clone().requires_grad_(True), xi.clone().requires_grad_(True))

    def grads(fuse):
        conv._LRT_DX_FUSE = fuse
        try:
            x.real.grad = x.imag.grad = None
            noise.manual_seed(11)
            y = layer(x)
            torch.autograd.backward((y.real, y.imag), (gr, gi))
            return x.real.grad.clone(), x.imag.grad.clone()
        finally:
            conv._LRT_DX_FUSE = True
    old = conv._CL_FORCE
    conv._CL_FORCE = True
    try:
        if conv._cl_layer_ok(geom, xr):
            a, b = grads(True), grads(False)
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        else:
            assert Ci % 64 or Co % 64                        # (the weight-gradient kernel wants 64-channel multiples)
    finally:
        conv._CL_FORCE = old


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_split_relu_one_launch_matches_torch(dtype):
    """CplxToCplx[torch.nn.ReLU] on device tensors = ONE launch for both planes each way (cplxamd_split_relu), with
    torch's semantics: NaN passes, the backward masks on the output; in-place modules keep torch's kernels."""
    from cplxmodule_amd import Cplx, nn
    dev = "cuda"
    torch.manual_seed(3)
    xr = torch.randn(5, 7, 9, 11, device=dev).to(dtype)
    xi = torch.randn(5, 7, 9, 11, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    xr.view(-1)[:4] = torch.tensor([float("nan"), -0.0, 0.0, float("inf")], device=dev, dtype=dtype)
    xr.requires_grad_(True); xi.requires_grad_(True)
    gr, gi = torch.randn_like(xr), torch.randn_like(xi)
    act = nn.CplxToCplx[torch.nn.ReLU]()
    y = act(Cplx(xr, xi))
    torch.autograd.backward((y.real, y.imag), (gr, gi))
    got = (y.real.detach(), y.imag.detach(), xr.grad.clone(), xi.grad.clone())
    xr.grad = xi.grad = None
    rr, ri = torch.relu(xr), torch.relu(xi)
    torch.autograd.backward((rr, ri), (gr, gi))
    for a, b in zip(got, (rr.detach(), ri.detach(), xr.grad, xi.grad)):
        assert torch.equal(torch.nan_to_num(a.float(), nan=7.0), torch.nan_to_num(b.float(), nan=7.0))
    assert type(y.real.grad_fn).__name__.startswith("SplitReluFn")
    z = nn.CplxToCplx[torch.nn.ReLU](inplace=True)(Cplx(xr.detach().clone(), xi.detach().clone()))
    assert torch.equal(torch.nan_to_num(z.real.float(), nan=7.0), torch.nan_to_num(rr.detach().float(), nan=7.0))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(256, 8, 28, 28), (3, 5, 7, 9), (70, 33, 1, 13), (1, 1, 1, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_complex_bias_gradient_both_planes_per_launch(shape, dtype):
    """cplxamd_chansum2 (both planes in each of its two launches) == two cplxamd_chansum calls, bit for bit."""
    from cplxmodule_amd import conv
    dev = "cuda"
    torch.manual_seed(sum(shape))
    gr, gi = (torch.randn(*shape, device=dev).to(dtype) for _ in range(2))
    a, b = conv.chansum2(gr, gi)
    assert torch.equal(a, conv.chansum(gr)) and torch.equal(b, conv.chansum(gi))
    ref = gr.double().sum(dim=(0, 2, 3))
    assert (a.double() - ref).abs().max() <= 1e-6 * gr.double().abs().sum(dim=(0, 2, 3)).max() + 1e-30


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(32, 8, 8, 28, 28, 2, 1, True),      # N = 72: the sums ride as GEMM column 72
                                 (16, 64, 32, 9, 9, 1, 1, True),      # N = 576 = 9 x 64: own tile -> the library sums apart
                                 (8, 6, 10, 11, 13, 1, 2, True),      # groups = 2
                                 (8, 5, 7, 12, 10, 1, 1, False)])     # real-valued
def test_bias_gradient_from_the_weight_gradient_gemm(cfg):
    """cplxamd_conv2d_wgrad_bias: the bias gradient (sums of G over batch and pixels) as one more column of the generic
    weight-gradient GEMM == float64 sums; the weight gradient itself is bit-identical to cplxamd_conv2d_wgrad's."""
    from cplxmodule_amd import conv
    B, Ci, Co, H, W, stride, groups, cplx = cfg
    dev = "cuda"
    torch.manual_seed(Ci * Co)
    x = [torch.randn(B, Ci, H, W, device=dev) for _ in range(2 if cplx else 1)]
    wshape = (Co, Ci // groups, 3, 3)
    geom, oshape = conv._geom(x[0].shape, wshape, (stride, stride), (1, 1), (1, 1), groups)
    g = [torch.randn(oshape, device=dev) for _ in range(2 if cplx else 1)]
    gi, xi = (g[1], x[1]) if cplx else (None, None)
    bsum = []
    dwr, dwi = conv.conv_wgrad(g[0], gi, x[0], xi, geom, wshape, bias_out=bsum)
    pr, pi = conv.conv_wgrad(g[0], gi, x[0], xi, geom, wshape)
    assert torch.equal(dwr, pr) and (not cplx or torch.equal(dwi, pi))
    assert len(bsum) == (2 if cplx else 1)
    for got, t in zip(bsum, g):
        ref = t.double().sum(dim=(0, 2, 3))
        assert (got.double() - ref).abs().max() <= 2e-6 * t.double().abs().sum(dim=(0, 2, 3)).max()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(5, 8, 8, 28, 28, 2, 1, 1),      # 4 x 4 x 1 path, one channel group (cfg5 layer 2)
                                 (4, 8, 16, 14, 14, 1, 1, 1),     # two groups forward / weight gradient, one for the data gradient
                                 (3, 12, 10, 9, 11, 2, 2, 1),     # ragged channel counts, stride 2, dilation 2
                                 (2, 24, 32, 7, 7, 1, 1, 2),      # groups = 2: 12 / 16 channels per group
                                 (70, 16, 32, 7, 7, 1, 1, 1),     # 32 x 32 tiles (few pixels), 32 rows: the 32 x 32 x 2 path
                                 (3, 16, 16, 13, 10, (2, 1), 1, 1),   # data gradient by stride phases: 2 x 1 classes
                                 (2, 40, 48, 9, 12, (1, 2), (1, 2), 1),   # ... 1 x 2 classes, dilated taps, 64-row tiles
                                 (2, 8, 12, 9, 11, 2, 1, 2),      # stride phases with groups = 2, odd image sizes
                                 (2, 6, 5, 8, 8, 3, 1, 1)])       # stride 3: the plain data-gradient path
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_generic_conv_narrow_layers_vs_float64(cfg, dtype):
    """The generic fp32 conv kernels on narrow layers (v_mfma_f32_4x4x1 for <= 16 output rows, 32 x 32 tiles with a 4-way K
    split for few pixels): forward, data gradient, weight and bias gradient against torch's float64 convolution."""
    from cplxmodule_amd import conv
    B, Ci, Co, H, W, stride, dil, groups = cfg
    dev = "cuda"
    torch.manual_seed(Ci + Co)
    xr, xi = (torch.randn(B, Ci, H, W, device=dev).to(dtype) for _ in range(2))
    wr, wi = ((torch.randn(Co, Ci // groups, 3, 3, device=dev) * 0.2).to(dtype) for _ in range(2))
    br, bi = torch.randn(Co, device=dev), torch.randn(Co, device=dev)
    stride, dil = conv._pair(stride), conv._pair(dil)
    geom, oshape = conv._geom(xr.shape, wr.shape, stride, dil, dil, groups)
    gr, gi = (torch.randn(oshape, device=dev).to(dtype) for _ in range(2))
    lib, P, st = conv._lib.load(), conv.ptr, conv.stream_ptr
    code = conv.dtype_code(xr)
    # straight through the C ABI: the generic kernels (the layer-level wrappers prefer the channels-last ones for bf16)
    yr, yi = torch.empty(oshape, device=dev, dtype=dtype), torch.empty(oshape, device=dev, dtype=dtype)
    conv.call("cplxamd_conv2d_fwd", P(xr), P(xi), P(wr), P(wi), P(br), P(bi), P(yr), P(yi), geom, code, st())
    dxr, dxi = torch.empty_like(xr), torch.empty_like(xi)
    conv.call("cplxamd_conv2d_dgrad", P(gr), P(gi), P(wr), P(wi), P(dxr), P(dxi), geom, code, st())
    ws = torch.empty(int(lib.cplxamd_conv2d_wgrad_ws_bytes(geom, 1)), dtype=torch.uint8, device=dev)
    dwr, dwi = (torch.empty(wr.shape, device=dev) for _ in range(2))
    db = torch.empty(2, Co, device=dev)
    conv.call("cplxamd_conv2d_wgrad_bias", P(gr), P(gi), P(xr), P(xi), None, P(dwr), P(dwi), P(db[0]), P(db[1]), geom, code,
              P(ws), ws.numel(), st())
    bsum = [db[0], db[1]]
    d = lambda t: t.double().cpu().requires_grad_(True)  # noqa: E731
    Xr, Xi, Wr, Wi = d(xr), d(xi), d(wr), d(wi)
    c = lambda x, w: torch.nn.functional.conv2d(x, w, None, stride, dil, dil, groups)  # noqa: E731
    Yr = c(Xr, Wr) - c(Xi, Wi) + br.double().cpu()[None, :, None, None]
    Yi = c(Xr, Wi) + c(Xi, Wr) + bi.double().cpu()[None, :, None, None]
    torch.autograd.backward((Yr, Yi), (gr.double().cpu(), gi.double().cpu()))
    for got, ref in ((yr, Yr.detach()), (yi, Yi.detach()), (dxr, Xr.grad), (dxi, Xi.grad), (dwr, Wr.grad), (dwi, Wi.grad),
                     (bsum[0], gr.double().cpu().sum((0, 2, 3))), (bsum[1], gi.double().cpu().sum((0, 2, 3)))):
        tol = 2e-6 * max(1.0, (B * oshape[2] * oshape[3]) ** 0.5 / 30)
        if dtype == torch.bfloat16 and got.dtype == torch.bfloat16:
            tol = 2.0 ** -8                                  # (float32 accumulation, outputs rounded to bf16)
        assert (got.double().cpu() - ref).abs().max() <= tol * ref.abs().max()


@pytest.mark.gpu
@pytest.mark.parametrize("div", [False, True])
@pytest.mark.parametrize("n", [(7, 33), (4, 16, 9, 9), (1,)])
def test_cplx_product_and_quotient_in_one_launch(div, n):
    """Cplx * Cplx and Cplx / Cplx on same-shape device tensors run cplxamd_cplx_mul: values bit-identical to the
    reference's chain of elementwise torch kernels (cplxmodule/cplx.py:135-165: every product / sum / quotient rounded on
    its own), gradients equal to autograd through that chain; broadcasting operands keep the chain."""
    from cplxmodule_amd import Cplx
    dev = "cuda"
    torch.manual_seed(len(n) + int(div))
    mk = lambda: torch.randn(*n, device=dev).requires_grad_(True)  # noqa: E731
    ar, ai, br, bi = mk(), mk(), mk(), mk()
    gr, gi = torch.randn(*n, device=dev), torch.randn(*n, device=dev)
    z = Cplx(ar, ai) / Cplx(br, bi) if div else Cplx(ar, ai) * Cplx(br, bi)
    assert type(z.real.grad_fn).__name__.startswith("CplxMulFn")
    torch.autograd.backward((z.real, z.imag), (gr, gi))
    got = [t.grad.clone() for t in (ar, ai, br, bi)]
    for t in (ar, ai, br, bi):
        t.grad = None
    if div:                                                  # the reference's lines, op for op
        den = br * br + bi * bi
        cr, ci = br / den, (-bi) / den
    else:
        cr, ci = br, bi
    rr, ri = ar * cr - ai * ci, ai * cr + ar * ci
    assert torch.equal(z.real.detach(), rr.detach()) and torch.equal(z.imag.detach(), ri.detach())
    torch.autograd.backward((rr, ri), (gr, gi))
    for a, t in zip(got, (ar, ai, br, bi)):
        assert (a - t.grad).abs().max() <= 4e-6 * t.grad.abs().max()
    if n[0] > 1:
        w = Cplx(ar, ai) * Cplx(br[:1], bi[:1])              # broadcast: torch's kernels
        assert not type(w.real.grad_fn).__name__.startswith("CplxMulFn")


@pytest.mark.gpu
@pytest.mark.parametrize("O", [4096, 4064, 4128])          # contraction over O: (O / 32) % 3 = 2, 1, 0 -> all three ring phases
def test_fused_real_lrt_input_gradient_is_bit_identical(O):
    """cplxamd_rgemm_lrt_dx (LinearVD / LinearARD): dX = G W + 2 X (*) ga in the epilogue of the persistent real (N,T)
    kernel == cplxamd_rgemm followed by cplxamd_lrt_dx_accum, bit for bit; through the layer's backward as well."""
    from cplxmodule_amd import ops
    from cplxmodule_amd._lib import BF16, ptr, stream_ptr, try_call
    dev, bf = "cuda", torch.bfloat16
    B, I = 8192, 4096                                        # 32 x 16 = 512 tiles of 256 x 256: two rounds on 256 CUs
    torch.manual_seed(O)
    g = torch.randn(B, O, device=dev).to(bf)
    w = torch.randn(O, I, device=dev).mul(0.02).to(bf)
    x = torch.randn(B, I, device=dev).to(bf)
    ga = torch.randn(B, I, device=dev).mul(0.3).to(bf)
    dx = torch.empty(B, I, device=dev, dtype=bf)
    assert try_call("cplxamd_rgemm_lrt_dx", ptr(g), O, 1, ptr(w), 1, I, ptr(x), ptr(ga), I, ptr(dx), I, B, I, O, BF16,
                    stream_ptr()), "the persistent kernel must take this shape"
    ref = ops._real_linear_dx(g, w, bf)
    ops.lrt_dx_accum(ref, None, x, None, ga)
    assert torch.equal(dx, ref)
    assert torch.equal(ops._real_lrt_dx(g, w, x, ga), ref)
    rows = torch.randint(0, B, (8,), device=dev)
    f64 = g[rows].double() @ w.double() + 2 * x[rows].double() * ga[rows].double()
    assert (dx[rows].double() - f64).abs().max() <= 1.2e-2 * f64.abs().max()
    small = torch.empty(B - 8, I, device=dev, dtype=bf)      # partial tiles: the one-tile kernel's staged epilogue, same bits
    assert try_call("cplxamd_rgemm_lrt_dx", ptr(g), O, 1, ptr(w), 1, I, ptr(x), ptr(ga), I, ptr(small), I, B - 8, I, O, BF16,
                    stream_ptr())
    assert torch.equal(small, ref[:B - 8])
