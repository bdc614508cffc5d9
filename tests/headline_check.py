"""Value check of the HEADLINE step where it runs: CplxLinearVD(4096, 4096), bf16 activations, batch 8192, behind
dp.DataParallel exactly as bench.py builds it, as ONE hipGraph replay (or eager): sampled rows of y, of dX (with the fused
`2 x g_a` term), three rows of dW WITH the fused KL accumulate, of dlog_sigma2, and the KL total -- against float64 numpy
with the numpy statement of the Philox stream (oracle/philox.py) for the noise of exactly the sampled outputs.
Reference arithmetic: cplxmodule/nn/relevance/complex/base.py:43-56 differentiated + complex/vd.py:95-99.

TEST INFRASTRUCTURE (imports the oracle): used by tests/test_gpu_headline.py and, untimed, by `bench.py --check`.
"""
import numpy as np
import torch

from oracle import cplx_oracle as orc
from oracle import philox

KLW = 1e-3


def _bf16r(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).bfloat16().double().numpy()


def _cplx_noise_at(elems, seed, offset):
    """The in-kernel noise of a complex layer at linear output indices `elems` (element e is pair e & 1 of Philox group
    e >> 1, scaled by sqrt(1/2): DESIGN "noise stream")."""
    e = np.asarray(elems, dtype=np.uint64)
    x = philox.philox4x32(e >> np.uint64(1), offset, seed)
    u = philox._u01(x)
    odd = (e & np.uint64(1)).astype(bool)
    u0, u1 = np.where(odd, u[:, 2], u[:, 0]), np.where(odd, u[:, 3], u[:, 1])
    r = np.sqrt(-2 * np.log(u0)) * np.sqrt(0.5)
    return r * np.cos(2 * np.pi * u1), r * np.sin(2 * np.pi * u1)


def _n(t):
    return t.detach().double().cpu().numpy()


def check_headline_step(B=8192, F=4096, graph=True, seed=4321, dev="cuda", full_kl=True, kl_weight_visible=3e3):
    """Runs one step of bench.py's workload and compares; returns {name: (max |err| / max |ref|, asserted bound)}.
    Raises AssertionError on a miss."""
    from cplxmodule_amd import Cplx, dp
    from cplxmodule_amd.nn import relevance as rel
    from cplxmodule_amd.nn.relevance import noise
    from cplxmodule_amd.utils.graphs import GraphedStep

    prev_mode = noise.mode
    torch.manual_seed(0)
    layer = rel.CplxLinearVD(F, F).to(dev)
    with torch.no_grad():
        layer.log_sigma2.uniform_(-12, 4)
    model = dp.DataParallel(layer, overlap=True)
    torch.manual_seed(1)
    x = Cplx(torch.randn(B, F, device=dev).bfloat16().requires_grad_(True),
             torch.randn(B, F, device=dev).bfloat16().requires_grad_(True))
    klw = torch.tensor(KLW, device=dev)
    layer.train()

    def step():
        model.zero_grad()
        x.real.grad = x.imag.grad = None
        y = model(x)
        kl = sum(rel.penalties(layer, reduction="sum"))
        torch.autograd.backward((y.real, y.imag, kl), (y.real.detach() * 2, y.imag.detach() * 2, klw))
        model.sync_gradients()
        return kl, y.real, y.imag

    try:
        noise.manual_seed(seed)
        if graph:
            noise.set_mode("philox-device")
            gs = GraphedStep(step, modules=[layer], warmup=3)
            gs.replay()
            torch.cuda.synchronize()
            st = noise.device_state(torch.device(dev)).cpu().numpy().astype(np.uint64)
            pseed, poff = int(st[0]), int(st[1])                 # the position the NEXT replay will draw from
            kl, yr, yi = gs.replay()
        else:
            noise.set_mode("philox")
            step()
            pseed, poff = noise.seed & 0xFFFFFFFFFFFFFFFF, noise.counter + 1
            kl, yr, yi = step()
        torch.cuda.synchronize()
        kl, yr, yi = kl.detach(), yr.detach(), yi.detach()
        w = layer.weight
        Wr, Wi = _n(w.real.detach().bfloat16()), _n(w.imag.detach().bfloat16())
        br, bi = _n(layer.bias.real), _n(layer.bias.imag)
        ls2 = _n(layer.log_sigma2)
        S16 = _n(layer.log_sigma2.detach().exp().bfloat16())
        res = {}

        def close(name, got, ref, tol):
            scale = float(np.abs(ref).max())
            err = float(np.abs(got - ref).max()) / scale
            res[name] = (err, tol)
            assert err <= tol, f"headline step: {name} off by {err:.3e} of max|ref| (bound {tol:.1e})"

        # ---- y and dX on sampled rows (first / tile boundaries / last)
        rows = np.array([0, 1, 255, 256, 4099 % B, B // 2] + list(range(B - 4, B)))
        tr = torch.from_numpy(rows).to(dev)
        Xr, Xi = _n(x.real[tr]), _n(x.imag[tr])
        mu_r = _bf16r(Xr @ Wr.T - Xi @ Wi.T + br)
        mu_i = _bf16r(Xr @ Wi.T + Xi @ Wr.T + bi)
        a = _bf16r(Xr * Xr + Xi * Xi)
        s2 = _bf16r(a @ S16.T)
        el = (rows[:, None].astype(np.uint64) * np.uint64(F) + np.arange(F, dtype=np.uint64)[None]).ravel()
        e_r, e_i = (v.reshape(len(rows), F) for v in _cplx_noise_at(el, pseed, poff))
        sd = np.sqrt(np.maximum(s2, 1e-8))
        close("y.real", _n(yr[tr]), mu_r + e_r * sd, 2e-2)
        close("y.imag", _n(yi[tr]), mu_i + e_i * sd, 2e-2)
        assert float(np.abs(e_r * sd).max()) > 0.05 * float(np.abs(mu_r).max())      # the noise is a visible part of y
        Gr, Gi = 2 * _n(yr[tr]), 2 * _n(yi[tr])                                       # the upstream gradient the step used
        gs2 = _bf16r(np.where(s2 >= 1e-8, (Gr * e_r + Gi * e_i) * 0.5 / sd, 0.0))
        g_a = _bf16r(gs2 @ S16)
        ref_r = _bf16r(Gr @ Wr + Gi @ Wi) + 2 * Xr * g_a
        ref_i = _bf16r(-Gr @ Wi + Gi @ Wr) + 2 * Xi * g_a
        assert float(np.abs(2 * Xr * g_a).max()) > 0.02 * float(np.abs(ref_r).max())   # ... and the fused term of dX
        close("dX.real (G conj W + 2 x ga)", _n(x.real.grad[tr]), ref_r, 2e-2)
        close("dX.imag (G conj W + 2 x ga)", _n(x.imag.grad[tr]), ref_i, 2e-2)
        # ---- three rows of dW = G^T conj(X) + klw dKL/dW and of dlog_sigma2, float64 over the whole batch.  At the
        # bench's KL weight (1e-3) the KL part is ~1e-5 of these gradients -- below what any tolerance on the sum can see --
        # so the same rows are checked AGAIN on one more step (same graph, the device scalar holding the weight overwritten)
        # with a weight that makes the fused accumulate a visible share.
        o_rows = np.array([0, 777 % F, F - 1])
        to = torch.from_numpy(o_rows).to(dev)
        Xd_r, Xd_i = x.real.detach().double(), x.imag.detach().double()
        w32r, w32i = _n(w.real)[o_rows], _n(w.imag)[o_rows]
        S3 = layer.log_sigma2.detach().exp().bfloat16().double()[to]                  # [3, F]
        a_c = (x.real.detach().float() ** 2 + x.imag.detach().float() ** 2).bfloat16().double()
        s2_c = (a_c @ S3.t()).float().bfloat16().double()
        sd_c = s2_c.clamp_min(1e-8).sqrt()

        def param_rows(tag, weight, yr, yi, poff, want_visible):
            Gc_r, Gc_i = 2 * yr[:, to].double(), 2 * yi[:, to].double()               # [B, 3]
            dWr = (Gc_r.t() @ Xd_r + Gc_i.t() @ Xd_i).cpu().numpy()
            dWi = (Gc_i.t() @ Xd_r - Gc_r.t() @ Xd_i).cpu().numpy()
            klg = orc.penalty_bwd("cplx_vd", np.full_like(ls2[o_rows], weight), ls2[o_rows], w32r, w32i)
            share = float(np.abs(klg["dwr"]).max()) / float(np.abs(dWr + klg["dwr"]).max())
            res[f"KL share of max|dW| {tag}"] = (share, 1.0)
            if want_visible:
                assert share > 0.05, f"the KL accumulate is not a visible part of dW ({share:.2e})"
            close(f"dW.real (+ KL accumulate) {tag}", _n(w.real.grad[to]), dWr + klg["dwr"], 2e-4)
            close(f"dW.imag (+ KL accumulate) {tag}", _n(w.imag.grad[to]), dWi + klg["dwi"], 2e-4)
            el = (np.arange(B, dtype=np.uint64)[:, None] * np.uint64(F) + o_rows[None].astype(np.uint64)).ravel()
            er, ei = (torch.from_numpy(v.reshape(B, 3)).to(dev) for v in _cplx_noise_at(el, pseed, poff))
            gsd = Gc_r * er + Gc_i * ei
            gs2_c = torch.where(s2_c >= 1e-8, gsd * 0.5 / sd_c, torch.zeros_like(gsd)).float().bfloat16().double()
            data = (gs2_c.t() @ a_c * layer.log_sigma2.detach().double()[to].exp()).cpu().numpy()
            close(f"dlog_sigma2 (+ KL accumulate) {tag}", _n(layer.log_sigma2.grad[to]), data + klg["dlog_sigma2"], 5e-3)

        param_rows("@ KL weight 1e-3", KLW, yr, yi, poff, want_visible=False)
        # ---- the KL total (scipy Ei on the full weight: ~8 s for 4096^2)
        if full_kl:
            want = orc.penalty("cplx_vd", ls2, _n(w.real), _n(w.imag)).sum()
            err = abs(float(kl.detach()) - want) / abs(want)
            res["KL total"] = (err, 2e-6)
            assert err <= 2e-6, f"headline step: KL total off by {err:.3e}"
        # ---- bias gradient: column sums of G (rides in the noise backward)
        gb = (2 * yr.double()).sum(0).cpu().numpy()
        close("dbias.real", _n(layer.bias.real.grad), gb, 1e-4)
        # ---- one more step with the KL weight (a device scalar the captured graph reads) made large enough to see
        big = kl_weight_visible
        klw.fill_(big)
        if graph:
            _, yr2, yi2 = gs.replay()
        else:
            _, yr2, yi2 = step()
        torch.cuda.synchronize()
        param_rows(f"@ KL weight {big:g}", big, yr2.detach(), yi2.detach(), poff + 1, want_visible=True)
        klw.fill_(KLW)
        return res
    finally:
        noise.set_mode(prev_mode)
