"""Generate tests/golden/*.npz by running the REAL reference (/root/reference).

Run in the build container only (the reference does not travel to the GPU
box):  ``python oracle/gen_golden.py``.  Every fixture holds seeded inputs
and the outputs / autograd gradients the reference produced for them, in
float32 and float64.  Fixtures are data only.

TEST INFRASTRUCTURE: never imported by the product package.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    # cplxmodule/__init__.py:2 imports a setup.py-generated __version__ module
    sys.path.insert(0, REF)
    m = types.ModuleType("cplxmodule.__version__")
    m.__version__ = open(os.path.join(REF, "VERSION")).read().strip()
    sys.modules["cplxmodule.__version__"] = m
    import cplxmodule  # noqa: F401
    return cplxmodule


cm = import_reference()
from cplxmodule import cplx  # noqa: E402
from cplxmodule.nn import CplxLinear, CplxConv2d, CplxBatchNorm1d, CplxBatchNorm2d  # noqa: E402
from cplxmodule.nn.modules.batchnorm import cplx_batch_norm  # noqa: E402
from cplxmodule.nn import relevance as rel  # noqa: E402
from cplxmodule.nn.relevance.complex import torch_expi  # noqa: E402

DT = {"f32": torch.float32, "f64": torch.float64}


def npy(t):
    return t.detach().cpu().numpy().copy()


def C(re, im):
    return cplx.Cplx(re, im)


def leaf(*shape, dtype, scale=1.0):
    return (torch.randn(*shape, dtype=dtype) * scale).requires_grad_(True)


def save(name, d):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"{name}: {len(d)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


# --------------------------------------------------------------------------- #
def gen_linear():
    d = {}
    for tag, dt in DT.items():
        for case, (lead, I, O) in {"a": ((5, 5), 200, 321), "b": ((64,), 128, 128),
                                   "c": ((7,), 33, 19)}.items():
            torch.manual_seed(11)
            xr, xi = leaf(*lead, I, dtype=dt), leaf(*lead, I, dtype=dt)
            wr, wi = leaf(O, I, dtype=dt, scale=0.1), leaf(O, I, dtype=dt, scale=0.1)
            br, bi = leaf(O, dtype=dt), leaf(O, dtype=dt)
            gr, gi = torch.randn(*lead, O, dtype=dt), torch.randn(*lead, O, dtype=dt)
            k = f"{tag}_{case}_"
            for nm, t in dict(xr=xr, xi=xi, wr=wr, wi=wi, br=br, bi=bi, gr=gr, gi=gi).items():
                d[k + nm] = npy(t)
            for algo in ("naive", "3m", "cat"):
                y = getattr(cplx, "linear_" + algo)(C(xr, xi), C(wr, wi), C(br, bi))
                d[k + f"y_{algo}_r"], d[k + f"y_{algo}_i"] = npy(y.real), npy(y.imag)
            y = cplx.linear(C(xr, xi), C(wr, wi), C(br, bi))
            grads = torch.autograd.grad((y.real * gr).sum() + (y.imag * gi).sum(),
                                        [xr, xi, wr, wi, br, bi])
            for nm, g in zip(["dxr", "dxi", "dwr", "dwi", "dbr", "dbi"], grads):
                d[k + nm] = npy(g)
            y = cplx.linear(C(xr, xi), C(wr, wi), None)
            d[k + "y_nobias_r"], d[k + "y_nobias_i"] = npy(y.real), npy(y.imag)
        # Cplx.__matmul__ (batched)
        torch.manual_seed(12)
        ur, ui = torch.randn(3, 9, 17, dtype=dt), torch.randn(3, 9, 17, dtype=dt)
        vr, vi = torch.randn(3, 17, 6, dtype=dt), torch.randn(3, 17, 6, dtype=dt)
        m = C(ur, ui) @ C(vr, vi)
        for nm, t in dict(ur=ur, ui=ui, vr=vr, vi=vi, mr=m.real, mi=m.imag).items():
            d[f"{tag}_mm_{nm}"] = npy(t)
    save("linear", d)


def _mixed_vd_params(O, I, dt, cplx_w=True):
    wr = torch.empty(O, I, dtype=dt).uniform_(-0.09, 0.09)
    wi = torch.empty(O, I, dtype=dt).uniform_(-0.09, 0.09)
    ls2 = torch.empty(O, I, dtype=dt).uniform_(-12, 4)
    # special values: exact zeros, tiny weights, huge/small variances
    wr.view(-1)[:3] = 0.0
    wi.view(-1)[:2] = 0.0
    wr.view(-1)[5], wi.view(-1)[5] = 1e-20, 0.0
    wr.view(-1)[6], wi.view(-1)[6] = 3e-7, -2e-7
    ls2.view(-1)[7], ls2.view(-1)[8], ls2.view(-1)[9] = -30.0, 9.0, -10.0
    return wr, wi, ls2


def gen_lrt_linear():
    d = {}
    for tag, dt in DT.items():
        torch.set_default_dtype(dt)
        B, I, O = 64, 128, 128
        # complex VD, training mode
        torch.manual_seed(21)
        layer = rel.CplxLinearVD(I, O, bias=True)
        with torch.no_grad():
            layer.log_sigma2.uniform_(-10, 1)
        x = cplx.randn(B, I)
        xr, xi = x.real.clone().requires_grad_(True), x.imag.clone().requires_grad_(True)
        gr, gi = torch.randn(B, O), torch.randn(B, O)
        layer.train()
        torch.manual_seed(77)
        y = layer(C(xr, xi))
        torch.manual_seed(77)
        tape = torch.randn(2, B, O)
        ps = [xr, xi, layer.weight.real, layer.weight.imag, layer.bias.real,
              layer.bias.imag, layer.log_sigma2]
        grads = torch.autograd.grad((y.real * gr).sum() + (y.imag * gi).sum(), ps)
        k = f"{tag}_cplx_"
        for nm, t in dict(xr=xr, xi=xi, wr=ps[2], wi=ps[3], br=ps[4], bi=ps[5],
                          ls2=ps[6], gr=gr, gi=gi, tape=tape, yr=y.real, yi=y.imag).items():
            d[k + nm] = npy(t)
        for nm, g in zip(["dxr", "dxi", "dwr", "dwi", "dbr", "dbi", "dls2"], grads):
            d[k + nm] = npy(g)
        layer.eval()
        y = layer(C(xr, xi))
        d[k + "yr_eval"], d[k + "yi_eval"] = npy(y.real), npy(y.imag)
        # clamp boundary: tiny activations so that s2 < 1e-8 for some rows
        xs = C(xr.detach() * 1e-3, xi.detach() * 1e-3)
        xs_r, xs_i = xs.real.clone().requires_grad_(True), xs.imag.clone().requires_grad_(True)
        with torch.no_grad():
            layer.log_sigma2.fill_(-9.0)
            layer.log_sigma2[: O // 2] = -13.5
        layer.train()
        torch.manual_seed(78)
        y = layer(C(xs_r, xs_i))
        torch.manual_seed(78)
        tape2 = torch.randn(2, B, O)
        grads = torch.autograd.grad((y.real * gr).sum() + (y.imag * gi).sum(),
                                    [xs_r, xs_i, layer.log_sigma2])
        d[k + "clamp_ls2"] = npy(layer.log_sigma2)
        for nm, t in dict(clamp_xr=xs_r, clamp_xi=xs_i, clamp_tape=tape2, clamp_yr=y.real,
                          clamp_yi=y.imag, clamp_dxr=grads[0], clamp_dxi=grads[1],
                          clamp_dls2=grads[2]).items():
            d[k + nm] = npy(t)

        # real VD, training mode
        torch.manual_seed(22)
        layer = rel.LinearVD(I, O, bias=True)
        with torch.no_grad():
            layer.log_sigma2.uniform_(-10, 1)
        x = torch.randn(B, I).requires_grad_(True)
        g = torch.randn(B, O)
        layer.train()
        torch.manual_seed(79)
        y = layer(x)
        torch.manual_seed(79)
        eps = torch.randn(B, O)
        grads = torch.autograd.grad((y * g).sum(), [x, layer.weight, layer.bias, layer.log_sigma2])
        k = f"{tag}_real_"
        for nm, t in dict(x=x, w=layer.weight, b=layer.bias, ls2=layer.log_sigma2, g=g,
                          eps=eps, y=y, dx=grads[0], dw=grads[1], db=grads[2],
                          dls2=grads[3]).items():
            d[k + nm] = npy(t)
    torch.set_default_dtype(torch.float32)
    save("lrt_linear", d)


def gen_penalty():
    d = {}
    kinds = {"real_vd": rel.LinearVD, "real_ard": rel.LinearARD,
             "cplx_vd": rel.CplxLinearVD, "cplx_ard": rel.CplxLinearARD}
    for tag, dt in DT.items():
        torch.set_default_dtype(dt)
        O, I = 48, 40
        torch.manual_seed(31)
        wr, wi, ls2 = _mixed_vd_params(O, I, dt)
        g = torch.rand(O, I)
        k = f"{tag}_"
        for nm, t in dict(wr=wr, wi=wi, ls2=ls2, g=g).items():
            d[k + nm] = npy(t)
        for kind, cls in kinds.items():
            layer = cls(I, O, bias=False)
            with torch.no_grad():
                layer.log_sigma2.copy_(ls2)
                if kind.startswith("cplx"):
                    layer.weight.real.copy_(wr)
                    layer.weight.imag.copy_(wi)
                    wps = [layer.weight.real, layer.weight.imag]
                else:
                    layer.weight.copy_(wr)
                    wps = [layer.weight]
            pen = layer.penalty
            d[k + kind + "_log_alpha"] = npy(layer.log_alpha)
            d[k + kind + "_penalty"] = npy(pen)
            d[k + kind + "_sum"] = npy(sum(rel.penalties(layer, reduction="sum")))
            d[k + kind + "_mean"] = npy(sum(rel.penalties(layer, reduction="mean")))
            grads = torch.autograd.grad((pen * g).sum(), [layer.log_sigma2] + wps)
            d[k + kind + "_dls2"] = npy(grads[0])
            d[k + kind + "_dwr"] = npy(grads[1])
            if len(grads) > 2:
                d[k + kind + "_dwi"] = npy(grads[2])
            gs = torch.autograd.grad(layer.penalty.sum(), [layer.log_sigma2] + wps)
            d[k + kind + "_sum_dls2"] = npy(gs[0])
            d[k + kind + "_sum_dwr"] = npy(gs[1])
            if len(gs) > 2:
                d[k + kind + "_sum_dwi"] = npy(gs[2])
            for th in (-0.5, 1.0, 3.0):
                m = layer.relevance(threshold=th)
                d[k + kind + f"_mask_{th}"] = npy(m)
                la = layer.log_alpha.detach()
                # near-threshold census: elements within 4 ulp of the threshold
                ulp = torch.abs(la) * torch.finfo(dt).eps
                d[k + kind + f"_near_{th}"] = np.array(int((torch.abs(la - th) < 4 * ulp).sum()))
            masks = rel.compute_ard_masks(layer, hard=False, threshold=1.0)
            assert list(masks) == ["mask"]
        # expi primitive on a grid (both signs) + the reference's backward
        x = torch.cat([-torch.logspace(-8, 3, 300), torch.logspace(-8, 1.5, 100),
                       torch.randn(200)]).to(dt).requires_grad_(True)
        y = torch_expi(x)
        gx, = torch.autograd.grad(y.sum(), x)
        d[k + "expi_x"], d[k + "expi_y"], d[k + "expi_dx"] = npy(x), npy(y), npy(gx)
    torch.set_default_dtype(torch.float32)
    save("penalty", d)


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden_cases import CONV_CASES  # noqa: E402


def gen_conv():
    d = {}
    for tag, dt in DT.items():
        torch.set_default_dtype(dt)
        for name, (B, Ci, Co, H, W, ks, st, pd, dl, gp, mode) in CONV_CASES.items():
            torch.manual_seed(41)
            layer = CplxConv2d(Ci, Co, ks, stride=st, padding=pd, dilation=dl,
                               groups=gp, bias=True, padding_mode=mode)
            x = cplx.randn(B, Ci, H, W)
            xr, xi = x.real.clone().requires_grad_(True), x.imag.clone().requires_grad_(True)
            y = layer(C(xr, xi))
            gr, gi = torch.randn_like(y.real), torch.randn_like(y.imag)
            ps = [xr, xi, layer.weight.real, layer.weight.imag, layer.bias.real, layer.bias.imag]
            grads = torch.autograd.grad((y.real * gr).sum() + (y.imag * gi).sum(), ps)
            k = f"{tag}_{name}_"
            for nm, t in dict(xr=xr, xi=xi, wr=ps[2], wi=ps[3], br=ps[4], bi=ps[5], gr=gr,
                              gi=gi, yr=y.real, yi=y.imag).items():
                d[k + nm] = npy(t)
            for nm, g in zip(["dxr", "dxi", "dwr", "dwi", "dbr", "dbi"], grads):
                d[k + nm] = npy(g)
        # LRT complex conv (training) and real conv VD
        torch.manual_seed(42)
        layer = rel.CplxConv2dVD(3, 4, 3, stride=1, padding=1)
        with torch.no_grad():
            layer.log_sigma2.uniform_(-10, 1)
        x = cplx.randn(2, 3, 8, 7)
        xr, xi = x.real.clone().requires_grad_(True), x.imag.clone().requires_grad_(True)
        layer.train()
        torch.manual_seed(91)
        y = layer(C(xr, xi))
        torch.manual_seed(91)
        tape = torch.randn(2, *y.real.shape)
        gr, gi = torch.randn_like(y.real), torch.randn_like(y.imag)
        ps = [xr, xi, layer.weight.real, layer.weight.imag, layer.bias.real,
              layer.bias.imag, layer.log_sigma2]
        grads = torch.autograd.grad((y.real * gr).sum() + (y.imag * gi).sum(), ps)
        k = f"{tag}_lrtc_"
        for nm, t in dict(xr=xr, xi=xi, wr=ps[2], wi=ps[3], br=ps[4], bi=ps[5], ls2=ps[6],
                          gr=gr, gi=gi, tape=tape, yr=y.real, yi=y.imag).items():
            d[k + nm] = npy(t)
        for nm, g in zip(["dxr", "dxi", "dwr", "dwi", "dbr", "dbi", "dls2"], grads):
            d[k + nm] = npy(g)
        d[k + "penalty_sum"] = npy(sum(rel.penalties(layer)))

        torch.manual_seed(43)
        layer = rel.Conv2dVD(3, 4, 3, stride=2, padding=1)
        with torch.no_grad():
            layer.log_sigma2.uniform_(-10, 1)
        x = torch.randn(2, 3, 9, 8).requires_grad_(True)
        layer.train()
        torch.manual_seed(92)
        y = layer(x)
        torch.manual_seed(92)
        eps = torch.randn(*y.shape)
        g = torch.randn_like(y)
        grads = torch.autograd.grad((y * g).sum(), [x, layer.weight, layer.bias, layer.log_sigma2])
        k = f"{tag}_lrtr_"
        for nm, t in dict(x=x, w=layer.weight, b=layer.bias, ls2=layer.log_sigma2, g=g,
                          eps=eps, y=y, dx=grads[0], dw=grads[1], db=grads[2],
                          dls2=grads[3]).items():
            d[k + nm] = npy(t)
    torch.set_default_dtype(torch.float32)
    save("conv", d)


def gen_batchnorm():
    d = {}
    for tag, dt in DT.items():
        torch.set_default_dtype(dt)
        for name, shape, cls in (("2d", (6, 5, 7, 4), CplxBatchNorm2d),
                                 ("1d", (16, 9), CplxBatchNorm1d),
                                 ("1d3", (4, 3, 10), CplxBatchNorm1d)):
            torch.manual_seed(51)
            F_ = shape[1]
            bn = cls(F_, eps=1e-5, momentum=0.1, affine=True)
            with torch.no_grad():
                bn.weight.add_(0.3 * torch.randn(2, 2, F_))
                bn.bias.add_(0.3 * torch.randn(2, F_))
            k = f"{tag}_{name}_"
            d[k + "weight"], d[k + "bias"] = npy(bn.weight), npy(bn.bias)
            bn.train()
            for step in range(3):
                base = torch.randn(*shape)
                xr = (1.5 * base + 0.4 * torch.randn(*shape) + 0.7).requires_grad_(True)
                xi = (0.8 * base - 0.5 * torch.randn(*shape) - 0.2).requires_grad_(True)
                y = bn(C(xr, xi))
                gr, gi = torch.randn_like(xr), torch.randn_like(xi)
                grads = torch.autograd.grad((y.real * gr).sum() + (y.imag * gi).sum(),
                                            [xr, xi, bn.weight, bn.bias])
                s = k + f"s{step}_"
                for nm, t in dict(xr=xr, xi=xi, yr=y.real, yi=y.imag, gr=gr, gi=gi,
                                  dxr=grads[0], dxi=grads[1], dweight=grads[2], dbias=grads[3],
                                  running_mean=bn.running_mean, running_var=bn.running_var,
                                  nbt=bn.num_batches_tracked).items():
                    d[s + nm] = npy(t)
            bn.eval()
            y = bn(C(xr, xi))
            grads = torch.autograd.grad((y.real * gr).sum() + (y.imag * gi).sum(),
                                        [xr, xi, bn.weight, bn.bias])
            for nm, t in dict(yr=y.real, yi=y.imag, dxr=grads[0], dxi=grads[1],
                              dweight=grads[2], dbias=grads[3]).items():
                d[k + "eval_" + nm] = npy(t)
        # functional form: no affine, no running stats; cumulative momentum
        torch.manual_seed(52)
        xr = torch.randn(8, 3, 5).requires_grad_(True)
        xi = (0.5 * xr.detach() + torch.randn(8, 3, 5)).requires_grad_(True)
        y = cplx_batch_norm(C(xr, xi), None, None, None, None, True, 0.1, 1e-3)
        gr, gi = torch.randn_like(xr), torch.randn_like(xi)
        grads = torch.autograd.grad((y.real * gr).sum() + (y.imag * gi).sum(), [xr, xi])
        for nm, t in dict(xr=xr, xi=xi, yr=y.real, yi=y.imag, gr=gr, gi=gi, dxr=grads[0],
                          dxi=grads[1]).items():
            d[f"{tag}_func_{nm}"] = npy(t)
        bn = CplxBatchNorm1d(3, momentum=None, affine=False)
        bn.train()
        for step in range(2):
            y = bn(C(xr.detach() + step, xi.detach() * (1 + step)))
        d[f"{tag}_cma_running_mean"] = npy(bn.running_mean)
        d[f"{tag}_cma_running_var"] = npy(bn.running_var)
        d[f"{tag}_cma_yr"], d[f"{tag}_cma_yi"] = npy(y.real), npy(y.imag)
    torch.set_default_dtype(torch.float32)
    save("batchnorm", d)


def gen_api():
    """Host-side contract: state-dict keys/shapes, init bounds, walker names."""
    d = {}
    torch.manual_seed(61)
    layers = {
        "CplxLinear": CplxLinear(100, 400),
        "CplxLinearVD": rel.CplxLinearVD(12, 7),
        "CplxLinearARD": rel.CplxLinearARD(12, 7, bias=False),
        "LinearVD": rel.LinearVD(12, 7),
        "LinearARD": rel.LinearARD(12, 7),
        "CplxConv2d": CplxConv2d(6, 4, (3, 2), groups=2),
        "CplxConv2dVD": rel.CplxConv2dVD(6, 4, 3),
        "Conv2dARD": rel.Conv2dARD(6, 4, 3),
        "CplxBatchNorm2d": CplxBatchNorm2d(5),
    }
    for name, layer in layers.items():
        sd = layer.state_dict()
        d[name + "__keys"] = np.array(list(sd.keys()))
        d[name + "__shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
        d[name + "__params"] = np.array([n for n, _ in layer.named_parameters()])
    lin = layers["CplxLinear"]
    d["CplxLinear__wmax"] = np.array(float(max(lin.weight.real.abs().max(), lin.weight.imag.abs().max())))
    d["CplxLinear__bmax"] = np.array(float(max(lin.bias.real.abs().max(), lin.bias.imag.abs().max())))
    cv = layers["CplxConv2d"]
    d["CplxConv2d__wmax"] = np.array(float(max(cv.weight.real.abs().max(), cv.weight.imag.abs().max())))
    d["CplxConv2d__bmax"] = np.array(float(max(cv.bias.real.abs().max(), cv.bias.imag.abs().max())))
    d["CplxLinearVD__ls2"] = npy(layers["CplxLinearVD"].log_sigma2)
    # walker names on a nested model
    model = torch.nn.Sequential(rel.CplxLinearVD(4, 5), torch.nn.Sequential(rel.CplxLinearARD(5, 3)))
    d["walk__penalty_names"] = np.array([n for n, _ in rel.named_penalties(model)])
    d["walk__mask_names"] = np.array(list(rel.compute_ard_masks(model, threshold=1.0)))
    d["walk__mask_names_prefix"] = np.array(list(rel.compute_ard_masks(model, prefix="net", threshold=1.0)))
    save("api", d)


def gen_bilinear():
    """cplx.bilinear (both conjugate modes), CplxBilinearVD / BilinearVD in
    training mode (noise tape recorded) and eval mode, values + autograd gradients."""
    d = {}
    for tag, dt in DT.items():
        torch.set_default_dtype(dt)
        B, I1, I2, O = 13, 9, 11, 7
        for conj in (True, False):
            torch.manual_seed(31)
            x1r, x1i, x2r, x2i = leaf(B, I1, dtype=dt), leaf(B, I1, dtype=dt), leaf(B, I2, dtype=dt), leaf(B, I2, dtype=dt)
            wr, wi = leaf(O, I1, I2, dtype=dt, scale=0.2), leaf(O, I1, I2, dtype=dt, scale=0.2)
            br, bi = leaf(O, dtype=dt), leaf(O, dtype=dt)
            gr, gi = torch.randn(B, O, dtype=dt), torch.randn(B, O, dtype=dt)
            y = cplx.bilinear(C(x1r, x1i), C(x2r, x2i), C(wr, wi), C(br, bi), conjugate=conj)
            grads = torch.autograd.grad((y.real * gr).sum() + (y.imag * gi).sum(),
                                        [x1r, x1i, x2r, x2i, wr, wi, br, bi])
            k = f"{tag}_fn_{'conj' if conj else 'plain'}_"
            for nm, t in dict(x1r=x1r, x1i=x1i, x2r=x2r, x2i=x2i, wr=wr, wi=wi, br=br, bi=bi, gr=gr,
                              gi=gi, yr=y.real, yi=y.imag).items():
                d[k + nm] = npy(t)
            for nm, g in zip(["dx1r", "dx1i", "dx2r", "dx2i", "dwr", "dwi", "dbr", "dbi"], grads):
                d[k + nm] = npy(g)
        # leading batch dims + no bias, through the layer
        torch.manual_seed(32)
        layer = cm.nn.CplxBilinear(I1, I2, O, bias=False, conjugate=True)
        a, b = cplx.randn(2, 3, I1), cplx.randn(2, 3, I2)
        y = layer(a, b)
        for nm, t in dict(x1r=a.real, x1i=a.imag, x2r=b.real, x2i=b.imag, wr=layer.weight.real,
                          wi=layer.weight.imag, yr=y.real, yi=y.imag).items():
            d[f"{tag}_layer_{nm}"] = npy(t)
        # complex VD bilinear: training (tape) + eval
        torch.manual_seed(33)
        layer = rel.CplxBilinearVD(I1, I2, O, bias=True, conjugate=True)
        with torch.no_grad():
            layer.log_sigma2.uniform_(-10, 1)
        a, b = cplx.randn(B, I1), cplx.randn(B, I2)
        x1r, x1i = a.real.clone().requires_grad_(True), a.imag.clone().requires_grad_(True)
        x2r, x2i = b.real.clone().requires_grad_(True), b.imag.clone().requires_grad_(True)
        gr, gi = torch.randn(B, O), torch.randn(B, O)
        layer.train()
        torch.manual_seed(81)
        y = layer(C(x1r, x1i), C(x2r, x2i))
        torch.manual_seed(81)
        tape = torch.randn(2, B, O)
        ps = [x1r, x1i, x2r, x2i, layer.weight.real, layer.weight.imag, layer.bias.real,
              layer.bias.imag, layer.log_sigma2]
        grads = torch.autograd.grad((y.real * gr).sum() + (y.imag * gi).sum(), ps)
        k = f"{tag}_vd_"
        for nm, t in dict(x1r=x1r, x1i=x1i, x2r=x2r, x2i=x2i, wr=ps[4], wi=ps[5], br=ps[6], bi=ps[7],
                          ls2=ps[8], gr=gr, gi=gi, tape=tape, yr=y.real, yi=y.imag).items():
            d[k + nm] = npy(t)
        for nm, g in zip(["dx1r", "dx1i", "dx2r", "dx2i", "dwr", "dwi", "dbr", "dbi", "dls2"], grads):
            d[k + nm] = npy(g)
        layer.eval()
        y = layer(C(x1r, x1i), C(x2r, x2i))
        d[k + "yr_eval"], d[k + "yi_eval"] = npy(y.real), npy(y.imag)
        d[k + "pen"] = npy(layer.penalty)
        # real VD bilinear
        torch.manual_seed(34)
        layer = rel.BilinearVD(I1, I2, O, bias=True)
        with torch.no_grad():
            layer.log_sigma2.uniform_(-10, 1)
        x1, x2 = torch.randn(B, I1).requires_grad_(True), torch.randn(B, I2).requires_grad_(True)
        g = torch.randn(B, O)
        layer.train()
        torch.manual_seed(82)
        y = layer(x1, x2)
        torch.manual_seed(82)
        eps = torch.randn(B, O)
        grads = torch.autograd.grad((y * g).sum(), [x1, x2, layer.weight, layer.bias, layer.log_sigma2])
        k = f"{tag}_real_"
        for nm, t in dict(x1=x1, x2=x2, w=layer.weight, b=layer.bias, ls2=layer.log_sigma2, g=g,
                          eps=eps, y=y, dx1=grads[0], dx2=grads[1], dw=grads[2], db=grads[3],
                          dls2=grads[4]).items():
            d[k + nm] = npy(t)
        layer.eval()
        d[k + "y_eval"] = npy(layer(x1, x2))
    torch.set_default_dtype(torch.float32)
    save("bilinear", d)


from gen_golden_cases import CONV3D_CASES, POOL3D_CASES  # noqa: E402


def gen_conv3d():
    """cplx.conv3d (values + gradients), CplxConv3dVD / Conv3dVD training with a recorded noise tape
    and eval mode, cplx.max_pool3d."""
    from cplxmodule.nn import CplxMaxPool3d  # noqa: F401
    d = {}
    for tag, dt in DT.items():
        torch.set_default_dtype(dt)
        for name, c in CONV3D_CASES.items():
            torch.manual_seed(41)
            xr, xi = leaf(*c["x"], dtype=dt), leaf(*c["x"], dtype=dt)
            wr, wi = leaf(*c["w"], dtype=dt, scale=0.2), leaf(*c["w"], dtype=dt, scale=0.2)
            br, bi = leaf(c["w"][0], dtype=dt), leaf(c["w"][0], dtype=dt)
            y = cplx.conv3d(C(xr, xi), C(wr, wi), C(br, bi), **c["kw"])
            gr, gi = torch.randn_like(y.real), torch.randn_like(y.imag)
            grads = torch.autograd.grad((y.real * gr).sum() + (y.imag * gi).sum(), [xr, xi, wr, wi, br, bi])
            k = f"{tag}_{name}_"
            for nm, t in dict(xr=xr, xi=xi, wr=wr, wi=wi, br=br, bi=bi, gr=gr, gi=gi, yr=y.real, yi=y.imag).items():
                d[k + nm] = npy(t)
            for nm, g in zip(["dxr", "dxi", "dwr", "dwi", "dbr", "dbi"], grads):
                d[k + nm] = npy(g)
        # complex VD conv3d
        torch.manual_seed(42)
        layer = rel.CplxConv3dVD(3, 4, (2, 3, 2), stride=(1, 2, 1), padding=(1, 1, 0))
        with torch.no_grad():
            layer.log_sigma2.uniform_(-10, 1)
        x = cplx.randn(2, 3, 5, 7, 6)
        xr, xi = x.real.clone().requires_grad_(True), x.imag.clone().requires_grad_(True)
        layer.train()
        torch.manual_seed(91)
        y = layer(C(xr, xi))
        torch.manual_seed(91)
        tape = torch.randn(2, *y.shape)
        gr, gi = torch.randn_like(y.real), torch.randn_like(y.imag)
        ps = [xr, xi, layer.weight.real, layer.weight.imag, layer.bias.real, layer.bias.imag, layer.log_sigma2]
        grads = torch.autograd.grad((y.real * gr).sum() + (y.imag * gi).sum(), ps)
        k = f"{tag}_vd_"
        for nm, t in dict(xr=xr, xi=xi, wr=ps[2], wi=ps[3], br=ps[4], bi=ps[5], ls2=ps[6], gr=gr, gi=gi,
                          tape=tape, yr=y.real, yi=y.imag).items():
            d[k + nm] = npy(t)
        for nm, g in zip(["dxr", "dxi", "dwr", "dwi", "dbr", "dbi", "dls2"], grads):
            d[k + nm] = npy(g)
        layer.eval()
        y = layer(C(xr, xi))
        d[k + "yr_eval"], d[k + "yi_eval"] = npy(y.real), npy(y.imag)
        # real VD conv3d
        torch.manual_seed(43)
        layer = rel.Conv3dVD(3, 4, 2, padding=1)
        with torch.no_grad():
            layer.log_sigma2.uniform_(-10, 1)
        x = torch.randn(2, 3, 4, 5, 6).requires_grad_(True)
        layer.train()
        torch.manual_seed(92)
        y = layer(x)
        torch.manual_seed(92)
        eps = torch.randn(*y.shape)
        g = torch.randn_like(y)
        grads = torch.autograd.grad((y * g).sum(), [x, layer.weight, layer.bias, layer.log_sigma2])
        k = f"{tag}_real_"
        for nm, t in dict(x=x, w=layer.weight, b=layer.bias, ls2=layer.log_sigma2, g=g, eps=eps, y=y,
                          dx=grads[0], dw=grads[1], db=grads[2], dls2=grads[3]).items():
            d[k + nm] = npy(t)
        layer.eval()
        d[k + "y_eval"] = npy(layer(x))
        # abs-max pooling 3-d
        torch.manual_seed(44)
        zr, zi = leaf(2, 3, 7, 8, 9, dtype=dt), leaf(2, 3, 7, 8, 9, dtype=dt)
        d[f"{tag}_mp_zr"], d[f"{tag}_mp_zi"] = npy(zr), npy(zi)
        for name, kw in POOL3D_CASES.items():
            for t in (zr, zi):
                t.grad = None
            y = cplx.max_pool3d(C(zr, zi), **kw)
            gr, gi = torch.randn_like(y.real), torch.randn_like(y.imag)
            torch.autograd.backward((y.real, y.imag), (gr, gi))
            k = f"{tag}_mp_{name}_"
            d[k + "yr"], d[k + "yi"], d[k + "gr"], d[k + "gi"] = npy(y.real), npy(y.imag), npy(gr), npy(gi)
            d[k + "dzr"], d[k + "dzi"] = npy(zr.grad), npy(zi.grad)
    torch.set_default_dtype(torch.float32)
    save("conv3d", d)


def gen_conv_transpose():
    """cplx.conv_transpose2d / conv_transpose1d (functional; the reference LAYER does not run on
    torch >= 2, its functional does): values + autograd gradients."""
    from gen_golden_cases import CONVT_CASES
    d = {}
    for tag, dt in DT.items():
        for name, c in CONVT_CASES.items():
            torch.manual_seed(51)
            xr, xi = leaf(*c["x"], dtype=dt), leaf(*c["x"], dtype=dt)
            wr, wi = leaf(*c["w"], dtype=dt, scale=0.3), leaf(*c["w"], dtype=dt, scale=0.3)
            co = c["w"][1] * c["kw"]["groups"]
            br, bi = leaf(co, dtype=dt), leaf(co, dtype=dt)
            y = cplx.conv_transpose2d(C(xr, xi), C(wr, wi), C(br, bi), **c["kw"])
            gr, gi = torch.randn_like(y.real), torch.randn_like(y.imag)
            grads = torch.autograd.grad((y.real * gr).sum() + (y.imag * gi).sum(), [xr, xi, wr, wi, br, bi])
            k = f"{tag}_{name}_"
            for nm, t in dict(xr=xr, xi=xi, wr=wr, wi=wi, br=br, bi=bi, gr=gr, gi=gi, yr=y.real, yi=y.imag).items():
                d[k + nm] = npy(t)
            for nm, g in zip(["dxr", "dxi", "dwr", "dwi", "dbr", "dbi"], grads):
                d[k + nm] = npy(g)
        torch.manual_seed(52)
        xr, xi = leaf(2, 3, 9, dtype=dt), leaf(2, 3, 9, dtype=dt)
        wr, wi = leaf(3, 4, 4, dtype=dt, scale=0.3), leaf(3, 4, 4, dtype=dt, scale=0.3)
        y = cplx.conv_transpose1d(C(xr, xi), C(wr, wi), None, stride=2, padding=1, output_padding=1, groups=1)
        for nm, t in dict(xr=xr, xi=xi, wr=wr, wi=wi, yr=y.real, yi=y.imag).items():
            d[f"{tag}_1d_{nm}"] = npy(t)
    save("conv_transpose", d)


def gen_extras():
    """layout converters, modReLU (+ learnable thresholds), CplxDropout."""
    from cplxmodule.nn import CplxModReLU, CplxAdaptiveModReLU, CplxDropout  # noqa: F401
    d = {}
    for tag, dt in DT.items():
        torch.manual_seed(11)
        # modReLU: generic values, values inside the dead zone, on the clamp and exact zeros
        zr, zi = leaf(6, 5, 8, dtype=dt), leaf(6, 5, 8, dtype=dt)
        with torch.no_grad():
            zr[0, 0, :4] *= 1e-6
            zi[0, 0, :4] *= 1e-6
            zr[0, 1, :2] = 0.0
            zi[0, 1, :2] = 0.0
            zr[0, 2, :3] *= 0.3
            zi[0, 2, :3] *= 0.3
        gr, gi = torch.randn(6, 5, 8, dtype=dt), torch.randn(6, 5, 8, dtype=dt)
        cases = {"scalar": 0.5, "one": torch.tensor([0.3], dtype=dt, requires_grad=True),
                 "chan": (torch.rand(5, 1, dtype=dt) * 0.8 - 0.1).requires_grad_(True)}
        d[f"{tag}_mr_zr"], d[f"{tag}_mr_zi"], d[f"{tag}_mr_gr"], d[f"{tag}_mr_gi"] = npy(zr), npy(zi), npy(gr), npy(gi)
        for name, tau in cases.items():
            for t in (zr, zi):
                t.grad = None
            y = cplx.modrelu(C(zr, zi), tau)
            torch.autograd.backward((y.real, y.imag), (gr, gi))
            k = f"{tag}_mr_{name}_"
            d[k + "tau"] = np.asarray(tau, dtype=np.float64) if isinstance(tau, float) else npy(tau)
            d[k + "yr"], d[k + "yi"], d[k + "dzr"], d[k + "dzi"] = npy(y.real), npy(y.imag), npy(zr.grad), npy(zi.grad)
            if not isinstance(tau, float):
                d[k + "dtau"] = npy(tau.grad)
        # layout converters (pure data movement)
        x = torch.randn(3, 4, 10, dtype=dt)
        z = cplx.from_interleaved_real(x, True, -1)
        d[f"{tag}_il_x"], d[f"{tag}_il_re"], d[f"{tag}_il_im"] = npy(x), npy(z.real), npy(z.imag)
        d[f"{tag}_il_back"] = npy(cplx.to_interleaved_real(z, True, -1))
        d[f"{tag}_il_stack"] = npy(cplx.to_interleaved_real(z, False, -1))
        zc = cplx.from_concatenated_real(x, True, -1)
        d[f"{tag}_cat_re"], d[f"{tag}_cat_im"] = npy(zc.real), npy(zc.imag)
        d[f"{tag}_cat_back"] = npy(cplx.to_concatenated_real(zc, None, -1))
    # extension penalties: value + gradients on the mixed-regime parameters
    from cplxmodule.nn.relevance.extensions import complex as ext
    for tag, dt in DT.items():
        torch.manual_seed(17)
        for name, cls in (("cplx_vd_approx", ext.CplxLinearVDApprox), ("cplx_vd_scalefree", ext.CplxLinearVDScaleFree),
                          ("cplx_vd_bogus", ext.CplxLinearVDBogus)):
            layer = cls(24, 20).to(dt)
            wr, wi, ls2 = _mixed_vd_params(20, 24, dt)
            with torch.no_grad():
                layer.weight.real.copy_(wr)
                layer.weight.imag.copy_(wi)
                layer.log_sigma2.copy_(ls2)
            g = torch.randn(20, 24, dtype=dt)
            pen = layer.penalty
            (pen * g).sum().backward()
            k = f"{tag}_ext_{name}_"
            d[k + "wr"], d[k + "wi"], d[k + "ls2"], d[k + "g"] = npy(wr), npy(wi), npy(ls2), npy(g)
            d[k + "pen"] = npy(pen)
            d[k + "dls2"] = npy(layer.log_sigma2.grad)
            d[k + "dwr"], d[k + "dwi"] = npy(layer.weight.real.grad), npy(layer.weight.imag.grad)
    # 1-d complex convolution (cplx.conv1d): zeros and circular padding, values + gradients
    for tag, dt in DT.items():
        torch.manual_seed(19)
        for name, kw in (("s2p1", dict(stride=2, padding=1)), ("d2", dict(dilation=2)),
                         ("circ", dict(padding=3, padding_mode="circular")), ("g2", dict(groups=2, padding=2))):
            xr, xi = leaf(3, 4, 21, dtype=dt), leaf(3, 4, 21, dtype=dt)
            cg = 2 if name == "g2" else 4
            wr, wi = leaf(6, cg, 3, dtype=dt, scale=0.3), leaf(6, cg, 3, dtype=dt, scale=0.3)
            br, bi = leaf(6, dtype=dt), leaf(6, dtype=dt)
            y = cplx.conv1d(C(xr, xi), C(wr, wi), C(br, bi), **kw)
            gr, gi = torch.randn_like(y.real), torch.randn_like(y.imag)
            torch.autograd.backward((y.real, y.imag), (gr, gi))
            k = f"{tag}_c1_{name}_"
            for n, t in dict(xr=xr, xi=xi, wr=wr, wi=wi, br=br, bi=bi, yr=y.real, yi=y.imag, gr=gr, gi=gi).items():
                d[k + n] = npy(t)
            for n, t in dict(dxr=xr, dxi=xi, dwr=wr, dwi=wi, dbr=br, dbi=bi).items():
                d[k + n] = npy(t.grad)
    # abs-max pooling (values, gradients); one case has ties (a constant block)
    torch.manual_seed(13)
    pools = {"k2": dict(kernel_size=2), "k3s2p1": dict(kernel_size=3, stride=2, padding=1),
             "rect": dict(kernel_size=(3, 2), stride=(2, 1), padding=(1, 0), dilation=(1, 2), ceil_mode=True)}
    for tag, dt in DT.items():
        zr, zi = leaf(2, 3, 9, 11, dtype=dt), leaf(2, 3, 9, 11, dtype=dt)
        with torch.no_grad():
            zr[0, 0, :4, :4] = 1.0
            zi[0, 0, :4, :4] = -1.0
        d[f"{tag}_mp_zr"], d[f"{tag}_mp_zi"] = npy(zr), npy(zi)
        for name, kw in pools.items():
            for t in (zr, zi):
                t.grad = None
            y = cplx.max_pool2d(C(zr, zi), **kw)
            gr, gi = torch.randn_like(y.real), torch.randn_like(y.imag)
            torch.autograd.backward((y.real, y.imag), (gr, gi))
            k = f"{tag}_mp_{name}_"
            d[k + "yr"], d[k + "yi"], d[k + "gr"], d[k + "gi"] = npy(y.real), npy(y.imag), npy(gr), npy(gi)
            d[k + "dzr"], d[k + "dzi"] = npy(zr.grad), npy(zi.grad)
        x1r, x1i = leaf(2, 4, 17, dtype=dt), leaf(2, 4, 17, dtype=dt)
        y1 = cplx.max_pool1d(C(x1r, x1i), 3, 2, 1)
        d[f"{tag}_mp1_zr"], d[f"{tag}_mp1_zi"], d[f"{tag}_mp1_yr"], d[f"{tag}_mp1_yi"] = npy(x1r), npy(x1i), npy(y1.real), npy(y1.imag)
    # dropout: the reference drops (re, im) jointly and rescales by 1/(1-p); record one realisation's
    # invariants (which elements share a fate, the scale) -- the Bernoulli stream itself is torch's.
    torch.manual_seed(5)
    layer = CplxDropout(0.3)
    layer.train()
    z = C(torch.randn(64, 50), torch.randn(64, 50))
    y = layer(z)
    d["do_zr"], d["do_zi"], d["do_yr"], d["do_yi"] = npy(z.real), npy(z.imag), npy(y.real), npy(y.imag)
    save("extras", d)


def gen_r02():
    """Round-2 fixtures: abs(Cplx) and log_alpha as differentiable ops (exact zeros included), penalties
    under a SIGNED cotangent, the masked layers (nn/masked) and the mask plumbing."""
    from cplxmodule.nn import masked
    d = {}
    for tag, dt in DT.items():
        torch.set_default_dtype(dt)
        # --- abs(Cplx): value + gradient; zeros must give gradient 0 (cplx.py:183-192: stack + norm)
        torch.manual_seed(41)
        zr, zi = leaf(37, 29, dtype=dt), leaf(37, 29, dtype=dt)
        with torch.no_grad():
            zr.view(-1)[:5] = 0.0
            zi.view(-1)[:5] = 0.0
            zr.view(-1)[7] = 0.0
            zi.view(-1)[9] = 0.0
            zr.view(-1)[11], zi.view(-1)[11] = 1e-20, -1e-21
        g = torch.randn(37, 29)
        a = abs(C(zr, zi))
        (a * g).sum().backward()
        for nm, t in dict(zr=zr, zi=zi, g=g, abs=a, dzr=zr.grad, dzi=zi.grad).items():
            d[f"{tag}_abs_{nm}"] = npy(t)
        # --- log_alpha (differentiable) and penalties under a signed cotangent, all four kinds
        O, I = 24, 20
        torch.manual_seed(42)
        wr, wi, ls2 = _mixed_vd_params(O, I, dt)
        gs = torch.randn(O, I)                         # mixed signs
        for nm, t in dict(wr=wr, wi=wi, ls2=ls2, g=gs).items():
            d[f"{tag}_sg_{nm}"] = npy(t)
        kinds = {"real_vd": rel.LinearVD, "real_ard": rel.LinearARD,
                 "cplx_vd": rel.CplxLinearVD, "cplx_ard": rel.CplxLinearARD}
        for kind, cls in kinds.items():
            layer = cls(I, O, bias=False)
            with torch.no_grad():
                layer.log_sigma2.copy_(ls2)
                if kind.startswith("cplx"):
                    layer.weight.real.copy_(wr)
                    layer.weight.imag.copy_(wi)
                    wps = [layer.weight.real, layer.weight.imag]
                else:
                    layer.weight.copy_(wr)
                    wps = [layer.weight]
            k = f"{tag}_sg_{kind}_"
            grads = torch.autograd.grad((layer.penalty * gs).sum(), [layer.log_sigma2] + wps)
            for nm, t in zip(["dls2", "dwr", "dwi"], grads):
                d[k + "pen_" + nm] = npy(t)
            grads = torch.autograd.grad(-0.37 * layer.penalty.sum(), [layer.log_sigma2] + wps)
            for nm, t in zip(["dls2", "dwr", "dwi"], grads):
                d[k + "negsum_" + nm] = npy(t)
            la = layer.log_alpha
            grads = torch.autograd.grad((la * gs).sum(), [layer.log_sigma2] + wps)
            d[k + "la"] = npy(la)
            for nm, t in zip(["dls2", "dwr", "dwi"], grads):
                d[k + "la_" + nm] = npy(t)
        # --- masked layers: forward + gradients with a soft and a hard mask
        torch.manual_seed(43)
        B, I, O = 9, 16, 12
        lay = masked.CplxLinearMasked(I, O, bias=True)
        mask = (torch.rand(O, I) > 0.4).to(dt)
        soft = torch.rand(O, I) * mask
        x = cplx.randn(B, I)
        xr, xi = x.real.clone().requires_grad_(True), x.imag.clone().requires_grad_(True)
        gr, gi = torch.randn(B, O), torch.randn(B, O)
        for mname, m in (("hard", mask), ("soft", soft)):
            lay.mask = m
            for t in (xr, xi, *lay.parameters()):
                t.grad = None
            y = lay(C(xr, xi))
            torch.autograd.backward((y.real, y.imag), (gr, gi))
            k = f"{tag}_mk_cl_{mname}_"
            d[k + "mask"] = npy(m)
            for nm, t in dict(yr=y.real, yi=y.imag, dxr=xr.grad, dxi=xi.grad, dwr=lay.weight.real.grad,
                              dwi=lay.weight.imag.grad, dbr=lay.bias.real.grad, dbi=lay.bias.imag.grad).items():
                d[k + nm] = npy(t)
        for nm, t in dict(xr=xr, xi=xi, gr=gr, gi=gi, wr=lay.weight.real, wi=lay.weight.imag, br=lay.bias.real,
                          bi=lay.bias.imag).items():
            d[f"{tag}_mk_cl_{nm}"] = npy(t)
        d[f"{tag}_mk_cl_state_keys"] = np.array(sorted(lay.state_dict().keys()))
        d[f"{tag}_mk_cl_sparsity_hard"] = np.array([v for _, v in lay.sparsity(hard=True)])
        d[f"{tag}_mk_cl_sparsity_soft"] = np.array([v for _, v in lay.sparsity(hard=False)])
        rl = masked.LinearMasked(I, O, bias=True)
        rl.mask = soft
        xx = torch.randn(B, I).requires_grad_(True)
        y = rl(xx)
        y.backward(gr)
        for nm, t in dict(x=xx, w=rl.weight, b=rl.bias, y=y, dx=xx.grad, dw=rl.weight.grad, db=rl.bias.grad).items():
            d[f"{tag}_mk_rl_{nm}"] = npy(t)
        # conv masked layers (complex + real), hard mask
        torch.manual_seed(44)
        cl = masked.CplxConv2dMasked(4, 6, 3, padding=1)
        cm = (torch.rand(6, 4, 3, 3) > 0.5).to(dt)
        cl.mask = cm
        cx = cplx.randn(2, 4, 7, 8)
        cxr, cxi = cx.real.clone().requires_grad_(True), cx.imag.clone().requires_grad_(True)
        y = cl(C(cxr, cxi))
        cgr, cgi = torch.randn_like(y.real), torch.randn_like(y.imag)
        torch.autograd.backward((y.real, y.imag), (cgr, cgi))
        for nm, t in dict(mask=cm, xr=cxr, xi=cxi, gr=cgr, gi=cgi, wr=cl.weight.real, wi=cl.weight.imag,
                          br=cl.bias.real, bi=cl.bias.imag, yr=y.real, yi=y.imag, dxr=cxr.grad, dxi=cxi.grad,
                          dwr=cl.weight.real.grad, dwi=cl.weight.imag.grad, dbr=cl.bias.real.grad,
                          dbi=cl.bias.imag.grad).items():
            d[f"{tag}_mk_cc_{nm}"] = npy(t)
        rc = masked.Conv2dMasked(4, 6, 3, padding=1)
        rc.mask = cm
        rx = torch.randn(2, 4, 7, 8).requires_grad_(True)
        y = rc(rx)
        y.backward(cgr)
        for nm, t in dict(x=rx, w=rc.weight, b=rc.bias, y=y, dx=rx.grad, dw=rc.weight.grad, db=rc.bias.grad).items():
            d[f"{tag}_mk_rc_{nm}"] = npy(t)
        # --- binarize_masks (incl. the -0.0 clean-up, base.py:257-258) and deploy_masks naming
        torch.manual_seed(45)
        src = torch.nn.Sequential()
        src.add_module("a", rel.CplxLinearARD(6, 5))
        src.add_module("b", rel.LinearARD(5, 4))
        with torch.no_grad():
            src.a.log_sigma2.uniform_(-6, 4)
            src.b.log_sigma2.uniform_(-6, 4)
            src.a.weight.real[0, :3] = -src.a.weight.real[0, :3].abs()   # negative weights under a zero mask -> -0.0
        masks = rel.compute_ard_masks(src, hard=False, threshold=0.5)
        sd, hard = masked.binarize_masks(src.state_dict(), masks)
        d[f"{tag}_bz_mask_keys"] = np.array(sorted(masks.keys()))
        for k_, v in src.state_dict().items():
            d[f"{tag}_bz_in_{k_}"] = npy(v)
        for k_, v in masks.items():
            d[f"{tag}_bz_softmask_{k_}"] = npy(v)
        for k_, v in sd.items():
            d[f"{tag}_bz_out_{k_}"] = npy(v)
            d[f"{tag}_bz_signbit_{k_}"] = np.signbit(npy(v))
        for k_, v in hard.items():
            d[f"{tag}_bz_hard_{k_}"] = npy(v)
        dst = torch.nn.Sequential()
        dst.add_module("a", masked.CplxLinearMasked(6, 5))
        dst.add_module("b", masked.LinearMasked(5, 4))
        missing = dst.load_state_dict(sd, strict=False)
        d[f"{tag}_bz_missing"] = np.array(sorted(missing.missing_keys))
        d[f"{tag}_bz_unexpected"] = np.array(sorted(missing.unexpected_keys))
        masked.deploy_masks(dst, state_dict=hard)
        d[f"{tag}_bz_deployed_keys"] = np.array(sorted(dst.state_dict().keys()))
        d[f"{tag}_bz_named_masks"] = np.array([n for n, _ in masked.named_masks(dst)])
    torch.set_default_dtype(torch.float32)
    save("r02", d)


def gen_trajectory():
    """the reference's train -> sparsify -> fine-tune harness
    (tests/test_relevance.py:52-84 train step, :216-229 phase hand-off) on a small 2-layer model, first
    20 Adam steps of each phase: fixed init, fixed data, the noise tape of every stochastic forward,
    and per step loss / mse / kl / sparsity@tau; the hand-off state dicts and masks; final masks."""
    import torch.nn.functional as F
    from collections import OrderedDict
    from cplxmodule.nn import RealToCplx, CplxToReal, CplxModReLU
    from cplxmodule.nn import masked
    from cplxmodule.nn.utils.sparsity import sparsity
    import warnings
    warnings.simplefilter("ignore")
    d = {}
    N_STEPS, B, NF, NH, NO = 20, 32, 24, 10, 8      # real features 24 -> complex 12 -> 10 -> 4 -> real 8
    tau = 0.73105
    threshold = float(np.log(tau) - np.log(1 - tau))
    d["threshold"] = np.array(threshold)

    tape = []
    real_randn, real_randn_like = torch.randn, torch.randn_like

    def rec_randn(*a, **k):
        t = real_randn(*a, **k)
        tape.append(npy(t))
        return t

    def rec_randn_like(x, **k):
        t = real_randn_like(x, **k)
        tape.append(npy(t))
        return t

    def build(kind, l1, l2):
        if kind == "cplx":
            return torch.nn.Sequential(OrderedDict([
                ("cplx", RealToCplx()), ("l1", l1(NF // 2, NH, bias=True)), ("act", CplxModReLU(0.05)),
                ("l2", l2(NH, NO // 2, bias=False)), ("real", CplxToReal())]))
        return torch.nn.Sequential(OrderedDict([
            ("l1", l1(NF, NH, bias=True)), ("act", torch.nn.LeakyReLU()), ("l2", l2(NH, NO, bias=False))]))

    tracks = {
        "cplx_ard": ("cplx", [CplxLinear, rel.CplxLinearARD, masked.CplxLinearMasked], [0.0, 1e-1, 0.0], "mean"),
        "cplx_vd": ("cplx", [CplxLinear, rel.CplxLinearVD, masked.CplxLinearMasked], [0.0, 1e-1, 0.0], "mean"),
        "real_vd": ("real", [torch.nn.Linear, rel.LinearVD, masked.LinearMasked], [0.0, 2e-2, 0.0], "sum"),
    }
    for tname, (kind, layers, klws, reduction) in tracks.items():
        torch.manual_seed(1234)
        X = torch.randn(B, NF)
        y = -X[:, :NO].clone()
        d[f"{tname}_X"], d[f"{tname}_y"] = npy(X), npy(y)
        prev = None
        for ph, (cls, klw) in enumerate(zip(layers, klws)):
            torch.manual_seed(100 + ph)
            model = build(kind, cls, cls)
            k = f"{tname}_p{ph}_"
            if prev is not None:
                state_dict = prev.state_dict()
                masks = rel.compute_ard_masks(prev, hard=False, threshold=threshold)
                state_dict, masks = masked.binarize_masks(state_dict, masks)
                model.load_state_dict(state_dict, strict=False)
                model = masked.deploy_masks(model, state_dict=masks)
                for kk, v in masks.items():
                    d[k + "deploy_" + kk] = npy(v)
            if ph == 1:
                # start the sparsification phase from a spread of relevances so that masks are non-trivial
                with torch.no_grad():
                    for m in model.modules():
                        if hasattr(m, "log_sigma2"):
                            m.log_sigma2.uniform_(-8.0, 1.0)
            for kk, v in model.state_dict().items():
                d[k + "init_" + kk] = npy(v)
            model.train()
            optim = torch.optim.Adam(model.parameters())
            rows = []
            tape.clear()
            torch.randn, torch.randn_like = rec_randn, rec_randn_like
            try:
                for _ in range(N_STEPS):
                    optim.zero_grad()
                    y_pred = model(X)
                    mse = F.mse_loss(y_pred, y)
                    kl_d = sum(rel.penalties(model, reduction=reduction))
                    loss = mse + klw * kl_d
                    loss.backward()
                    optim.step()
                    f_sp = sparsity(model, hard=True, threshold=threshold)
                    rows.append([float(loss), float(mse), float(kl_d), float(f_sp)])
            finally:
                torch.randn, torch.randn_like = real_randn, real_randn_like
            d[k + "traj"] = np.array(rows, dtype=np.float64)      # [step, (loss, mse, kl, sparsity)]
            d[k + "klw"] = np.array(klw)
            d[k + "n_tape"] = np.array(len(tape))
            for j, t in enumerate(tape):
                d[k + f"tape_{j:03d}"] = t
            for kk, v in model.state_dict().items():
                d[k + "final_" + kk] = npy(v)
            fm = rel.compute_ard_masks(model, hard=False, threshold=threshold)
            for kk, v in fm.items():
                d[k + "finalmask_" + kk] = npy(v)
            prev = model
        d[f"{tname}_reduction"] = np.array(reduction)
    save("trajectory", d)


def gen_trajectory_conv():
    """BASELINE configs[4] in miniature, pinned to the reference: a Deep-Complex-Net style stack
    (CplxConv2d + CplxBatchNorm2d + split ReLU) x 2 with a complex linear head, trained with the reference's own
    modules through dense -> ARD -> masked (tests/test_relevance.py:98-253 / tests/test_mnist.py:199-245 harness
    shape), 12 Adam steps per phase on fixed synthetic complex images.  Track "head": only the head is
    CplxLinear -> CplxLinearARD -> CplxLinearMasked.  Track "conv": the second convolution is
    CplxConv2d -> CplxConv2dARD -> CplxConv2dMasked as well (the conv LRT path, nn/relevance/complex/base.py:120-135).
    Stored: data, every phase's initial state dict, the raw noise draws, per step (loss, ce, kl, sparsity@tau),
    batch-norm running statistics, final parameters and masks."""
    import torch.nn.functional as F
    from collections import OrderedDict
    from cplxmodule.nn import CplxToCplx, masked
    from cplxmodule.nn.utils.sparsity import sparsity
    import warnings
    warnings.simplefilter("ignore")
    d = {}
    N_STEPS, B, HW, C1, C2, NCLS = 12, 16, 12, 4, 8, 10
    tau = 0.73105
    threshold = float(np.log(tau) - np.log(1 - tau))
    d["threshold"] = np.array(threshold)
    tape = []
    real_randn, real_randn_like = torch.randn, torch.randn_like

    def rec_randn(*a, **k):
        t = real_randn(*a, **k)
        tape.append(npy(t))
        return t

    def rec_randn_like(x, **k):
        t = real_randn_like(x, **k)
        tape.append(npy(t))
        return t

    class Net(torch.nn.Module):
        def __init__(self, conv2, head):
            super().__init__()
            self.features = torch.nn.Sequential(OrderedDict([
                ("conv1", CplxConv2d(1, C1, 3, padding=1)), ("bn1", CplxBatchNorm2d(C1)), ("act1", CplxToCplx[torch.nn.ReLU]()),
                ("conv2", conv2(C1, C2, 3, stride=2, padding=1)), ("bn2", CplxBatchNorm2d(C2)), ("act2", CplxToCplx[torch.nn.ReLU]())]))
            self.head = head(C2 * (HW // 2) * (HW // 2), NCLS)

        def forward(self, x):
            z = self.features(x)
            z = self.head(cplx.Cplx(z.real.flatten(1), z.imag.flatten(1)))
            return abs(z)

    tracks = {
        "head": ([CplxConv2d] * 3, [CplxLinear, rel.CplxLinearARD, masked.CplxLinearMasked], [0.0, 2e-3, 0.0]),
        "conv": ([CplxConv2d, rel.CplxConv2dARD, masked.CplxConv2dMasked],
                 [CplxLinear, rel.CplxLinearARD, masked.CplxLinearMasked], [0.0, 2e-3, 0.0]),
    }
    g = torch.Generator().manual_seed(77)
    protos = torch.rand(NCLS, HW, HW, generator=g)
    labels = torch.randint(0, NCLS, (B,), generator=g)
    imgs = protos[labels] + 0.3 * torch.rand(B, HW, HW, generator=g)
    z = torch.fft.fft2(imgs) / HW
    xr, xi = z.real.unsqueeze(1).float().contiguous(), z.imag.unsqueeze(1).float().contiguous()
    d["xr"], d["xi"], d["labels"] = npy(xr), npy(xi), npy(labels)
    x = cplx.Cplx(xr, xi)
    for tname, (convs, heads, klws) in tracks.items():
        prev = None
        for ph, (conv2, head, klw) in enumerate(zip(convs, heads, klws)):
            torch.manual_seed(300 + ph)
            model = Net(conv2, head)
            k = f"{tname}_p{ph}_"
            if prev is not None:
                state_dict = prev.state_dict()
                masks = rel.compute_ard_masks(prev, hard=False, threshold=threshold)
                state_dict, masks = masked.binarize_masks(state_dict, masks)
                model.load_state_dict(state_dict, strict=False)
                if ph == 2:
                    model = masked.deploy_masks(model, state_dict=masks)
                    for kk, v in masks.items():
                        d[k + "deploy_" + kk] = npy(v)
            if ph == 1:
                with torch.no_grad():
                    for m in model.modules():
                        if hasattr(m, "log_sigma2"):
                            m.log_sigma2.uniform_(-8.0, 1.0)
            for kk, v in model.state_dict().items():
                d[k + "init_" + kk] = npy(v)
            model.train()
            optim = torch.optim.Adam(model.parameters(), lr=2e-3)
            rows = []
            tape.clear()
            torch.randn, torch.randn_like = rec_randn, rec_randn_like
            try:
                for _ in range(N_STEPS):
                    optim.zero_grad()
                    ce = F.cross_entropy(model(x), labels)
                    kl_d = sum(rel.penalties(model, reduction="sum"), torch.zeros(()))
                    loss = ce + klw * kl_d
                    loss.backward()
                    optim.step()
                    f_sp = sparsity(model, hard=True, threshold=threshold)
                    rows.append([float(loss), float(ce), float(kl_d), float(f_sp)])
            finally:
                torch.randn, torch.randn_like = real_randn, real_randn_like
            d[k + "traj"] = np.array(rows, dtype=np.float64)
            d[k + "klw"] = np.array(klw)
            d[k + "n_tape"] = np.array(len(tape))
            for j, t in enumerate(tape):
                d[k + f"tape_{j:03d}"] = t
            for kk, v in model.state_dict().items():
                d[k + "final_" + kk] = npy(v)
            fm = rel.compute_ard_masks(model, hard=False, threshold=threshold)
            for kk, v in fm.items():
                d[k + "finalmask_" + kk] = npy(v)
            prev = model
    save("trajectory_conv", d)


def gen_x3():
    """Shapes the float32 split-operand products take (every dimension a multiple of 32, ragged against the kernels'
    256 / 128 tiles): CplxLinear, CplxLinearVD and LinearVD forward + autograd gradients in float32 (the reference's own
    numbers) and float64 (the same seeds: what the float32 numbers approximate)."""
    d = {}
    B, I, O = 32, 64, 96
    for tag, dt in DT.items():
        torch.set_default_dtype(dt)
        torch.manual_seed(111)
        xr, xi = leaf(B, I, dtype=dt), leaf(B, I, dtype=dt)
        wr, wi = leaf(O, I, dtype=dt, scale=0.1), leaf(O, I, dtype=dt, scale=0.1)
        br, bi = leaf(O, dtype=dt), leaf(O, dtype=dt)
        gr, gi = torch.randn(B, O, dtype=dt), torch.randn(B, O, dtype=dt)
        k = f"{tag}_lin_"
        for nm, t in dict(xr=xr, xi=xi, wr=wr, wi=wi, br=br, bi=bi, gr=gr, gi=gi).items():
            d[k + nm] = npy(t)
        y = cplx.linear(C(xr, xi), C(wr, wi), C(br, bi))
        d[k + "yr"], d[k + "yi"] = npy(y.real), npy(y.imag)
        grads = torch.autograd.grad((y.real * gr).sum() + (y.imag * gi).sum(), [xr, xi, wr, wi, br, bi])
        for nm, g in zip(["dxr", "dxi", "dwr", "dwi", "dbr", "dbi"], grads):
            d[k + nm] = npy(g)
        # complex VD layer, training mode, noise tape recorded
        torch.manual_seed(121)
        layer = rel.CplxLinearVD(I, O, bias=True)
        with torch.no_grad():
            layer.log_sigma2.uniform_(-8, 1)
        x = cplx.randn(B, I)
        xr, xi = x.real.clone().requires_grad_(True), x.imag.clone().requires_grad_(True)
        gr, gi = torch.randn(B, O), torch.randn(B, O)
        layer.train()
        torch.manual_seed(177)
        y = layer(C(xr, xi))
        torch.manual_seed(177)
        tape = torch.randn(2, B, O)
        kl = sum(rel.penalties(layer))
        ps = [xr, xi, layer.weight.real, layer.weight.imag, layer.bias.real, layer.bias.imag, layer.log_sigma2]
        grads = torch.autograd.grad((y.real * gr).sum() + (y.imag * gi).sum() + 1e-2 * kl, ps)
        k = f"{tag}_cvd_"
        for nm, t in dict(xr=xr, xi=xi, wr=ps[2], wi=ps[3], br=ps[4], bi=ps[5], ls2=ps[6], gr=gr, gi=gi, tape=tape,
                          yr=y.real, yi=y.imag, kl=kl).items():
            d[k + nm] = npy(t)
        for nm, g in zip(["dxr", "dxi", "dwr", "dwi", "dbr", "dbi", "dls2"], grads):
            d[k + nm] = npy(g)
        # real VD layer
        torch.manual_seed(122)
        layer = rel.LinearVD(I, O, bias=True)
        with torch.no_grad():
            layer.log_sigma2.uniform_(-8, 1)
        x = torch.randn(B, I).requires_grad_(True)
        g = torch.randn(B, O)
        layer.train()
        torch.manual_seed(179)
        y = layer(x)
        torch.manual_seed(179)
        eps = torch.randn(B, O)
        kl = sum(rel.penalties(layer))
        grads = torch.autograd.grad((y * g).sum() + 1e-2 * kl, [x, layer.weight, layer.bias, layer.log_sigma2])
        k = f"{tag}_rvd_"
        for nm, t in dict(x=x, w=layer.weight, b=layer.bias, ls2=layer.log_sigma2, g=g, eps=eps, y=y, kl=kl,
                          dx=grads[0], dw=grads[1], db=grads[2], dls2=grads[3]).items():
            d[k + nm] = npy(t)
    torch.set_default_dtype(torch.float32)
    save("x3", d)


if __name__ == "__main__":
    torch.set_num_threads(1)  # reproducible summation order
    gens = dict(linear=gen_linear, lrt_linear=gen_lrt_linear, penalty=gen_penalty, conv=gen_conv,
                batchnorm=gen_batchnorm, api=gen_api, extras=gen_extras, bilinear=gen_bilinear, conv3d=gen_conv3d, conv_transpose=gen_conv_transpose,
                r02=gen_r02, trajectory=gen_trajectory, trajectory_conv=gen_trajectory_conv, x3=gen_x3)
    for name in (sys.argv[1:] or list(gens)):
        gens[name]()
