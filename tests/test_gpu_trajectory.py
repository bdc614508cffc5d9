"""the reference's train -> sparsify -> fine-tune harness replayed on the GPU.

tests/golden/trajectory.npz (oracle/gen_golden.py:gen_trajectory, run against /root/reference) holds,
for three tracks (complex ARD, complex VD with the exact Ei penalty, real VD) and the three phases
dense -> VD/ARD -> masked of tests/test_relevance.py:52-84 / :216-229: the fixed data, every phase's
initial state dict (after the reference's own binarize / load / deploy hand-off), the raw noise draw of
every stochastic forward, and per Adam step (loss, mse, kl, sparsity@tau), plus the final parameters
and masks.  Here the same model is built from cplxmodule_amd, started from the same state, fed the
same noise tape, and must reproduce the trajectory to 1e-5 relative and the final masks exactly."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NF, NH, NO = 24, 10, 8
RTOL = 1e-5


def _build(kind, cls):
    from cplxmodule_amd import nn
    if kind == "cplx":
        return torch.nn.Sequential(OrderedDict([
            ("cplx", nn.RealToCplx()), ("l1", cls(NF // 2, NH, bias=True)), ("act", nn.CplxModReLU(0.05)),
            ("l2", cls(NH, NO // 2, bias=False)), ("real", nn.CplxToReal())]))
    return torch.nn.Sequential(OrderedDict([
        ("l1", cls(NF, NH, bias=True)), ("act", torch.nn.LeakyReLU()), ("l2", cls(NH, NO, bias=False))]))


@pytest.mark.parametrize("track", ["cplx_ard", "cplx_vd", "real_vd"])
def test_trajectory_matches_reference(golden, track):
    import torch.nn.functional as F
    from gpu_util import T, N
    from cplxmodule_amd import nn
    from cplxmodule_amd.nn import masked, relevance as rel
    from cplxmodule_amd.nn.relevance import noise
    from cplxmodule_amd.nn.utils.sparsity import sparsity
    g = golden("trajectory")
    threshold = float(g["threshold"])
    kind = "cplx" if track.startswith("cplx") else "real"
    layers = {"cplx_ard": [nn.CplxLinear, rel.CplxLinearARD, masked.CplxLinearMasked],
              "cplx_vd": [nn.CplxLinear, rel.CplxLinearVD, masked.CplxLinearMasked],
              "real_vd": [torch.nn.Linear, rel.LinearVD, masked.LinearMasked]}[track]
    reduction = str(g[f"{track}_reduction"])
    X, y = T(g[f"{track}_X"]), T(g[f"{track}_y"])
    report = []
    prev_mode = noise.mode
    try:
        for ph, cls in enumerate(layers):
            k = f"{track}_p{ph}_"
            if kind == "real" and ph == 0:
                # torch.nn.Linear is not ours: the dense real phase only provides the start of phase 1
                continue
            model = _build(kind, cls).to("cuda")
            state = {n[len(k) + 5:]: T(v) for n, v in g.items() if n.startswith(k + "init_")}
            res = model.load_state_dict(state, strict=True)
            assert not res.missing_keys and not res.unexpected_keys
            if ph == 2:      # the masks arrived through the state dict (BaseMasked._load_from_state_dict)
                for n, m in masked.named_masks(model):
                    np.testing.assert_array_equal(N(m), g[k + "deploy_" + n + ".mask"])
            n_tape = int(g[k + "n_tape"])
            noise.set_tape([torch.from_numpy(g[k + f"tape_{j:03d}"]) for j in range(n_tape)])
            klw = float(g[k + "klw"])
            model.train()
            optim = torch.optim.Adam(model.parameters())
            ref = g[k + "traj"]
            rows = []
            for step in range(ref.shape[0]):
                optim.zero_grad()
                y_pred = model(X)
                mse = F.mse_loss(y_pred, y)
                kl_d = sum(rel.penalties(model, reduction=reduction))
                loss = mse + klw * kl_d
                loss.backward()
                optim.step()
                rows.append([float(loss), float(mse), float(kl_d), float(sparsity(model, hard=True, threshold=threshold))])
            rows = np.array(rows)
            assert not noise._tape, "the whole reference tape must have been consumed"
            err = np.abs(rows - ref) / np.maximum(np.abs(ref), 1e-12)
            err[ref == 0] = np.abs(rows - ref)[ref == 0]
            report.append((ph, err.max(axis=0)))
            np.testing.assert_allclose(rows[:, :3], ref[:, :3], rtol=RTOL, atol=1e-7,
                                       err_msg=f"{track} phase {ph}: loss / mse / kl trajectory")
            np.testing.assert_array_equal(rows[:, 3], ref[:, 3], err_msg=f"{track} phase {ph}: sparsity@tau")
            fm = rel.compute_ard_masks(model, hard=False, threshold=threshold)
            for n, m in fm.items():
                np.testing.assert_array_equal(N(m), g[k + "finalmask_" + n], err_msg=f"final mask {n}")
            for n, v in model.state_dict().items():
                refv = g[k + "final_" + n]
                np.testing.assert_allclose(N(v), refv, rtol=2e-4, atol=2e-6 * max(1.0, np.abs(refv).max()),
                                           err_msg=f"final parameter {n}")
    finally:
        noise.set_mode(prev_mode)
    print(f"\n[trajectory {track}] max relative error per phase (loss, mse, kl, sparsity): " +
          "; ".join(f"p{ph}: {np.array2string(e, precision=2)}" for ph, e in report))
