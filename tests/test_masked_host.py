"""nn.masked host logic (CPU): mask setter, deploy / binarize, state-dict contract."""
import pytest
import torch

from cplxmodule_amd import nn
from cplxmodule_amd.nn import masked, relevance as rel
from cplxmodule_amd.nn.utils import sparsity


def test_mask_setter_and_weight_masked():
    layer = masked.CplxLinearMasked(4, 3)
    assert not layer.is_sparse and not masked.is_sparse(layer)
    with pytest.raises(RuntimeError, match="has no sparsity mask"):
        layer.weight_masked
    with pytest.raises(TypeError):
        layer.mask = 1.0
    layer.mask = torch.tensor([1.0, 0.0, 1.0, 0.0])          # broadcast to the weight shape
    assert layer.is_sparse and layer.mask.shape == (3, 4) and layer.mask.is_contiguous()
    wm = layer.weight_masked
    assert float(wm.real[:, 1].abs().max()) == 0 and float(wm.imag[:, 3].abs().max()) == 0
    assert "mask" in layer.state_dict()
    layer.mask = None
    assert not layer.is_sparse and "mask" not in layer.state_dict()


def test_state_dict_contract():
    src = masked.LinearMasked(5, 2)
    src.mask = (torch.rand(2, 5) > 0.5).float()
    dst = masked.LinearMasked(5, 2)
    dst.load_state_dict(src.state_dict())
    assert torch.equal(dst.mask, src.mask)
    dense = masked.LinearMasked(5, 2)
    with pytest.raises(RuntimeError, match="mask"):
        dst.load_state_dict(dense.state_dict())              # strict: absent mask is reported
    dst.load_state_dict(dense.state_dict(), strict=False)    # not strict: mask kept
    assert dst.is_sparse


def test_deploy_binarize_pipeline_names():
    ard = torch.nn.Sequential(rel.CplxLinearARD(6, 5), torch.nn.Sequential(rel.CplxLinearARD(5, 3)))
    fine = torch.nn.Sequential(masked.CplxLinearMasked(6, 5),
                               torch.nn.Sequential(masked.CplxLinearMasked(5, 3)))
    masks = {"0.mask": (torch.rand(5, 6) > 0.5).float() * 0.7, "1.0.mask": torch.ones(3, 5)}
    sd, hard = masked.binarize_masks(ard.state_dict(), masks)
    assert set(hard["0.mask"].unique().tolist()) <= {0.0, 1.0}
    dropped = masks["0.mask"] == 0
    assert float(sd["0.weight.real"][dropped].abs().max()) == 0
    assert not torch.signbit(sd["0.weight.real"][dropped]).any()
    missing = fine.load_state_dict(sd, strict=False)
    assert "0.log_sigma2" in missing.unexpected_keys
    masked.deploy_masks(fine, state_dict=hard)
    assert [n for n, _ in masked.named_masks(fine)] == ["0", "1.0"]
    assert masked.is_sparse(fine[0]) and torch.equal(fine[0].mask, hard["0.mask"])
    s = sparsity(fine, hard=True)
    assert 0 < s < 1
    masked.deploy_masks(fine, state_dict={}, reset=True)
    assert not masked.is_sparse(fine[0])
    assert nn.masked is masked


def test_binarize_and_deploy_match_reference_fixture(golden):
    """binarize_masks (incl. the -0.0 clean-up, nn/masked/base.py:257-258), the hand-off's missing /
    unexpected keys and deploy_masks naming against what the reference produced (golden r02)."""
    import numpy as np
    g = golden("r02")
    pre = "f32_bz_"
    sd = {n[len(pre) + 3:]: torch.from_numpy(v) for n, v in g.items() if n.startswith(pre + "in_")}
    masks = {n[len(pre) + 9:]: torch.from_numpy(v) for n, v in g.items() if n.startswith(pre + "softmask_")}
    out, hard = masked.binarize_masks(sd, masks)
    for n, v in out.items():
        np.testing.assert_array_equal(v.numpy(), g[pre + "out_" + n])
        np.testing.assert_array_equal(np.signbit(v.numpy()), g[pre + "signbit_" + n])
    for n, v in hard.items():
        np.testing.assert_array_equal(v.numpy(), g[pre + "hard_" + n])
    dst = torch.nn.Sequential()
    dst.add_module("a", masked.CplxLinearMasked(6, 5))
    dst.add_module("b", masked.LinearMasked(5, 4))
    res = dst.load_state_dict(out, strict=False)
    assert sorted(res.missing_keys) == list(g[pre + "missing"])
    assert sorted(res.unexpected_keys) == list(g[pre + "unexpected"])
    masked.deploy_masks(dst, state_dict=hard)
    assert sorted(dst.state_dict().keys()) == list(g[pre + "deployed_keys"])
    assert [n for n, _ in masked.named_masks(dst)] == list(g[pre + "named_masks"])
