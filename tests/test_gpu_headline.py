"""The headline step (BASELINE configs[1] + VD: CplxLinearVD 4096 -> 4096, bf16, batch 8192) value-checked at its own shape,
as the hipGraph replay bench.py times and as eager launches (tests/headline_check.py)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("graph", (True, False))
def test_headline_step_values(graph):
    from headline_check import check_headline_step
    res = check_headline_step(graph=graph, full_kl=graph)
    assert {"y.real", "dX.real (G conj W + 2 x ga)", "dW.real (+ KL accumulate) @ KL weight 1e-3", "dlog_sigma2 (+ KL accumulate) @ KL weight 3000"} <= set(res), sorted(res)
    for name, (err, tol) in res.items():
        assert err <= tol, (name, err, tol)


def test_headline_check_small_shape_eager():
    """The same check at a shape the one-tile kernels take (partial tiles), so that a failure at full size can be told
    from a failure of the checker."""
    from headline_check import check_headline_step
    check_headline_step(B=512, F=256, graph=False, full_kl=True)
