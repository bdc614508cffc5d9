"""hipGraph capture of a stochastic training step: the noise position lives on the device
("philox-device" mode), so every replay of the captured fwd+bwd draws fresh noise, and each
replay is bit-identical to the eager step taken from the same stream position."""
import pytest
import torch

from gpu_util import DEV

pytestmark = pytest.mark.gpu


def _warm_and_capture(step, modules):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    for m in modules:
        m.zero_grad(set_to_none=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    return g, out


@pytest.mark.parametrize("kind", ["cplx", "real"])
def test_graph_replay_matches_eager(kind):
    from cplxmodule_amd import Cplx
    from cplxmodule_amd.nn import relevance as rel
    torch.manual_seed(3)
    rel.noise.manual_seed(11)
    rel.noise.set_mode("philox-device")
    try:
        if kind == "cplx":
            layer = rel.CplxLinearVD(96, 80).to(DEV)
            x = Cplx(torch.randn(32, 96, device=DEV), torch.randn(32, 96, device=DEV))
        else:
            layer = rel.LinearARD(96, 80).to(DEV)
            x = torch.randn(32, 96, device=DEV)

        def step():
            y = layer(x)
            sq = (y.real ** 2).sum() + (y.imag ** 2).sum() if kind == "cplx" else (y ** 2).sum()
            loss = sq + 1e-2 * sum(rel.penalties(layer))
            loss.backward()
            return y

        g, y = _warm_and_capture(step, [layer])
        state = rel.noise.device_state(torch.device(DEV))
        outs, grads, offs = [], [], []
        for _ in range(3):
            offs.append(int(state[1].item()))
            layer.log_sigma2.grad.zero_()
            g.replay()
            torch.cuda.synchronize()
            outs.append((y.real.clone(), y.imag.clone()) if kind == "cplx" else (y.clone(),))
            grads.append(layer.log_sigma2.grad.clone())
        assert offs == [offs[0], offs[0] + 1, offs[0] + 2]          # one offset per replay
        assert not torch.equal(outs[0][0], outs[1][0])               # fresh noise each replay
        assert not torch.equal(outs[1][0], outs[2][0])

        # the eager host-counter path from the same stream position gives the same bits
        rel.noise.set_mode("philox")
        for k in range(3):
            rel.noise.counter = offs[k] - 1
            layer.zero_grad(set_to_none=True)
            ye = step()
            got = (ye.real, ye.imag) if kind == "cplx" else (ye,)
            for a, b in zip(got, outs[k]):
                assert torch.equal(a, b)
            assert torch.equal(layer.log_sigma2.grad, grads[k])
    finally:
        rel.noise.set_mode("philox")
