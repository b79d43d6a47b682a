"""World.getStateJacobian / getActionJacobian (World.cpp:2210-2243, tested upstream by
unit/test_RL_API.cpp:146-189 against finite differences): dense Jacobians assembled from 2n vector-Jacobian products,
checked against central finite differences of the GPU step itself and against J^T g = backward(g)."""
import numpy as np
import pytest

from util import contact_inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("contact", [False, True])
def test_state_and_action_jacobians_vs_finite_differences(contact):
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    B = 8
    md, s0, a0 = contact_inputs("atlas20", B, 51)
    if not contact:
        md = na.atlas("atlas20", ground=False)
    world = na.World(md, device="cuda:0")
    n2, k = s0.shape[1], a0.shape[1]
    st = torch.tensor(s0, device="cuda:0", requires_grad=True)
    at = torch.tensor(a0, device="cuda:0", requires_grad=True)
    g = np.random.default_rng(1).normal(0, 1, s0.shape)
    out = timestep(world, st, at)
    JS = world.getStateJacobian().cpu().numpy()          # [B, 2n, 2n]
    JA = world.getActionJacobian().cpu().numpy()         # [B, 2n, k]
    out.backward(torch.tensor(g, device="cuda:0"))
    assert JS.shape == (B, n2, n2) and JA.shape == (B, n2, k)
    # J^T g == backward(g)
    assert np.allclose(np.einsum("bij,bi->bj", JS, g), st.grad.cpu().numpy(), rtol=1e-9, atol=1e-9 * np.abs(st.grad.cpu().numpy()).max())
    assert np.allclose(np.einsum("bij,bi->bj", JA, g), at.grad.cpu().numpy(), rtol=1e-9, atol=1e-9 * max(np.abs(at.grad.cpu().numpy()).max(), 1e-30))

    def f(s, a):
        w2 = na.World(md, device="cuda:0")                # cold LCP start like the differentiated step
        return timestep(w2, torch.tensor(s, device="cuda:0"), torch.tensor(a, device="cuda:0")).cpu().numpy()

    eps = 1e-6
    cols = list(range(6, n2, 5)) if contact else list(range(0, n2, 3))   # with contact the free-joint rotation columns sit on LCP kinks less often for joint DOFs
    for j in cols:
        d = np.zeros_like(s0); d[:, j] = eps
        fd = (f(s0 + d, a0) - f(s0 - d, a0)) / (2 * eps)
        scale = max(np.abs(JS[:, :, j]).max(), 1.0)
        assert np.abs(fd - JS[:, :, j]).max() <= (2e-5 if contact else 2e-6) * scale, (j, np.abs(fd - JS[:, :, j]).max(), scale)
    for j in range(0, k, 4):
        d = np.zeros_like(a0); d[:, j] = eps
        fd = (f(s0, a0 + d) - f(s0, a0 - d)) / (2 * eps)
        scale = max(np.abs(JA[:, :, j]).max(), 1e-3)
        assert np.abs(fd - JA[:, :, j]).max() <= 2e-5 * scale
