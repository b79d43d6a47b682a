"""Size-independent properties of the hot path at benchmark-scale batches (no oracle needed): worlds are independent, so
the step must be deterministic, equivariant under any permutation of the worlds, independent of the batch size a world is
computed in, and the backward pass must be linear in the cotangent.  Checked on the metric configuration (Atlas-20 on the
ground, 8 contacts) at B = 16384 and on the free-falling model."""
import numpy as np
import pytest

from util import contact_inputs

pytestmark = pytest.mark.gpu


def _fwd_bwd(md, s, a, g):
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    world = na.World(md, device="cuda:0")
    st = torch.tensor(s, device="cuda:0", requires_grad=True)
    at = torch.tensor(a, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at)
    out.backward(torch.tensor(g, device="cuda:0"))
    return out.detach().cpu().numpy(), st.grad.cpu().numpy(), at.grad.cpu().numpy(), world.last_status.cpu().numpy()


@pytest.mark.parametrize("ground", [True, False])
def test_determinism_permutation_equivariance_and_batch_independence(ground):
    import nimblephysics_amd as na
    B = 16384
    md, s, a = contact_inputs("atlas20", B, 61, joint_noise=0.01)      # a few per cent of the worlds leave stage 0
    if not ground:
        md = na.atlas("atlas20", ground=False)
    g = np.random.default_rng(2).normal(0, 1, s.shape)
    r1 = _fwd_bwd(md, s, a, g)
    r2 = _fwd_bwd(md, s, a, g)
    for x, y in zip(r1, r2):
        assert np.array_equal(x, y)                                      # deterministic (no atomics in the data path)
    perm = np.random.default_rng(3).permutation(B)
    rp = _fwd_bwd(md, s[perm], a[perm], g[perm])
    for x, y in zip(r1, rp):
        assert np.array_equal(x[perm], y)                                # worlds do not see each other
    sub = np.sort(np.random.default_rng(4).choice(B, 64, replace=False))
    rs = _fwd_bwd(md, s[sub], a[sub], g[sub])
    for x, y in zip(r1, rs):
        assert np.array_equal(x[sub], y)                                 # same world, any batch size, same bits
    for odd in (1, 3, 63):                                               # odd batches: the last wavefront of the row kernel (two worlds per
        sub = np.sort(np.random.default_rng(5 + odd).choice(B, odd, replace=False))   # wavefront) works with an idle second half
        rs = _fwd_bwd(md, s[sub], a[sub], g[sub])
        for x, y in zip(r1, rs):
            assert np.array_equal(x[sub], y)


def test_backward_is_linear_in_the_cotangent():
    import torch
    import nimblephysics_amd as na
    B = 16384
    md, s, a = contact_inputs("atlas20", B, 62)
    world = na.World(md, device="cuda:0")
    s_soa, a_soa = world.to_soa(torch.tensor(s, device="cuda:0")), world.to_soa(torch.tensor(a, device="cuda:0"))
    nxt, saved, status = world.step_soa(s_soa, a_soa)
    rng = np.random.default_rng(5)
    g1 = torch.tensor(rng.normal(0, 1, (s.shape[1], B)), device="cuda:0")
    g2 = torch.tensor(rng.normal(0, 1, (s.shape[1], B)), device="cuda:0")
    a1, a2 = 0.7, -1.9
    gs1, ga1 = world.backward_soa(saved, g1)
    gs2, ga2 = world.backward_soa(saved, g2)
    gs, ga = world.backward_soa(saved, a1 * g1 + a2 * g2)
    for lhs, rhs in ((gs, a1 * gs1 + a2 * gs2), (ga, a1 * ga1 + a2 * ga2)):
        scale = rhs.abs().amax(0).clamp_min(1e-300)
        assert ((lhs - rhs).abs().amax(0) / scale).max().item() < 1e-9
