"""dL/dmass of the device path (nbl_backward_inertia, closed form) against central differences of the CPU oracle's step with
respect to the same parameters - which is how the reference itself obtains / checks this gradient
(BackpropSnapshot::finiteDifferenceMassVelJacobian, BackpropSnapshot.cpp:2407-; Skeleton.cpp:1826-1829, 2078-2081)."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_fd(md, entries, s, a, g, eps_rel=1e-6, fd_relative=False):
    """d(g . step(s, a; theta))/dtheta per world by central differences of the oracle, [B, dims]."""
    from oracle import OracleWorld
    from nimblephysics_amd.mass import WithRespectToMass
    md = copy.deepcopy(md)
    w = WithRespectToMass(md)
    for body, t in entries:
        w.registerNode(body, t)
    x0 = w.get()
    out = np.zeros((s.shape[0], w.dim()))
    for p in range(w.dim()):
        # (fd_relative: the step relative to the parameter itself - masses x 1e-2 leave inertia entries of 1e-6, below an absolute step)
        eps = eps_rel * (max(1.0, abs(x0[p])) if not fd_relative or x0[p] == 0 else 100.0 * abs(x0[p]))
        vals = []
        for sgn in (+1, -1):
            x = x0.copy(); x[p] += sgn * eps
            w.set(x)
            vals.append(OracleWorld(md).step_batch(s, a, threads=8)["next"])
        w.set(x0)
        out[:, p] = ((vals[0] - vals[1]) * g).sum(1) / (2 * eps)
    return out


def _device(md, entries, s, a, g, masses=None):
    import torch
    import nimblephysics_amd as na
    world = na.World(md, device="cuda:0")
    for body, t in entries:
        world.tuneMass(body, t)
    if masses is not None:
        world.setMasses(masses)
    st = world.to_soa(torch.tensor(s, device="cuda:0")); at = world.to_soa(torch.tensor(a, device="cuda:0"))
    nxt, saved, status = world.step_soa(st, at)
    world.backward_soa(saved, world.to_soa(torch.tensor(g, device="cuda:0")))
    gm = world.backward_inertia_soa(saved, s.shape[0])
    return gm.T.cpu().numpy(), world.from_soa(nxt).cpu().numpy(), world


def _check(md, entries, s, a, seed, tol=2e-6, second_eps=None, max_outliers=0.0, fd_relative=False):
    g = np.random.default_rng(seed).normal(0, 1, s.shape)
    dev, _, _ = _device(md, entries, s, a, g)
    ref = _oracle_fd(md, entries, s, a, g, fd_relative=fd_relative)
    if second_eps is not None:
        # with contacts a perturbed oracle step can land on another LCP branch (status bit 0x100, non-unique solutions):
        # that difference quotient is ~1e5 at one step size and fine at the other.  Keep, per entry, the quotient closer
        # to the device value: an entry passes if EITHER step size confirms it.
        ref2 = _oracle_fd(md, entries, s, a, g, eps_rel=second_eps)
        ref = np.where(np.abs(ref2 - dev) < np.abs(ref - dev), ref2, ref)
    # per-parameter scale, floored: a fully clamped body (a foot held by 4 sticking contacts) has a gradient that is
    # exactly zero on the device and finite-difference noise in the oracle
    scale = np.maximum(np.abs(ref).max(0), 1e-3 * np.abs(ref).max()) + 1e-9
    err = np.abs(dev - ref) / scale
    # max_outliers: share of (world, parameter) entries allowed to miss: a finite difference of the ORACLE across an LCP kink
    # (the perturbed step resolves on another branch at both step sizes) is not a derivative
    bad = err >= tol
    assert bad.mean() <= max_outliers, (float(bad.mean()), err.max(), np.unravel_index(err.argmax(), err.shape), dev[0], ref[0])


def test_pendulum_and_cartpole_all_entry_types():
    from nimblephysics_amd.mass import WrtMassBodyNodeEntryType as T
    from util import cfg_inputs
    md, s, a = cfg_inputs("pendulum", 64, 21)
    md = copy.deepcopy(md); md.bodies[0].com = (0.1, -0.3, 0.05)
    for t in (T.INERTIA_MASS, T.INERTIA_COM, T.INERTIA_DIAGONAL, T.INERTIA_OFF_DIAGONAL, T.INERTIA_FULL):
        _check(md, [(0, t)], s, a, 22)
    mu = copy.deepcopy(md); mu.bodies[0].beta = (0.5, -1.5, 0.25); mu.bodies[0].com = (0.1, -0.3, 0.05)   # COM = beta * 0.2
    _check(mu, [(0, T.INERTIA_COM_MU)], s, a, 22)
    md, s, a = cfg_inputs("cartpole", 64, 23)
    _check(md, [(0, T.INERTIA_MASS), (1, T.INERTIA_FULL)], s, a, 24)


def test_atlas_free_fall_mass_and_com():
    from nimblephysics_amd.mass import WrtMassBodyNodeEntryType as T
    from util import cfg_inputs
    md, s, a = cfg_inputs("atlas20", 32, 25)
    names = [b.name for b in md.bodies]
    welded = [i for i, b in enumerate(md.bodies) if b.joint_type == "weld"]
    _check(md, [(0, T.INERTIA_MASS), (names[5], T.INERTIA_COM), (welded[0], T.INERTIA_MASS), (len(names) - 1, T.INERTIA_DIAGONAL)], s, a, 26)


def test_standing_atlas_with_contacts():
    """The clamping contact rows change the adjoint (A_c^T v' = 0): the same lambda = dL/dtau carries it."""
    from nimblephysics_amd.mass import WrtMassBodyNodeEntryType as T
    from util import contact_inputs
    md, s, a = contact_inputs("atlas20", 32, 27)
    feet = [i for i, b in enumerate(md.bodies) if "foot" in b.name and b.joint_type != "weld"]
    _check(md, [(0, T.INERTIA_MASS), (feet[0], T.INERTIA_MASS), (feet[-1], T.INERTIA_COM), (3, T.INERTIA_FULL)], s, a, 28, tol=2e-4, second_eps=1e-5)


def test_box_stack_with_contacts():
    from nimblephysics_amd.mass import WrtMassBodyNodeEntryType as T
    from util import box_stack_inputs
    md, s, a = box_stack_inputs(64, 29)
    # the 0.1 kg cubes of box_stacking.skel slide (friction rows on their bounds): more worlds sit next to a kink than with the
    # Atlas feet; 2 % of the quotients may land across one
    _check(md, [("box1", T.INERTIA_MASS), ("box2", T.INERTIA_FULL)], s, a, 30, tol=2e-4, second_eps=1e-5, max_outliers=0.02)


def test_set_masses_changes_the_step_like_the_oracle():
    """World::setMasses re-uploads the body constants: the next step must equal the oracle's on the edited model."""
    from nimblephysics_amd.mass import WrtMassBodyNodeEntryType as T, WithRespectToMass
    from oracle import OracleWorld
    from util import contact_inputs
    md, s, a = contact_inputs("atlas20", 64, 31)
    entries = [(0, T.INERTIA_MASS), (4, T.INERTIA_FULL)]
    md2 = copy.deepcopy(md)
    w = WithRespectToMass(md2)
    for b, t in entries:
        w.registerNode(b, t)
    x = w.get()
    x[0] *= 1.3; x[1] *= 0.7; x[2:5] += 0.01; x[5:8] *= 1.2
    w.set(x)
    ref = OracleWorld(md2).step_batch(s, a, threads=8)["next"]
    g = np.zeros_like(s)
    _, nxt, world = _device(md, entries, s, a, g, masses=x)
    assert np.abs(nxt - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())
    assert np.allclose(world.getMasses().numpy(), x)
    base = OracleWorld(md).step_batch(s, a, threads=8)["next"]
    assert np.abs(ref - base).max() > 1e-6      # the edit matters


def test_timestep_mass_argument_autograd():
    """timestep(world, state, action, mass) returns dMass (reference timestep.py:33-35, 57-60); the shared mass vector
    receives the sum over the worlds."""
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.mass import WrtMassBodyNodeEntryType as T
    from nimblephysics_amd.timestep import timestep
    from util import cfg_inputs
    md, s, a = cfg_inputs("cartpole", 16, 33)
    world = na.World(md, device="cuda:0")
    world.tuneMass(1, T.INERTIA_MASS)
    world.tuneMass(0, T.INERTIA_MASS)
    assert world.getMassDims() == 2
    mass = torch.tensor([1.5, 0.8], dtype=torch.float64, requires_grad=True)
    st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0")
    out = timestep(world, st, at, mass)
    g = np.random.default_rng(34).normal(0, 1, s.shape)
    out.backward(torch.tensor(g, device="cuda:0"))
    md2 = copy.deepcopy(md)
    md2.bodies[1].inertia = tuple(x * 1.5 / md2.bodies[1].mass for x in md2.bodies[1].inertia); md2.bodies[1].mass = 1.5
    md2.bodies[0].inertia = tuple(x * 0.8 / md2.bodies[0].mass for x in md2.bodies[0].inertia); md2.bodies[0].mass = 0.8
    ref = _oracle_fd(md2, [(1, T.INERTIA_MASS), (0, T.INERTIA_MASS)], s, a, g).sum(0)
    assert mass.grad.shape == (2,)
    assert np.abs(mass.grad.numpy() - ref).max() < 2e-6 * np.abs(ref).max()


def test_rollout_mass_gradient_equals_the_chain_of_timesteps():
    """rollout(..., mass=) sums the per-step inertia gradients on the device (nbl_rollout_backward_inertia); the chain of
    timestep(..., mass) calls through autograd must give the same state / action / mass gradients."""
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.mass import WrtMassBodyNodeEntryType as T
    from nimblephysics_amd.timestep import rollout, timestep
    from util import contact_inputs
    md, s, a = contact_inputs("atlas20", 128, 41)
    Tn = 6
    rng = np.random.default_rng(42)
    acts = rng.normal(0, 0.1, (128, Tn, a.shape[1]))
    gT = rng.normal(0, 1, (128, Tn + 1, s.shape[1]))
    res = []
    for mode in ("rollout", "chain"):
        world = na.World(md, device="cuda:0")
        world.tuneMass(0, T.INERTIA_MASS); world.tuneMass(3, T.INERTIA_COM)
        mass = world.getMasses().clone().requires_grad_(True)
        st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(acts, device="cuda:0", requires_grad=True)
        if mode == "rollout":
            states = rollout(world, st, at, warm_start=False, mass=mass)
        else:
            xs = [st]
            for t in range(Tn):
                world.reset_lcp_cache()
                xs.append(timestep(world, xs[-1], at[:, t], mass))
            states = torch.stack(xs, 1)
        (states * torch.tensor(gT, device="cuda:0")).sum().backward()
        res.append((states.detach().cpu().numpy(), st.grad.cpu().numpy(), at.grad.cpu().numpy(), mass.grad.numpy()))
    for x, y in zip(res[0], res[1]):
        assert np.abs(x - y).max() <= 1e-12 * max(1.0, np.abs(y).max())
    assert np.abs(res[0][3]).max() > 0


def test_clone_and_action_space_keep_the_mass_registration():
    """World::clone copies model, action space, registered mass parameters and current masses; setActionSpace re-uploads the
    model without losing the registration."""
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.mass import WrtMassBodyNodeEntryType as T
    from util import cfg_inputs
    md, s, a = cfg_inputs("cartpole", 8, 51)
    w = na.World(md, device="cuda:0")
    w.tuneMass(1, T.INERTIA_MASS)
    w.setMasses([2.5])
    w.setActionSpace([0])
    assert w.getMassDims() == 1 and float(w.getMasses()[0]) == 2.5 and w.getActionSize() == 1
    c = w.clone()
    assert c.getMassDims() == 1 and float(c.getMasses()[0]) == 2.5 and c.getActionSpace() == [0]
    st = torch.tensor(s, device="cuda:0"); at = torch.tensor(a[:, :1], device="cuda:0")
    g = torch.randn(2 * w.n, 8, dtype=torch.float64, device="cuda:0")
    outs = []
    for world in (w, c):
        nxt, saved, _ = world.step_soa(world.to_soa(st), world.to_soa(at))
        gs, ga = world.backward_soa(saved, g)
        gm = world.backward_inertia_soa(saved, 8)
        outs.append((nxt, gs, ga, gm))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    assert outs[0][3].abs().max() > 0
