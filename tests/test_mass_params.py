"""Host logic of the inertia ("mass") parameters (nimblephysics_amd/mass.py): the closed-form directions dG/dtheta against
finite differences of the spatial tensor, through weld merging (WithRespectToMass.cpp:50-140, Inertia.cpp:157-179, 1368-1383)."""
import copy

import numpy as np
import pytest

import nimblephysics_amd as na
from nimblephysics_amd.mass import WithRespectToMass, WrtMassBodyNodeEntryType as T, spatial_inertia


def _merged_G(md, body):
    m = md.merge_welds() if md.has_welds() else md
    b = m.bodies[body]
    return spatial_inertia(b.mass, b.com, b.inertia)


@pytest.mark.parametrize("etype", [T.INERTIA_MASS, T.INERTIA_COM, T.INERTIA_COM_MU, T.INERTIA_DIAGONAL, T.INERTIA_OFF_DIAGONAL, T.INERTIA_FULL])
def test_directions_match_finite_differences_through_welds(etype):
    md = copy.deepcopy(na.atlas("atlas20", ground=True))   # arms welded: composite bodies
    targets, _ = md.weld_targets()
    welded = [i for i, b in enumerate(md.bodies) if b.joint_type == "weld" and targets[i] >= 0]
    moving = [i for i, b in enumerate(md.bodies) if b.joint_type == "revolute"]
    for body in (welded[0], welded[-1], moving[3]):
        md.bodies[body].com = (0.01, -0.02, 0.03)
        if etype == T.INERTIA_COM_MU:   # the COM of a scaling-group body lies on its beta axis
            md.bodies[body].beta = (0.5, -1.0, 1.5)
            md.bodies[body].com = (0.01, -0.02, 0.03)
        md.bodies[body].inertia = (0.11, 0.12, 0.13, 0.004, -0.003, 0.002)
        w = WithRespectToMass(md)
        w.registerNode(body, etype)
        bodies, dG = w.device_table()
        assert dG.shape == (w.dim(), 36) and np.all(bodies == targets[body])
        x0 = w.get()
        for p in range(w.dim()):
            eps = 1e-6
            xp, xm = x0.copy(), x0.copy()
            xp[p] += eps; xm[p] -= eps
            w.set(xp); Gp = _merged_G(md, targets[body])
            w.set(xm); Gm = _merged_G(md, targets[body])
            w.set(x0)
            fd = (Gp - Gm) / (2 * eps)
            assert np.abs(fd - dG[p].reshape(6, 6)).max() < 1e-8 * max(1.0, np.abs(fd).max())


def test_mass_vector_layout_and_bounds():
    md = copy.deepcopy(na.cartpole())
    w = WithRespectToMass(md)
    w.registerNode(1, T.INERTIA_MASS, upperBound=[5.0], lowerBound=[0.1])
    w.registerNode(md.bodies[0].name, T.INERTIA_COM)
    assert w.dim() == 4
    assert np.allclose(w.get(), [md.bodies[1].mass, *md.bodies[0].com])
    w.set([2.0, 0.1, 0.2, 0.3])
    assert md.bodies[1].mass == 2.0 and tuple(md.bodies[0].com) == (0.1, 0.2, 0.3)
    assert np.allclose(w.upperBound()[:1], [5.0]) and np.allclose(w.lowerBound()[:1], [0.1])
    with pytest.raises(ValueError):
        w.set([1.0])
    with pytest.raises(ValueError):
        w.registerNode(1, T.INERTIA_COM)


def test_com_mu_entry_follows_beta():
    """INERTIA_COM_MU: one scalar mu, COM = beta * mu; the getter divides by the first non-zero beta (WithRespectToMass.cpp:76-92, 157-166)."""
    md = copy.deepcopy(na.cartpole())
    w = WithRespectToMass(md)
    assert tuple(md.bodies[1].beta) == (1.0, 1.0, 1.0)   # BodyNode's default
    e = w.registerNode(1, T.INERTIA_COM_MU)
    assert e.dim() == 1 and w.dim() == 1
    w.set([0.25])
    assert tuple(md.bodies[1].com) == (0.25, 0.25, 0.25) and np.allclose(w.get(), [0.25])
    md.bodies[1].beta = (0.0, -2.0, 0.5)
    w.set([0.1])
    assert np.allclose(md.bodies[1].com, (0.0, -0.2, 0.05)) and np.allclose(w.get(), [0.1])
    md.bodies[1].beta = (0.0, 0.0, 4.0)
    w.set([0.1])
    assert np.allclose(md.bodies[1].com, (0.0, 0.0, 0.4)) and np.allclose(w.get(), [0.1])
    # the description round-trips the non-default beta and omits the default one
    d = md.to_json()
    assert d["bodies"][1]["beta"] == [0.0, 0.0, 4.0] and "beta" not in d["bodies"][0]
    assert tuple(type(md).from_json(d).bodies[1].beta) == (0.0, 0.0, 4.0)


def test_setting_the_mass_rescales_the_moment():
    """BodyNode::setMass keeps the box dimensions (Inertia::setMass, preserveDimsAndEuler): the whole tensor scales."""
    md = copy.deepcopy(na.single_pendulum())
    G0 = spatial_inertia(md.bodies[0].mass, md.bodies[0].com, md.bodies[0].inertia)
    w = WithRespectToMass(md)
    w.registerNode(0, T.INERTIA_MASS)
    w.set([2 * md.bodies[0].mass])
    G1 = spatial_inertia(md.bodies[0].mass, md.bodies[0].com, md.bodies[0].inertia)
    assert np.allclose(G1, 2 * G0)
