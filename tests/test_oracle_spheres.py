"""Sphere colliders in the CPU oracle: narrow phase (collideBoxSphere / collideSphereBox / collideSphereSphere,
DARTCollide.cpp:1482-1880) and the contact-geometry gradient model of sphere contacts (DCC.cpp:116-228, 328-403, 594-709)
pinned the way the reference pins its gradients: VJP == J^T g with J by central differences of the step."""
import numpy as np
import pytest

import nimblephysics_amd as na
from oracle import OracleWorld
from util import ball_state as _state, ball_world, rel_err


def _fd_check(md, s0, a0, seed, tol=2e-6):
    w = OracleWorld(md); n = w.n

    def step(x, u):
        w.reset_lcp_cache()
        return w.step(x, u)

    step(s0, a0)
    assert w.last_status & 0x1, "no contact"
    g = np.random.default_rng(seed).normal(0, 1, 2 * n)
    gs, ga = w.backprop(g)
    eps = 1e-7
    Js = np.zeros((2 * n, 2 * n)); Ja = np.zeros((2 * n, w.k))
    for j in range(2 * n):
        xp, xm = s0.copy(), s0.copy(); xp[j] += eps; xm[j] -= eps
        Js[:, j] = (step(xp, a0) - step(xm, a0)) / (2 * eps)
    for j in range(w.k):
        up, um = a0.copy(), a0.copy(); up[j] += eps; um[j] -= eps
        Ja[:, j] = (step(s0, up) - step(s0, um)) / (2 * eps)
    assert rel_err(gs, Js.T @ g) < tol and rel_err(ga, Ja.T @ g) < tol, (rel_err(gs, Js.T @ g), rel_err(ga, Ja.T @ g))
    return w


@pytest.mark.parametrize("order", ["box_first", "sphere_first"])
def test_ball_on_ground_contact_and_gradient(order):
    md = ball_world(order)
    s0, a0 = _state(md, [(0.3, -0.2)], 1)
    w = _fd_check(md, s0, a0, 2)
    c = w.last_contacts()
    assert c.shape[0] == 1
    assert int(c[0, 7]) == (5 if order == "box_first" else 4)            # BOX_SPHERE / SPHERE_BOX (Contact.hpp:57-58)
    nrm = c[0, 3:6]
    # the normal points from the second object towards the first
    assert np.allclose(nrm, [0, -1, 0] if order == "box_first" else [0, 1, 0], atol=1e-12)
    assert abs(c[0, 6] - 2e-3) < 1e-12 and abs(c[0, 1]) < 1e-12           # depth, contact point on the top face


def test_ball_on_the_rim_locks_two_faces():
    """Centre beyond the +x face and above the +y face: the contact point sits on the box edge, two face normals locked."""
    md = ball_world("box_first")
    s0, a0 = _state(md, [(2.0 + 0.06, 0.1)], 3)
    s0[4] = 0.07                      # |centre - edge| = sqrt(0.06^2 + 0.07^2) = 0.0922 < r
    w = _fd_check(md, s0, a0, 4, tol=5e-6)
    c = w.last_contacts()
    assert c.shape[0] == 1 and abs(c[0, 0] - 2.0) < 1e-12 and abs(c[0, 1]) < 1e-12


def test_centre_inside_the_box_degenerates_to_a_vertex_face_contact():
    md = ball_world("box_first")
    md.contact_clipping_depth = 0.2
    s0, a0 = _state(md, [(0.0, 0.0)], 5)
    s0[4] = -0.01
    w = OracleWorld(md); w.step(s0, a0)
    c = w.last_contacts()
    assert c.shape[0] == 1 and int(c[0, 7]) == 2 and abs(c[0, 6] - 0.11) < 1e-12   # FACE_VERTEX, depth = min + r


def test_two_balls_and_the_ground():
    """Ball 1 rests on the ground and ball 0 leans on ball 1: SPHERE_SPHERE + two box-sphere contacts."""
    md = ball_world("box_first", n_balls=2)
    s0, a0 = _state(md, [(0.0, 0.0), (0.198, 0.0)], 6)
    w = _fd_check(md, s0, a0, 7, tol=5e-6)
    types = sorted(int(t) for t in w.last_contacts()[:, 7])
    assert types == [5, 5, 6]


def test_articulated_spheres():
    md = ball_world("sphere_first", n_balls=1, arm=True)
    s0, a0 = _state(md, [(0.0, 0.0)], 8)
    s0[0:3] = 0.0; s0[6] = 0.0        # ball upright, arm horizontal: both spheres touch the ground
    w = _fd_check(md, s0, a0, 9, tol=5e-6)
    assert w.last_contacts().shape[0] == 2
