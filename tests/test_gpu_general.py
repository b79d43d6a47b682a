"""The GENERAL instantiation of the contact stage (csrc/gen_contact.hip, gen_lcp_dev.hpp, gen_dantzig_dev.hpp: up to 64 contacts = 192 LCP
rows, 64 colliders, 512 collider pairs per world; a model runs on it when it asks for more than 16 contact slots) on the device against the
oracle: the worlds the 24- / 48-row builds truncate - towers of five and ten cubes, the reference's own
data/skel/test/box_stacking.skel in the configuration it ships in (ten cubes stacked face to face: 28 contacts in constrained groups of
several cubes from the first step, 40 in ONE group once the tower stands on the ground) - and, so that the new kernels are judged where the answer is known best, the
metric distribution and the earlier many-contact scenes run through the general kernels as well."""
import ctypes as C

import numpy as np
import pytest

from parity import assert_match_or_reference_unstable, world_errors

pytestmark = pytest.mark.gpu
TOL = 1e-7


def _fwd_bwd(md, s, a, seed, lcp=None):
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    world = na.World(md, device="cuda:0")
    g = np.random.default_rng(seed).normal(0, 1, s.shape)
    st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at)
    status = world.last_status.cpu().numpy().astype(np.uint32)
    out.backward(torch.tensor(g, device="cuda:0"))
    dev = {"next": out.detach().cpu().numpy(), "grad_state": st.grad.cpu().numpy(), "grad_action": at.grad.cpu().numpy()}
    for k in dev:
        assert np.isfinite(dev[k]).all(), (k, "non-finite device output")
    return world, dev, status, g


def test_the_general_dantzig_driver_is_bit_identical_to_the_reference_on_the_device():
    """nbl_selftest_lcp_dantzig with n > 48 runs gen_dantzig_dev.hpp::genDantzigSeq on the GPU (lane 0 of a wavefront per problem, the
    problem in HBM): success flag and every bit of x equal to the reference's own dSolveLCP (oracle/_ref), 51 .. 192 rows, rank-deficient
    contact problems included."""
    import oracle
    from nimblephysics_amd import _lib
    from util import contact_lcp, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref not built")
    L = _lib.lib()
    OL = oracle._lib()
    pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    rng = np.random.default_rng(0)
    solved = failed = 0
    for n_contacts, count in ((17, 24), (27, 16), (40, 12), (64, 8)):
        n = 3 * n_contacts
        probs = []
        for k in range(count):
            ndof = n + int(rng.integers(0, 6)) if k % 3 == 0 else int(rng.choice([12, 30, 60]))
            A, b, lo, hi, fi = contact_lcp(rng, n_contacts, ndof)
            if k % 4 == 1:
                b = np.abs(b)
            probs.append((A, b, lo, hi, fi))
        A = np.ascontiguousarray(np.stack([p[0] for p in probs])); b = np.ascontiguousarray(np.stack([p[1] for p in probs]))
        lo = np.ascontiguousarray(np.stack([p[2] for p in probs])); hi = np.ascontiguousarray(np.stack([p[3] for p in probs]))
        fi = np.ascontiguousarray(np.stack([p[4] for p in probs]).astype(np.int32))
        x = np.zeros((count, n)); rc = np.zeros(count, np.int32)
        r = L.nbl_selftest_lcp_dantzig(count, n, A.ctypes.data_as(pd), b.ctypes.data_as(pd), lo.ctypes.data_as(pd), hi.ctypes.data_as(pd),
                                       fi.ctypes.data_as(pi), x.ctypes.data_as(pd), rc.ctypes.data_as(pi))
        assert r == 0, L.nbl_last_error()
        for k, (Ak, bk, lok, hik, fik) in enumerate(probs):
            xr = np.zeros(n)
            okr = OL.nbo_lcp_dantzig(n, Ak.ctypes.data_as(pd), xr.ctypes.data_as(pd), bk.copy().ctypes.data_as(pd), lok.copy().ctypes.data_as(pd),
                                     hik.copy().ctypes.data_as(pd), fik.copy().ctypes.data_as(pi), 1)
            if rc[k] == -1:
                assert okr == 0 or not np.all(np.isfinite(xr))
                continue
            assert okr == rc[k], (n, k, okr, rc[k])
            if okr == 1:
                solved += 1
                assert np.array_equal(xr, x[k]), (n, k, np.abs(xr - x[k]).max())
            else:
                failed += 1
    print(f"[general Dantzig on the device] {solved} problems of 51 .. 192 rows solved bit for bit, {failed} early exits, all flags equal")
    assert solved >= 20


def test_the_metric_distribution_on_the_general_build():
    """Atlas-20 + 8 contacts at sigma = 0.02 (half of the worlds leave stage 0 and run reduce + Dantzig / CFM + PGS / the frictionless
    fallback) with 24 contact slots requested: the SAME worlds through rows / cascade / adjoint of gen_contact.hip.  Every world's next
    state and both gradients against the oracle, the same stage-0 set."""
    import nimblephysics_amd as na
    from oracle import OracleWorld
    from util import contact_inputs
    B = 256
    md, s, a = contact_inputs("atlas20", B, 13, joint_noise=0.02, vel_noise=0.01, action_noise=0.0)
    md.max_contacts = 24
    world, dev, st, g = _fwd_bwd(md, s, a, 14)
    assert world._L.nbl_model_max_contacts(world._h) == 64
    ow = OracleWorld(md)
    ref = ow.step_batch(s, a, g, threads=8)
    assert (st & 1).all() and not (st & 0x80).any()
    assert np.array_equal((st & 0x2) != 0, (ref["status"] & 0x2) != 0)
    assert 0.2 < ((st & 0x2) != 0).mean() < 0.8
    e, _ = world_errors(dev, ref)
    print("[metric distribution on the general build] stages:", {hex(int(k)): int(c) for k, c in zip(*np.unique(st & 0x13e, return_counts=True))},
          "max errors:", {k: float(v.max()) for k, v in e.items()})
    bad, _ = assert_match_or_reference_unstable("metric distribution, general build", ow, s, a, g, dev, ref, TOL, max_unstable=4)


@pytest.mark.parametrize("n_cubes,B", [(5, 64), (10, 24)])
def test_cube_towers_of_twenty_and_forty_contacts(n_cubes, B):
    """A tower of n cubes on the ground plate: 4 contacts per interface, one constrained group of n skeletons - 20 contacts (60 rows) with
    five cubes, 40 (120 rows) with ten: what the 48-row build truncated and flagged.  Nothing overflows now; every world's next state and
    gradients against the oracle."""
    from oracle import OracleWorld
    from util import cube_tower_inputs
    md, s, a = cube_tower_inputs(B, 7 + n_cubes, n_cubes, max_contacts=4 * n_cubes + 8)
    world, dev, st, g = _fwd_bwd(md, s, a, 2)
    assert world._L.nbl_model_max_contacts(world._h) == 64
    ow = OracleWorld(md)
    ref = ow.step_batch(s, a, g, threads=8)
    ow.step(s[0], a[0])
    assert len(ow.last_contacts()) == 4 * n_cubes
    assert (st & 1).all() and not (st & 0x80).any() and not (ref["status"] & 0x80).any(), "overflow"
    e, _ = world_errors(dev, ref)
    print(f"[{n_cubes}-cube tower] stages:", {hex(int(k)): int(c) for k, c in zip(*np.unique(st & 0x13e, return_counts=True))},
          "max errors:", {k: float(v.max()) for k, v in e.items()})
    bad, _ = assert_match_or_reference_unstable(f"{n_cubes}-cube tower, {4 * n_cubes} contacts", ow, s, a, g, dev, ref, TOL, ulps=16, max_unstable=max(3, int(0.1 * B)))


def three_groups_scene(B, towers=2, table=True, seed=21):
    """`towers` towers of three cubes and (table) a five-footed table side by side on the ground plate"""
    import nimblephysics_amd as na
    from util import cube_tower_inputs
    rng = np.random.default_rng(seed)
    side, mass = 0.2, 0.1
    I = mass * side * side / 6.0
    ncubes = 3 * towers
    bodies = [na.BodySpec(f"cube{i}", -1, "free", f"cube{i}_joint", mass=mass, inertia=(I, I, I, 0, 0, 0)) for i in range(ncubes)]
    boxes = [na.BoxSpec(-1, na.make_transform((0, -0.5, 0)), (30.0, 1.0, 30.0), 1.0)]
    boxes += [na.BoxSpec(i, np.eye(4), (side, side, side), 1.0) for i in range(ncubes)]
    if table:
        bodies.append(na.BodySpec("table", -1, "free", "table_joint", mass=2.0, inertia=(0.05, 0.08, 0.05, 0, 0, 0)))
        for (cx, cz) in [(-0.3, -0.2), (0.3, -0.2), (-0.3, 0.2), (0.3, 0.2), (0.0, 0.0)]:
            boxes.append(na.BoxSpec(ncubes, na.make_transform((cx, 0.05, cz)), (0.1, 0.1, 0.1), 1.0))
    md = na.ModelDescription("three_groups", bodies, boxes, gravity=(0.0, -9.81, 0.0), dt=1e-3, max_contacts=12 * towers + (20 if table else 0) + 4)
    n = md.num_dofs
    q = np.zeros((B, n)); v = np.zeros((B, n))
    for t in range(towers):
        _, sT, _ = cube_tower_inputs(B, 31 + t, 3)
        q[:, 18 * t:18 * t + 18] = sT[:, 0:18]; v[:, 18 * t:18 * t + 18] = sT[:, 18:36]
        q[:, [18 * t + 3, 18 * t + 9, 18 * t + 15]] += 3.0 * (t + 1) * (1 if t % 2 == 0 else -1)      # the towers 3 m apart, the table at the origin
    if table:
        o = 6 * ncubes
        q[:, o + 1] = rng.uniform(-1, 1, B); q[:, o + 4] = -rng.uniform(1e-4, 2e-3, B)
        v[:, o:o + 6] = rng.normal(0, 0.05, (B, 6))
    return md, np.concatenate([q, v], 1), np.zeros((B, n))


def test_two_towers_and_a_table_are_three_constrained_groups():
    """constrained groups on the general build: two towers of three cubes and a five-footed table side by side - 12 + 12 + 20 contacts in
    three groups, each solved on its own (one that needs the fallback stages does not change the others)."""
    from oracle import OracleWorld
    B = 48
    md, s, a = three_groups_scene(B)
    world, dev, st, g = _fwd_bwd(md, s, a, 22)
    ow = OracleWorld(md)
    ref = ow.step_batch(s, a, g, threads=8)
    ow.step(s[0], a[0])
    assert len(ow.last_contacts()) == 44
    assert (st & 1).all() and not ((st | ref["status"]) & 0x80).any()
    e, _ = world_errors(dev, ref)
    print("[three constrained groups, 44 contacts] max errors:", {k: float(v_.max()) for k, v_ in e.items()})
    assert_match_or_reference_unstable("three groups", ow, s, a, g, dev, ref, TOL, ulps=16, max_unstable=int(0.15 * B), closeness=1.0, max_by_closeness=4)


def test_box_stacking_skel_as_it_ships_until_the_tower_rests_on_the_ground():
    """data/skel/test/box_stacking.skel in the file's OWN configuration (nimblephysics_amd/data/box_stacking_full.json, every coordinate
    zero): ten cubes stacked face to face at depth 0, 0.395 m above the ground box.  28 contacts from the first step (the reference keeps
    contacts at depth 0, ConstraintSolver.cpp:598-601; two of the nine interfaces round to a gap of one ulp), 40 in ONE constrained group of ten
    skeletons once the tower stands.  Rolled out for 420 steps
    (free fall 284 steps, impact, rest), the LCP warm start carried from step to step; EVERY step's next state and both gradients against the
    oracle started from the device's state and warm start (teacher forcing: errors do not accumulate) with the parity criterion of every
    other test; no world ever overflows."""
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    from oracle import OracleWorld
    md = na.ModelDescription.load("box_stacking_full")
    md.max_contacts = 40
    n = md.num_dofs
    assert n == 60 and len(md.boxes) == 11
    B, T = 2, 420
    s0 = np.zeros((B, 2 * n))
    rng = np.random.default_rng(5)
    yaw = 0.3
    for k in range(10):       # the second world: the tower turned as a whole and every cube shifted a little (still face to face at depth 0, four
        s0[1, 6 * k + 1] = yaw   # points per interface: cubes turned AGAINST each other meet in octagons, 8 points each - 80 contacts)
        dx, dz = rng.uniform(-0.01, 0.01, 2)
        s0[1, 6 * k + 3] = np.cos(yaw) * dx + np.sin(yaw) * dz; s0[1, 6 * k + 5] = -np.sin(yaw) * dx + np.cos(yaw) * dz
    a = np.zeros((B, n))
    world = na.World(md, device="cuda:0")
    assert world._L.nbl_model_max_contacts(world._h) == 64
    ow = OracleWorld(md)
    stride = 3 * md.max_contacts                                   # (the oracle takes 3 x max_contacts entries; the device buffer has room for 64 contacts)
    world.reset_lcp_cache()
    x = torch.tensor(s0, device="cuda:0")
    at = torch.tensor(a, device="cuda:0")
    worst = 0.0
    contacts = []
    unstable_steps = []
    for t in range(T):
        cache = world.lcp_cache.clone().cpu().numpy() if world.lcp_cache is not None else None
        kw = {}
        if cache is not None:
            kw = {"lcp_in": np.ascontiguousarray(cache[:stride].T), "lcp_len_in": cache[-1].astype(np.int32)}
        xin = x.detach().cpu().numpy()
        xt = x.detach().clone().requires_grad_(True); att = at.clone().requires_grad_(True)
        y = timestep(world, xt, att)
        g = np.random.default_rng(100 + t).normal(0, 1, xin.shape)
        y.backward(torch.tensor(g, device="cuda:0"))
        dev = {"next": y.detach().cpu().numpy(), "grad_state": xt.grad.cpu().numpy(), "grad_action": att.grad.cpu().numpy()}
        ref = ow.step_batch(xin, a, g, threads=2, **kw)
        e, _ = world_errors(dev, ref)
        err = max(float(v_.max()) for v_ in e.values())
        if err <= TOL:
            worst = max(worst, err)
        # (the impact of ten exactly aligned cubes is as degenerate as an LCP gets - 40 contacts of rank 6 per interface, every pivot a tie:
        #  for a handful of steps around it the reference's own answer moves under 16-ulp perturbations of the state; the criterion proves
        #  that per step and holds the device to one of the reference's outcomes)
        bad, _ = assert_match_or_reference_unstable(f"box_stacking.skel step {t}", ow, xin, a, g, dev, ref, TOL, lcp=kw.get("lcp_in"), lcp_len=kw.get("lcp_len_in"),
                                                    ulps=16, max_unstable=B, closeness=1.0, max_by_closeness=B, verbose=False)
        if bad:
            unstable_steps.append(t)
        st = world.last_status.cpu().numpy().astype(np.uint32)
        assert not (st & 0x80).any() and not (ref["status"] & 0x80).any(), (t, "contact overflow")
        assert np.array_equal(st & 0x1, ref["status"] & 0x1), t
        x = y.detach()
        if t in (0, 283, 300, 419):
            ow.step(xin[0], a[0])
            contacts.append((t, len(ow.last_contacts())))
    print(f"[box_stacking.skel as shipped] steps with a reference-unstable world ({len(unstable_steps)} of {T}): {unstable_steps}")
    assert len(unstable_steps) <= 40 and all(270 <= t_ <= 360 for t_ in unstable_steps), unstable_steps
    fin = x.cpu().numpy()
    print(f"[box_stacking.skel as shipped] contacts of the untouched world at steps {contacts}; worst error (next state and gradients) over the stable steps {worst:.1e}; "
          f"bottom cube at y = {fin[0, 4]:.4f} (rest: -0.395), |v| at the end {np.abs(fin[:, n:]).max():.2e}")
    assert contacts[0][1] >= 24 and contacts[-1][1] == 40       # (step 0: the interfaces whose 0.2 m spacing rounds to a gap of 1 ulp are out)
    assert abs(fin[0, 4] + 0.395) < 2e-3 and np.abs(fin[:, n:]).max() < 0.2
