"""The one-world-per-lane TREE kernels (k_step_forward, k_step_backward, k_bwd_recompute, k_bwd_final, k_tree_to_lanes) are the
path of models that do not fit a wavefront (more than 64 bodies or DOFs) and of the memory-lean mode.  The default path of every
named config is one world per wavefront, so these kernels only run when a switch or the model size selects them: this file runs
them — each switch on a tree without contact and on the Atlas contact config, plus a 70-body chain — against the CPU oracle to
the same tolerance as the default path."""
import numpy as np
import pytest

from test_gpu_random_trees import random_tree, _compare
from util import contact_inputs

pytestmark = pytest.mark.gpu

MODES = [{"NBL_COOP_TREE": "0"}, {"NBL_COOP_FINAL": "0"}, {"NBL_SAVE_TREE": "0"}, {"NBL_COOP_TREE": "0", "NBL_SAVE_TREE": "0"},
         {"NBL_TREE_PACK": "1"}, {"NBL_DETECT_SPLIT": "0"}, {"NBL_AUX_OVERLAP": "1"}]
IDS = ["-".join(f"{k[4:]}={v}" for k, v in m.items()) for m in MODES]


@pytest.mark.parametrize("mode", MODES, ids=IDS)
def test_tree_without_contact_in_every_mode(mode, monkeypatch):
    for k, v in mode.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(77)
    md = random_tree(rng, 17, "random", True)
    _compare(md, 96, 5)


@pytest.mark.parametrize("mode", MODES, ids=IDS)
def test_atlas_contact_in_every_mode(mode, monkeypatch):
    """Atlas-20 on the ground (8 contacts, easy distribution: every world resolves at stage 0), forward and backward."""
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    from oracle import OracleWorld
    for k, v in mode.items():
        monkeypatch.setenv(k, v)
    B = 128
    md, s, a = contact_inputs("atlas20", B, 3)
    g = np.random.default_rng(4).normal(0, 1, s.shape)
    world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
    st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at)
    status = world.last_status.cpu().numpy().astype(np.uint32)
    out.backward(torch.tensor(g, device="cuda:0"))
    ref = ow.step_batch(s, a, g, threads=8)
    ok = ((status & 0x2) != 0) & ((ref["status"] & 0x2) != 0)
    assert ok.mean() > 0.9
    sc = lambda x: max(np.abs(x[ok]).max(), 1e-30)
    for name, dev, r in (("next", out.detach().cpu().numpy(), ref["next"]), ("grad_state", st.grad.cpu().numpy(), ref["grad_state"]),
                         ("grad_action", at.grad.cpu().numpy(), ref["grad_action"])):
        assert np.abs(dev[ok] - r[ok]).max() / sc(r) < 1e-7, (mode, name)


@pytest.mark.parametrize("shape,free_root", [("chain", False), ("random", True)])
def test_a_70_body_tree_runs_one_world_per_lane(shape, free_root):
    """More bodies than a wavefront has lanes: the library must pick the one-world-per-lane tree kernels by itself."""
    rng = np.random.default_rng(4242)
    md = random_tree(rng, 70, shape, free_root)
    _compare(md, 64, 11, tol=1e-6 if shape == "chain" else 1e-7)


def test_fused_cascade_launch_is_bit_identical_to_the_two_launches(monkeypatch):
    """NBL_FUSED_CASCADE=1: stages 1-3 and the final part of the LCP cascade in ONE launch (k_contact_cascade_fused: the wavefront of a
    world that finishes its stage second goes on with select + standardise + outputs, candidates through LDS) against the default
    two launches (k_contact_cascade_stages, k_contact_cascade_final): next states, status words, warm starts and gradients bit for bit,
    on the metric distribution (half of the worlds in the cascade) and on two cubes side by side (several constrained groups)."""
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    from util import box_stack_inputs
    cases = [contact_inputs("atlas20", 1024, 13, joint_noise=0.02, vel_noise=0.01, action_noise=0.0), box_stack_inputs(512, 32)]
    md2 = na.box_stack(); n2 = md2.num_dofs; rng = np.random.default_rng(21)
    s2 = np.zeros((512, 2 * n2))
    gb = md2.boxes[0]; top = (md2.bodies[0].T_pj @ gb.T)[1, 3] + 0.5 * gb.size[1]; half = 0.5 * gb.size[0]
    for k, x0 in enumerate((-0.4, 0.4)):
        o = 6 * k; c0 = md2.bodies[1 + k].T_pj[:3, 3]
        s2[:, o + 1] = rng.uniform(-1, 1, 512); s2[:, o + 3] = x0 * half + rng.uniform(-0.15, 0.15, 512) * half - c0[0]
        s2[:, o + 4] = top + 0.1 - rng.uniform(1e-4, 1e-3, 512) - c0[1]; s2[:, o + 5] = rng.uniform(-0.5, 0.5, 512) * half - c0[2]
        s2[:, n2 + o:n2 + o + 6] = rng.normal(0, 0.05, (512, 6))
    cases.append((md2, s2, rng.normal(0, 0.1, (512, n2))))
    results = {}
    for fused in ("0", "1"):
        monkeypatch.setenv("NBL_FUSED_CASCADE", fused)
        out = []
        for md, s, a in cases:
            g = np.random.default_rng(5).normal(0, 1, s.shape)
            world = na.World(md, device="cuda:0")
            st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
            y = timestep(world, st, at)
            status = world.last_status.clone(); cache = world.lcp_cache.clone()
            y.backward(torch.tensor(g, device="cuda:0"))
            out.append((y.detach().clone(), status, cache, st.grad.clone(), at.grad.clone()))
        results[fused] = out
    for ra, rb in zip(results["0"], results["1"]):
        assert ((ra[1] & 0x2) == 0).float().mean() > 0.2            # the cascade is exercised
        for ta, tb in zip(ra, rb):
            assert torch.equal(ta, tb)


LAUNCH_SHAPES = [{"NBL_FUSED_DETECT": "0"}, {"NBL_TREE_WPB": "1", "NBL_DETECT_WL": "8"}, {"NBL_TREE_WPB": "2", "NBL_DETECT_WL": "16"}, {"NBL_DETECT_SPLIT": "0"},
                 {"NBL_ROWS_PACK": "1"}, {"NBL_TREE_PACK": "1"}]


@pytest.mark.parametrize("mode", LAUNCH_SHAPES, ids=["-".join(f"{k[4:]}={v}" for k, v in m.items()) for m in LAUNCH_SHAPES])
def test_the_launch_shapes_do_not_change_a_bit(mode, monkeypatch):
    """How the worlds are dealt to workgroups and lanes is a launch decision, not arithmetic: the narrow phase as a launch of its own (world
    transforms from the tree block) or next to the tree kernel (its own forward kinematics from LDS-resident joint transforms, all threads
    of the workgroup), 8 / 16 / 32 worlds per narrow-phase workgroup (lane stride of its LDS buffers), one / two / three tree wavefronts per
    workgroup, one lane or several per world in the narrow phase, one or two worlds per wavefront in the contact-row kernel, one or four
    worlds per wavefront in the tree kernels: next states, status words, warm starts and both gradients bit for bit against the default,
    on the metric distribution (half of the worlds in the cascade)."""
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    md, s, a = contact_inputs("atlas20", 1024, 13, joint_noise=0.02, vel_noise=0.01, action_noise=0.0)
    g = np.random.default_rng(5).normal(0, 1, s.shape)
    results = []
    for env in ({}, mode):
        for k in ("NBL_FUSED_DETECT", "NBL_TREE_WPB", "NBL_DETECT_WL", "NBL_DETECT_SPLIT", "NBL_ROWS_PACK", "NBL_TREE_PACK"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        world = na.World(md, device="cuda:0")
        st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
        y = timestep(world, st, at)
        status = world.last_status.clone(); cache = world.lcp_cache.clone()
        y.backward(torch.tensor(g, device="cuda:0"))
        results.append((y.detach().clone(), status, cache, st.grad.clone(), at.grad.clone()))
    assert ((results[0][1] & 0x2) == 0).float().mean() > 0.2            # the cascade is exercised
    for name, ta, tb in zip(("next", "status", "warm start", "grad_state", "grad_action"), results[0], results[1]):
        assert torch.equal(ta, tb), (mode, name)


def test_a_call_cut_into_stream_slices_gives_the_same_bits_and_can_be_captured():
    """From 4096 worlds on, a call of a model with colliders is cut into two slices on two HIP streams by default (fork / join events
    around the caller's stream): next states, status words, warm starts and gradients bit for bit as with one slice and with four, and
    the sliced step captured in a HIP graph replays to the same bits."""
    import torch
    import nimblephysics_amd as na
    B = 4096
    md, s, a = contact_inputs("atlas20", B, 17, joint_noise=0.02, vel_noise=0.01, action_noise=0.0)
    g = np.random.default_rng(6).normal(0, 1, s.shape)
    out = {}
    for slices in (0, 1, 4):
        world = na.World(md, device="cuda:0")
        world.set_slices(slices)
        assert world.slices_for(B) == {0: 2, 1: 1, 4: 4}[slices] and world.slices_for(1024) == (1 if slices < 4 else 4)
        st = world.to_soa(torch.tensor(s, device="cuda:0")); at = world.to_soa(torch.tensor(a, device="cuda:0")); gt = world.to_soa(torch.tensor(g, device="cuda:0"))
        nxt, sv, status = world.step_soa(st, at)
        gs, ga = world.backward_soa(sv, gt)
        out[slices] = (nxt.clone(), status.clone(), world.lcp_cache.clone(), gs.clone(), ga.clone())
    for slices in (1, 4):
        for name, ta, tb in zip(("next", "status", "warm start", "grad_state", "grad_action"), out[0], out[slices]):
            assert torch.equal(ta, tb), (slices, name)
    assert ((out[0][1] & 0x2) == 0).float().mean() > 0.2
    world = na.World(md, device="cuda:0")
    graphed = na.GraphedStep(world, B).capture()
    graphed.state.copy_(world.to_soa(torch.tensor(s, device="cuda:0"))); graphed.action.copy_(world.to_soa(torch.tensor(a, device="cuda:0")))
    graphed.grad_next.copy_(world.to_soa(torch.tensor(g, device="cuda:0")))
    nxt, dstate, daction = graphed.replay()
    torch.cuda.synchronize()
    assert torch.equal(nxt, out[0][0]) and torch.equal(dstate, out[0][3]) and torch.equal(daction, out[0][4])


FEATURE_MODES = [{"NBL_COOP_TREE": "0"}, {"NBL_COOP_FINAL": "0"}, {"NBL_SAVE_TREE": "0"}, {"NBL_FUSED_DETECT": "0"}, {"NBL_DETECT_SPLIT": "0"}]


@pytest.mark.parametrize("mode", FEATURE_MODES, ids=["-".join(f"{k[4:]}={v}" for k, v in m.items()) for m in FEATURE_MODES])
def test_round3_features_in_the_fallback_modes(mode, monkeypatch):
    """Joint-limit rows next to contacts, capsule contacts and self-collision (the narrow phase appends the limit rows from the saved q
    when it does not run inside the forward launch; the colliders' world transforms come from the tree block instead of the kernel's own
    forward kinematics): every world against the oracle like in the default mode."""
    for k, v in mode.items():
        monkeypatch.setenv(k, v)
    import test_gpu_capsules as tc
    import test_gpu_joint_limits as tl
    import test_gpu_self_collision as ts
    from util import capsule_world, folding_arm, limited_arm
    md = limited_arm(ground=True)
    s, a = tl._states(md, 256, 5, at_limit=0.35)
    s[:, 0] = np.random.default_rng(6).uniform(-0.025, 0.008, len(s))
    tl._compare(f"limited arm on the ground {mode}", md, s, a, 7, min_limit=0.5, min_contact=0.3)
    md = capsule_world(order="fixed_first", kinds=("capsule", "capsule", "sphere"))

    def pose(rng):
        y0 = 0.35 - rng.uniform(1e-3, 3e-3)
        return [((0.0, np.pi / 2 + rng.normal(0, 0.02), 0.0), (0.0, y0, 0.0)),
                ((np.pi / 2 + rng.normal(0, 0.05), 0.0, 0.0), (0.15 + rng.normal(0, 0.01), y0 + 0.1 + 0.1 + 0.2 - rng.uniform(1e-3, 3e-3), rng.normal(0, 0.01))),
                (rng.normal(0, 0.3, 3), (-0.15 + rng.normal(0, 0.01), y0 + 0.1 + 0.1 - rng.uniform(1e-3, 3e-3), rng.normal(0, 0.01)))]
    s, a = tc._states(md, pose, 128, 7)
    tc._compare(f"capsule pile {mode}", md, s, a, 8, [13, 15])
    md = folding_arm(True, "box")
    s, a = ts._states(256, 1, 1.872, 2.14)
    ts._compare(f"folding arm {mode}", md, s, a, 2, 0.7)


def test_the_fused_cascade_launch_equals_the_two_launch_cascade_bit_for_bit(monkeypatch):
    """NBL_FUSED_CASCADE=1 (stages 1-3 and the final part of the LCP cascade in ONE launch, the two stage wavefronts of a world handing
    their candidates over through LDS with a release / acquire arrival counter; opt-in: measured slower with four slices in flight) runs
    the same device functions as the default k_contact_cascade_stages + k_contact_cascade_final: on the metric distribution, where a
    third of the worlds go through it, every bit of the next state, the status words and both gradients must be equal."""
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    md, s, a = contact_inputs("atlas20", 1024, 13, joint_noise=0.02, vel_noise=0.01, action_noise=0.0)
    g = np.random.default_rng(14).normal(0, 1, s.shape)
    res = []
    for fused in ("0", "1"):
        monkeypatch.setenv("NBL_FUSED_CASCADE", fused)
        world = na.World(md, device="cuda:0")
        st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
        out = timestep(world, st, at)
        status = world.last_status.cpu().numpy().astype(np.uint32)
        out.backward(torch.tensor(g, device="cuda:0"))
        res.append((out.detach().cpu().numpy(), status, st.grad.cpu().numpy(), at.grad.cpu().numpy()))
    assert 0.2 < ((res[0][1] & 0x2) == 0).mean() < 0.8                          # the cascade is exercised
    for x, y in zip(*res):
        assert np.array_equal(x, y)
