"""The randomised soaks' own harness (tools/soak_parity.py, tools/soak_stress.py) on the CPU: what the GPU soaks feed the device must be
well formed - a NaN compares as "not above the tolerance", so a mutation that produces one hides the worlds it touches (round 3: the
`limits` mode put ball-joint velocities on their infinite velocity limits)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("variant", ["balls", "big", "multi"])
def test_every_stress_mode_produces_finite_states_and_a_model_the_oracle_steps(variant):
    import soak_parity
    import soak_stress
    from oracle import OracleWorld
    stepped = 0
    for k, mode in enumerate(soak_stress.MODES + ("mix", "adjacent+limits+mass")):
        for seed in range(900 + 7 * k, 903 + 7 * k):
            case = soak_parity.make_case(seed, 4, variant == "big", variant == "multi", variant == "balls", False)
            if case is None:
                continue
            md, s, a, g = soak_stress.mutator(mode)(seed, *case)
            assert np.isfinite(s).all() and np.isfinite(a).all() and np.isfinite(g).all(), (mode, seed)
            assert s.shape == (4, 2 * md.num_dofs) and a.shape == (4, len(md.action_map)), (mode, seed)
            r = OracleWorld(md).step_batch(s, a, g, threads=2)
            assert all(np.isfinite(r[x]).all() for x in ("next", "grad_state", "grad_action")), (mode, seed)
            stepped += 1
    assert stepped >= 30


def test_the_mixed_mode_keeps_the_draws_of_its_first_eight_mutations():
    """tests/test_gpu_stress.py pins the world that exposed the duplicate filter's memory by the explicit list of mutations the mixed
    mode drew for its seed; the draw of a seed must not change when mutations are appended to MIX_ORDER."""
    import soak_stress
    assert soak_stress.MIX_ORDER[:8] == ("capsule", "geom", "mass", "mu", "selfcol", "limits", "subset", "dt")
    pick = np.random.default_rng(160020 + 77).random(len(soak_stress.MIX_ORDER)) < 0.5
    assert [m for m, p in zip(soak_stress.MIX_ORDER[:8], pick[:8]) if p] == ["geom", "mass", "selfcol", "limits", "dt"]
