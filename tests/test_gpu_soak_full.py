"""The randomised soaks at FULL size inside the GPU suite ("the headline soak numbers are builder-only"): the 22 world
modes of tools/final_soak.sh on fixed seeds - five model families of tools/soak_parity.py, the warm-start soak, the fourteen stress modes and
the three mixed-feature families of tools/soak_stress.py - 2 x 1.1 M worlds, every one of them compared with the oracle (next state and both
gradients; a world above 1e-6 must be PROVEN reference-unstable, tools/soak_parity.py::prove_reference_unstable), plus the five model families
once more with every model forced onto the GENERAL instantiation of the contact stage.  About 50 s on an MI355X box with 16 host cores;
NBL_SKIP_FULL_SOAK=1 leaves it out."""
import os
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("NBL_SKIP_FULL_SOAK") == "1", reason="NBL_SKIP_FULL_SOAK=1")]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
SEEDS = [300000, 400000]    # seed offsets of two passes (300000: profiles/r05_final_soak.log = `bash tools/final_soak.sh 300000`)
FAMILIES = {"": {}, "big": {"big": True}, "multi": {"multi": True}, "balls": {"balls": True}, "far": {"balls": True, "far": True}}


@pytest.mark.parametrize("S", SEEDS)
@pytest.mark.parametrize("family", list(FAMILIES))
def test_parity_soak_full_size(family, S):
    import soak_parity
    tot = soak_parity.run(40000 + S, 300, 256, verbose=False, **FAMILIES[family])
    print(f"parity:{family}", tot)
    assert tot["MISMATCH"] == 0 and tot["nonfinite"] == 0 and tot["worlds"] >= 76000, tot
    assert tot["contact"] > 0.15 * tot["worlds"] and tot["cascade"] > 0.1 * tot["worlds"], tot
    if family == "big":
        assert tot["overflow"] == 0, tot                                  # (16 slots)
    # worlds that overflow the 8 slots of their model are not left out: they run again on the general build and are judged there
    assert tot.get("rerun_on_general_build", 0) == tot["overflow"], tot


@pytest.mark.parametrize("S", SEEDS)
def test_warm_start_soak_full_size(S):
    import soak_warm
    for args in ((41000 + S, 300, 256, "balls"), (45000 + S, 300, 256, "balls", False, "mix")):
        tot = soak_warm.run(*args[:4], verbose=False, stress=args[5] if len(args) > 5 else None)
        print("warm", args[3:], tot)
        assert tot["MISMATCH"] == 0 and tot["worlds"] >= 76000 and tot["contact2"] > 5000, tot


@pytest.mark.parametrize("mode", ["dt", "tinydt", "fast", "torque", "mass", "nograv", "geom", "mu", "subset", "atlimit", "capsule", "limits", "selfcol", "adjacent"])
@pytest.mark.parametrize("S", SEEDS)
def test_stress_soak_full_size(mode, S):
    import soak_stress
    tot = soak_stress.run(mode, 43000 + S, 120, 256)
    print(f"stress:{mode}", tot)
    assert tot["MISMATCH"] == 0 and tot["nonfinite"] == 0 and tot["worlds"] >= 30000, tot


@pytest.mark.parametrize("variant", ["balls", "big", "multi"])
@pytest.mark.parametrize("S", SEEDS)
def test_mixed_feature_soak_full_size(variant, S):
    import soak_stress
    tot = soak_stress.run("mix", 44000 + S, 200, 256, variant=variant)
    print(f"stress:mix:{variant}", tot)
    assert tot["MISMATCH"] == 0 and tot["nonfinite"] == 0 and tot["worlds"] >= 50000, tot


@pytest.mark.parametrize("family", list(FAMILIES))
def test_parity_soak_on_the_general_build(family):
    """Every model of the family on the general instantiation (64 contact slots: rows looped over, matrices in HBM scratch, sequential
    Dantzig): no overflow, no mismatch."""
    import soak_parity
    tot = soak_parity.run(48000, 100, 128, verbose=False, slots=64, **FAMILIES[family])
    print(f"general parity:{family}", tot)
    assert tot["MISMATCH"] == 0 and tot["overflow"] == 0 and tot["nonfinite"] == 0 and tot["worlds"] >= 12000, tot


def test_parity_soak_on_the_384_row_general_build():
    """The same general code compiled for 128 contact slots (models that ask for more than 64): two families, every world against the oracle."""
    import soak_parity
    for fam in ("multi", "big"):
        tot = soak_parity.run(49000, 30, 64, verbose=False, slots=128, **FAMILIES[fam])
        print(f"384-row general parity:{fam}", tot)
        assert tot["MISMATCH"] == 0 and tot["overflow"] == 0 and tot["nonfinite"] == 0 and tot["worlds"] >= 1800, tot
