"""The conditioning claim DESIGN.md section 4 rests the chained-density tolerance on, as a CPU test of the ORACLE alone (no GPU, no HIP library):
the reference arithmetic's own eval-mode density moves by more than the 1e-4 north-star tolerance when its field sample bins move by one fp32
ulp.  The oracle is pinned to the reference's outputs (tests/test_oracle_vs_golden.py), so this is a property of the reference path."""
import pytest

from helpers import oracle_density_sensitivity


@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_oracle_density_moves_more_than_the_tolerance_under_one_ulp_of_its_bins(golden_dir, mode):
    sens = oracle_density_sensitivity(golden_dir, mode, "tiny")
    for sfx, s in sens.items():
        # measured: shared 1.2e-4 max, 0.1-0.2 % of the samples above 1e-4 (density scale ~1e2)
        assert 2e-5 < s["max"] < 2e-3, (sfx, s)
        assert s["frac"] < 0.02, (sfx, s)
        assert s["scale"] > 1.0, (sfx, s)
    # the claim itself: a 1-ulp move of the sampler's output is enough to exceed a chained max-abs bound of 1e-4 somewhere
    assert max(s["max"] for s in sens.values()) > 1e-4 * 0.5, sens


def test_default_size_sensitivity_is_reported(golden_dir):
    """Same measurement at the shipped table sizes (16 x 2^19, 5 x 2^17), 64 rays: the bound the default-size chained tests derive theirs from."""
    s = oracle_density_sensitivity(golden_dir, "shared", "default")[""]
    assert 1e-6 < s["max"] < 5e-3, s


def test_conditioning_fixture_is_what_the_oracle_measures(golden_dir):
    """tests/golden/conditioning.json (oracle/make_conditioning.py) -- the file bench.py reads for `parity.density_err_over_1ulp_response` --
    against a fresh measurement on the default-size, 64-ray set."""
    import json
    import os

    with open(os.path.join(golden_dir, "conditioning.json")) as f:
        fx = json.load(f)
    assert set(fx) == {"shared/tiny", "shared/default", "shared/default256"}
    s1 = oracle_density_sensitivity(golden_dir, "shared", "default", ulps=1)[""]
    assert fx["shared/default"]["ulp1_max"] == pytest.approx(s1["max"], rel=1e-3)
    assert fx["shared/default"]["ulp1_frac_above_1e-4"] == pytest.approx(s1["frac"], abs=1e-4)
    for v in fx.values():
        assert 1e-4 < v["ulp1_max"] < v["ulp4_max"] < 1e-3 and v["ulp1_frac_above_1e-4"] < v["ulp4_frac_above_1e-4"] < 0.05
