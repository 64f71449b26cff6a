"""Fixture for the conditioning of the chained density: how far the ORACLE's own eval-mode density moves when its field
sample bins move by 1 and by 4 fp32 ulps, per golden set -- tests/golden/conditioning.json.  The oracle is pinned to the reference's outputs
(tests/test_oracle_vs_golden.py), so these are properties of the reference arithmetic.  bench.py reads the FILE (data) to print the ratio
"max chained density error / 1-ulp response" in its parity block; tests/test_conditioning_cpu.py checks that the file is what this script
produces.  Runs on the CPU in ~1 min:  python oracle/make_conditioning.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from helpers import oracle_density_sensitivity  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
SETS = (("shared", "tiny"), ("shared", "default"), ("shared", "default256"))


def measure():
    out = {}
    for mode, size in SETS:
        s1 = oracle_density_sensitivity(GOLDEN, mode, size, ulps=1)[""]
        s4 = oracle_density_sensitivity(GOLDEN, mode, size, ulps=4)[""]
        out[f"{mode}/{size}"] = {"ulp1_max": s1["max"], "ulp1_frac_above_1e-4": s1["frac"], "ulp4_max": s4["max"], "ulp4_frac_above_1e-4": s4["frac"],
                                 "density_scale": s1["scale"]}
    return out


if __name__ == "__main__":
    res = measure()
    with open(os.path.join(GOLDEN, "conditioning.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(json.dumps(res, indent=1))
