#!/usr/bin/env python3
"""Markdown table of per-launch PMC counters from the passes of scripts/pmc_passes.sh (rocprofv3 counter_collection.csv files).

    python scripts/pmc_summary.py gpurun_out > profiles/r01_pmc_summary.md

Counters are averaged over the launches of (kernel, grid X x Y): the three hash grids share k_grid_scatter and differ by grid shape
(main 786432x2 = 16 levels in two interleaved level groups, prop0 2097152x1, prop1 1572864x1)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
    acc = defaultdict(lambda: defaultdict(list))  # (kernel, grid) -> counter -> values
    for path in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        dims = {}  # dispatch id -> "X x Y" from the kernel trace of the same run (the counter file only has the total grid size)
        trace = path.replace("counter_collection", "kernel_trace")
        if os.path.exists(trace):
            with open(trace) as f:
                for row in csv.DictReader(f):
                    dims[row["Dispatch_Id"]] = f'{row["Grid_Size_X"]}x{row["Grid_Size_Y"]}'
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                if not name.startswith("k_"):
                    continue
                acc[(name, dims.get(row["Dispatch_Id"], row["Grid_Size"]))][row["Counter_Name"]].append(float(row["Counter_Value"]))
    cols = ["FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_ATOMIC_sum", "TCC_HIT_sum", "TCC_MISS_sum"]
    print("# PMC counters per launch (rocprofv3 --pmc, separate passes; bench.py --steps 4 --warmup 2, N = 4096 rays, shared mode)\n")
    print("FETCH_SIZE / WRITE_SIZE are in KB as rocprofv3 reports them. On gfx950 FETCH_SIZE under-reports wide coalesced streams by 2x")
    print("(MI355X_MICROARCH.md, HBM); the gather kernels here read 8-byte entries, for which TCC_EA0_RDREQ x 64 B is the calibrated figure")
    print("(`fetch_MB_rdreq`). Values are means over the launches of one (kernel, grid size).\n")
    print("| kernel | grid (threads) | launches | FETCH_SIZE KB | WRITE_SIZE KB | RDREQ | fetch_MB_rdreq | ATOMIC req | L2 hit | L2 miss |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for (name, grid), ctr in sorted(acc.items()):
        m = {c: (sum(ctr[c]) / len(ctr[c]) if ctr.get(c) else float("nan")) for c in cols}
        n = max(len(v) for v in ctr.values())
        print(f"| `{name}` | {grid} | {n} | {m['FETCH_SIZE']:.0f} | {m['WRITE_SIZE']:.0f} | {m['TCC_EA0_RDREQ_sum']:.0f} | "
              f"{m['TCC_EA0_RDREQ_sum'] * 64 / 1e6:.1f} | {m['TCC_EA0_ATOMIC_sum']:.0f} | {m['TCC_HIT_sum']:.0f} | {m['TCC_MISS_sum']:.0f} |")


if __name__ == "__main__":
    main()
