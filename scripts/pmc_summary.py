#!/usr/bin/env python3
"""profiles/<prefix>_pmc.json + <prefix>_pmc_summary.md from the passes of scripts/pmc_passes.sh.

    python scripts/pmc_summary.py gpurun_out/pmc profiles/r02

Per-launch means of every counter, keyed by kernel and grid (X x Y threads); a k_grid_fold launch is filed under the k_grid_bin launch it
follows (the two kernels + nothing else make one tn_hash_scatter call).  The JSON carries the hash of the kernel sources it was measured on
(bench.source_hash): bench.py reports `traffic` only while that hash matches the sources it runs."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short(name):
    return name.split("(")[0].replace("void ", "")


def main():
    root, prefix = sys.argv[1], sys.argv[2]
    acc = defaultdict(lambda: defaultdict(list))
    for path in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
        trace = path.replace("counter_collection", "kernel_trace")
        dims, order = {}, []
        with open(trace) as f:
            for row in csv.DictReader(f):
                dims[row["Dispatch_Id"]] = (short(row["Kernel_Name"]), f'{row["Grid_Size_X"]}x{row["Grid_Size_Y"]}', int(row["Start_Timestamp"]))
        # file the fold under the preceding bin launch
        owner, last_bin = {}, None
        for did, (name, grid, _) in sorted(dims.items(), key=lambda kv: kv[1][2]):
            if name.startswith("k_grid_bin"):
                last_bin = grid
            owner[did] = (name, last_bin if name.startswith("k_grid_fold") else grid)
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Dispatch_Id"] not in owner:
                    continue
                name, grid = owner[row["Dispatch_Id"]]
                if not name.startswith("k_"):
                    continue
                acc[f"{name} {grid}"][row["Counter_Name"]].append(float(row["Counter_Value"]))
    kernels = {k: {c: sum(v) / len(v) for c, v in ctr.items()} for k, ctr in sorted(acc.items())}
    for k, m in kernels.items():
        # HBM-side bytes per launch: reads = RDREQ x 64 B (calibrated for these 8-byte gathers; FETCH_SIZE (KB) under-reports wide streams 2x on
        # gfx950, so for streaming kernels max(FETCH_SIZE*2, RDREQ*64) is the safer figure); writes = WRITE_SIZE KB
        rd = max(m.get("TCC_EA0_RDREQ_sum", 0.0) * 64.0, m.get("FETCH_SIZE", 0.0) * 1024.0)
        m["read_bytes"] = rd
        m["write_bytes"] = m.get("WRITE_SIZE", 0.0) * 1024.0
        m["traffic_bytes"] = rd + m["write_bytes"]
        if m.get("SQ_BUSY_CU_CYCLES"):
            # SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD (4 matrix cores per CU), SQ_BUSY_CU_CYCLES per CU: k_field_mlp_fwd reads 2.6 without the 4
            m["mfma_busy_frac"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * m["SQ_BUSY_CU_CYCLES"])
        if m.get("SQ_BUSY_CYCLES") and m.get("SQ_LDS_IDX_ACTIVE") is not None:
            m["lds_conflict_frac"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(m.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)
    import bench

    out = {"source_hash": bench.source_hash(), "workload": "thermal-nerfacto shared, 4096 rays, stand-alone launches (scripts/pmc_target.py)", "kernels": kernels}
    with open(prefix + "_pmc.json", "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    cols = ["read_bytes", "write_bytes", "TCC_EA0_ATOMIC_sum", "TCC_HIT_sum", "TCC_MISS_sum", "mfma_busy_frac", "SQ_INSTS_LDS", "lds_conflict_frac"]
    with open(prefix + "_pmc_summary.md", "w") as f:
        f.write("# PMC counters per launch (rocprofv3 --pmc, separate passes; scripts/pmc_passes.sh -> scripts/pmc_summary.py)\n\n")
        f.write(f"kernel sources hash `{out['source_hash']}`; N = 4096 rays, shared mode, every kernel launched alone.\n")
        f.write("read = max(TCC_EA0_RDREQ x 64 B, FETCH_SIZE KB); write = WRITE_SIZE KB; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES);\n")
        f.write("lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.  A k_grid_fold row is filed under the grid of the k_grid_bin launch it follows.\n\n")
        f.write("| kernel, grid | read MB | write MB | atomic req | L2 hit | L2 miss | MFMA busy | LDS instr | LDS conflict |\n|---|---|---|---|---|---|---|---|---|\n")
        for k, m in kernels.items():
            g = lambda c: m.get(c, float("nan"))  # noqa: E731
            f.write(f"| `{k}` | {g('read_bytes') / 1e6:.1f} | {g('write_bytes') / 1e6:.1f} | {g('TCC_EA0_ATOMIC_sum'):.0f} | {g('TCC_HIT_sum'):.0f} | "
                    f"{g('TCC_MISS_sum'):.0f} | {g('mfma_busy_frac'):.3f} | {g('SQ_INSTS_LDS'):.0f} | {g('lds_conflict_frac'):.3f} |\n")
    print("wrote", prefix + "_pmc.json", prefix + "_pmc_summary.md", len(kernels), "kernels")


if __name__ == "__main__":
    main()
