#!/usr/bin/env python3
"""Per-kernel summary (calls / avg / total / share) of a rocprofv3 --kernel-trace run stored in rocpd sqlite format.

    python scripts/rocpd_stats.py gpurun_out/prof_x/x_results.db [out.csv] [--split-grid]

--split-grid keeps launches of one kernel with different grid sizes apart (the three hash grids share k_grid_scatter).
--tail N additionally prints the mean over the LAST N launches of every (kernel, grid): bench.py times its roofline kernels alone after the
training steps (12 launches each), so the tail is the stand-alone duration, while the all-launch mean includes the in-step launches that
overlap with other streams (companion-stream GEMMs, proposal-network backward) and are stretched by them.
"""
import csv
import sqlite3
import sys


def main():
    args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--tail"]
    split = "--split-grid" in sys.argv
    con = sqlite3.connect(args[0])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    key = "name" + (f", {gx}" if split and gx else "")
    rows = con.execute(f"select {key}, count(*), avg(end-start), sum(end-start) from kernels group by {key} order by 4 desc").fetchall()
    total = sum(r[-1] for r in rows)
    print(f"total kernel ms {total / 1e6:.3f}")
    out = []
    for r in rows:
        name = r[0][:60] + (f" [grid {r[1]}]" if split and gx else "")
        calls, avg, tot = r[-3], r[-2], r[-1]
        out.append((name, calls, avg / 1e3, tot / 1e6, 100.0 * tot / total))
    for name, calls, avg, tot, pct in out[:28]:
        print(f"{name:<78} calls={calls:5d} avg_us={avg:9.1f} tot_ms={tot:8.2f} {pct:5.1f}%")
    if "--tail" in sys.argv:
        n = int(sys.argv[sys.argv.index("--tail") + 1])
        # Only the kernels bench.py launches ALONE after the steps (kernel_roofline: the proposal forward, the three table scatters, the main field's
        # forward launch group and its backward MLP phase) have a stand-alone tail; for every other kernel the last launches are in-step launches
        # (round 4 printed those under the same heading: profiles/r04_bench_n1_kernel_stats_tail.txt lists k_prop_bwd_mlp at 191.7 us there).
        alone = ("k_prop_fwd", "k_grid_bin", "k_grid_fold", "k_seg_bin", "k_seg_fold", "k_field_prep", "k_field_encode_xcd", "k_field_mlp_fwd", "k_field_bwd_fused", "k_field_pack",
                 "k_field_emb_finish")
        is_alone = lambda name: any(a in name for a in alone)  # noqa: E731

        def tail(r):
            q = f"select end-start from kernels where name=? {'and ' + gx + '=?' if split and gx else ''} order by start desc limit {n}"
            d = [x[0] for x in con.execute(q, (r[0], r[1]) if split and gx else (r[0],))]
            return f"{r[0][:60]:<62}{(' [grid %d]' % r[1]) if split and gx else '':<16} tail_avg_us={sum(d) / len(d) / 1e3:9.1f}"

        print(f"\nSTAND-ALONE launches (last {n} of each; kernels bench.py launches alone after the steps -- reproduces roofline.avg_launch_ms / roofline.mfma; a k_seg_bin<false> row with a LARGER grid than the stand-alone one is the in-step launch that carries the d-position co-work blocks):")
        for r in [r for r in rows if is_alone(r[0])][:14]:
            print(tail(r))
        print(f"\nIN-STEP launches (last {n} of each; kernels that only ever run inside a training step, beside whatever the step runs on other streams -- NOT stand-alone):")
        for r in [r for r in rows if not is_alone(r[0])][:12]:
            print(tail(r))
    if len(args) > 1:
        with open(args[1], "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "avg_us", "total_ms", "percent"])
            for row in out:
                w.writerow([row[0], row[1], f"{row[2]:.2f}", f"{row[3]:.3f}", f"{row[4]:.2f}"])


if __name__ == "__main__":
    main()
