#!/usr/bin/env python3
"""Kernel timeline of ONE training step from a rocprofv3 --kernel-trace run (rocpd sqlite): which stream / queue every kernel ran on, when it
started relative to the step's first kernel and how long it took.  Used to show the order of the RCCL all-reduce kernels relative to the
scatter launches of the data-parallel schedule (bench.py --force-dp).

    python scripts/rocpd_timeline.py gpurun_out/prof_dp/x_results.db [out.md] [--step-from-end 3] [--mark k_field_prep --step-index 30]

A step is delimited by consecutive launches of k_sample_pixels / k_sample_rays (the first kernel of every bench step), or of k_pose_spaced_bins
when the pixel sampling rides in the previous step's optimiser launch, or of k_field_prep when the whole sampling front does.
"""
import re
import sqlite3
import sys


def main():
    opts = ("--step-from-end", "--mark", "--step-index")
    args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] not in opts]
    back = int(sys.argv[sys.argv.index("--step-from-end") + 1]) if "--step-from-end" in sys.argv else 3
    mark = sys.argv[sys.argv.index("--mark") + 1] if "--mark" in sys.argv else None            # kernel-name substring that opens an iteration
    index = int(sys.argv[sys.argv.index("--step-index") + 1]) if "--step-index" in sys.argv else None  # iteration counted from the FIRST mark
    con = sqlite3.connect(args[0])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    lane = [c for c in ("stream_id", "queue_id", "stream", "queue", "tid") if c in cols]
    sel = ", ".join(["name", "start", "end"] + lane)
    rows = con.execute(f"select {sel} from kernels order by start").fetchall()
    if mark is not None:
        marks = [i for i, r in enumerate(rows) if mark in r[0]]
        if index is not None:
            if len(marks) < index + 2:
                print("not enough steps in the trace", len(marks))
                return
            back = len(marks) - 1 - index
    else:
        marks = [i for i, r in enumerate(rows) if r[0].startswith(("k_sample_pixels", "k_sample_rays"))]
    if len(marks) < back + 1:
        # the data manager hands the next batch to the training step (TnTrainStep.next_sample: sampled inside the optimiser launch): the first
        # kernel of an iteration is then the pose correction + level-0 bins
        marks = [i for i, r in enumerate(rows) if r[0].startswith("k_pose_spaced_bins")]
    if len(marks) < back + 1:
        # ... and with the sampling front in the previous step's optimiser launch too (TnTrainStep.next_sampling) an iteration starts at the field
        marks = [i for i, r in enumerate(rows) if "k_field_prep" in r[0]]
    if len(marks) < back + 1:
        print("not enough steps in the trace", len(marks))
        return
    a, b = marks[-back - 1], marks[-back]
    step = rows[a:b]
    t0 = step[0][1]
    lines = [f"step of {len(step)} kernels, {(max(r[2] for r in step) - t0) / 1e3:.1f} us from first start to last end; columns: start_us, dur_us, "
             + "/".join(lane) + ", kernel", ""]
    lines.append("| start us | dur us | " + " | ".join(lane) + " | kernel |")
    lines.append("|---|---|" + "---|" * len(lane) + "---|")
    for r in step:
        nm = re.sub(r"\(anonymous namespace\)::", "", r[0])
        nm = (nm.split("(")[0] if len(nm.split("(")[0]) > 8 else nm)[:70]
        lines.append(f"| {(r[1] - t0) / 1e3:8.1f} | {(r[2] - r[1]) / 1e3:7.1f} | " + " | ".join(str(x) for x in r[3:]) + f" | `{nm}` |")
    text = "\n".join(lines)
    print(text)
    if len(args) > 1:
        with open(args[1], "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
