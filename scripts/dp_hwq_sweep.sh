# bench.py --force-dp (the N > 1 schedule on a one-rank RCCL group, AVG = a real RCCL kernel beside the folds) and the plain step for a range of
# GPU_MAX_HW_QUEUES settings -> gpurun_out/<round>/dp_hwq_sweep.json (copied to profiles/<round>_dp_hwq_sweep.json).  Legs: force_dp = what
# bench.py --force-dp runs (parallel.ScheduleGuard picks the schedule), force_dp_overlapped_only = the overlapped schedule with the guard off,
# plain = the single-GPU step.   usage: bash scripts/dp_hwq_sweep.sh r05
R=${1:-r05}
mkdir -p gpurun_out/$R
out=gpurun_out/$R/dp_hwq_sweep.json
echo "[" > $out
first=1
for q in unset 4 5 7 8 16; do
  for leg in force_dp force_dp_overlapped_only plain; do
    if [ "$q" = "unset" ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    if [ "$leg" = "force_dp" ]; then flags="--force-dp"; elif [ "$leg" = "force_dp_overlapped_only" ]; then flags="--force-dp --no-dp-guard"; else flags=""; fi
    line=$(python bench.py $flags --no-cpu-baseline --no-extras --steps 50 --warmup 10 --long-steps 60 2>/dev/null | grep '^{' | tail -1)
    [ $first -eq 1 ] || echo "," >> $out
    first=0
    python - "$q" "$leg" <<PY >> $out
import json, sys
d = json.loads('''$line''') if '''$line''' else {}
lr = d.get("long_run") or {}
print(json.dumps({"GPU_MAX_HW_QUEUES": sys.argv[1], "leg": sys.argv[2], "ms_per_step": d.get("ms_per_step"), "rays_per_s": d.get("value"),
                  "median_ms": lr.get("median_ms_per_step"), "median_ms_update_steps": lr.get("median_ms_update_steps"),
                  "median_ms_other_steps": lr.get("median_ms_other_steps"), "dp": d.get("dp"), "hw_queues_env": d.get("hw_queues_env")}))
PY
    echo "$q $leg done" >&2
  done
done
echo "]" >> $out
python - <<PY
import json
rows = json.load(open("$out"))
for r in rows: print(r["GPU_MAX_HW_QUEUES"], r["leg"], r["ms_per_step"], r["median_ms"], r["median_ms_update_steps"], r["median_ms_other_steps"])
PY
