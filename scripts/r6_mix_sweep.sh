# sweep of the interleaved chain / Adam layouts (TN_NEXT_SAMPLING=4:c:a:adam_blocks) against the default co-work row (1) and in-line sampling (0)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${R:-r6mix}; mkdir -p $O
for m in ${MODES:-0 1 4:3:1:1024 4:2:2:1024 4:1:1:1024 4:2:1:1024 4:3:1:2048 4:2:2:512 4:3:1:512}; do
  f=$O/bench_$(echo $m | tr ':' '_').json
  TN_NEXT_SAMPLING=$m timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $f
  python - <<PY
import json
try:
    d=json.load(open("$f")); l=d.get("long_run") or {}
    print("$m", round(d["value"]), round(d["ms_per_step"],4), "update", round(l.get("median_ms_update_steps"),4), "other", round(l.get("median_ms_other_steps"),4))
except Exception as e:
    print("$m", "failed", e)
PY
done
