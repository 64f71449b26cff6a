# same-box A/B of two builds of the library: TN_LIB=<other .so> against the in-tree one, interleaved repeats
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${R:-ab}; mkdir -p $O
for rep in 1 2 3; do
for v in prev cur; do
  if [ $v = prev ]; then export TN_LIB=$GRAFT_REPO_ROOT/nerfstudio-thermal_amd/libthermal_nerf_hip_prev.so; else unset TN_LIB; fi
  python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | grep '^{' | tail -1 > $O/bench_${v}_$rep.json
  python - <<PY
import json
d=json.load(open("$O/bench_${v}_$rep.json")); l=d.get("long_run") or {}
print("$v", $rep, round(d["value"]), round(d["ms_per_step"],4), "update", round(l.get("median_ms_update_steps") or 0,4), "other", round(l.get("median_ms_other_steps") or 0,4))
PY
done; done
