# TN_NEXT_SAMPLING=4 (Adam inside the chain's waves) against the default on one MI355X: the chain test in every mode, bench lines, kernel stats
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${R:-r6fused}; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_datamanager_gpu.py tests/test_trainer_sequence_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -4 $O/tests.log
for m in 1 4 1 4; do
  TN_NEXT_SAMPLING=$m timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_ns$m.json 2> $O/bench_ns$m.err; echo "bench $m rc $?"
  python - <<PY
import json
d=json.loads(open("$O/bench_ns$m.json").read().strip().splitlines()[-1])
l=d.get("long_run") or {}
print($m, round(d["value"]), round(d["ms_per_step"],4), "long: update", l.get("median_ms_update_steps"), "other", l.get("median_ms_other_steps"))
PY
done
for m in 4; do
rm -rf gpurun_out/prof_ns
TN_NEXT_SAMPLING=$m rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ns -o a -- python3 bench.py --no-cpu-baseline --no-extras --steps 50 --warmup 10 --long-steps 0 > $O/prof_$m.log 2>&1
DB=$(find gpurun_out/prof_ns -name '*.db' | head -1)
python scripts/rocpd_stats.py $DB $O/kernel_stats_$m.csv --split-grid --tail 10 > $O/kernel_stats_tail_$m.txt 2>&1
find gpurun_out/prof_ns -name '*.db' -delete
grep -E "k_adam|k_next_sampling" $O/kernel_stats_tail_$m.txt | cut -c1-150
done
