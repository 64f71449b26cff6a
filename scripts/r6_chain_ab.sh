# A/B of TnTrainStep::next_sampling on one MI355X: tests, bench lines for modes 0 (in line) / 1 (co-work) / 2 (serial), a kernel timeline of mode 1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${R:-r6b}; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_datamanager_gpu.py tests/test_trainer_sequence_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
for m in ${MODES:-0 1 2 3}; do
  TN_NEXT_SAMPLING=$m timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_ns$m.json 2> $O/bench_ns$m.err; echo "bench $m rc $?"
  python - <<PY
import json
d=json.loads(open("$O/bench_ns$m.json").read().strip().splitlines()[-1])
l=d.get("long_run") or {}
print($m, round(d["value"]), round(d["ms_per_step"],4), "long: update", l.get("median_ms_update_steps"), "other", l.get("median_ms_other_steps"))
PY
done
for m in ${PROF_MODES:-1 3}; do
rm -rf gpurun_out/prof_ns
TN_NEXT_SAMPLING=$m rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ns -o a -- python3 bench.py --no-cpu-baseline --no-extras --steps 50 --warmup 10 --long-steps 0 > $O/prof_$m.log 2>&1
DB=$(find gpurun_out/prof_ns -name '*.db' | head -1)
python scripts/rocpd_timeline.py $DB $O/timeline_${m}_a.md --step-from-end 30 > /dev/null 2> $O/timeline.err
python scripts/rocpd_timeline.py $DB $O/timeline_${m}_b.md --step-from-end 31 > /dev/null 2>> $O/timeline.err
python scripts/rocpd_stats.py $DB $O/kernel_stats_$m.csv --split-grid --tail 10 > $O/kernel_stats_tail_$m.txt 2>&1
find gpurun_out/prof_ns -name '*.db' -delete
grep -E "k_adam|k_next_sampling|k_sample_rays" $O/kernel_stats_tail_$m.txt | cut -c1-150
done
