# kernel stats + one-step timelines of the fused step:  gpurun -- 'bash scripts/gpu_timeline.sh <tag>'
# (the LAST 20 steps of a bench run are the in-step measurement, which issues the backward phase by phase: the timelines show steps of the timed region)
tag=${1:-t}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
rm -rf gpurun_out/$tag/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/$tag/prof -o a -- python3 bench.py --no-cpu-baseline --steps 40 --warmup 10 --long-steps 0 $BENCH_ARGS > gpurun_out/$tag/prof.log 2>&1
db=$(find gpurun_out/$tag/prof -name '*.db' | head -1)
python scripts/rocpd_stats.py $db gpurun_out/$tag/kernel_stats.csv --split-grid --tail 10 > gpurun_out/$tag/kernel_stats.txt 2>&1
python scripts/rocpd_timeline.py $db gpurun_out/$tag/timeline.md --step-from-end ${STEP_A:-26} > /dev/null 2> gpurun_out/$tag/timeline.err
python scripts/rocpd_timeline.py $db gpurun_out/$tag/timeline_update.md --step-from-end ${STEP_B:-25} > /dev/null 2>> gpurun_out/$tag/timeline.err
find gpurun_out/$tag/prof -name '*.db' -delete
grep '^{' gpurun_out/$tag/prof.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'])"
