cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for a in "--force-dp" "" "--force-dp --dp-chunks 3"; do python bench.py $a --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1:], round(d['value']), d['ms_per_step'])" $a; done
rm -rf gpurun_out/prof_dp
rocprofv3 --kernel-trace -d gpurun_out/prof_dp -o dp -- python3 bench.py --force-dp --no-cpu-baseline --steps 20 --warmup 10 > gpurun_out/prof_dp.log 2>&1
python scripts/rocpd_timeline.py $(find gpurun_out/prof_dp -name '*.db' | head -1) gpurun_out/dp_timeline.md --step-from-end 4 | head -3
python scripts/rocpd_timeline.py $(find gpurun_out/prof_dp -name '*.db' | head -1) gpurun_out/dp_timeline_update.md --step-from-end 5 | head -3
cat /proc/loadavg
