# PMC passes + summary + a bench line:  gpurun --timeout 1200 -- 'bash scripts/gpu_pmc.sh r04'
pfx=${1:-r04}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
bash scripts/pmc_passes.sh gpurun_out/pmc | tail -12
python scripts/pmc_summary.py gpurun_out/pmc profiles/$pfx > gpurun_out/pmc_summary.log 2>&1; tail -2 gpurun_out/pmc_summary.log
cp profiles/${pfx}_pmc.json profiles/${pfx}_pmc_summary.md gpurun_out/
python bench.py --steps 50 --warmup 10 > gpurun_out/bench_${pfx}.json 2> gpurun_out/bench_${pfx}.err
grep '^{' gpurun_out/bench_${pfx}.json | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['value']), d['ms_per_step'], d['long_run']['median_ms_per_step'], 'frac', round(r['frac'],4), 'in-step', r['avg_launch_ms_in_step'], 'traffic', r['traffic'], r.get('traffic_unavailable'))
print(d['cpu_baseline']['value'], d['gpu_over_cpu'])"
