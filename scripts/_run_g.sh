cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for a in "--path model-api" "" "--path model-api" "" "--path model-api --api-optimizer torch"; do python bench.py $a --no-cpu-baseline --steps 200 --warmup 60 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1:], round(d['value']), d['ms_per_step'])" $a; done
python scripts/debug_api_metrics.py 2>&1 | grep "wall"
nproc; cat /proc/loadavg
