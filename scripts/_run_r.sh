cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for a in "" "--steps 200 --warmup 60" "--force-dp" "--force-dp --dp-chunks 0" "--force-dp --dp-adam-per-range" "--force-dp --dp-chunks 5" "--path model-api --steps 200 --warmup 60" "--path model-api --api-optimizer torch --steps 200 --warmup 60" "--mode separate --rays 8192" "--rays 1024"; do python bench.py $a --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1:], round(d['value']), round(d['ms_per_step'],4))" $a; done
