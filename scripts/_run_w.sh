cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for q in 8 12 16 24; do for a in "" "--force-dp" "--path model-api --steps 200 --warmup 60"; do GPU_MAX_HW_QUEUES=$q python bench.py $a --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1:], round(d['value']), round(d['ms_per_step'],4))" Q=$q $a; done; done
