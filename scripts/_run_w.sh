cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in zp2 z2 p2 n2 zp1 zp0; do TN_EXPERIMENT_AUX=$m python bench.py --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1:], round(d['value']), round(d['ms_per_step'],4))" $m; done
