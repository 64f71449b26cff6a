cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for e in "A=1" "TN_EXPERIMENT_PROP1_STREAM=1"; do for a in "" "--force-dp" "--mode separate --rays 8192"; do env $e python bench.py $a --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1:], round(d['value']), round(d['ms_per_step'],4))" $e $a; done; done
