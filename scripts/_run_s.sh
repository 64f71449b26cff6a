cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for w in _wt/d650ad1 _wt/f534db0 .; do
  for a in "--force-dp" ; do (cd $w && python bench.py $a --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1:], round(d['value']), round(d['ms_per_step'],4))" $w $a); done
done
