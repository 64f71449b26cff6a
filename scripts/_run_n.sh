cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -c "import os; print('affinity', len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /proc/loadavg
for i in 1 2; do python bench.py --force-dp --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('dp', round(d['value']), d['ms_per_step'])"; done
git stash -q 2>/dev/null
python bench.py --path model-api --no-cpu-baseline --steps 100 --warmup 60 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('api', round(d['value']), d['ms_per_step'])"
