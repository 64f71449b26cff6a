cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_ops_gpu.py tests/test_fullsize_parity_gpu.py tests/test_model_gpu.py tests/test_dp_gpu.py -x -q 2>&1 | tail -3
for rep in 1 2; do for v in 0 1; do TN_DPOS_SPLIT=$v python bench.py --no-cpu-baseline --steps 200 --warmup 40 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1:], round(d['value']), round(d['ms_per_step'],4))" split=$v; done; done
for v in 0 1; do TN_DPOS_SPLIT=$v python bench.py --mode separate --rays 8192 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1:], round(d['value']), round(d['ms_per_step'],4))" sep split=$v; done
for v in 0 1; do TN_DPOS_SPLIT=$v python bench.py --force-dp --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1:], round(d['value']), round(d['ms_per_step'],4))" dp split=$v; done
