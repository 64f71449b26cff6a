cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python scripts/time_wgrad.py 2>&1 | grep "field bwd"
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fused', round(d['value']), d['ms_per_step'])"
python -m pytest tests/test_hip_ops_gpu.py tests/test_fullsize_parity_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -3
