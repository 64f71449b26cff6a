cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_ops_gpu.py tests/test_fullsize_parity_gpu.py -x -q -k "scatter or bwd or backward or full" 2>&1 | tail -3
python bench.py --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(round(d['value']), round(d['ms_per_step'],4), {k: round(v['ms'],4) for k,v in d['roofline']['all_kernels'].items()})"
python bench.py --mode separate --rays 8192 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(round(d['value']), round(d['ms_per_step'],4))"
