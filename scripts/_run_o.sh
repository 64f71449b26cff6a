cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "init_scale" 2>&1 | tail -5
python bench.py --nerf-samples 96 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('96 samples', round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['roofline']['step'])"
