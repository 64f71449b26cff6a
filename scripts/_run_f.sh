cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_model_api_gpu.py tests/test_components_gpu.py tests/test_model_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -15
for a in "--path model-api" ""; do python bench.py $a --no-cpu-baseline --steps 200 --warmup 60 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1:], round(d['value']), d['ms_per_step'])" $a; done
python scripts/time_api_host.py 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP\|amdgpu.ids" | head -11
