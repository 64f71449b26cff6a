cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s4
timeout 900 python -m pytest tests/test_hip_ops_gpu.py tests/test_fullsize_parity_gpu.py tests/test_model_gpu.py tests/test_model_api_gpu.py tests/test_dp_gpu.py -m gpu -x -q > gpurun_out/s4/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/s4/tests.log
python scripts/time_field_bwd.py
python scripts/time_ops.py 2>&1 | tail -2
TN_FIELD_DPOS_JAC=0 python scripts/time_ops.py 2>&1 | tail -2
for j in 0 1; do
  TN_FIELD_DPOS_JAC=$j python bench.py --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/s4/bench_jac$j.json 2> gpurun_out/s4/bench_jac$j.err
done
python - <<PY
import json
for s in (0,1):
    d=json.loads([l for l in open(f"gpurun_out/s4/bench_jac{s}.json") if l.startswith("{")][-1])
    print(s, round(d["value"]), d["ms_per_step"], d["long_run"]["median_ms_per_step"], d["long_run"]["median_ms_update_steps"], d["long_run"]["median_ms_other_steps"], d["roofline"]["avg_launch_ms"])
PY
