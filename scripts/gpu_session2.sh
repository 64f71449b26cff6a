cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
timeout 900 python -m pytest tests/test_hip_ops_gpu.py tests/test_fullsize_parity_gpu.py tests/test_model_gpu.py tests/test_model_api_gpu.py -m gpu -x -q > gpurun_out/s2/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/s2/tests.log
for f in 0 1; do
  TN_FIELD_BWD_FUSED=$f python scripts/time_ops.py > gpurun_out/s2/ops_fused$f.log 2>&1; tail -4 gpurun_out/s2/ops_fused$f.log
  TN_FIELD_BWD_FUSED=$f python bench.py --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/s2/bench_fused$f.json 2> gpurun_out/s2/bench_fused$f.err
done
python - <<PY
import json
for s in (0,1):
    d=json.loads([l for l in open(f"gpurun_out/s2/bench_fused{s}.json") if l.startswith("{")][-1])
    print(s, round(d["value"]), d["ms_per_step"], d["long_run"]["median_ms_per_step"], d["long_run"]["median_ms_update_steps"], d["long_run"]["median_ms_other_steps"], d["roofline"]["avg_launch_ms"])
PY
