cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s1
for st in 1 16; do
  TN_BIN_COUNT_STRIDE=$st python scripts/scatter_time.py > gpurun_out/s1/scatter_stride$st.log 2>&1; tail -3 gpurun_out/s1/scatter_stride$st.log
done
TN_BIN_COUNT_STRIDE=1 python bench.py --no-cpu-baseline --steps 50 --warmup 10 --ops > gpurun_out/s1/bench_stride1.json 2> gpurun_out/s1/bench_stride1.err
python bench.py --no-cpu-baseline --steps 50 --warmup 10 --ops > gpurun_out/s1/bench_stride16.json 2> gpurun_out/s1/bench_stride16.err
python - <<PY
import json
for s in (1,16):
    d=json.loads([l for l in open(f"gpurun_out/s1/bench_stride{s}.json") if l.startswith("{")][-1])
    print(s, round(d["value"]), d["ms_per_step"], d["long_run"]["median_ms_per_step"], d["long_run"]["median_ms_update_steps"], d["long_run"]["median_ms_other_steps"], d["roofline"]["avg_launch_ms"])
PY
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s1/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/s1/tests.log
