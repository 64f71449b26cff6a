set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cw
timeout -k 10 900 python -m pytest tests/test_hip_ops_gpu.py tests/test_model_gpu.py tests/test_trainer_sequence_gpu.py tests/test_fused_trainer_gpu.py -x -q > gpurun_out/cw/tests.log 2>&1 || { tail -40 gpurun_out/cw/tests.log; exit 1; }
tail -2 gpurun_out/cw/tests.log
for c in 1 2; do timeout -k 10 200 python scripts/step_times.py 60 | tail -1; done
TOP=40 bash scripts/prof_kernels.sh cw/k scripts/step_times.py 40 | grep -E "render|clip|losses"
