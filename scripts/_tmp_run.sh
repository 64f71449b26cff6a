set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cw
timeout -k 10 900 python -m pytest tests/test_hip_ops_gpu.py tests/test_model_gpu.py -x -q > gpurun_out/cw/tests.log 2>&1 || { tail -40 gpurun_out/cw/tests.log; exit 1; }
tail -2 gpurun_out/cw/tests.log
TOP=40 bash scripts/prof_kernels.sh cw/k scripts/step_times.py 40 | grep -E "sample_rays|pose_spaced|pose_bwd|field_prep"
