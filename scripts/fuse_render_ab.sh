# A/B of tn_train_step with (TN_FUSE_RENDER=1) / without tn_render_losses_bwd (default: render_fwd, train_losses, render_bwd as three launches)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fr
timeout -k 10 900 python -m pytest tests/test_hip_ops_gpu.py tests/test_trainer_sequence_gpu.py tests/test_model_gpu.py -x -q -k "render or losses or one_call or train_step or launch" > gpurun_out/fr/tests.log 2>&1 || { tail -40 gpurun_out/fr/tests.log; exit 1; }
tail -2 gpurun_out/fr/tests.log
for f in 0 1 0 1; do
  TN_FUSE_RENDER=$f timeout -k 10 200 python scripts/step_times.py 60 | tail -1
done
