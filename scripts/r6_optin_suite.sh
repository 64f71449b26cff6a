# The training-level GPU tests under each opt-in switch (the per-kernel tests of the switches are in the suite proper):
#   TN_NEXT_SAMPLING=4  training curve vs the oracle, fused trainer, trainer sequence, held-out quality, pipeline
#   TN_HEAD_BF16X3=1    training curve, fused trainer, held-out quality, the model goldens, pipeline
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${R:-r6env}; mkdir -p $O
TN_NEXT_SAMPLING=4 timeout -k 10 500 python -m pytest tests/test_training_curve_gpu.py tests/test_fused_trainer_gpu.py tests/test_trainer_sequence_gpu.py tests/test_heldout_quality_gpu.py tests/test_pipeline_gpu.py -q -m gpu > $O/mode4.log 2>&1; echo "mode4 rc $?"; tail -1 $O/mode4.log
TN_HEAD_BF16X3=1 timeout -k 10 500 python -m pytest tests/test_training_curve_gpu.py tests/test_fused_trainer_gpu.py tests/test_heldout_quality_gpu.py tests/test_model_gpu.py tests/test_pipeline_gpu.py -q -m gpu > $O/bf3.log 2>&1; echo "bf3 rc $?"; tail -1 $O/bf3.log
