# same-box A/B of the drop-in path's host-side switches (bench.py --path model-api, autocast + GradScaler), interleaved, REPS rounds:
#   TN_FUSED_SCALER_STEP=0/1 (optim.Optimizers._fused_scaler_step) x --api-single-thread-backward off/on
mkdir -p gpurun_out/r04
for i in $(seq ${REPS:-3}); do
for v in "0 " "1 " "1 --api-single-thread-backward"; do
  set -- $v
  TN_FUSED_SCALER_STEP=$1 python bench.py --path model-api --steps 200 --warmup 60 --no-cpu-baseline --no-extras $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused_scaler_step=$1 $2', round(d['ms_per_step'],4), 'ms')" || exit 1
done
done
