# same-box A/B of environment switches on the fused step:  gpurun -- 'bash scripts/ab_env.sh "A=1" "A=2 B=3" ...'   ("-" = no switch)
# prints per variant: value (rays/s), ms/step, median update / other step of the 200-step long run
mkdir -p gpurun_out
for v in "$@"; do
  [ "$v" = "-" ] && v=""
  line=$(env $v python3 bench.py --no-cpu-baseline --steps 100 --warmup 20 --long-steps 200 $BENCH_ARGS 2>gpurun_out/ab_env.err | grep '^{' | tail -1)
  echo "$line" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); l=d['long_run']; print('%-40s %9.0f rays/s  %.4f ms/step  update %.4f  other %.4f  median %.4f' % ('${v:-base}', d['value'], d['ms_per_step'], l['median_ms_update_steps'], l['median_ms_other_steps'], l['median_ms_per_step']))"
done
