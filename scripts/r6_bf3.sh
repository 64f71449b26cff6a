# TN_HEAD_BF16X3 (split-bf16 colour head) on one MI355X: its tests, the field launches stand-alone with and without, kernel timelines of the step with and without
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${R:-r6bf3}; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_head_bf16x3_gpu.py tests/test_hip_ops_gpu.py -q -m gpu -k "bf16x3 or field" > $O/tests.log 2>&1; echo "tests rc $?"; tail -5 $O/tests.log
for f in 0 1; do
  TN_HEAD_BF16X3=$f timeout -k 10 200 python scripts/time_field_fwd.py > $O/time_$f.log 2>&1; echo "time $f rc $?"; tail -2 $O/time_$f.log
done
for f in 0 1; do
  rm -rf gpurun_out/prof_bf3
  TN_HEAD_BF16X3=$f rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bf3 -o a -- python3 bench.py --no-cpu-baseline --no-extras --steps 50 --warmup 10 --long-steps 0 > $O/prof_$f.log 2>&1
  DB=$(find gpurun_out/prof_bf3 -name '*.db' | head -1)
  python scripts/rocpd_timeline.py $DB $O/timeline_${f}_update.md --mark k_field_prep --step-index 31 > /dev/null 2> $O/timeline.err
  python scripts/rocpd_stats.py $DB $O/kernel_stats_$f.csv --split-grid --tail 10 > $O/kernel_stats_tail_$f.txt 2>&1
  find gpurun_out/prof_bf3 -name '*.db' -delete
  grep -E "k_field_mlp_fwd|k_field_bwd_fused" $O/kernel_stats_tail_$f.txt | cut -c1-160
  head -1 $O/timeline_${f}_update.md
done
