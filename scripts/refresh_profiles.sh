# Everything under profiles/<round>_* in one gpurun call (tests, PMC passes, bench lines, kernel stats, timelines):
#   gpurun --timeout 1200 -- 'bash scripts/refresh_profiles.sh > gpurun_out/refresh.log 2>&1'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${R:-r06}
mkdir -p gpurun_out/$R
PART=${PART:-all}   # 3: kernel traces / timelines only; 1: tests, PMC passes, hardware-queue sweep, forward ablation, microbenchmark; 2: bench lines, kernel stats, timelines (two gpurun calls: each stays under the 20-minute limit)
if [ "$PART" != "2" ] && [ "$PART" != "3" ]; then
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/$R/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/$R/tests.log
bash scripts/pmc_passes.sh gpurun_out/pmc | tail -12
python scripts/pmc_summary.py gpurun_out/pmc profiles/$R > gpurun_out/$R/pmc_summary.log 2>&1 && cp profiles/${R}_pmc.json profiles/${R}_pmc_summary.md gpurun_out/$R/
bash scripts/dp_hwq_sweep.sh $R 2> gpurun_out/$R/sweep.err | tail -20
if [ -z "$SKIP_ABLATION" ]; then  # (SKIP_ABLATION=1: the forward kernels and the microbenchmark are unchanged since the last run -- their files under profiles/ stay)
bash scripts/fwd_ablation.sh gpurun_out/$R/fwd_ablation > gpurun_out/$R/fwd_ablation.log 2>&1; grep -E '^==' gpurun_out/$R/fwd_ablation.log
find gpurun_out/$R/fwd_ablation -name '*.csv' ! -name '*kernel_stats.csv' -delete; find gpurun_out/$R/fwd_ablation -name '*.db' -delete
timeout -k 5 120 scripts/microbench/mfma_valu_overlap > gpurun_out/$R/mfma_valu_overlap.txt 2>&1; tail -18 gpurun_out/$R/mfma_valu_overlap.txt
fi
fi
if [ "$PART" = "1" ]; then exit 0; fi
if [ "$PART" != "3" ]; then  # (PART=3: only the kernel traces / timelines below)
last() { grep '^{' "$1" | tail -1 > "$1.tmp"; mv "$1.tmp" "$1"; }
b() { out=gpurun_out/$R/bench_$1.json; shift; python bench.py "$@" > $out 2>/dev/null; last $out; }
b n1_driver_invocation --steps 20 --warmup 5
b n1_fused
b n1_fused_200steps --no-cpu-baseline --no-extras --steps 200 --warmup 60
b n1_separate_8192 --mode separate --rays 8192
b n1_model_api --path model-api --no-cpu-baseline --steps 200 --warmup 60
b n1_model_api_single_thread_backward --path model-api --api-single-thread-backward --no-cpu-baseline --steps 200 --warmup 60
b n1_model_api_no_scaler --path model-api --no-grad-scaler --no-cpu-baseline --steps 200 --warmup 60
b n1_model_api_torch_adam --path model-api --api-optimizer torch --no-cpu-baseline --steps 200 --warmup 60
b n1_no_grad_scaler --no-grad-scaler --no-cpu-baseline --no-extras
b n1_force_dp --force-dp --no-cpu-baseline --no-extras
b n1_force_dp_overlapped --force-dp --no-dp-guard --no-cpu-baseline --no-extras
b n1_force_dp_overlapped_sum --force-dp --no-dp-guard --force-dp-sum --no-cpu-baseline --no-extras
b n1_force_dp_sharded --force-dp --dp-shard-optimizer --no-grad-scaler --no-cpu-baseline --no-extras
b n1_1024rays --rays 1024 --no-cpu-baseline
b n1_96samples --nerf-samples 96 --no-cpu-baseline
b n1_splat_1080p --workload splat
python scripts/rccl_latency.py 2>/dev/null | grep '^{' > gpurun_out/$R/rccl_1rank_latency.json
# round 6: the next iteration's sampling front inside the optimiser launch -- same-box A/B of the launch modes (0 = in line, 1 = co-work (default), 2 = serial, 3 = companion stream, 4 = opt-in: Adam inside the chain's waves)
for m in 0 1 2 3 4; do TN_NEXT_SAMPLING=$m python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > gpurun_out/$R/bench_n1_next_sampling_mode$m.json; done
python scripts/eval_probe.py > gpurun_out/$R/eval_render.json 2>/dev/null
python scripts/sampler_ulps.py gpurun_out/$R/sampler_ulps.md > /dev/null 2>&1
fi
rm -rf gpurun_out/prof_a gpurun_out/prof_dp gpurun_out/prof_sep
# (the last 20 steps of a fused / separate run are the in-step measurement, which issues the backward phase by phase: the timelines show steps of the timed region)
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_a -o a -- python3 bench.py --no-cpu-baseline --no-extras --steps 50 --warmup 10 --long-steps 0 > gpurun_out/$R/prof_a.log 2>&1
rocprofv3 --kernel-trace -d gpurun_out/prof_dp -o dp -- python3 bench.py --force-dp --no-dp-guard --no-cpu-baseline --steps 20 --warmup 10 --long-steps 0 > gpurun_out/$R/prof_dp.log 2>&1
rocprofv3 --kernel-trace -d gpurun_out/prof_sep -o sep -- python3 bench.py --mode separate --rays 8192 --no-cpu-baseline --steps 20 --warmup 10 --long-steps 0 > gpurun_out/$R/prof_sep.log 2>&1
python scripts/rocpd_stats.py $(find gpurun_out/prof_a -name '*.db' | head -1) gpurun_out/$R/bench_n1_kernel_stats.csv --split-grid --tail 10 > gpurun_out/$R/bench_n1_kernel_stats_tail.txt 2>&1
# (round 6: an iteration of the one-call path starts at k_field_prep -- its sampling front ran in the previous iteration's optimiser launch; iterations
# 0-9 are the warm-up, 10-59 the timed region, then the phased and stand-alone launches: two consecutive timed iterations are picked by index)
python scripts/rocpd_timeline.py $(find gpurun_out/prof_a -name '*.db' | head -1) gpurun_out/$R/fused_timeline.md --mark k_field_prep --step-index 30 > /dev/null 2> gpurun_out/$R/timeline.err
python scripts/rocpd_timeline.py $(find gpurun_out/prof_a -name '*.db' | head -1) gpurun_out/$R/fused_timeline_update_step.md --mark k_field_prep --step-index 31 > /dev/null 2>> gpurun_out/$R/timeline.err
# (which of two consecutive steps updates the proposal networks depends on the run: the one with fewer launches is the step without an update)
na=$(head -1 gpurun_out/$R/fused_timeline.md | sed 's/step of \([0-9]*\) kernels.*/\1/'); nb=$(head -1 gpurun_out/$R/fused_timeline_update_step.md | sed 's/step of \([0-9]*\) kernels.*/\1/')
if [ "$na" -gt "$nb" ]; then mv gpurun_out/$R/fused_timeline.md gpurun_out/$R/tmp.md; mv gpurun_out/$R/fused_timeline_update_step.md gpurun_out/$R/fused_timeline.md; mv gpurun_out/$R/tmp.md gpurun_out/$R/fused_timeline_update_step.md; fi
python scripts/rocpd_timeline.py $(find gpurun_out/prof_dp -name '*.db' | head -1) gpurun_out/$R/dp_timeline.md --step-from-end 4 > /dev/null 2>> gpurun_out/$R/timeline.err
python scripts/rocpd_timeline.py $(find gpurun_out/prof_sep -name '*.db' | head -1) gpurun_out/$R/separate_timeline.md --step-from-end 25 > /dev/null 2>> gpurun_out/$R/timeline.err
find gpurun_out/prof_a gpurun_out/prof_dp gpurun_out/prof_sep -name '*.db' -delete
for f in gpurun_out/$R/bench_*.json; do python - <<PY
import json
try:
    d=json.load(open("$f")); print("$f".split("bench_")[1], round(d["value"],1), round(d["ms_per_step"],4), (d.get("long_run") or {}).get("median_ms_per_step"))
except Exception as e:
    print("$f", "unreadable", e)
PY
done
cat gpurun_out/$R/rccl_1rank_latency.json
