# Everything under profiles/r02_* in one gpurun call (tests, PMC passes, bench lines, kernel stats, timelines):
#   gpurun --timeout 1200 -- 'bash scripts/refresh_profiles.sh > gpurun_out/refresh.log 2>&1'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/tests_a.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/tests_a.log
bash scripts/pmc_passes.sh gpurun_out/pmc | tail -8
# summarise on the box so that the bench lines below can quote roofline.traffic for the kernels they run (stamped with the csrc hash)
python scripts/pmc_summary.py gpurun_out/pmc profiles/r02 > gpurun_out/pmc_summary.log 2>&1 && cp profiles/r02_pmc.json profiles/r02_pmc_summary.md gpurun_out/
last() { grep '^{' "$1" | tail -1 > "$1.tmp"; mv "$1.tmp" "$1"; }
python bench.py > gpurun_out/bench_fused.json 2> gpurun_out/bench_fused.err; last gpurun_out/bench_fused.json
python bench.py --mode separate --rays 8192 > gpurun_out/bench_sep.json 2>/dev/null; last gpurun_out/bench_sep.json
python bench.py --path model-api --no-cpu-baseline --steps 200 --warmup 60 > gpurun_out/bench_api.json 2>/dev/null; last gpurun_out/bench_api.json
python bench.py --path model-api --api-optimizer torch --no-cpu-baseline --steps 200 --warmup 60 > gpurun_out/bench_api_torch.json 2>/dev/null; last gpurun_out/bench_api_torch.json
python bench.py --no-cpu-baseline --steps 200 --warmup 60 > gpurun_out/bench_fused200.json 2>/dev/null; last gpurun_out/bench_fused200.json
python bench.py --force-dp --no-cpu-baseline > gpurun_out/bench_dp.json 2>/dev/null; last gpurun_out/bench_dp.json
python bench.py --rays 1024 --no-cpu-baseline > gpurun_out/bench_1024.json 2>/dev/null; last gpurun_out/bench_1024.json
python bench.py --nerf-samples 96 --no-cpu-baseline > gpurun_out/bench_96.json 2>/dev/null; last gpurun_out/bench_96.json
python bench.py --workload splat > gpurun_out/bench_splat.json 2>/dev/null; last gpurun_out/bench_splat.json
python scripts/rccl_latency.py 2>/dev/null | grep '^{' > gpurun_out/rccl_latency.json
rm -rf gpurun_out/prof_a gpurun_out/prof_dp gpurun_out/prof_s
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_a -o a -- python3 bench.py --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/prof_a.log 2>&1
rocprofv3 --kernel-trace -d gpurun_out/prof_dp -o dp -- python3 bench.py --force-dp --no-cpu-baseline --steps 20 --warmup 10 > gpurun_out/prof_dp.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_s -o s -- python3 scripts/time_splat.py > gpurun_out/prof_s.log 2>&1
python scripts/rocpd_stats.py $(find gpurun_out/prof_a -name '*.db' | head -1) gpurun_out/kernel_stats.csv --split-grid --tail 10 > gpurun_out/kernel_stats.txt 2>&1
python scripts/rocpd_stats.py $(find gpurun_out/prof_s -name '*.db' | head -1) gpurun_out/splat_stats.csv > gpurun_out/splat_stats.txt 2>&1
python scripts/rocpd_timeline.py $(find gpurun_out/prof_dp -name '*.db' | head -1) gpurun_out/dp_timeline.md --step-from-end 4 > /dev/null 2>gpurun_out/dp_timeline.err
python scripts/rocpd_timeline.py $(find gpurun_out/prof_dp -name '*.db' | head -1) gpurun_out/dp_timeline_update.md --step-from-end 5 > /dev/null 2>>gpurun_out/dp_timeline.err
python scripts/rocpd_timeline.py $(find gpurun_out/prof_a -name '*.db' | head -1) gpurun_out/fused_timeline.md --step-from-end 4 > /dev/null 2>>gpurun_out/dp_timeline.err
python scripts/rocpd_timeline.py $(find gpurun_out/prof_a -name '*.db' | head -1) gpurun_out/fused_timeline_update.md --step-from-end 5 > /dev/null 2>>gpurun_out/dp_timeline.err
find gpurun_out/prof_a gpurun_out/prof_dp gpurun_out/prof_s -name '*.db' -delete
for f in fused sep api api_torch fused200 dp 1024 96 splat; do python - <<PY
import json;d=json.load(open("gpurun_out/bench_$f.json"));print("$f",round(d["value"],1),d["ms_per_step"])
PY
done
cat gpurun_out/rccl_latency.json
