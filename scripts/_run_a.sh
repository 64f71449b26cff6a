cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/tests_a.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/tests_a.log
bash scripts/pmc_passes.sh gpurun_out/pmc
python bench.py > gpurun_out/bench_fused.json 2> gpurun_out/bench_fused.err; tail -c 600 gpurun_out/bench_fused.json
python bench.py --mode separate --rays 8192 --no-cpu-baseline > gpurun_out/bench_sep.json 2>/dev/null
python bench.py --path model-api --no-cpu-baseline > gpurun_out/bench_api.json 2>/dev/null
python bench.py --force-dp --no-cpu-baseline > gpurun_out/bench_dp.json 2>/dev/null
python bench.py --rays 1024 --no-cpu-baseline > gpurun_out/bench_1024.json 2>/dev/null
python scripts/rccl_latency.py > gpurun_out/rccl_latency.json 2>&1
rm -rf gpurun_out/prof_a gpurun_out/prof_dp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_a -o a -- python3 bench.py --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/prof_a.log 2>&1
rocprofv3 --kernel-trace -d gpurun_out/prof_dp -o dp -- python3 bench.py --force-dp --no-cpu-baseline --steps 20 --warmup 10 > gpurun_out/prof_dp.log 2>&1
python scripts/rocpd_stats.py $(find gpurun_out/prof_a -name '*.db' | head -1) gpurun_out/kernel_stats.csv --split-grid --tail 10 > gpurun_out/kernel_stats.txt 2>&1
python scripts/rocpd_timeline.py $(find gpurun_out/prof_dp -name '*.db' | head -1) gpurun_out/dp_timeline.md > /dev/null 2>gpurun_out/dp_timeline.err
find gpurun_out/prof_a gpurun_out/prof_dp -name '*.db' -size +20M -delete
for f in sep api dp 1024; do python - <<PY
import json;d=json.load(open("gpurun_out/bench_$f.json"));print("$f",d["value"],d["ms_per_step"])
PY
done
cat gpurun_out/rccl_latency.json | tail -1
