cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_c2
rocprofv3 --kernel-trace -d gpurun_out/prof_c2 -o c -- python3 bench.py --mode separate --rays 8192 --no-cpu-baseline --steps 30 --warmup 10 > gpurun_out/prof_c2.log 2>&1
python scripts/rocpd_stats.py $(find gpurun_out/prof_c2 -name '*.db' | head -1) gpurun_out/c2_stats.csv --split-grid 2>&1 | head -24
python scripts/rocpd_timeline.py $(find gpurun_out/prof_c2 -name '*.db' | head -1) gpurun_out/tl_c2.md --step-from-end 4 > /dev/null
find gpurun_out/prof_c2 -name '*.db' -delete
