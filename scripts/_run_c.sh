cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_c
rocprofv3 --kernel-trace -d gpurun_out/prof_c -o c -- python3 bench.py --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/prof_c.log 2>&1
python scripts/rocpd_stats.py $(find gpurun_out/prof_c -name '*.db' | head -1) gpurun_out/kernel_stats_c.csv --split-grid --tail 10 2>&1 | head -16
python scripts/rocpd_timeline.py $(find gpurun_out/prof_c -name '*.db' | head -1) gpurun_out/tl_c.md --step-from-end 4 | grep "wgrad\|grid_bin\|grid_fold\|mlp_bwd\|adam\|step of"
