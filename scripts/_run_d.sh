cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_d
rocprofv3 --kernel-trace -d gpurun_out/prof_d -o d -- python3 bench.py --no-cpu-baseline --rays 1024 --steps 50 --warmup 10 > gpurun_out/prof_d.log 2>&1
python scripts/rocpd_timeline.py $(find gpurun_out/prof_d -name '*.db' | head -1) gpurun_out/tl_d.md --step-from-end 4 | cut -c1-150
python scripts/rocpd_timeline.py $(find gpurun_out/prof_d -name '*.db' | head -1) gpurun_out/tl_d2.md --step-from-end 5 | cut -c1-150
