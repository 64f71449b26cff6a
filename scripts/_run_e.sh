cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_e
rocprofv3 --kernel-trace -d gpurun_out/prof_e -o e -- python3 bench.py --no-cpu-baseline --path model-api --steps 100 --warmup 60 > gpurun_out/prof_e.log 2>&1
python scripts/rocpd_timeline.py $(find gpurun_out/prof_e -name '*.db' | head -1) gpurun_out/tl_e.md --step-from-end 4 | cut -c1-150
python scripts/rocpd_timeline.py $(find gpurun_out/prof_e -name '*.db' | head -1) gpurun_out/tl_e2.md --step-from-end 5 | cut -c1-150
