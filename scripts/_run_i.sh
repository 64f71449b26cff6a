cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_splat_gpu.py -x -q 2>&1 | tail -5
python scripts/time_splat.py 2>&1 | tail -1
rm -rf gpurun_out/prof_s
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_s -o s -- python3 scripts/time_splat.py > gpurun_out/prof_s.log 2>&1
python scripts/rocpd_stats.py $(find gpurun_out/prof_s -name '*.db' | head -1) gpurun_out/splat_stats.csv 2>&1 | head -14
