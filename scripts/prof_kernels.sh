# per-kernel average durations of one script under rocprofv3:  gpurun -- 'bash scripts/prof_kernels.sh <tag> scripts/time_prop.py [args]'   (TN_LIB is honoured)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag && rm -rf gpurun_out/$tag/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/$tag/prof -o a -- python3 "$@" > gpurun_out/$tag/run.log 2>&1
db=$(find gpurun_out/$tag/prof -name '*.db' | head -1)
python scripts/rocpd_stats.py $db gpurun_out/$tag/kernel_stats.csv --split-grid > gpurun_out/$tag/kernel_stats.txt 2>&1
find gpurun_out/$tag/prof -name '*.db' -delete
head -${TOP:-14} gpurun_out/$tag/kernel_stats.csv | cut -c1-150
