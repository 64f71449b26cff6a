# A/B of the two table-scatter paths (TN_SCATTER_MODE=1 binned, 2 segmented): parity tests, entry-point times, step times, per-kernel times
# usage: gpurun -- 'bash scripts/seg_ab.sh [tag:ENV=VALUE ...]'   (each extra leg is mode 2 with that environment setting; TN_LIB=... selects a variant library)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/seg
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 600 python -m pytest tests/test_hip_ops_gpu.py tests/test_fullsize_parity_gpu.py -x -q > gpurun_out/seg/tests.log 2>&1 || { tail -30 gpurun_out/seg/tests.log; exit 1; }
tail -2 gpurun_out/seg/tests.log
fi
run() {  # run <tag> <mode> [ENV=VALUE]
  echo "== $1"
  export TN_SCATTER_MODE=$2
  [ -n "$3" ] && export "$3"
  timeout -k 10 200 python scripts/scatter_time.py
  timeout -k 10 200 python scripts/step_times.py 60 | tail -1
  TOP=14 bash scripts/prof_kernels.sh seg/$1 scripts/scatter_time.py | grep -E "k_grid_|k_seg_" || true
  [ -n "$3" ] && unset "${3%%=*}"
  return 0
}
[ -z "$SKIP_M1" ] && run m1 1
run m2 2
for v in "$@"; do run "${v%%:*}" 2 "${v#*:}"; done
