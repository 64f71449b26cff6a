#!/bin/bash
# PMC passes for profiles/rNN_pmc_summary.md (run on the GPU box, from the repo root):
#   gpurun -- 'bash scripts/pmc_passes.sh'   then   python scripts/pmc_summary.py gpurun_out > profiles/rNN_pmc_summary.md
# One rocprofv3 run per counter group (separate passes, kernel-trace only -- never combined with sys/hip traces), the program itself after "--".
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  rm -rf gpurun_out/pmc_$n
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$n -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_$n.log 2>&1
  echo "$n: rc=$? $(find gpurun_out/pmc_$n -name '*counter_collection.csv' | head -1)"
done
