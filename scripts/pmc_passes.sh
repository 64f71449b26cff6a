#!/bin/bash
# PMC passes behind profiles/rNN_pmc.json (run on the GPU box, from the repo root):
#   gpurun -- 'bash scripts/pmc_passes.sh gpurun_out/pmc'   then   python scripts/pmc_summary.py gpurun_out/pmc profiles/r02
# One rocprofv3 run per counter group (separate passes, kernel-trace only -- never combined with sys/hip traces), the program itself after "--".
out=${1:-gpurun_out/pmc}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p $out
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" \
         "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_INSTS_MFMA"; do
  i=$((i+1))
  rm -rf $out/p$i
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/p$i -- python3 scripts/pmc_target.py > $out/p$i.log 2>&1
  echo "pass $i [$c] rc=$? $(find $out/p$i -name '*counter_collection.csv' | head -1)"
  # calibration kernels (known byte counts per access shape) under the same traffic counters
  if [ $i -le 3 ]; then
    rm -rf $out/c$i
    timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/c$i -- scripts/microbench/pmc_calib > $out/c$i.log 2>&1
    echo "calib $i rc=$?"
  fi
done
