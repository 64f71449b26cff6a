#!/bin/bash
# SQ counters of the scatter kernels (LDS utilisation, waits) -- run on the GPU box from the repo root:  bash scripts/pmc_sq.sh <outdir>
out=${1:-gpurun_out/pmc_sq}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rm -rf $out/p$i
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/p$i -- python3 scripts/scatter_pmc.py > $out.p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - "$out" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_grid" not in name: continue
        acc[(name, row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, ctr in sorted(acc.items()):
    print(k)
    for c, v in sorted(ctr.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.0f}")
PY
