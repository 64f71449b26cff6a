cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
rm -rf gpurun_out/pmc_s
timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_s -- python3 scripts/time_splat.py > gpurun_out/pmc_s.log 2>&1
python - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/pmc_s/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no counters:", open("gpurun_out/pmc_s.log").read()[-600:])
else:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:30]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for k in acc:
        if "splat_raster" in k or "splat_project" in k:
            print(k, {c: round(v / cnt[k][c]) for c, v in acc[k].items()})
PY
done
