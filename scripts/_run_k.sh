cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py --workload splat 2>/dev/null | tail -1 > gpurun_out/bench_splat.json; cat gpurun_out/bench_splat.json | cut -c1-1500
for c in "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "WRITE_SIZE" "FETCH_SIZE"; do
rm -rf gpurun_out/pmc_s
timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_s -- python3 scripts/time_splat.py > gpurun_out/pmc_s.log 2>&1
python - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/pmc_s/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"][:34]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in acc:
    if "splat" in k or "rocprim" in k:
        print(k, {c: round(v / cnt[k][c]) for c, v in acc[k].items()})
PY
done
