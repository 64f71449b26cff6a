# copies what scripts/refresh_profiles.sh left under gpurun_out/<round>/ into profiles/<round>_* (run in the build container after the gpurun call)
R=${1:-r05}
cd "$(dirname "$0")/.."
for f in gpurun_out/$R/bench_*.json; do cp $f profiles/${R}_bench_$(basename $f | sed 's/^bench_//'); done
cp gpurun_out/$R/bench_n1_kernel_stats.csv profiles/${R}_bench_n1_kernel_stats.csv
cp gpurun_out/$R/bench_n1_kernel_stats_tail.txt profiles/${R}_bench_n1_kernel_stats_tail.txt
for t in fused_timeline fused_timeline_update_step dp_timeline separate_timeline; do cp gpurun_out/$R/$t.md profiles/${R}_$t.md; done
cp gpurun_out/$R/rccl_1rank_latency.json profiles/${R}_rccl_1rank_latency.json
cp gpurun_out/$R/${R}_pmc.json profiles/${R}_pmc.json
cp gpurun_out/$R/${R}_pmc_summary.md profiles/${R}_pmc_summary.md
cp gpurun_out/$R/dp_hwq_sweep.json profiles/${R}_dp_hwq_sweep.json
[ -f gpurun_out/$R/mfma_valu_overlap.txt ] && cp gpurun_out/$R/mfma_valu_overlap.txt profiles/${R}_mfma_valu_overlap.txt
[ -f gpurun_out/$R/fwd_ablation.log ] && grep -E '^==' gpurun_out/$R/fwd_ablation.log > profiles/${R}_fwd_ablation.txt
python - <<PY
import json, sys
sys.path.insert(0, ".")
import bench
print("pmc hash", json.load(open("profiles/${R}_pmc.json"))["source_hash"], "sources", bench.source_hash())
d = json.loads(open("profiles/${R}_bench_n1_fused.json").read())
r = d["roofline"]
print("fused", round(d["value"]), d["ms_per_step"], "frac", round(r["frac"], 4), "in-step", r["avg_launch_ms_in_step"], "traffic", r["traffic"], r.get("traffic_unavailable"))
PY
