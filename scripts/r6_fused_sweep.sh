# TN_NEXT_SAMPLING=4: how many batches per ray the chain's waves should take (TN_FUSED_SITES), bench lines only
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${R:-r6fsweep}; mkdir -p $O
for cfg in "1 0" "4 10" "1 0" "4 10" "1 0" "4 10"; do
  set -- $cfg
  TN_NEXT_SAMPLING=$1 TN_FUSED_SITES=$2 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/b.json 2> $O/b.err
  python - <<PY
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
l=d.get("long_run") or {}
print("mode $1 sites $2:", round(d["value"]), round(d["ms_per_step"],4), "update", round(l.get("median_ms_update_steps"),4), "other", round(l.get("median_ms_other_steps"),4))
PY
done
