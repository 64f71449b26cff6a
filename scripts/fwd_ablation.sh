# k_field_mlp_fwd<true> stand-alone under the FWD_ABLATE / FRAG_NT variant builds (scripts/build_variant.sh fwd_<name> tn_field -DFWD_ABLATE=<bits>):
# rocprofv3 kernel trace of scripts/time_field_fwd.py per variant, mean duration of the training chain and of the gather.
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=${1:-gpurun_out/fwd_ablation}
mkdir -p $out
export TIME_BWD=0
for v in default fwd_a1 fwd_a2 fwd_a4 fwd_a8 fwd_a17 fwd_a31 fwd_nt0; do
  if [ $v = default ]; then unset TN_LIB; else export TN_LIB=nerfstudio-thermal_amd/build/libtn_$v.so; fi
  rm -rf $out/$v
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$v -o t -- python3 scripts/time_field_fwd.py > $out/$v.log 2>&1
  f=$(find $out/$v -name '*kernel_stats.csv' | head -1)
  echo "== $v: $(grep -h 'field fwd' $out/$v.log | tail -1)"
  grep -E 'k_field_mlp_fwd|k_field_encode_xcd|k_field_prep' $f | awk -F, '{printf "   %-60s calls %s avg_ns %s\n", substr($1,1,60), $2, $4}'
done
