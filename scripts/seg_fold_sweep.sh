# per-kernel times of the segmented scatter under a list of settings: gpurun -- 'bash scripts/seg_fold_sweep.sh TN_SEG_FOLD_GW=5 TN_LIB=nerfstudio-thermal_amd/build/libtn_x.so ...'
cd $GRAFT_REPO_ROOT
export TN_SCATTER_MODE=2
for v in "$@"; do
  echo "== $v"
  export $v
  TOP=14 bash scripts/prof_kernels.sh seg/sweep scripts/scatter_time.py ${WHICH:-main} | grep -E "k_seg_fold" || true
  unset "${v%%=*}"
done
