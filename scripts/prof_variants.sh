# per-kernel durations of one timing script for several builds of the library:  gpurun -- 'bash scripts/prof_variants.sh <tag> <script> <variant> ...'
# ("-" = the default build; a variant name selects nerfstudio-thermal_amd/build/libtn_<name>.so)
tag=$1; script=$2; shift 2
for v in "$@"; do
  if [ "$v" = "-" ]; then unset TN_LIB; name=default; else export TN_LIB=nerfstudio-thermal_amd/build/libtn_$v.so; name=$v; fi
  echo "== $name"
  TOP=${TOP:-8} bash scripts/prof_kernels.sh $tag/$name $script | grep -v "^kernel,calls"
done
