cd $GRAFT_REPO_ROOT
for lib in "" nerfstudio-thermal_amd/build/lib_base.so "" nerfstudio-thermal_amd/build/lib_base.so; do TN_LIB=$lib timeout -k 10 200 python scripts/step_times.py 60 | tail -1; done
