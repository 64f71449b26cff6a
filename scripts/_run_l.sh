cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_ops_gpu.py tests/test_fullsize_parity_gpu.py tests/test_model_gpu.py tests/test_dp_gpu.py tests/test_model_api_gpu.py -x -q 2>&1 | tail -6
for a in "" "--mode separate --rays 8192" "--force-dp"; do python bench.py $a --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1:], round(d['value']), d['ms_per_step'])" $a; done
rm -rf gpurun_out/prof_c
rocprofv3 --kernel-trace -d gpurun_out/prof_c -o c -- python3 bench.py --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/prof_c.log 2>&1
python scripts/rocpd_timeline.py $(find gpurun_out/prof_c -name '*.db' | head -1) gpurun_out/tl_c.md --step-from-end 5 | grep "prop_bwd\|wgrad\|grid_bin\|grid_fold\|mlp_bwd\|adam\|step of\|weights_bwd"
