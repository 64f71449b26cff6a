cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
J='import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1:], round(d["value"]), round(d["ms_per_step"],4))'
for rep in 1 2 3; do for f in 0 1; do
TN_FUSE_SMALL=$f python bench.py --no-cpu-baseline --steps 200 --warmup 40 2>/dev/null | python -c "$J" fuse=$f
done; done
for f in 0 1; do
TN_FUSE_SMALL=$f python bench.py --no-cpu-baseline --rays 1024 2>/dev/null | python -c "$J" 1024 fuse=$f
TN_FUSE_SMALL=$f python bench.py --mode separate --rays 8192 --no-cpu-baseline 2>/dev/null | python -c "$J" sep fuse=$f
TN_FUSE_SMALL=$f python bench.py --no-cpu-baseline --path model-api 2>/dev/null | python -c "$J" model-api fuse=$f
done
python bench.py --force-dp --no-cpu-baseline 2>/dev/null | python -c "$J" dp
rocprofv3 --kernel-trace -d gpurun_out/prof_f -o f -- python3 bench.py --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/prof_f.log 2>&1
python scripts/rocpd_timeline.py $(find gpurun_out/prof_f -name '*.db' | head -1) gpurun_out/s2_fused_timeline.md --step-from-end 4 > /dev/null 2>gpurun_out/tl.err
python scripts/rocpd_timeline.py $(find gpurun_out/prof_f -name '*.db' | head -1) gpurun_out/s2_fused_timeline_update.md --step-from-end 5 > /dev/null 2>>gpurun_out/tl.err
rm -rf gpurun_out/prof_f
rocprofv3 --kernel-trace -d gpurun_out/prof_f -o f -- python3 bench.py --no-cpu-baseline --path model-api --steps 50 --warmup 10 > gpurun_out/prof_f.log 2>&1
python scripts/rocpd_timeline.py $(find gpurun_out/prof_f -name '*.db' | head -1) gpurun_out/s2_api_timeline.md --step-from-end 4 > /dev/null 2>gpurun_out/tl.err
rm -rf gpurun_out/prof_f
