cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 6 12; do TN_BIN_BLOCKS_PER_CU=$v python bench.py --no-cpu-baseline --steps 200 --warmup 40 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1:], round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['all_kernels']['scatter(main grid)']['ms'],4))" v=$v; done; done
for v in 6 12; do TN_BIN_BLOCKS_PER_CU=$v python bench.py --mode separate --rays 8192 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1:], round(d['value']), round(d['ms_per_step'],4))" sep v=$v; done
