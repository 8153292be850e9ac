# usage: bash profiles/ab_sparse_libs.sh <lib.so> [<lib.so> ...]: bench.py --workload sparse140 | sparse2k with each library in turn, two
# alternations on the same box: sweep ms, fraction of the HBM peak (sweep, largest wave), whole-call wall ms
cp genlib.jl_amd/lib/libgenphi.so /tmp/keep.so
for rep in 1 2; do for v in "$@"; do cp $v genlib.jl_amd/lib/libgenphi.so; for w in sparse140 sparse2k; do python bench.py --workload $w --steps 8 --warmup 2 2>/dev/null | python -c "import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v'.split('/')[-1], '$w', round(j['ms_per_step'], 4), round(j['roofline']['frac'], 3), round(j['roofline']['largest_wave']['frac'], 3), round(j['config']['call_wall_ms_mean'], 2))"; done; done; done
cp /tmp/keep.so genlib.jl_amd/lib/libgenphi.so
