#!/bin/bash
# usage: pmc.sh tag lib
TAG=$1; LIB=$2
cp $LIB genlib.jl_amd/lib/libgenphi.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pass in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD"; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --output-format csv -d gpurun_out/r3g/$TAG/$n -- python3 bench.py --workload cfg4 --steps 1 --warmup 0 --no-cpu-baseline --no-d2h > gpurun_out/r3g/$TAG/$n.log 2>&1
done
python3 - $TAG <<'PY'
import csv,glob,sys,collections
tag=sys.argv[1]
for f in sorted(glob.glob(f'gpurun_out/r3g/{tag}/*/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=(r['Kernel_Name'].split('(anonymous namespace)::')+[r['Kernel_Name']])[1].split('(')[0]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
    for k in agg:
        if 'fast' in k and '1024, 24, 6' in k or '512, 52, 16' in k:
            print(tag,k,{c:(v/cnt[(k,c)]) for c,v in agg[k].items()})
PY
rm -rf gpurun_out/r3g/$TAG/*/
