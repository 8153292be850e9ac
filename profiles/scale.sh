#!/bin/bash
# The multi-GPU measurement in one go, for the first lease of an 8-GPU node (nothing here has ever run on more than one GPU):
#   (i)  bench.py --gpus N, N = 1 2 4 8: final-level row shards, upper levels replicated, no collective; kernel-only and end to end
#        with every rank's copy of its own row block to its host;
#   (ii) bench.py --gpus N --workload cfg3s --exchange --backend nccl: storage-sharded levels (column panels), one RCCL all-to-all
#        of parent columns per level step over xGMI; next to the 7 x 153 GB/s prediction the line carries.
# One JSON line per run under $OUT, then a table (profiles/scale_summary.py).  Launch it BEFORE anything else touches the GPUs.
#   SCALE_NS="1 2 4 8"   the rank counts            SCALE_DRY=1   CPU rehearsal: gloo, --dry-run / one shared GPU is never needed
#   SCALE_STEPS=10       timed sweeps per run
set -u
cd "$(dirname "$0")/.."
NS=${SCALE_NS:-"1 2 4 8"}; OUT=${SCALE_OUT:-gpurun_out/scale}; STEPS=${SCALE_STEPS:-10}; DRY=${SCALE_DRY:-0}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
port=29610
run() {   # run N name args...
  local n=$1 name=$2; shift; shift
  port=$((port + 1))
  if [ "$n" = 1 ] && [ "$DRY" = 0 ]; then
    python bench.py --gpus 1 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "$port" bench.py --gpus "$n" "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
  fi
  echo "$name rc=$?"
}
for n in $NS; do
  if [ "$DRY" = 1 ]; then
    run "$n" "rows_n$n" --dry-run --workload "${SCALE_DRY_WORKLOAD:-cfg2}"
  else
    run "$n" "rows_n$n" --steps "$STEPS" --warmup 2 --no-cpu-baseline --no-others
    run "$n" "panels_cfg3s_n$n" --workload cfg3s --exchange --backend nccl --pg --steps "$STEPS" --warmup 2 --no-cpu-baseline
  fi
done
python profiles/scale_summary.py "$OUT" $NS
