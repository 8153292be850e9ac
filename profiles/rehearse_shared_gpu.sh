#!/bin/bash
# Rehearsal of the N > 1 GPU path on ONE GPU: N ranks (gloo for the barrier and the max-reduce, every rank on cuda:0) each compute their
# row shard of the final level with the upper levels replicated -- the code the 8-GPU bench runs, minus RCCL and minus the other GPUs.
# Not a scaling measurement (the ranks share one device).  usage (GPU box, repo root): bash profiles/rehearse_shared_gpu.sh "2 4" [workload]
set -u
NS=${1:-"2 4"}; W=${2:-cfg4}; OUT=gpurun_out/rehearse; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
port=29710
for n in $NS; do
  port=$((port + 1))
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n \
      --workload $W --backend gloo --single-device --steps 5 --warmup 2 --no-cpu-baseline --no-d2h > $OUT/shared_${W}_n$n.json 2> $OUT/shared_${W}_n$n.err
  echo "n=$n rc=$? $(tail -c 400 $OUT/shared_${W}_n$n.json | tr -d '\n' | tail -c 300)"
done
# ... and the storage-sharded path (column panels, one all-to-all of parent columns per level step -- through host memory here)
if [ "${REHEARSE_PANELS:-1}" = 1 ]; then
  for n in ${REHEARSE_PANEL_NS:-2}; do
    port=$((port + 1))
    timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n \
        --workload cfg3s --exchange --backend gloo --single-device --steps 3 --warmup 1 --no-cpu-baseline --no-d2h > $OUT/shared_panels_cfg3s_n$n.json 2> $OUT/shared_panels_cfg3s_n$n.err
    echo "panels n=$n rc=$? $(tail -c 300 $OUT/shared_panels_cfg3s_n$n.json | tr -d '\n')"
  done
fi
