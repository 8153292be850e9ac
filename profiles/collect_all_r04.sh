#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: every artefact of round 4 that profiles/ holds.
# Bench lines (one JSON line each), rocprofv3 kernel stats + PMC level tables (collect.sh + summarize.py) for the dense
# workloads, bench lines of the secondary paths.  usage: collect_all_r04.sh [tag]
set -u
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${TAG}_artifacts; rm -rf "$O"; mkdir -p "$O"
# traffic first (bench.py reads profiles/traffic_<workload>.json of the same kernel sources)
for w in cfg4 cfg3s cfg3 cfg2 cfg5 cfg4o cfg2all; do
  bash profiles/collect.sh $TAG $w > "$O/collect_$w.out" 2>&1
  python profiles/summarize.py $TAG $w >> "$O/collect_$w.out" 2>&1
  cp profiles/${TAG}_${w}_kernel_stats.csv profiles/${TAG}_${w}_levels.csv profiles/traffic_$w.json "$O/" 2>/dev/null
  rm -rf gpurun_out/prof_${TAG}_$w
  echo "collected $w"
done
python bench.py --steps 20 --warmup 5 > "$O/${TAG}_bench_cfg4.json" 2> "$O/bench_cfg4.err"; echo "bench cfg4"
for w in cfg3s cfg2 cfg3 cfg5 cfg2all cfg2q; do python bench.py --workload $w --steps 20 --warmup 3 > "$O/${TAG}_bench_$w.json" 2> "$O/bench_$w.err"; echo "bench $w"; done
GENPHI_STAY_NARROW=0 python bench.py --workload cfg3s --steps 20 --warmup 3 --no-cpu-baseline > "$O/${TAG}_bench_cfg3s_nothing_in_place.json" 2>/dev/null
GENPHI_STAY_NARROW=0 python bench.py --workload cfg2all --steps 10 --warmup 2 --no-cpu-baseline --no-d2h > "$O/${TAG}_bench_cfg2all_round3_plan.json" 2>/dev/null
GENPHI_NO_STAY=1 python bench.py --workload cfg2all --steps 10 --warmup 2 --no-cpu-baseline --no-d2h > "$O/${TAG}_bench_cfg2all_nothing_in_place.json" 2>/dev/null
GENPHI_STAY_LAST=0 python bench.py --workload cfg2all --steps 10 --warmup 2 --no-cpu-baseline --no-d2h > "$O/${TAG}_bench_cfg2all_proband_cut_compacted.json" 2>/dev/null
python bench.py --workload cfg4o --steps 5 --warmup 1 --no-cpu-baseline > "$O/${TAG}_bench_cfg4o.json" 2> "$O/bench_cfg4o.err"; echo "bench cfg4o"
for w in cfg3 cfg2; do python bench.py --workload $w --storage f64 --steps 10 --warmup 2 > "$O/${TAG}_bench_${w}_f64.json" 2> "$O/bench_${w}_f64.err"; done
for w in sparse140 sparse2k; do python bench.py --workload $w --steps 5 --warmup 1 > "$O/${TAG}_bench_$w.json" 2> "$O/bench_$w.err"; done
python bench.py --workload cfg3 --exchange --steps 20 --warmup 3 > "$O/${TAG}_bench_panel_cfg3_w1.json" 2> "$O/panel_w1.err"
python bench.py --workload cfg3 --exchange --steps 20 --warmup 3 --panel-host-sync > "$O/${TAG}_bench_panel_cfg3_w1_host_sync_per_step.json" 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 1 --backend nccl --pg --exchange --workload cfg3 --steps 10 --warmup 2 2> "$O/panel_rccl_w1.err" | grep '^{' > "$O/${TAG}_bench_panel_cfg3_w1_rccl_process_group.json"
for n in 2 3; do python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2952$n bench.py --gpus $n --backend gloo --single-device --exchange --workload cfg3 --steps 3 --warmup 1 2> "$O/panel_w$n.err" | grep '^{' > "$O/${TAG}_bench_panel_cfg3_w$n.json"; done
ls "$O"
