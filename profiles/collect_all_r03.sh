#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: every artefact of round 3 that profiles/ holds.
# Bench lines (one JSON line each), rocprofv3 kernel stats + PMC level tables (collect.sh + summarize.py) for the dense
# workloads, kernel stats and bench lines of the secondary paths (sparse_phi, column panels).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_artifacts; rm -rf "$O"; mkdir -p "$O"
python bench.py --steps 20 --warmup 5 > "$O/r03_bench_cfg4.json" 2> "$O/bench_cfg4.err"
for w in cfg2 cfg3 cfg5; do python bench.py --workload $w --steps 20 --warmup 3 > "$O/r03_bench_$w.json" 2> "$O/bench_$w.err"; done
python bench.py --workload cfg4o --steps 5 --warmup 1 --no-cpu-baseline > "$O/r03_bench_cfg4o.json" 2> "$O/bench_cfg4o.err"
for w in cfg4 cfg3 cfg2 cfg5 cfg4o; do
  bash profiles/collect.sh r03 $w > "$O/collect_$w.out" 2>&1
  python profiles/summarize.py r03 $w >> "$O/collect_$w.out" 2>&1
  cp profiles/r03_${w}_kernel_stats.csv profiles/r03_${w}_levels.csv profiles/traffic_$w.json "$O/" 2>/dev/null
  rm -rf gpurun_out/prof_r03_$w
done
# secondary paths
for w in cfg3 cfg2; do python bench.py --workload $w --storage f64 --steps 10 --warmup 2 > "$O/r03_bench_${w}_f64.json" 2> "$O/bench_${w}_f64.err"; done
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_f64" -- python3 bench.py --workload cfg3 --storage f64 --steps 5 --warmup 1 > "$O/kt_f64.log" 2>&1
find "$O/kt_f64" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$O/r03_cfg3_f64_kernel_stats.csv"; rm -rf "$O/kt_f64"
for w in sparse140 sparse2k; do python bench.py --workload $w --steps 5 --warmup 1 > "$O/r03_bench_$w.json" 2> "$O/bench_$w.err"; done
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_sparse" -- python3 bench.py --workload sparse140 --steps 3 --warmup 1 > "$O/kt_sparse.log" 2>&1
find "$O/kt_sparse" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$O/r03_sparse140_kernel_stats.csv"; rm -rf "$O/kt_sparse"
python bench.py --workload cfg3 --exchange --steps 10 --warmup 2 > "$O/r03_bench_panel_cfg3_w1.json" 2> "$O/panel_w1.err"
GENPHI_PANEL_NAIVE=1 python bench.py --workload cfg3 --exchange --steps 5 --warmup 1 > "$O/r03_bench_panel_cfg3_w1_per_entry_kernel.json" 2>/dev/null
for n in 2 3; do python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2952$n bench.py --gpus $n --backend gloo --single-device --exchange --workload cfg3 --steps 3 --warmup 1 2> "$O/panel_w$n.err" | grep '^{' > "$O/r03_bench_panel_cfg3_w$n.json"; done
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_panel" -- python3 bench.py --workload cfg3 --exchange --steps 5 --warmup 1 > "$O/kt_panel.log" 2>&1
find "$O/kt_panel" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$O/r03_panel_cfg3_w1_kernel_stats.csv"; rm -rf "$O/kt_panel"
ls "$O"
