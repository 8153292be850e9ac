#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: the artefacts of round 5 that profiles/ holds.
# rocprofv3 kernel stats + PMC level tables + traffic files (collect.sh + summarize.py) for the BASELINE configurations, then the
# bench lines: the default line (cfg4 + the other configurations + call walls), one line per configuration, the same-box
# A/B "every level dense" (--no-sparse), secondary paths.  usage: collect_all_r05.sh [tag] [part]   part: prof | bench | all
set -u
export GENPHI_ENV_HOOKS=1      # (A/B variants below use environment hooks)
TAG=${1:-r05}; PART=${2:-all}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${TAG}_artifacts; mkdir -p "$O"
if [ "$PART" = prof ] || [ "$PART" = all ]; then
  for w in cfg4 cfg2 cfg3 cfg3s cfg5; do
    bash profiles/collect.sh $TAG $w > "$O/collect_$w.out" 2>&1
    python profiles/summarize.py $TAG $w >> "$O/collect_$w.out" 2>&1
    cp profiles/${TAG}_${w}_kernel_stats.csv profiles/${TAG}_${w}_levels.csv profiles/traffic_$w.json "$O/" 2>/dev/null
    rm -rf gpurun_out/prof_${TAG}_$w
    echo "collected $w: $(tail -1 $O/collect_$w.out)"
  done
fi
if [ "$PART" = bench ] || [ "$PART" = all ]; then
  python bench.py --steps 20 --warmup 5 > "$O/${TAG}_bench_cfg4.json" 2> "$O/bench_cfg4.err"; echo "bench cfg4 (default line)"
  python bench.py --steps 20 --warmup 5 --no-sparse --no-cpu-baseline --no-d2h > "$O/${TAG}_bench_cfg4_every_level_dense.json" 2>/dev/null; echo "bench cfg4 dense"
  for w in cfg2 cfg3 cfg3s cfg5; do
    python bench.py --workload $w --steps 20 --warmup 3 > "$O/${TAG}_bench_$w.json" 2> "$O/bench_$w.err"
    python bench.py --workload $w --steps 20 --warmup 3 --no-sparse --no-cpu-baseline > "$O/${TAG}_bench_${w}_every_level_dense.json" 2>/dev/null
    echo "bench $w"
  done
  for w in cfg2all cfg2q; do python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > "$O/${TAG}_bench_$w.json" 2> "$O/bench_$w.err"; echo "bench $w"; done
  python bench.py --workload cfg4o --steps 5 --warmup 1 --no-cpu-baseline > "$O/${TAG}_bench_cfg4o.json" 2> "$O/bench_cfg4o.err"; echo "bench cfg4o"
  for w in cfg3 cfg2; do
    python bench.py --workload $w --storage f64 --steps 10 --warmup 2 > "$O/${TAG}_bench_${w}_f64.json" 2> "$O/bench_${w}_f64.err"
    python bench.py --workload $w --storage f64 --steps 10 --warmup 2 --no-sparse > "$O/${TAG}_bench_${w}_f64_every_level_dense.json" 2>/dev/null
  done
  for w in sparse140 sparse2k; do python bench.py --workload $w --steps 5 --warmup 1 > "$O/${TAG}_bench_$w.json" 2> "$O/bench_$w.err"; done
  python bench.py --workload cfg3 --exchange --steps 20 --warmup 3 > "$O/${TAG}_bench_panel_cfg3_w1.json" 2> "$O/panel_w1.err"
  GENPHI_PLAN_CACHE=0 python profiles/microbench/call_wall.py cfg2 cfg3 cfg3s cfg5 > "$O/${TAG}_call_wall_one_shot.out" 2>&1
  python profiles/microbench/call_wall.py cfg2 cfg3 cfg3s cfg5 > "$O/${TAG}_call_wall_repeated.out" 2>&1
fi
ls "$O"
